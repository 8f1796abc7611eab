# usage (GPU box): bash tools/pmc_one.sh <outdir> <cin> <cout> <spatial> <k> <fwd|dgrad|wgrad>   (RUNET_PREC=bf16 for the bf16 kernels)
# two SQ passes + FETCH_SIZE + WRITE_SIZE, each in its own rocprofv3 --pmc run (no tracing domains), condensed to one table
set -e
OUT=$1; shift
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
         "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $ROOT/$OUT/p$i -- python3 $ROOT/tools/${PMC_SCRIPT:-conv_one.py} "$@" ${PMC_ITERS-2} > /dev/null 2>&1
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if n.startswith("at::") or "elementwise" in n or "distribution" in n: continue
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    print(n)
    for k, v in sorted(c.items()):
        print("   %-28s %14.0f  (avg of %d launches)" % (k, sum(v) / len(v), len(v)))
PY
