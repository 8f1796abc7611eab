# usage (GPU box, from the repo root): bash tools/collect_profiles.sh <outdir> <tag> [extra bench.py args]
# rocprofv3 kernel-trace stats (13 steps) + FETCH_SIZE / WRITE_SIZE PMC passes (3 steps each, no tracing domains) of bench.py,
# condensed by tools/prof_summary.py and tools/pmc_traffic.py into <outdir>/<tag>_kernel_stats.txt and <outdir>/<tag>_pmc_traffic.json
set -e
OUT=$1; TAG=$2; shift; shift
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/${TAG}_trace -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/${TAG}_fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/${TAG}_write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
cd $ROOT
python3 tools/prof_summary.py $OUT/${TAG}_trace 13 > $OUT/${TAG}_kernel_stats.txt
python3 tools/pmc_traffic.py $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_pmc_traffic.json 3 "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $*" > $OUT/${TAG}_pmc_top.txt
cp $(ls $OUT/${TAG}_trace/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write
