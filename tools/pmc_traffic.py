
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes): pmc_traffic.py <fetch_dir> <write_dir> <out.json>.
bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE reports half of a wide coalesced read)."""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*$", "", name).strip()
        acc[name].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                   "--no-roofline`; per-launch averages; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reports half of a wide "
                   "coalesced read, MI355X_MICROARCH.md HBM section)", "kernels": {}}
    for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
        if k.startswith("at::") or "elementwise" in k or len(fetch[k]) < 2:
            continue
        f = sum(fetch[k]) / len(fetch[k])
        w = sum(write.get(k, [0.0])) / max(1, len(write.get(k, [0.0])))
        out["kernels"][k] = {"launches": len(fetch[k]), "fetch_kb_raw_avg": round(f, 1), "write_kb_avg": round(w, 1),
                             "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out["kernels"].items())[:12]:
        print(f"{k[:60]:60s} {v['launches']:4d} launches  {v['hbm_bytes_per_launch_corrected'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
