
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes): pmc_traffic.py <fetch_dir> <write_dir> <out.json> [steps in the profiled run] [command label].
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
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    label = sys.argv[5] if len(sys.argv) > 5 else "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline"
    out = {"note": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `{label}`; per-launch averages; "
                   "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md HBM section)",
           "kernels": {}}
    tot = sum((2 * sum(fetch[k]) + sum(write.get(k, [0.0]))) * 1024 for k in fetch if not (k.startswith("at::") or "elementwise" in k))
    out["step"] = {"hbm_bytes_per_step": int(tot / steps), "gb_per_step": round(tot / steps / 1e9, 2), "steps_profiled": steps,
                   "source": "sum over all kernels of the run / steps"}
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
