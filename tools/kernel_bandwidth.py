"""Join a single-stream rocprofv3 kernel-stats CSV with the PMC traffic file: achieved HBM GB/s per kernel (traffic / standalone duration).
    python tools/kernel_bandwidth.py <kernel_stats.csv> <pmc_traffic.json> <steps>
Finds the streaming kernels that sit far below the ~5 TB/s a copy reaches (too few workgroups, too little in flight)."""
import csv
import json
import sys

stats, traffic, steps = sys.argv[1], json.load(open(sys.argv[2]))["kernels"], float(sys.argv[3])


def norm(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


rows = []
for r in csv.DictReader(open(stats)):
    n = norm(r["Name"])
    t = traffic.get(n)
    if t is None:
        continue
    us = float(r["AverageNs"]) / 1e3
    by = t["hbm_bytes_per_launch_corrected"]
    rows.append((float(r["TotalDurationNs"]) / 1e6 / steps, n, float(r["Calls"]) / steps, us, by / 1e6, by / us / 1e3))
print(f"{'kernel':64s} {'calls':>6s} {'ms/step':>8s} {'avg us':>8s} {'MB':>8s} {'GB/s':>7s}")
for ms, n, calls, us, mb, gbs in sorted(rows, reverse=True)[:60]:
    print(f"{n[:64]:64s} {calls:6.1f} {ms:8.3f} {us:8.1f} {mb:8.1f} {gbs:7.0f}")
