
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a per-kernel table (ms per step)."""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {f}: total kernel time {tot / 1e6:.2f} ms over {steps:.0f} steps = {tot / 1e6 / steps:.2f} ms/step")
print(f"{'kernel':84s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>9s} {'%':>6s}")
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6
    if ms / tot * 1e6 < 0.0015:
        continue
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:84]
    print(f"{name:84s} {float(r['Calls']) / steps:10.1f} {ms / steps:9.3f} {float(r['AverageNs']) / 1e3:9.1f} {100 * ms / tot:6.2f}")
