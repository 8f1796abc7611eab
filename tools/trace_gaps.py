"""Gap analysis of a rocprofv3 --kernel-trace CSV: over the last `frac` of the dispatches (steady state), the wall span, the summed kernel
time, how much of the span has at least one kernel running, and the distribution of idle gaps between consecutive kernels.
usage: trace_gaps.py <dir with *kernel_trace.csv> [frac=0.4]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
rows = rows[int(len(rows) * (1 - frac)):]
span = rows[-1][1] - rows[0][0]
busy, cur_end, gaps, overl, where, last = 0, rows[0][0], [], 0, [], ""
for s, e, name in rows:
    if s > cur_end:
        gaps.append(s - cur_end)
        where.append((s - cur_end, last, name, s - rows[0][0]))
        busy += e - s
        cur_end = e
    else:
        overl += min(e, cur_end) - s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    if e >= cur_end:
        last = name
ksum = sum(e - s for s, e, _ in rows)
gaps.sort()
print(f"{len(rows)} dispatches, span {span / 1e6:.2f} ms, summed kernel time {ksum / 1e6:.2f} ms, >=1 kernel running {busy / 1e6:.2f} ms "
      f"({100 * busy / span:.0f} %), overlapped time {overl / 1e6:.2f} ms")
if gaps:
    q = lambda p: gaps[min(len(gaps) - 1, int(p * len(gaps)))] / 1e3
    print(f"idle gaps: {len(gaps)}, total {sum(gaps) / 1e6:.2f} ms, median {q(0.5):.1f} us, p90 {q(0.9):.1f} us, max {gaps[-1] / 1e3:.1f} us")
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
print("largest gaps (us, kernel that ended last before it -> kernel that started after it, offset in the window):")
for g, a, b, off in sorted(where, reverse=True)[:16]:
    print(f"   {g / 1e3:8.1f}  {short(a):44s} -> {short(b):44s} at {off / 1e6:8.2f} ms")
