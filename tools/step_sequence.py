"""The launch sequence of ONE steady-state step from a rocprofv3 --kernel-trace CSV (single stream: RUNET_NO_WGRAD_STREAM=1): one line per
dispatch - start offset (us), duration (us), grid, short kernel name.  Steps are cut at `adam_multi_kernel`.  With a second argument
only the dispatches whose name contains it are listed, each with the two dispatches in front of it (who issues the torch-side fills and
copies of a step).   usage: step_sequence.py <dir with *kernel_trace.csv> [name filter]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else None
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in csv.DictReader(open(f))),
              key=lambda r: r[0])
cuts = [i for i, r in enumerate(rows) if "adam_multi_kernel" in r[2]]
if len(cuts) < 3:
    sys.exit("fewer than three optimizer launches in the trace")
step = rows[cuts[-2] + 1:cuts[-1] + 1]


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::(?:\(anonymous namespace\)::)?", "at::", name)
    return name[:110]


t0 = step[0][0]
print(f"# {len(step)} dispatches, {(step[-1][1] - t0) / 1e6:.3f} ms")
for i, (s, e, name, grid) in enumerate(step):
    if flt is None:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {grid:>9} {short(name)}")
    elif flt in name:
        for s2, e2, n2, g2 in step[max(0, i - 2):i]:
            print(f"      {(s2 - t0) / 1e3:10.1f} {(e2 - s2) / 1e3:8.1f} {g2:>9} {short(n2)}")
        print(f"  >>  {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {grid:>9} {short(name)}")
