"""Per-launch timeline of a kernel-trace CSV (rocprofv3 --kernel-trace --output-format csv): for every launch whose name contains
argv[2] (default k_elim_step) the duration and the idle time since the previous kernel's end -- what a step of the 64-block elimination
costs on the device, launch by launch.  python tools/trace_elim_steps.py <dir or csv> [name]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "k_elim_step"
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
for f in files:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    prev_end = None
    seq = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if name in r["Kernel_Name"]:
            seq.append((e - s, s - prev_end if prev_end else 0, r["Kernel_Name"][:40]))
        prev_end = e
    n = len(seq)
    print("%s: %d launches of *%s*" % (f, n, name))
    if not n:
        continue
    # the last evaluation's launches (a block of consecutive ones): print each
    tail = seq[-40:]
    print("  last %d: duration us / gap us" % len(tail))
    print("  " + " ".join("%.1f/%.1f" % (d / 1e3, g / 1e3) for d, g, _ in tail))
    ds = sorted(d for d, _, _ in seq)
    gs = sorted(g for _, g, _ in seq)
    print("  duration median %.1f us, mean %.1f; gap median %.1f us, mean %.1f" % (ds[n // 2] / 1e3, sum(ds) / n / 1e3, gs[n // 2] / 1e3, sum(gs) / n / 1e3))
