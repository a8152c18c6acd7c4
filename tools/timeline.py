#!/usr/bin/env python3
"""Kernel timeline of the steady-state steps of a bench run: where does the time between the kernels go?

    rocprofv3 --kernel-trace --output-format csv -d DIR -o kt -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary
    python tools/timeline.py DIR [first_kernel_regex]

Splits the trace into steps at every launch of the first kernel of a run (k_run_begin), prints the last full step kernel by kernel
(start offset, duration, gap to the previous kernel's end) and, over all steady-state steps, the mean of: step period, sum of kernel
durations, sum of the gaps inside a step, gap between two steps."""
import csv, glob, os, re, sys
d = sys.argv[1]
first = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"k_run_begin")
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("tnsx::", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if first.search(r[2])]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
if not steps:
    sys.exit("no steps found")
# (bench.py's default run appends ten steps on the same cloud in the OTHER input order behind the timed loop: the steps evaluated here are those of
#  the timed loop -- the second and third fifth of all steps, warm and in the main line's order)
steady = steps[len(steps) // 5: (3 * len(steps)) // 5] or steps[len(steps) // 2:]
def stats(st, nxt_start):
    ker = sum(e - s for s, e, _ in st)
    inner = sum(st[i][0] - st[i - 1][1] for i in range(1, len(st)))
    return (nxt_start - st[0][0], ker, inner, nxt_start - st[-1][1])
acc = [stats(st, steps[steps.index(st) + 1][0][0] if steps.index(st) + 1 < len(steps) else rows[starts[-1]][0]) for st in steady]
last = steady[-1]
t0 = last[0][0]
print(f"a steady-state step of the timed loop ({len(last)} kernels):")
prev_end = None
for s, e, name in last:
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:7.2f}"
    print(f"  +{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:9.2f} us  {gap:14s} {name[:70]}")
    prev_end = e
n = len(acc)
print(f"mean over {n} steady-state steps (us): period {sum(a[0] for a in acc) / n / 1e3:.1f} | kernels {sum(a[1] for a in acc) / n / 1e3:.1f} | "
      f"gaps inside a step {sum(a[2] for a in acc) / n / 1e3:.1f} | between steps {sum(a[3] for a in acc) / n / 1e3:.1f} | kernels/period "
      f"{sum(a[1] for a in acc) / sum(a[0] for a in acc):.4f}")
