"""Per-kernel breakdown of ONE replay of the benchmark step from a rocprofv3 kernel trace.

usage (on the GPU box):
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train \
            --no-volume --no-bf16 --no-rooflines
  python3 tools/trace_step.py gpurun_out/trace 5 > profiles/r04_step_breakdown.txt

The trace ends with `steps` graph replays of the same kernel sequence; the period K of the tail is found from the kernel names, the last
`steps` periods are averaged.  Printed: kernels per replay, sum of kernel time, sum of the gaps between consecutive kernels (launch
boundaries inside the graph), wall per replay, and the table by kernel name (launches, total us, share) -- the denominators every
"x % of the step" statement of DESIGN.md uses."""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no *kernel_trace.csv under {d}"
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def short(name):
    n = name.split("(")[0]
    return n if len(n) <= 90 else n[:87] + "..."


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    seq_out = sys.argv[3] if len(sys.argv) > 3 else None      # optional: the ordered kernel list of the last replay (start us, us, name)
    rows = load(d)
    names = [r[2] for r in rows]
    n = len(names)
    K = None
    for k in range(50, n // steps + 1):
        if all(names[n - 1 - i] == names[n - 1 - i - k] for i in range(k * (steps - 1))):
            K = k
            break
    assert K, "no periodic tail found"
    tail = rows[n - K * steps:]
    per = defaultdict(lambda: [0, 0.0])
    ksum = gap = 0.0
    for s in range(steps):
        seq = tail[s * K:(s + 1) * K]
        for i, (a, b, nm) in enumerate(seq):
            per[short(nm)][0] += 1
            per[short(nm)][1] += (b - a) / 1e3
            ksum += (b - a) / 1e3
            if i:
                gap += max(0.0, (a - seq[i - 1][1]) / 1e3)
    if seq_out:
        last = tail[(steps - 1) * K:]
        with open(seq_out, "w") as f:
            f.write(f"# ordered kernels of one replay of the benchmark step ({K} launches): index, start (us from the first), duration (us), kernel\n")
            for i, (a, b, nm) in enumerate(last):
                f.write(f"{i:3d} {(a - last[0][0]) / 1e3:8.1f} {(b - a) / 1e3:7.1f}  {short(nm)}\n")
    wall = sum((tail[(s + 1) * K - 1][1] - tail[s * K][0]) / 1e3 for s in range(steps)) / steps
    print(f"kernels per replay {K}; kernel time {ksum / steps:.1f} us; gaps between consecutive kernels {gap / steps:.1f} us; "
          f"first start -> last end {wall:.1f} us (average of the last {steps} replays)")
    print(f"{'kernel':92s} {'launches':>8s} {'us':>9s} {'share':>6s}")
    for nm, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{nm:92s} {c / steps:8.1f} {t / steps:9.1f} {100 * t / ksum:5.1f}%")


if __name__ == "__main__":
    main()
