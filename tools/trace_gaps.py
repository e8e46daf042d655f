"""GPU idle time inside the timed steps of a rocprofv3 --kernel-trace run of bench.py.

    rocprofv3 --kernel-trace -d gpurun_out/trace -o t --output-format csv -- python3 bench.py --steps 4 --warmup 2
    python tools/trace_gaps.py gpurun_out/trace

Prints span, busy time (union of kernel intervals), idle time and the largest gaps with the kernels either side."""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    files = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(files[-1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    # timed region = after the last adamw-free long pause: use the last N adamw launches as step ends
    ends = [i for i, e in enumerate(ev) if "adamw_kernel" in e[2]]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    lo, hi = ends[-steps - 1] + 1, ends[-1]
    seg = ev[lo:hi + 1]
    span = seg[-1][1] - seg[0][0]
    busy, cur_end, gaps = 0, seg[0][0], []
    for i, (s, e, n) in enumerate(seg):
        if s > cur_end:
            gaps.append((s - cur_end, seg[i - 1][2][:60], n[:60]))
            cur_end = s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    print(f"{steps} steps: span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms "
          f"({100 * (span - busy) / span:.2f} %), {len(seg)} launches, {len(gaps)} gaps")
    hist = {}
    for g, _, _ in gaps:
        k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else ">=20us"
        hist.setdefault(k, [0, 0]); hist[k][0] += 1; hist[k][1] += g
    for k, (c, t) in hist.items():
        print(f"  gaps {k}: {c} totalling {t / 1e6:.3f} ms")
    for g, a, b in sorted(gaps, reverse=True)[:15]:
        print(f"  {g / 1e3:8.1f} us  after {a}  before {b}")


if __name__ == "__main__":
    main()
