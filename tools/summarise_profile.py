"""Summarise a rocprofv3 --kernel-trace --stats run of bench.py: per-group kernel time split and the gate/up
GEMM call (main four-wave launch with the largest grid + the launches that finish its partial last wave).

  python tools/summarise_profile.py <rocprof output dir> <steps incl. warmup> <out prefix under profiles/>
"""
import csv, glob, os, re, shutil, sys


def main():
    d, steps, prefix = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    f = max(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    shutil.copy(f, f"profiles/{prefix}_kernel_stats.csv")
    rs = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rs)
    groups = {"gemm four-wave": 0, "gemm 8-wave": 0, "attention bwd": 0, "attention fwd": 0, "lora skinny": 0, "swiglu": 0, "norms": 0}
    for r in rs:
        n, t = r["Name"], float(r["TotalDurationNs"])
        if "gemm_w4" in n or "gemm_splitk" in n: groups["gemm four-wave"] += t
        elif "gemm_" in n: groups["gemm 8-wave"] += t
        elif "attn_bwd" in n or "attn_delta" in n or "attn_dkdv" in n: groups["attention bwd"] += t
        elif "attn_fwd" in n: groups["attention fwd"] += t
        elif "lora_" in n or "wgrad" in n or "transpose16" in n: groups["lora skinny"] += t
        elif "swiglu" in n: groups["swiglu"] += t
        elif "norm" in n: groups["norms"] += t
    print(f"kernel time {tot / 1e6 / steps:.1f} ms/step")
    for k, v in groups.items():
        print(f"  {k:16s} {100 * v / tot:5.1f} %  {v / 1e6 / steps:6.1f} ms/step")
    t = max(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    tr = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
    w4 = [r for r in tr if re.search(r"gemm_w4_kernel<8, 0, 8, false, false>|gemm_w4_kernel<8, 0, 8(, false)?>", r["Kernel_Name"])]
    key = "Grid_Size_X" if "Grid_Size_X" in w4[0] else "Grid_Size"
    gmax = max(int(r[key]) for r in w4)
    idx = {id(r): i for i, r in enumerate(tr)}
    main_l = [r for r in w4 if int(r[key]) == gmax]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    dm = [dur(r) for r in main_l]
    # the launches that complete the same vlb_gemm_bf16_ws call: a split-K tail (four-wave kernel on K ranges of the
    # partial wave's tiles + gemm_splitk_reduce_kernel) or a re-cut 256x128 tail launch
    tails, kinds = [], set()
    for r in main_l:
        t_us, j = 0.0, idx[id(r)] + 1
        while j < len(tr) and ("true>" in tr[j]["Kernel_Name"] and "gemm_w4_kernel<8, 0, 8, false, true>" in tr[j]["Kernel_Name"]
                               or "gemm_splitk_reduce" in tr[j]["Kernel_Name"] or "gemm_w4_kernel<4" in tr[j]["Kernel_Name"]):
            t_us += dur(tr[j]); kinds.add(re.search(r"gemm_\w+<[^>]*>", tr[j]["Kernel_Name"]).group(0)); j += 1
        if t_us:
            tails.append(t_us)
    print(f"gate/up: main grid {gmax} n={len(dm)} avg {sum(dm) / len(dm):.1f} us; tail n={len(tails)} avg {sum(tails) / max(1, len(tails)):.1f} us ({sorted(kinds)})")
    with open(f"profiles/{prefix}_gateup_gemm_launches.csv", "w") as o:
        o.write("# gate/up GEMM call = gemm_w4_kernel<8,0,8> main launch (largest grid) + the launches that finish the partial last wave "
                f"({' + '.join(sorted(kinds)) or 'none'}); us, from rocprofv3 --kernel-trace of bench.py --steps 8 --warmup 2 --no-cpu-baseline\n")
        o.write(f"main_launches,{len(dm)},avg_us,{sum(dm) / len(dm):.2f},min_us,{min(dm):.2f},max_us,{max(dm):.2f}\n")
        if tails:
            o.write(f"tail_launches,{len(tails)},avg_us,{sum(tails) / len(tails):.2f},min_us,{min(tails):.2f},max_us,{max(tails):.2f}\n")
        o.write(f"call_avg_us,{sum(dm) / len(dm) + (sum(tails) / len(tails) if tails else 0):.2f}\n")


main()
