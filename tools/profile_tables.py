"""Turn rocprofv3 outputs of `bench.py` into the tables committed under profiles/.

  python tools/profile_tables.py stats  <kernel-trace dir> <timed steps> <out prefix> [note]     # per-kernel table + roofline column;
                                                       # only the dispatches between bench.py's two vlb_profile_marker_kernel launches
                                                       # (the timed steps: no model construction, no warm-up) are counted
  python tools/profile_tables.py sq     <pmc dir> <out csv>                                      # SQ counters per kernel
  python tools/profile_tables.py traffic <fetch dir> <write dir> <out csv>                       # gate/up GEMM call: main + split-K tail + reduce

Algorithmic bytes / flops per call are stated here for the default workload (configs[2], B=3, packed
rows M=5861, dim 4096, ff 14336, 32/8 heads x 128): the roofline column is (algorithmic work per call) / (avg duration).
"""
import csv, glob, os, re, sys

M, E, FF, QD, KD, S, B, HQ = int(os.environ.get("VLB_ROWS", 5861)), 4096, 14336, 4096, 1024, 2048, int(os.environ.get("VLB_CLIPS", 3)), 32
PEAK_TF, PEAK_TB = 2500.0, 8.0
# kernel-name regex -> (bound, algorithmic unit per call, value)   [GB for hbm, TFLOP for mfma]
ATT_FWD = 4.0 * B * HQ * S * S * 128 / 2 / 1e12 * (M / (B * S)) ** 2          # causal, packed rows
WORK = [
    (r"rmsnorm_fwd", "hbm", 2 * M * E * 2 / 1e9), (r"rmsnorm_bwd", "hbm", 4 * M * E * 2 / 1e9),
    (r"swiglu_fwd", "hbm", 3 * M * FF * 2 / 1e9), (r"swiglu_bwd", "hbm", 5 * M * FF * 2 / 1e9),
    (r"rope_kernel", "hbm", 2 * M * (QD + KD) * 2 / 1e9), (r"attn_fwd_kernel<128", "mfma", ATT_FWD),
    (r"attn_bwd_dkdv", "mfma", ATT_FWD), (r"attn_bwd_dq", "mfma", 1.5 * ATT_FWD),
    (r"attn_delta", "hbm", 2 * M * QD * 2 / 1e9), (r"adamw_kernel", "hbm", 50.4e6 * 18 / 1e9), (r"sumsq_kernel", "hbm", 50.4e6 * 4 / 1e9),
    (r"splice_kernel", "hbm", M * E * 2 / 1e9), (r"head_pool_kernel", "hbm", M * E * 2 / 1e9),
]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:64]


def timed_window(tr):
    """Dispatches of the kernel trace between bench.py's two marker launches (sorted by start time), or None."""
    marks = [r for r in tr if "vlb_profile_marker_kernel" in r["Kernel_Name"]]
    if len(marks) < 2:
        return None
    lo, hi = int(marks[0]["End_Timestamp"]), int(marks[-1]["Start_Timestamp"])
    return [r for r in tr if int(r["Start_Timestamp"]) >= lo and int(r["End_Timestamp"]) <= hi]


def stats(d, steps, prefix, note=""):
    t = max(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    tr = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
    win = timed_window(tr)
    scope = f"the {steps} timed steps only (dispatches between bench.py's two vlb_profile_marker_kernel launches: no model construction, no warm-up)"
    if win is None:
        win, scope = tr, f"WHOLE trace (no markers found): {steps} steps assumed, model init included"
    acc = {}
    for r in win:
        a = acc.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rs = [{"Name": k, "Calls": v[0], "TotalDurationNs": v[1], "AverageNs": v[1] / v[0]} for k, v in acc.items()]
    rs.sort(key=lambda r: -r["TotalDurationNs"])
    tot = sum(float(r["TotalDurationNs"]) for r in rs)
    with open(f"profiles/{prefix}_kernel_table.csv", "w") as o:
        o.write(f"# rocprofv3 --kernel-trace of `python3 bench.py --steps {steps} ...`; {scope}; "
                "achieved = algorithmic work per call (tools/profile_tables.py WORK) / avg duration; frac = achieved / (8 TB/s | 2.5 PFLOP/s)" + (f"; {note}" if note else "") + "\n")
        o.write("kernel,calls_per_step,avg_us,ms_per_step,pct_of_kernel_time,bound,achieved,unit,frac_of_peak\n")
        for r in rs:
            t = float(r["TotalDurationNs"])
            if t / tot < 0.001:
                continue
            name = r["Name"]
            bound = ach = unit = frac = ""
            for pat, bd, val in WORK:
                if re.search(pat, name):
                    avg_s = float(r["AverageNs"]) * 1e-9
                    if bd == "hbm":
                        ach, unit, frac, bound = val / avg_s / 1e3, "TB/s", val / avg_s / 1e3 / PEAK_TB, "hbm"
                    else:
                        ach, unit, frac, bound = val / avg_s, "TFLOP/s", val / avg_s / PEAK_TF, "mfma"
                    ach, frac = f"{ach:.2f}", f"{frac:.3f}"
                    break
            if not bound and "gemm" in name:
                bound = "mfma"
            o.write(f"\"{short(name)}\",{int(r['Calls']) / steps:.1f},{float(r['AverageNs']) / 1e3:.1f},{t / 1e6 / steps:.2f},{100 * t / tot:.2f},{bound},{ach},{unit},{frac}\n")
    groups = {"gemm four-wave": 0, "gemm mx-fp8": 0, "gemm 8-wave": 0, "quantise/transpose": 0, "optimiser": 0, "attention bwd": 0, "attention fwd": 0, "lora skinny": 0, "swiglu": 0, "norms": 0, "other": 0}
    for r in rs:
        n, t = r["Name"], float(r["TotalDurationNs"])
        if "gemm_w4" in n or "gemm_splitk" in n: groups["gemm four-wave"] += t
        elif "gemm_mxfp8" in n: groups["gemm mx-fp8"] += t
        elif "gemm_" in n: groups["gemm 8-wave"] += t
        elif "quantize" in n or "transpose" in n: groups["quantise/transpose"] += t
        elif "adamw" in n or "sumsq" in n: groups["optimiser"] += t
        elif "attn_bwd" in n or "attn_delta" in n or "attn_dkdv" in n: groups["attention bwd"] += t
        elif "attn_fwd" in n: groups["attention fwd"] += t
        elif "lora_" in n or "wgrad" in n or "transpose16" in n: groups["lora skinny"] += t
        elif "swiglu" in n: groups["swiglu"] += t
        elif "norm" in n: groups["norms"] += t
        else: groups["other"] += t
    with open(f"profiles/{prefix}_kernel_groups.txt", "w") as o:
        o.write(f"kernel time {tot / 1e6 / steps:.1f} ms/step; {scope}" + (f"; {note}" if note else "") + "\n")
        for k, v in groups.items():
            o.write(f"  {k:16s} {100 * v / tot:5.1f} %  {v / 1e6 / steps:6.1f} ms/step\n")
    print(open(f"profiles/{prefix}_kernel_groups.txt").read())
    # gate/up call from the trace (timed steps only)
    tr = win
    f8 = [r for r in tr if "gemm_mxfp8" in r["Kernel_Name"]]
    if f8:          # --fp8 run: the gate/up projection is a vlb_gemm_mxfp8 call = whole-tile launch (largest grid) + the re-cut halves
        key = "Grid_Size_X" if "Grid_Size_X" in f8[0] else "Grid_Size"
        whole = [r for r in f8 if re.search(r"kernel<8", r["Kernel_Name"])]
        gmax = max(int(r[key]) for r in whole)
        idx = {id(r): i for i, r in enumerate(tr)}
        dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        main_l = [r for r in whole if int(r[key]) == gmax]
        # the forward gate/up call only: its dgrad twin (N = 4096, K = 28672) has a smaller grid; the wgrad one has another N
        dm, tails = [dur(r) for r in main_l], []
        for r in main_l:
            j = idx[id(r)] + 1
            tails.append(dur(tr[j]) if j < len(tr) and "gemm_mxfp8" in tr[j]["Kernel_Name"] and re.search(r"kernel<4", tr[j]["Kernel_Name"]) else 0.0)
        call = sum(dm) / len(dm) + sum(tails) / len(tails)
        with open(f"profiles/{prefix}_gateup_gemm_launches.csv", "w") as o:
            o.write("# gate/up GEMM call of the --fp8 run = gemm_mxfp8 whole-tile launch (largest grid) + the 256x128 re-cut launch of its partial last round; "
                    "us, from the kernel trace; peak = 5000 TFLOP/s (dense fp8 MFMA)\n")
            o.write(f"main_launches,{len(dm)},avg_us,{sum(dm) / len(dm):.2f},min_us,{min(dm):.2f},max_us,{max(dm):.2f}\n")
            o.write(f"tail_launches,{len(tails)},avg_us,{sum(tails) / len(tails):.2f}\n")
            o.write(f"call_avg_us,{call:.2f},tflops,{2.0 * M * 2 * FF * E / call / 1e6:.1f},frac_of_5000,{2.0 * M * 2 * FF * E / call / 1e6 / 5000.0:.4f}\n")
        print(open(f"profiles/{prefix}_gateup_gemm_launches.csv").read())
        return
    w4 = [r for r in tr if re.search(r"gemm_w4_kernel<8, 0, 8, false, false>", r["Kernel_Name"])]
    if not w4:
        return
    key = "Grid_Size_X" if "Grid_Size_X" in w4[0] else "Grid_Size"
    gmax = max(int(r[key]) for r in w4)
    idx = {id(r): i for i, r in enumerate(tr)}
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    main_l = [r for r in w4 if int(r[key]) == gmax]
    dm, tails = [dur(r) for r in main_l], []
    for r in main_l:
        t_us, j = 0.0, idx[id(r)] + 1
        while j < len(tr) and ("gemm_w4_kernel<8, 0, 8, false, true>" in tr[j]["Kernel_Name"] or "gemm_splitk_reduce" in tr[j]["Kernel_Name"]
                               or "gemm_w4_kernel<4" in tr[j]["Kernel_Name"]):
            t_us += dur(tr[j]); j += 1
        tails.append(t_us)
    call = sum(dm) / len(dm) + sum(tails) / len(tails)
    with open(f"profiles/{prefix}_gateup_gemm_launches.csv", "w") as o:
        o.write("# gate/up GEMM call = gemm_w4_kernel<8,0,8,false,false> main launch (largest grid) + the split-K launches of its partial last wave "
                "(gemm_w4_kernel<8,0,8,false,true> + gemm_splitk_reduce_kernel); us, from the kernel trace\n")
        o.write(f"main_launches,{len(dm)},avg_us,{sum(dm) / len(dm):.2f},min_us,{min(dm):.2f},max_us,{max(dm):.2f}\n")
        o.write(f"tail_launches,{len(tails)},avg_us,{sum(tails) / len(tails):.2f}\n")
        o.write(f"call_avg_us,{call:.2f},tflops,{2.0 * M * 2 * FF * E / call / 1e6:.1f},frac_of_2500,{2.0 * M * 2 * FF * E / call / 1e6 / PEAK_TF:.4f}\n")
    print(open(f"profiles/{prefix}_gateup_gemm_launches.csv").read())


def sq(d, out):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        a = acc.setdefault(k, {})
        c = a.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
    cols = ["SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
    with open(out, "w") as o:
        o.write("# rocprofv3 --pmc " + " ".join(cols) + " on `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline` (default LoRA workload); per-kernel means over dispatches\n")
        o.write("# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / 4 / 8 (SQ_BUSY_CYCLES accumulates per shader engine, 32 on the chip with 32 SIMDs each; MFMA busy per SIMD): matrix-pipe busy share of SIMD cycles\n")
        o.write("# lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; wait_inst_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; wait_any_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES\n")
        o.write("kernel,dispatches," + ",".join(cols) + ",mfma_busy_frac,lds_conflict_frac,wait_inst_frac,wait_any_frac\n")
        rows = []
        for k, a in acc.items():
            m = {c: (a[c][0] / a[c][1] if c in a else 0.0) for c in cols}
            n = max(v[1] for v in a.values())
            busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"] / 32 if m["SQ_BUSY_CYCLES"] else 0
            conf = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if m["SQ_LDS_IDX_ACTIVE"] else 0
            wi = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else 0
            wa = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else 0
            rows.append((m["SQ_BUSY_CYCLES"] * n, f"\"{k}\",{n}," + ",".join(f"{m[c]:.0f}" for c in cols) + f",{busy:.3f},{conf:.3f},{wi:.3f},{wa:.3f}\n"))
        for _, line in sorted(rows, reverse=True)[:40]:
            o.write(line)
    print(open(out).read()[:3000])


def traffic(dfetch, dwrite, out, workload="lora"):
    """workload 'lora' (default bench: vlb_gemm_swiglu_save, M = 5861) or 'frozen' (--workload frozen: the SwiGLU-fused
    vlb_gemm_bf16 call; run with VLB_ROWS=9447)."""
    res = {}
    for d, c in ((dfetch, "FETCH_SIZE"), (dwrite, "WRITE_SIZE")):
        f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
        rs = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
        rs.sort(key=lambda r: int(r["Dispatch_Id"]))
        main = [r for r in rs if "gemm_w4_kernel<8, 0, 8, false, false>" in r["Kernel_Name"]]
        grid = max(int(r["Grid_Size"]) for r in main)
        pos = {r["Dispatch_Id"]: i for i, r in enumerate(rs)}
        mains, tails = [], []
        for r in main:
            if int(r["Grid_Size"]) != grid:
                continue
            mains.append(float(r["Counter_Value"]))
            t, j = 0.0, pos[r["Dispatch_Id"]] + 1
            while j < len(rs) and ("gemm_w4_kernel<8, 0, 8, false, true>" in rs[j]["Kernel_Name"] or "gemm_splitk_reduce" in rs[j]["Kernel_Name"]):
                t += float(rs[j]["Counter_Value"]); j += 1
            tails.append(t)
        res[c] = (grid, len(mains), sum(mains) / len(mains), sum(tails) / len(tails))
    fm, ft = res["FETCH_SIZE"][2], res["FETCH_SIZE"][3]
    wm, wt = res["WRITE_SIZE"][2], res["WRITE_SIZE"][3]
    total = (2 * (fm + ft) + wm + wt) * 1024
    frozen = workload == "frozen"
    if frozen:
        alg = 2.0 * (M * E + 2 * FF * E) + 2.0 * M * FF                                      # A, W, h = silu(gate)*up [M,FF]
    else:
        alg = 2.0 * (M * (E + 64) + 2 * FF * (E + 64)) + 2.0 * M * 2 * FF + 2.0 * M * FF      # A|t, W|B, saved [gate|up] [M,2FF], h = silu(gate)*up [M,FF]
    with open(out, "w") as o:
        if frozen:
            o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --workload frozen --steps 1 --warmup 1 --no-cpu-baseline` (packed rows)\n")
            o.write("# one gate/up GEMM CALL (vlb_gemm_bf16_ws, SwiGLU fused in the epilogue) = main gemm_w4_kernel launch (full waves of tiles) + its split-K tail launch + the reduce launch; KB per call, means over the calls of the run\n")
        else:
            o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline` (default LoRA workload, packed rows)\n")
            o.write("# one gate/up GEMM CALL (vlb_gemm_swiglu_save: GEMM + SwiGLU + saved pre-activations) = main gemm_w4_kernel launch (10 full waves of tiles) + its split-K tail launch + the reduce launch; KB per call, means over the calls of the run\n")
        o.write("# gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM) -> corrected bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024\n")
        o.write("counter,main_grid_threads,calls,main_mean_kb,tail_mean_kb\n")
        for c, (g, n, a, b) in res.items():
            o.write(f"{c},{g},{n},{a:.1f},{b:.1f}\n")
        what_alg = "A + W + h [M,N/2], bf16" if frozen else "A|t + W|B + saved [gate|up] [M,N] + h [M,N/2], bf16"
        o.write(f"# traffic per call = {total:.4e} bytes (main launch alone {(2 * fm + wm) * 1024:.4e}); algorithmic {alg:.4e} ({what_alg}); ratio {total / alg:.2f}\n")
    print(open(out).read())
    # the tracked shape -> bytes table bench.py reads its roofline.traffic from
    import json
    jp = "profiles/gateup_traffic.json"
    tab = json.load(open(jp)) if os.path.exists(jp) else {}
    tab[f"{M},{2 * FF},{E}"] = {"bytes": float(f"{total:.4e}"), "source": out,
                               "what": ("--workload frozen (packed rows): SwiGLU-fused vlb_gemm_bf16_ws call = main launch + split-K tail + reduce" if frozen else
                                        "configs[2] default bench (LoRA, packed rows): vlb_gemm_swiglu_save call = main launch + split-K tail + reduce")}
    json.dump(tab, open(jp, "w"), indent=1)


def traffic_fp8(dfetch, dwrite, out):
    """Same for the --fp8 run: one vlb_gemm_mxfp8 gate/up call = whole-tile launch (largest grid) + the re-cut launch behind it."""
    res = {}
    for d, c in ((dfetch, "FETCH_SIZE"), (dwrite, "WRITE_SIZE")):
        f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
        rs = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
        rs.sort(key=lambda r: int(r["Dispatch_Id"]))
        whole = [r for r in rs if "gemm_mxfp8" in r["Kernel_Name"] and re.search(r"kernel<8", r["Kernel_Name"])]
        grid = max(int(r["Grid_Size"]) for r in whole)
        pos = {r["Dispatch_Id"]: i for i, r in enumerate(rs)}
        mains, tails = [], []
        for r in whole:
            if int(r["Grid_Size"]) != grid:
                continue
            mains.append(float(r["Counter_Value"]))
            j = pos[r["Dispatch_Id"]] + 1
            tails.append(float(rs[j]["Counter_Value"]) if j < len(rs) and "gemm_mxfp8" in rs[j]["Kernel_Name"] and re.search(r"kernel<4", rs[j]["Kernel_Name"]) else 0.0)
        res[c] = (grid, len(mains), sum(mains) / len(mains), sum(tails) / len(tails))
    fm, ft = res["FETCH_SIZE"][2], res["FETCH_SIZE"][3]
    wm, wt = res["WRITE_SIZE"][2], res["WRITE_SIZE"][3]
    total = (2 * (fm + ft) + wm + wt) * 1024
    alg = (M * E + 2 * FF * E) * (1 + 1 / 32) + 2.0 * M * 2 * FF
    with open(out, "w") as o:
        o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --workload full --fp8 --steps 1 --warmup 1`\n")
        o.write("# one gate/up vlb_gemm_mxfp8 CALL = whole-tile gemm_mxfp8_pipe_kernel launch (largest grid) + the re-cut half-tile launch of its partial last round; KB per call, means over the calls of the run\n")
        o.write("# gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM) -> corrected bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024\n")
        o.write("counter,main_grid_threads,calls,main_mean_kb,tail_mean_kb\n")
        for c, (g, n, a, b) in res.items():
            o.write(f"{c},{g},{n},{a:.1f},{b:.1f}\n")
        o.write(f"# traffic per call = {total:.4e} bytes; algorithmic {alg:.4e} (e4m3 A + W with E8M0 scales, C [M,N] bf16); ratio {total / alg:.2f}\n")
    print(open(out).read())
    import json
    jp = "profiles/gateup_traffic.json"
    tab = json.load(open(jp)) if os.path.exists(jp) else {}
    tab[f"fp8:{M},{2 * FF},{E}"] = {"bytes": float(f"{total:.4e}"), "source": out,
                                   "what": "--workload full --fp8: vlb_gemm_mxfp8 gate/up call = whole-tile launch + re-cut halves"}
    json.dump(tab, open(jp, "w"), indent=1)


if __name__ == "__main__":
    {"stats": lambda: stats(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else ""), "sq": lambda: sq(sys.argv[2], sys.argv[3]),
     "traffic": lambda: traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "lora"),
     "traffic_fp8": lambda: traffic_fp8(sys.argv[2], sys.argv[3], sys.argv[4])}[sys.argv[1]]()
