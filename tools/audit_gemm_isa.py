"""Static check of the generated gfx950 ISA of the four-wave GEMM kernels (csrc/gemm.hip).

The kernels issue their MFMAs from inline asm with the accumulators tied to AGPRs; two things silently ruin them and
both are invisible in the source: (1) the accumulator array not being promoted to registers (any loop over acc[][] that
fails to unroll, or too many epilogue variants, and the compiler keeps it in scratch memory - one scratch store behind
every MFMA), (2) accumulator tuples being copied around behind the asm statements.  This compiles gemm.hip to assembly
(device only, ~40 s) and checks every K loop: MFMA count, no scratch access, no v_accvgpr_* copy, the expected number
of LDS-DMA pieces and fragment reads.

    python tools/audit_gemm_isa.py            # prints one line per kernel, exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-function", "-mllvm", "-pragma-unroll-threshold=100000"]      # = csrc/Makefile (FLAGS + GEMM_FLAGS)


def k_loops(body):
    labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    for k, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            seg = body[labels[m.group(1)]:k + 1]
            if any("v_mfma" in x for x in seg) and any("global_load_lds" in x for x in seg):      # the K loop (epilogues hold MFMAs too)
                yield seg


def _vregs(text):
    """VGPR indices an instruction line names (v7, v[8:11])."""
    out = []
    for lo, hi in re.findall(r"\bv\[?(\d+)(?::(\d+))?\]?", text.split(";")[0]):
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def audit_wd(body):
    """gemm_wd_kernel: the W double buffer is hand-assigned to v[192:255] (loads and MFMAs name the registers literally).
    Checked on the generated ISA: (1) the hot loop (two K-tiles per iteration) holds exactly 256 MFMA, 16 LDS-DMA pieces, 16
    W loads, 64 ds_read_b128, 2 barriers, 4 x lgkmcnt(0) + 4 x counted vmcnt(8) and nothing else that waits, no scratch access,
    no v_accvgpr copy; (2) between the first W load and the last MFMA of the kernel NO compiler-issued instruction (outside
    the asm statements) names a register >= v192; (3) every W load writes, and every MFMA reads its W operand from, that range,
    and inside the loop the MFMAs of a K-tile read only the buffer that the PREVIOUS tile's loads filled."""
    report, bad = [], []
    hdr = [k for k, l in enumerate(body) if "Inner Loop Header" in l]
    if not hdr:
        return ["gemm_wd_kernel: no K loop found"], ["gemm_wd_kernel: no K loop found"]
    h = hdr[0]
    label = body[h].split(":")[0]
    back = [k for k, l in enumerate(body) if k > h and re.search(r"s_c?branch\w* " + re.escape(label) + r"\b", l)]
    loop = body[h:back[0] + 1]
    cnt = lambda seg, pat: sum(bool(re.search(pat, x.split(";")[0])) for x in seg)
    waits = [x.strip() for x in loop if "s_waitcnt" in x]
    got = dict(mfma=cnt(loop, r"v_mfma"), dma=cnt(loop, r"global_load_lds"), wload=cnt(loop, r"global_load_dwordx4 v\["),
               rd=cnt(loop, r"ds_read_b128"), scratch=cnt(loop, r"scratch_"), acc=cnt(loop, r"v_accvgpr"), barrier=cnt(loop, r"s_barrier"))
    want = dict(mfma=256, dma=16, wload=16, rd=64, scratch=0, acc=0, barrier=2)
    line = "gemm_wd_kernel: K loop (2 tiles) " + ", ".join(f"{v} {k}" for k, v in got.items()) + f", waits {sorted(set(waits))}"
    report.append(line)
    if got != want or sorted(waits) != sorted(["s_waitcnt lgkmcnt(0)"] * 4 + ["s_waitcnt vmcnt(8)"] * 4):
        bad.append(line)
    first = min(k for k, l in enumerate(body) if re.search(r"global_load_dwordx4 v\[", l))
    last = max(k for k, l in enumerate(body) if "v_mfma" in l)
    inasm, viol = False, []
    for x in body[first - 1:last + 1]:
        if "ASMSTART" in x:
            inasm = True
        elif "ASMEND" in x:
            inasm = False
        elif not inasm and not x.strip().startswith(";") and any(r >= 192 for r in _vregs(x)):
            viol.append(x.strip())
    for x in body[first:last + 1]:
        m = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", x)
        if m and not (192 <= int(m.group(1)) and int(m.group(2)) <= 255):
            viol.append("W load outside v[192:255]: " + x.strip())
        m = re.search(r"v_mfma_f32_16x16x32_bf16 a\[\d+:\d+\], v\[(\d+):(\d+)\]", x)
        if m and not (192 <= int(m.group(1)) and int(m.group(2)) <= 255):
            viol.append("MFMA W operand outside v[192:255]: " + x.strip())
    # buffer discipline inside the loop: tile of parity P reads v[192+32P, +32) and loads v[192+32(1-P), +32)
    seq = []
    for x in loop:
        m = re.search(r"global_load_dwordx4 v\[(\d+):", x)
        if m:
            seq.append(("L", (int(m.group(1)) - 192) // 32))
        m = re.search(r"v_mfma_f32_16x16x32_bf16 a\[\d+:\d+\], v\[(\d+):", x)
        if m:
            seq.append(("M", (int(m.group(1)) - 192) // 32))
    mf = [b for k, b in seq if k == "M"]
    half = len(mf) // 2
    ok = mf[:half] == [0] * half and mf[half:] == [1] * half
    pos = [i for i, (k, _) in enumerate(seq) if k == "M"]
    for i, (k, b) in enumerate(seq):
        if k == "L":
            tile_par = 0 if i < pos[half - 1] else 1          # a tile's first W load is issued in front of its first MFMA
            ok = ok and b == 1 - tile_par
    report.append(f"gemm_wd_kernel: {len(viol)} compiler accesses to v[192:255] in the K region; W buffer discipline {'ok' if ok else 'BROKEN'}")
    if viol or not ok:
        bad.append(report[-1] + " " + "; ".join(viol[:4]))
    return report, bad


def audit(asm_path):
    lines = open(asm_path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    report, bad = [], []
    for i, name in starts:
        if "gemm_wd_kernel" in name:
            r, b = audit_wd(lines[i:min(x for x in ends if x > i)])
            report += r
            bad += b
            continue
        if "gemm_w4_kernel" not in name:
            continue
        body = lines[i:min(x for x in ends if x > i)]
        tmpl = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)E", name)
        nt, abl, mt, masked, splitk = (int(x) for x in tmpl.groups())
        streamk = int(bool(abl & 256))               # tools build: the stream-K experiment (its segment loop wraps the K loop: only the innermost loop is the K loop)
        loops = list(k_loops(body))
        want_mfma = 2 * mt * nt                         # two k-steps of MT x NT fragments per K-tile
        for seg in loops:
            n_mfma = sum("v_mfma" in l for l in seg)
            n_scr = sum("scratch_" in l for l in seg)
            n_acc = sum("v_accvgpr" in l for l in seg)
            n_dma = sum("global_load_lds" in l for l in seg)
            n_rd = sum("ds_read_b128" in l for l in seg)
            report.append(f"gemm_w4_kernel<NT={nt}, MT={mt}, masked={masked}, splitk={splitk}, streamk={streamk}>: K loop {n_mfma} MFMA, {n_dma} LDS-DMA, "
                          f"{n_rd} ds_read_b128, {n_scr} scratch, {n_acc} v_accvgpr")
            if streamk and n_mfma > want_mfma:
                continue                             # the segment loop around the K loop (holds the peeled tiles and the epilogues)
            if n_mfma != want_mfma or n_scr or n_acc or n_dma != 2 * (mt + nt) // 2 or n_rd != 2 * (mt + nt):
                bad.append(report[-1])
        if not loops and not splitk:
            bad.append(f"gemm_w4_kernel<NT={nt}, MT={mt}, masked={masked}>: no K loop found")
    return report, bad


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gemm.s")
        # -DVLB_TOOLS: the tools build holds everything the product build holds plus the experiment kernels (gemm_wd_kernel)
        subprocess.run([HIPCC] + FLAGS + (["-DVLB_TOOLS"] if "--tools" in sys.argv else []) + ["-S", "--cuda-device-only",
                        os.path.join(ROOT, "phantom_vlb_amd", "csrc", "gemm.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        report, bad = audit(out)
    print("\n".join(report))
    if bad or not report:
        print("VIOLATIONS:\n" + "\n".join(bad or ["no four-wave kernel found"]))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
