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


def audit(asm_path):
    lines = open(asm_path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    report, bad = [], []
    for i, name in starts:
        if "gemm_w4_kernel" not in name:
            continue
        body = lines[i:min(x for x in ends if x > i)]
        tmpl = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)E", name)
        nt, _, mt, masked, splitk = (int(x) for x in tmpl.groups())
        loops = list(k_loops(body))
        want_mfma = 2 * mt * nt                         # two k-steps of MT x NT fragments per K-tile
        for seg in loops:
            n_mfma = sum("v_mfma" in l for l in seg)
            n_scr = sum("scratch_" in l for l in seg)
            n_acc = sum("v_accvgpr" in l for l in seg)
            n_dma = sum("global_load_lds" in l for l in seg)
            n_rd = sum("ds_read_b128" in l for l in seg)
            report.append(f"gemm_w4_kernel<NT={nt}, MT={mt}, masked={masked}, splitk={splitk}>: K loop {n_mfma} MFMA, {n_dma} LDS-DMA, "
                          f"{n_rd} ds_read_b128, {n_scr} scratch, {n_acc} v_accvgpr")
            if n_mfma != want_mfma or n_scr or n_acc or n_dma != 2 * (mt + nt) // 2 or n_rd != 2 * (mt + nt):
                bad.append(report[-1])
        if not loops and not splitk:
            bad.append(f"gemm_w4_kernel<NT={nt}, MT={mt}, masked={masked}>: no K loop found")
    return report, bad


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gemm.s")
        subprocess.run([HIPCC] + FLAGS + ["-S", "--cuda-device-only",
                        os.path.join(ROOT, "phantom_vlb_amd", "csrc", "gemm.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        report, bad = audit(out)
    print("\n".join(report))
    if bad or not report:
        print("VIOLATIONS:\n" + "\n".join(bad or ["no four-wave kernel found"]))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
