"""Static check of the generated gfx950 code of the four-wave GEMM kernels (no GPU needed: hipcc cross-compiles)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="hipcc not available")
def test_four_wave_gemm_k_loops_stay_in_registers(tmp_path):
    """Every K loop of gemm_w4_kernel holds exactly its 2 x MT x NT MFMAs, MT + NT LDS-DMA pieces and 2 (MT + NT) fragment
    reads, and touches neither scratch memory nor v_accvgpr_* copies (tools/audit_gemm_isa.py explains why that can break
    without any change to the loop's source).  (`python tools/audit_gemm_isa.py --tools` also audits the tools build's
    gemm_wd_kernel - hand-assigned W registers v[192:255]: counts and waits of its loop, no compiler-issued access to that range
    in the K region, W loads / MFMA W operands inside it, the double buffer's read / fill alternation.)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_gemm_isa as A
    import subprocess
    out = str(tmp_path / "gemm.s")
    mk = open(os.path.join(ROOT, "phantom_vlb_amd", "csrc", "Makefile")).read()
    assert "-pragma-unroll-threshold=100000" in mk and "-pragma-unroll-threshold=100000" in A.FLAGS      # audit what the Makefile builds
    subprocess.run([A.HIPCC if os.path.exists(A.HIPCC) else shutil.which("hipcc")] + A.FLAGS +
                   ["-S", "--cuda-device-only", os.path.join(ROOT, "phantom_vlb_amd", "csrc", "gemm.hip"), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    report, bad = A.audit(out)
    assert len(report) >= 8 and not any("gemm_wd_kernel" in r for r in report), report     # the W-direct experiment is not in the product
    assert not bad, "\n".join(bad)
