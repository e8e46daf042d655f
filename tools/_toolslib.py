"""Import FIRST in a tuning script: points phantom_vlb_amd at libvlb_tools.so (the -DVLB_TOOLS build: kernel-variant
switches, timing-only ablations with WRONG results, superseded kernels kept for A/B), building it if needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("VLB_TOOLS_LIB") or os.path.join(ROOT, "phantom_vlb_amd", "libvlb_tools.so")      # VLB_TOOLS_LIB: another tools build (A/B across builds)
if not os.path.exists(PATH):
    subprocess.run(["make", "-C", os.path.join(ROOT, "phantom_vlb_amd", "csrc"), "-j8", "tools"], check=True)
os.environ["VLB_LIB"] = PATH
