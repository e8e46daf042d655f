"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes for the gate/up GEMM launches.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [grid_workgroups]

Units are KB (MI355X_MICROARCH.md, HBM section); on gfx950 FETCH_SIZE under-reports wide coalesced
reads by 2x, so traffic = (2*FETCH + WRITE) * 1024 bytes per launch.  The gate/up launches are
picked as the gemm_w4_kernel launches with the largest grid (the main launch of each gate/up call; its
256x128 tail launch, if any, is not included).
"""
import csv
import glob
import sys


def rows(d):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    return list(csv.DictReader(open(f)))


def pick(rs, counter):
    gemm = [r for r in rs if "gemm_w4_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    grid = max(int(r["Grid_Size"]) for r in gemm)
    if len(sys.argv) > 3:
        grid = int(sys.argv[3]) * 512
    vals = [float(r["Counter_Value"]) for r in gemm if int(r["Grid_Size"]) == grid]
    return grid, vals


def main():
    out = {}
    for d, c in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
        grid, v = pick(rows(d), c)
        out[c] = (grid, len(v), sum(v) / len(v), min(v), max(v))
    print("counter,grid_threads,launches,mean_kb,min_kb,max_kb")
    for c, (grid, n, mean, lo, hi) in out.items():
        print(f"{c},{grid},{n},{mean:.1f},{lo:.1f},{hi:.1f}")
    t = (2 * out["FETCH_SIZE"][2] + out["WRITE_SIZE"][2]) * 1024
    print(f"# traffic per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 = {t:.4e} bytes")


main()
