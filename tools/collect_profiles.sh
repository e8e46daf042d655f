#!/bin/bash
# Re-collects every file of a round (ROUND=r04 by default) under profiles/ on a one-GPU MI355X box (run from the repo root: `gpurun -- 'bash tools/collect_profiles.sh'`).
# Kernel traces and PMC passes are separate rocprofv3 runs (counters never share a run with a trace); raw traces are deleted, the
# summaries written by tools/profile_tables.py land in profiles/ and are copied to gpurun_out/pf/out/ so that they travel back.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${ROUND:-r04}
PART=${PART:-all}      # a: LoRA + frozen workloads, b: full fine-tune (bf16, fp8) + HBM-bound kernel bench, all: both (needs ~20 min)
rm -rf gpurun_out/pf && mkdir -p gpurun_out/pf
if [ "$PART" != "b" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/lora -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/pf/lora.log 2>&1
python3 tools/profile_tables.py stats gpurun_out/pf/lora 8 ${R}_bench_lora7b "default command: the next step's vision side runs on a side stream under the backward pass, so kernel durations overlap (their sum exceeds the step time) and overlapped kernels read slower than alone" > gpurun_out/pf/lora_tables.log 2>&1
# the same step with everything on one stream: clean per-kernel durations for the kernel split
VLB_BENCH_VISION_PREFETCH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/lora_serial -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/pf/lora_serial.log 2>&1
python3 tools/profile_tables.py stats gpurun_out/pf/lora_serial 8 ${R}_bench_lora7b_serial "VLB_BENCH_VISION_PREFETCH=0: one stream, no overlap" > gpurun_out/pf/lora_serial_tables.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pf/sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pf/sq.log 2>&1
python3 tools/profile_tables.py sq gpurun_out/pf/sq profiles/${R}_lora_step_sq_counters.csv > gpurun_out/pf/sq_tables.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pf/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pf/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pf/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pf/write.log 2>&1
python3 tools/profile_tables.py traffic gpurun_out/pf/fetch gpurun_out/pf/write profiles/${R}_gemm_gateup_hbm_traffic_lora.csv > gpurun_out/pf/traffic_tables.log 2>&1
# the frozen workload's gate/up call (the round-1 collection bench.py used to quote predates the tile order and the compact epilogues)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pf/fetchf -- python3 bench.py --workload frozen --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pf/fetchf.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pf/writef -- python3 bench.py --workload frozen --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pf/writef.log 2>&1
VLB_ROWS=9447 VLB_CLIPS=5 python3 tools/profile_tables.py traffic gpurun_out/pf/fetchf gpurun_out/pf/writef profiles/${R}_gemm_gateup_hbm_traffic_frozen.csv frozen > gpurun_out/pf/trafficf_tables.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/frozen -- python3 bench.py --workload frozen --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/pf/frozen.log 2>&1
VLB_ROWS=9447 VLB_CLIPS=5 python3 tools/profile_tables.py stats gpurun_out/pf/frozen 8 ${R}_bench_frozen7b > gpurun_out/pf/frozen_tables.log 2>&1
fi
if [ "$PART" != "a" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/full8 -- python3 bench.py --workload full --fp8 --steps 4 --warmup 2 > gpurun_out/pf/full8.log 2>&1
python3 tools/profile_tables.py stats gpurun_out/pf/full8 4 ${R}_bench_full7b_fp8 > gpurun_out/pf/full8_tables.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/full -- python3 bench.py --workload full --steps 4 --warmup 2 > gpurun_out/pf/full.log 2>&1
python3 tools/profile_tables.py stats gpurun_out/pf/full 4 ${R}_bench_full7b > gpurun_out/pf/full_tables.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pf/fetch8 -- python3 bench.py --workload full --fp8 --steps 1 --warmup 1 > gpurun_out/pf/fetch8.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pf/write8 -- python3 bench.py --workload full --fp8 --steps 1 --warmup 1 > gpurun_out/pf/write8.log 2>&1
python3 tools/profile_tables.py traffic_fp8 gpurun_out/pf/fetch8 gpurun_out/pf/write8 profiles/${R}_gemm_gateup_hbm_traffic_fp8.csv > gpurun_out/pf/traffic8_tables.log 2>&1
python3 tools/bench_hbm_kernels.py > profiles/${R}_hbm_bound_kernels.txt 2> gpurun_out/pf/hbm.err
fi
mkdir -p gpurun_out/pf/out && cp profiles/${R}_* profiles/gateup_traffic.json gpurun_out/pf/out/
for f in gpurun_out/pf/*_tables.log; do echo "== $f"; tail -n 4 $f; done
grep -h '"metric"' gpurun_out/pf/*.log || true
find gpurun_out/pf -name "*kernel_trace.csv" -delete; find gpurun_out/pf -name "*counter_collection.csv" -delete; find gpurun_out/pf -name "*.db" -delete
