#!/bin/bash
# The rocprofv3 passes behind profiles/: kernel statistics of bench.py, then one counter pass per run of
# scripts/pmc_workload.py (FETCH_SIZE, WRITE_SIZE, a set of SQ counters).  Run on the GPU box from the repository root:
#     bash scripts/profile_round.sh gpurun_out/r03b
set -e
out=$(realpath -m "$1")
root=$(pwd)
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > "$out/bench_under_profiler.json" 2> "$out/stats.err"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$root/scripts/pmc_workload.py" cfg4 1.0 2 > "$out/fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$root/scripts/pmc_workload.py" cfg4 1.0 2 > "$out/write.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d "$out/pmc_sq" -- python3 "$root/scripts/pmc_workload.py" cfg4 1.0 2 > "$out/sq.log" 2>&1
# keep what the summaries read, drop the bulky traces
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*agent_info.csv" -delete
ls -la "$out"/*/*/ | head -40
