#!/bin/bash
# Refresh the measured evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [bench.py arguments, e.g. --unitigs 1000000]
# 1. bench.py as the driver runs it (CPU baseline included)  -> gpurun_out/<tag>_bench.json
# 2. the same command under rocprofv3 --kernel-trace --stats -> gpurun_out/prof_<tag>/, condensed by tools/summarize_prof.py
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only, as the pool requires)
#                                                            -> gpurun_out/pmc_{fetch,write}_<tag>/, tools/pmc_traffic.py
# The profiled invocations pass --no-cpu-baseline: the reference binary must not run as a child of the profiler.
set -e
TAG=${1:-r02}
shift || true
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
fail=0
python bench.py "$@" 2>"$O/${TAG}_bench.log" | tail -1 > "$O/${TAG}_bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$TAG" -- python3 bench.py --gen-in-process --no-cpu-baseline --steps 5 --warmup 2 "$@" > "$O/${TAG}_bench_under_rocprof.json" 2>"$O/prof_${TAG}_bench.log" || { echo "WARNING: rocprofv3 --stats run failed (see prof_${TAG}_bench.log)"; fail=1; }
python tools/summarize_prof.py "$O/prof_$TAG" "$O/${TAG}_kernel_trace" 5 > "$O/${TAG}_kernel_trace_summary.txt" 2>&1 || { echo "WARNING: no kernel trace summary"; fail=1; }
head -24 "$O/${TAG}_kernel_trace_summary.txt"; grep "device busy" "$O/${TAG}_kernel_trace_summary.txt" || true
[ -s "$O/${TAG}_kernel_trace_kernels.csv" ] || { echo "WARNING: empty kernel trace summary"; fail=1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$TAG" -- python3 bench.py --gen-in-process --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>"$O/pmc_fetch_$TAG.log" || { echo "WARNING: FETCH_SIZE pass failed"; fail=1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_$TAG" -- python3 bench.py --gen-in-process --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>"$O/pmc_write_$TAG.log" || { echo "WARNING: WRITE_SIZE pass failed"; fail=1; }
python tools/pmc_traffic.py "$O/pmc_fetch_$TAG" "$O/pmc_write_$TAG" "$O/pmc_traffic_$TAG.json" "$O/${TAG}_bench_under_rocprof.json" || { echo "WARNING: no PMC traffic summary"; fail=1; }
# SQ counters (their own pass): instructions issued per pipe and where the wave-cycles go -> the issue-rate roofline of bench.py
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d "$O/pmc_sq_$TAG" -- python3 bench.py --gen-in-process --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2>"$O/pmc_sq_$TAG.log" || { echo "WARNING: SQ pass failed"; fail=1; }
python tools/pmc_sq.py "$O/pmc_sq_$TAG.json" "$O/pmc_sq_$TAG" --bench "$O/${TAG}_bench_under_rocprof.json" || { echo "WARNING: no SQ summary"; fail=1; }
# keep only the condensed files: the raw per-dispatch CSVs are hundreds of MB
rm -rf "$O/prof_$TAG" "$O/pmc_fetch_$TAG" "$O/pmc_write_$TAG" "$O/pmc_sq_$TAG"
ls -la "$O" | grep "$TAG" || true
exit $fail
