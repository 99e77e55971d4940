#!/bin/bash
# Refresh the measured evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>         e.g. r01c
# 1. bench.py as the driver runs it                          -> gpurun_out/<tag>_bench_1M.json
# 2. the same command under rocprofv3 --kernel-trace --stats -> gpurun_out/prof_<tag>/, condensed by tools/summarize_prof.py
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only, as the pool requires)
#                                                            -> gpurun_out/pmc_{fetch,write}_<tag>/, tools/pmc_traffic.py
set -e
TAG=${1:-r01c}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
python bench.py --steps 8 --warmup 2 2>"$O/${TAG}_bench_1M.log" | tail -1 > "$O/${TAG}_bench_1M.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$TAG" -- python3 bench.py --steps 5 --warmup 2 > "$O/prof_${TAG}_bench.json" 2>"$O/prof_${TAG}_bench.log" || true
python tools/summarize_prof.py "$O/prof_$TAG" "$O/${TAG}_kernel_trace_1M" | head -20
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$TAG" -- python3 bench.py --steps 2 --warmup 1 > /dev/null 2>"$O/pmc_fetch_$TAG.log" || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_$TAG" -- python3 bench.py --steps 2 --warmup 1 > /dev/null 2>"$O/pmc_write_$TAG.log" || true
python tools/pmc_traffic.py "$O/pmc_fetch_$TAG" "$O/pmc_write_$TAG" "$O/pmc_traffic_$TAG.json" || true
# keep only the condensed files: the raw per-dispatch CSVs are hundreds of MB
rm -rf "$O/prof_$TAG" "$O/pmc_fetch_$TAG" "$O/pmc_write_$TAG"
ls -la "$O" | grep "$TAG"
