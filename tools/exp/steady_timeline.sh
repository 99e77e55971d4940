#!/bin/bash
# the dispatches of one steady step of bench.py in start order, with the device's idle gaps: gpurun_out/<tag>_steady_pass_timeline.txt
TAG=${1:-r5}; R=$(cd "$(dirname "$0")/../.." && pwd); cd $R; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl_$TAG -- python3 bench.py --gen-in-process --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/${TAG}_tl_bench.json 2> gpurun_out/${TAG}_tl_bench.log
python3 tools/pass_timeline.py gpurun_out/prof_tl_$TAG 20 > gpurun_out/${TAG}_steady_pass_timeline.txt 2>&1
rm -rf gpurun_out/prof_tl_$TAG
head -5 gpurun_out/${TAG}_steady_pass_timeline.txt; tail -25 gpurun_out/${TAG}_steady_pass_timeline.txt
