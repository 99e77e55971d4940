#!/bin/bash
# BASELINE.json configs[4] (10 M unitigs, k = 31, -z 16, insertions up to 50 bp) on a GPU box: the CLI's timings, the MD5 of its
# twelve files (to hold against the digests of the build whose files were compared with the reference's, profiles/history/
# r01p / r04s), and the kernel table of the same run under rocprofv3.
#   usage (through gpurun, from the repo root): tools/config4_profile.sh <tag> [unitigs]
set -e
TAG=${1:-r08}
N=${2:-10000000}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
export TMPDIR=/tmp
W=$(mktemp -d /tmp/pf_c4.XXXXXX)
python "$R/tools/make_graph.py" "$W/in" "$N" 7 31 50 4 | tail -1   # (seed 7: the graph of profiles/history/r01_fullscale_parity.txt, whose files were the reference's)
mkdir -p "$W/gpu" "$W/prof"
cd "$W/gpu"
for rep in 1 2; do
  t0=$(date +%s%N)
  "$R/ploidyfrost_amd/csrc/ploidyfrost" -g "$W/in/g.gfa" -d "$W/in/g_kmc" -o x -l 5 -u 1000 -t 32 -z 16 -v > gpu.log
  echo "cli wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
  grep -E "findSuperBubble\(\):  Real time|PloidyEstimation\(\):  Real time|\[device\]" gpu.log
done
( cd PloidyFrost_output && md5sum x_*.txt )
cd "$W/prof"
export PF_ORDERLY_EXIT=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$W/prof/out" -- "$R/ploidyfrost_amd/csrc/ploidyfrost" -g "$W/in/g.gfa" -d "$W/in/g_kmc" -o x -l 5 -u 1000 -t 32 -z 16 > prof.log 2>&1 || echo "WARNING: rocprofv3 run failed"
python "$R/tools/summarize_prof.py" "$W/prof/out" "$O/${TAG}_config4_kernel_trace" 1 2>&1 | head -30
rm -rf "$W"
