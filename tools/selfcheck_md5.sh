#!/bin/bash
# MD5 of the twelve output files of this repository's CLI on a bench-style synthetic graph (deterministic for a given
# size / seed / k / max_ins / ploidy).  Used to check a device-side change against the digests of a build whose outputs
# were compared with the reference's by tools/fullscale_parity.sh.
#   usage: tools/selfcheck_md5.sh [unitigs] [seed] [k] [max_ins] [ploidy] [extra PloidyFrost options...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-1000000}; SEED=${2:-77}; KK=${3:-25}; INS=${4:-6}; PL=${5:-4}; shift 5 2>/dev/null || shift $#; EXTRA="$@"
W=$(mktemp -d /tmp/pf_md5.XXXXXX)
python "$ROOT/tools/make_graph.py" "$W/in" "$N" "$SEED" "$KK" "$INS" "$PL" | tail -1
mkdir -p "$W/gpu"
for rep in $(seq 1 ${PF_REPEAT:-1}); do   # PF_REPEAT=n: n runs on the same inputs (timing spread)
( cd "$W/gpu" && "$ROOT/ploidyfrost_amd/csrc/ploidyfrost" -g "$W/in/g.gfa" -d "$W/in/g_kmc" -o x -l 5 -u 1000 -t 32 -v $EXTRA > gpu.log )
grep -E "findSuperBubble\(\):  Real time|PloidyEstimation\(\):  Real time|\[device\]|\[bfs\]" "$W/gpu/gpu.log"
done
( cd "$W/gpu/PloidyFrost_output" && md5sum x_*.txt )
rm -rf "$W"
