#!/bin/bash
# Full-scale parity check on a GPU box: a bench-style synthetic graph of <unitigs> unitigs (default
# 1 M, BASELINE.json configs[1]) is run through the real reference binary (oracle/_ref/PloidyFrost
# -t 1, single CPU core) and through this repository's CLI; all twelve output files must be
# byte-identical.  Prints both timings.   usage: tools/fullscale_parity.sh [unitigs] [seed] [k] [max_ins] [ploidy] [extra PloidyFrost options...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-1000000}; SEED=${2:-77}; KK=${3:-25}; INS=${4:-6}; PL=${5:-4}; shift 5 2>/dev/null || shift $#; EXTRA="$@"
W=$(mktemp -d /tmp/pf_parity.XXXXXX)
python "$ROOT/tools/make_graph.py" "$W/in" "$N" "$SEED" "$KK" "$INS" "$PL" | tail -1
mkdir -p "$W/ref" "$W/gpu"
t0=$(date +%s%N)
( cd "$W/ref" && "$ROOT/oracle/_ref/PloidyFrost" -g "$W/in/g.gfa" -d "$W/in/g_kmc" -o x -l 5 -u 1000 -t 1 $EXTRA > ref.log )
echo "reference wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
grep -E "findSuperBubble\(\):  Cpu time|PloidyEstimation\(\):  Cpu time|Alleles" "$W/ref/ref.log"
t0=$(date +%s%N)
( cd "$W/gpu" && "$ROOT/ploidyfrost_amd/csrc/ploidyfrost" -g "$W/in/g.gfa" -d "$W/in/g_kmc" -o x -l 5 -u 1000 -t 32 -v $EXTRA > gpu.log )
echo "ploidyfrost (MI355X) wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
grep -E "findSuperBubble\(\):  Real time|PloidyEstimation\(\):  Real time|Alleles|\[device\]|\[bfs\]" "$W/gpu/gpu.log"
bad=0
for f in "$W"/ref/PloidyFrost_output/*; do
  if cmp -s "$f" "$W/gpu/PloidyFrost_output/$(basename "$f")"; then echo "IDENTICAL $(basename "$f") $(stat -c %s "$f") bytes"; else echo "DIFFERENT $(basename "$f")"; bad=1; fi
done
rm -rf "$W"
exit $bad
