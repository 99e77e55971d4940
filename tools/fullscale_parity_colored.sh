#!/bin/bash
# Full-scale parity check of the colored path on a GPU box (BASELINE.json configs[3]): a synthetic colored graph of
# <unitigs> unitigs (default 2 M) from <samples> samples is run through the real reference binary
# (oracle/_ref/PloidyFrost -f ... -t 1, single CPU core) and through this repository's CLI; all twelve output files must
# be byte-identical.  Prints both timings.
#   usage: tools/fullscale_parity_colored.sh [unitigs] [seed] [k] [samples] [ploidy] [extra PloidyFrost options...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-2000000}; SEED=${2:-77}; KK=${3:-25}; SM=${4:-3}; PL=${5:-2}; shift 5 2>/dev/null || shift $#; EXTRA="$@"
W=$(mktemp -d /tmp/pf_cparity.XXXXXX)
python "$ROOT/tools/make_colored_graph.py" "$W/in" "$N" "$SEED" "$KK" "$SM" "$PL" | tail -1
mkdir -p "$W/ref" "$W/gpu"
t0=$(date +%s%N)
( cd "$W/ref" && "$ROOT/oracle/_ref/PloidyFrost" -g "$W/in/g.gfa" -f "$W/in/g.bfg_colors" -d "$W/in/dbs.txt" -C "$W/in/cutoffs.txt" -o x -t 1 $EXTRA | grep -v "can not find" > ref.log )
echo "reference wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
grep -E "Finding superbubbles Cpu time|PloidyEstimation\(\): Cpu time|Alleles|loading Real" "$W/ref/ref.log"
t0=$(date +%s%N)
( cd "$W/gpu" && "$ROOT/ploidyfrost_amd/csrc/ploidyfrost" -g "$W/in/g.gfa" -f "$W/in/g.bfg_colors" -d "$W/in/dbs.txt" -C "$W/in/cutoffs.txt" -o x -t 32 -v $EXTRA > gpu.log )
echo "ploidyfrost (MI355X) wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
grep -E "findSuperBubble\(\):  Real time|PloidyEstimation\(\):  Real time|Alleles|\[device\]|loading Real" "$W/gpu/gpu.log"
bad=0
for f in "$W"/ref/PloidyFrost_output/*; do
  if cmp -s "$f" "$W/gpu/PloidyFrost_output/$(basename "$f")"; then echo "IDENTICAL $(basename "$f") $(stat -c %s "$f") bytes"; else echo "DIFFERENT $(basename "$f")"; bad=1; fi
done
rm -rf "$W"
exit $bad
