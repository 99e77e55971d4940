#!/bin/bash
# the CLI's load at 5 M unitigs, step by step (PF_TRACE_LOAD): usage tools/exp/load_trace.sh [unitigs]
R=$(cd "$(dirname "$0")/../.." && pwd); N=${1:-5000000}
W=$(mktemp -d /tmp/pf_t.XXXX)
python3 $R/tools/make_graph.py $W/in $N 1000 | tail -1
cd $W
for i in 1 2; do
t0=$(date +%s%N)
env PF_TRACE_LOAD=1 $EXTRA_ENV $R/ploidyfrost_amd/csrc/ploidyfrost -g $W/in/g.gfa -d $W/in/g_kmc -o x -l 5 -u 1000 -t 32 > log.txt 2> err.txt
echo "== run $i: wall $(( ($(date +%s%N) - t0) / 1000000 )) ms; $(grep -E 'findSuperBubble\(\):  Real|PloidyEstimation\(\):  Real' log.txt | tr '\n' ' ')"
grep -i "load\|trace\|\[" err.txt | head -60
done
cd $R; rm -rf $W
