#!/bin/bash
# Where the CLI's wall time goes outside the two timed phases: tools/load_trace.sh [unitigs]   (GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-1000000}
W=$(mktemp -d /tmp/pf_lt.XXXX)
python $R/tools/make_graph.py $W/in $N 77 | tail -1
cd $W
for i in 1 2; do
t0=$(date +%s%N)
PF_TRACE_FIND=1 PF_TRACE_PLOIDY=1 PF_TRACE_LOAD=1 $R/ploidyfrost_amd/csrc/ploidyfrost -g $W/in/g.gfa -d $W/in/g_kmc -o x -l 5 -u 1000 -t 32 -v > log.txt 2> trace.txt
echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
cat trace.txt | grep -E "^\[(load|main|find|ploidy)\]"
grep -E "loading Real|Real time" log.txt
done
rm -rf $W
