#!/bin/bash
# Timeline of a one-shot CLI run (first pass over a fresh graph): tools/first_pass_trace.sh   (GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/pf_ft.XXXX)
python $R/tools/make_graph.py $W/in 1000000 77 | tail -1
cd $W
PF_TRACE_FIND=1 PF_TRACE_PLOIDY=1 PF_TRACE_BFS=1 PF_TRACE_LOAD=1 $R/ploidyfrost_amd/csrc/ploidyfrost -g $W/in/g.gfa -d $W/in/g_kmc -o x -l 5 -u 1000 -t 32 -v > log.txt 2> trace.txt
grep -E "^\[find\]|^\[ploidy\] (coverage|scan|files|pipeline)|^\[bfs\] device|^\[load\] device" trace.txt
grep -E "Real time" log.txt
rm -rf $W
