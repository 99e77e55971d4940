#!/bin/bash
# Where a one-shot CLI run at BASELINE.json configs[4] (10 M unitigs, k = 31, -z 16) spends its wall time: [load] / [main] lines of two runs
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
W=$(mktemp -d /tmp/pf_c4t.XXXX)
python $R/tools/make_graph.py $W/in 10000000 7 31 50 4 | tail -1
ls -la $W/in | head
cd $W
for rep in 1 2; do
  t0=$(date +%s%N)
  PF_TRACE_LOAD=1 $R/ploidyfrost_amd/csrc/ploidyfrost -g $W/in/g.gfa -d $W/in/g_kmc -o x -l 5 -u 1000 -t 32 -z 16 -v > log.txt 2> trace.txt
  echo "cli wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
  grep -E "^\[load\]|^\[main\]|unitig ids" trace.txt
  grep -E "Real time" log.txt
done
rm -rf $W
