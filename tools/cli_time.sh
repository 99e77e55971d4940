#!/bin/bash
# CLI wall time at 1 M unitigs, two runs: tools/cli_time.sh   (GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/pf_t.XXXX)
python $R/tools/make_graph.py $W/in 1000000 77 | tail -1
cd $W
for i in 1 2; do
t0=$(date +%s%N)
$R/ploidyfrost_amd/csrc/ploidyfrost -g $W/in/g.gfa -d $W/in/g_kmc -o x -l 5 -u 1000 -t 32 -v > log.txt
echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms"
grep -E "loading Real|Real time|\[device\]" log.txt
done
rm -rf $W
