set -e
R=$GRAFT_REPO_ROOT
W=$(mktemp -d /tmp/pf_ft.XXXX)
python $R/tools/make_graph.py $W/in 5000000 1000 | tail -1
cd $W
PF_TRACE_FIND=1 PF_TRACE_PLOIDY=1 PF_TRACE_ALIGN=1 PF_TRACE_LOAD=1 $R/ploidyfrost_amd/csrc/ploidyfrost -g $W/in/g.gfa -d $W/in/g_kmc -o x -l 5 -u 1000 -t 32 -v > log.txt 2> trace.txt
cat trace.txt
grep -E "Real time" log.txt
rm -rf $W
