#!/bin/bash
# K-COV-JOIN under rocprofv3: kernel trace + PMC passes (each its own run, as the pool requires).  usage: tools/exp/join_prof.sh <tag> [unitigs]
TAG=${1:-join}; N=${2:-1000000}
R=$(cd "$(dirname "$0")/../.." && pwd); O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"; export TMPDIR=/tmp
python3 tools/exp/join_exp.py $N 5 > "$O/plain.txt" 2>&1; tail -2 "$O/plain.txt"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  d="$O/pmc_$(echo $set | cut -d' ' -f1)"
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$d" -- python3 tools/exp/join_exp.py $N 3 > "$d.log" 2>&1 || echo "pass $set failed"
  python3 - "$d" <<'PY' >> "$O/counters.txt"
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_cov_join" in n or "k_table_build" in n:
            a = acc[n.split("(")[0].split("<")[0].split(" ")[-1]][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    for c, (v, n) in cs.items():
        print("%-16s %-24s %16.1f per launch (%d launches)" % (k, c, v / n, n))
PY
  rm -rf "$d"
done
cat "$O/counters.txt"
