#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd); cd "$R"; mkdir -p gpurun_out/$1
for cfg in "64 -" "0 -" "64 0" "0 0" "512 -" "16 -"; do set -- $cfg; export PF_JOIN_CHUNK=$1; if [ "$2" = "-" ]; then unset PF_JOIN_PRIO; else export PF_JOIN_PRIO=$2; fi
python3 bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk=$1 prio=$2', 'ms_per_step', d['ms_per_step'], 'excl', d['ms_per_step_excl_join'], 'find', d['host_phases_s_per_step']['find_total_s'], 'ploidy', d['host_phases_s_per_step']['ploidy_total_s'], 'join', d['roofline_k_cov_join']['avg_ms_each'], 'bfs_thread', d['kernels']['k_bfs_thread']['avg_ms'])
" | tee -a gpurun_out/${TAG:-r5k}/overlap.txt; done
