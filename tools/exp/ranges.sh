#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd); cd "$R"; mkdir -p gpurun_out/$1
for cfg in "2 2" "3 2" "4 2" "6 2" "2 1" "3 1" "4 1" "6 1" "4 3" "8 2"; do set -- $cfg; 
PF_ALIGN_RANGES=$1 PF_ALIGN_THREADS=$2 python3 bench.py --steps 20 --no-cpu-baseline ${EXTRA} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ranges=$1 threads=$2', 'incl', d['ms_per_step'], 'excl', d['ms_per_step_excl_join'], 'median', d['ms_per_step_excl_join_dist']['median'], 'align', d['host_phases_s_per_step']['align_s'], 'ploidy', d['host_phases_s_per_step']['ploidy_total_s'], d['output_check']['identical_to_reference'])
" | tee -a gpurun_out/${TAG:-r5}/ranges.txt; done
