#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for cfg in "2 2" "3 2" "3 3" "4 2"; do
  set -- $cfg
  PF_ALIGN_RANGES=$1 PF_ALIGN_THREADS=$2 python bench.py --no-cpu --steps 10 > gpurun_out/rng.json 2> gpurun_out/rng.log
  python3 - $1 $2 <<'PY'
import json, sys
d = json.loads(open('gpurun_out/rng.json').read().strip().splitlines()[-1])
hp = d['host_phases_s_per_step']
print('ranges', sys.argv[1], 'threads', sys.argv[2], 'step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'excl', d['ms_per_step_excl_join'], 'ploidy', hp['ploidy_total_s'], 'align', hp['align_s'], 'ok', d['output_check'].get('identical_to_reference'))
PY
done
