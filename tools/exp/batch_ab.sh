#!/bin/bash
# text pieces of 4 x PF_BATCH_BUBBLES bubbles: K-TEXT's launches fill the device only from a certain size on
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for b in 32768 49152 65536 98304 131072; do
  PF_BATCH_BUBBLES=$b python bench.py --no-cpu --steps 10 > gpurun_out/batch.json 2> gpurun_out/batch.log
  python3 - $b <<'PY'
import json, sys
d = json.loads(open('gpurun_out/batch.json').read().strip().splitlines()[-1])
hp = d['host_phases_s_per_step']; kf = d['kernels']['k_call_format']
print('batch', sys.argv[1].ljust(7), 'step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'excl', d['ms_per_step_excl_join'], 'ploidy', hp['ploidy_total_s'], 'align', hp['align_s'], 'format', kf['avg_ms'], 'x', kf['launches_per_step'], 'ok', d['output_check'].get('identical_to_reference'))
PY
done
