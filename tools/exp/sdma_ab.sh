#!/bin/bash
# does the pass depend on how device-to-host copies are made (SDMA engines or blit kernels)?
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for v in default 1 0; do
  if [ $v = default ]; then unset HSA_ENABLE_SDMA; else export HSA_ENABLE_SDMA=$v; fi
  python bench.py --no-cpu --steps 10 > gpurun_out/sdma_$v.json 2> gpurun_out/sdma_$v.log
  python3 - $v <<'PY'
import json, sys
d = json.loads(open('gpurun_out/sdma_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
hp = d['host_phases_s_per_step']
print('HSA_ENABLE_SDMA', sys.argv[1].ljust(8), 'step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'excl', d['ms_per_step_excl_join'], 'find', hp['find_total_s'], 'ploidy', hp['ploidy_total_s'], 'scan', hp['scan_s'], 'align', hp['align_s'],
      'copy', d['kernels'].get('copy_text_to_host', {}).get('avg_ms'))
PY
done
