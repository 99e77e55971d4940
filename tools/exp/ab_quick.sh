#!/bin/bash
# one library, three workloads: step and the kernels named on the command line.  usage: tools/exp/ab_quick.sh k_call_sites k_call_paths ...
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for wl in "--workload stress" "--workload stress --unitigs 1000000" ""; do
  python bench.py $wl --no-cpu --steps 8 > gpurun_out/abq.json 2> gpurun_out/abq.log
  python3 - "$wl" "$@" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/abq.json').read().strip().splitlines()[-1])
ks = " ".join("%s %.3f x%g" % (k, d['kernels'][k]['avg_ms'], d['kernels'][k]['launches_per_step']) for k in sys.argv[2:] if k in d['kernels'])
print((sys.argv[1] or 'default').ljust(40), 'step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'excl', d['ms_per_step_excl_join'], 'align_s', d['host_phases_s_per_step']['align_s'], '|', ks, '| ok', d['output_check'].get('identical_to_reference'))
PY
done
