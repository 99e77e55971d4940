#!/bin/bash
# K-BUBBLE's register budget: wavefronts a SIMD by launch bounds (2 = 212 registers, 3 = 168 + spills, 4 = 128 + spills).  usage: tools/exp/bub_waves.sh
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for w in 2 3 4; do
  (cd ploidyfrost_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I../../include -c -o pf_bubble.o <(sed "s|__launch_bounds__(64, 3) void k_bubble|__launch_bounds__(64, $w) void k_bubble|" pf_bubble.hip) -x hip && make -s libploidyfrost_hip.so libploidyfrost_host.so > /dev/null 2>&1)
  for wl in "--workload stress --unitigs 1000000" ""; do
    python bench.py $wl --no-cpu --steps 10 > gpurun_out/bw_$w.json 2> gpurun_out/bw_$w.log
    python3 -c "
import json
d=json.loads(open('gpurun_out/bw_$w.json').read().strip().splitlines()[-1])
kb=d['kernels']['k_bubble']
print('waves/SIMD $w', '$wl' or 'default', 'step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'excl', d['ms_per_step_excl_join'], 'align_s', d['host_phases_s_per_step']['align_s'], 'k_bubble avg', kb['avg_ms'], 'union', kb.get('union_ms_per_step'), 'ok', d['output_check'].get('identical_to_reference'))"
  done
done
