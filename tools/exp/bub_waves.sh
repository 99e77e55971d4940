#!/bin/bash
# K-BUBBLE's register budget: wavefronts a SIMD by launch bounds (2 = 212 registers, 3 = 168 + 45 spilled, 4 = 128 + 112 spilled) on
# two workloads.  usage (GPU box): tools/exp/bub_waves.sh   -- rebuilds the library three times; leaves the tree's own build behind
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R
for w in 2 3 4; do
  sed "s|__launch_bounds__(64, 3) void k_bubble|__launch_bounds__(64, $w) void k_bubble|" ploidyfrost_amd/csrc/pf_bubble.hip > ploidyfrost_amd/csrc/pf_bubble_w.hip
  (cd ploidyfrost_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -I../../include -c -o pf_bubble.o pf_bubble_w.hip && rm pf_bubble_w.hip && make -s libploidyfrost_hip.so libploidyfrost_host.so > /dev/null 2>&1)
  for wl in "--workload stress --unitigs 1000000" ""; do
    python bench.py $wl --no-cpu --steps 10 > gpurun_out/bw_$w.json 2> gpurun_out/bw_$w.log
    python3 - "$w" "$wl" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/bw_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
kb = d['kernels']['k_bubble']
print('waves/SIMD', sys.argv[1], (sys.argv[2] or 'default').ljust(38), 'step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'align_s', d['host_phases_s_per_step']['align_s'],
      'k_bubble avg', kb['avg_ms'], 'union', kb.get('union_ms_per_step'), 'ok', d['output_check'].get('identical_to_reference'))
PY
  done
done
(cd ploidyfrost_amd/csrc && touch pf_bubble.hip && make -s > /dev/null 2>&1)
