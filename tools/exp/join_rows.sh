#!/bin/bash
R=$(cd "$(dirname "$0")/../.." && pwd); cd "$R"; mkdir -p gpurun_out/$1
for pl in 4 2; do PF_TAB_PER_LINE=$pl python3 tools/exp/join_exp.py ${2:-1000000} 5 2>&1 | grep -E "k_cov_join|load|Error|error" | sed "s/^/per_line<=$pl /" | tee -a gpurun_out/$1/rows.txt; done
