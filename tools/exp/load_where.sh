#!/bin/bash
# Where does the load of the 5 M-unitig graph go?  The same files (in /dev/shm, as bench.py keeps them) through the CLI and through
# the Python facade, each twice, with the host layer's load trace.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=$(mktemp -d /dev/shm/pf_lw.XXXXXX)
python "$ROOT/tools/make_graph.py" "$W/in" ${1:-5000000} 1000 | tail -1
for rep in 1 2; do
  echo "== CLI, files in /dev/shm (run $rep)"
  ( cd "$W" && PF_TRACE_LOAD=1 "$ROOT/ploidyfrost_amd/csrc/ploidyfrost" -g "$W/in/g.gfa" -d "$W/in/g_kmc" -o x -l 5 -u 1000 -t 32 2>&1 | grep -E "^\[load\]|Graph loading" )
done
echo "== Python facade (hostapi.Run), same files, torch imported and the GPU used before"
PF_TRACE_LOAD=1 python - "$W" <<'PY'
import sys, time
sys.path.insert(0, "/root/repo")
import torch
x = torch.zeros(1 << 28, device="cuda"); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
from ploidyfrost_amd import hostapi
W = sys.argv[1]
for rep in range(2):
    t = time.time()
    run = hostapi.Run(W + "/in/g.gfa", W + "/in/g_kmc", z=8, device=0)
    print("hostapi.Run: %.3f s" % (time.time() - t), flush=True)
    run.close()
PY
rm -rf "$W"
