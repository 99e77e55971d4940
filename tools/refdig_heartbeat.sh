#!/bin/bash
# tools/reference_digests.py with a line of progress a minute (gpurun takes seven silent minutes for a hang; the reference says
# nothing for ten).  usage: tools/refdig_heartbeat.sh <out.json> <unitigs> [reference_digests.py options]
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
python3 tools/reference_digests.py "$@" > "${1%.json}.log" 2>&1 &
pid=$!
t0=$(date +%s)
while kill -0 $pid 2>/dev/null; do sleep 30; echo "[refdig] $(( $(date +%s) - t0 )) s: $(tail -1 "${1%.json}.log" | cut -c1-160)"; done
wait $pid; rc=$?
tail -3 "${1%.json}.log" | cut -c1-400
exit $rc
