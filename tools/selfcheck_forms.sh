#!/bin/bash
# One bench-style graph, the CLI twice: per-unitig coverages streamed from the joined SoA (default) and probed in the hash
# table at call time (PF_KCOV_SCAN=probe).  Prints sizes + MD5 of the twelve files of both runs and whether they agree.
#   usage: tools/selfcheck_forms.sh [unitigs] [seed] [k] [max_ins] [ploidy] [extra PloidyFrost options...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-1000000}; SEED=${2:-77}; KK=${3:-25}; INS=${4:-6}; PL=${5:-4}; shift 5 2>/dev/null || shift $#; EXTRA="$@"
W=$(mktemp -d /tmp/pf_forms.XXXXXX)
python "$ROOT/tools/make_graph.py" "$W/in" "$N" "$SEED" "$KK" "$INS" "$PL" | tail -1
for form in stream probe; do
  mkdir -p "$W/$form"
  ( cd "$W/$form" && if [ $form = probe ]; then export PF_KCOV_SCAN=probe; fi; "$ROOT/ploidyfrost_amd/csrc/ploidyfrost" -g "$W/in/g.gfa" -d "$W/in/g_kmc" -o x -l 5 -u 1000 -t 32 -v $EXTRA > run.log )
  echo "== $form"
  grep -E "findSuperBubble\(\):  Real time|PloidyEstimation\(\):  Real time|Alleles|\[device\]" "$W/$form/run.log"
  ( cd "$W/$form/PloidyFrost_output" && for f in x_*.txt; do echo "$(md5sum < $f | cut -d' ' -f1) $(stat -c %s $f) $f"; done ) | tee "$W/$form.md5"
done
if cmp -s "$W/stream.md5" "$W/probe.md5"; then echo "FORMS_IDENTICAL"; rc=0; else echo "FORMS_DIFFER"; rc=1; fi
rm -rf "$W"
exit $rc
