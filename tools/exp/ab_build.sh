#!/bin/bash
# A/B of compile-time variants on ONE box: for every -D flag given (or "-" for none) the device library is rebuilt there and
# tools/ab_pass.py measures the pass.   usage (through gpurun): tools/exp/ab_build.sh <rounds> <flag|-> [<flag|-> ...]
R=$(cd "$(dirname "$0")/../.." && pwd)
ROUNDS=$1; shift
for f in "$@"; do
  D=""; [ "$f" != "-" ] && D="-D$f"
  touch "$R/ploidyfrost_amd/csrc/pf_bubble_launch.hpp"
  make -C "$R/ploidyfrost_amd/csrc" -j16 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-value -Wno-unused-result $D" CXXFLAGS="-O2 -g -std=c++17 -fPIC -Wall -Wextra -Wno-sign-compare -Wno-implicit-fallthrough $D" > /dev/null 2>&1 || { echo "build failed for $f"; continue; }
  echo "== $f"
  python "$R/tools/ab_pass.py" --rounds "$ROUNDS" "-" 2>&1 | tail -1
done
