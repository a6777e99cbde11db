#!/bin/bash
# Builds diagnostic variants of the stem kernel: build_stem.sh <name>:<extra -D flags> ...
#   e.g. build_stem.sh 0:-DTT_STEM_SKIP=0 60:-DTT_STEM_SKIP=60
cd "$(dirname "$0")"
F="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DTT_STEM_STAMP"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  hipcc $F ${flags//,/ } -o stem_parts_$name stem_parts.hip 2>&1 | grep -E "error" -A3
done
