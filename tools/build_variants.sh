#!/bin/bash
# Diagnostic builds of the engine for tools/probe_variants.py (never used by the product).
set -e
cd "$(dirname "$0")/../pygrank_amd/csrc"
OUT=../../tools/variants
mkdir -p $OUT
build() { # name, flags
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $2 -shared pgh_runtime.hip pgh_graph.hip pgh_spmv.hip pgh_graphgen.hip pgh_bsf.hip pgh_spmm.hip pgh_pb.hip pgh_bsf64.hip pgh_dist.hip -ldl -o $OUT/libpgh_$1.so
}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  build "$name" "$flags" &
done
wait
ls -la $OUT
