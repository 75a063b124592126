#!/bin/bash
# A/B baseline: build the library from the sources of a git revision -> build/libpclseg_<name>.so
# (no #ifdef forks in the tree: an older kernel set is compared by building the older commit).
#   scripts/build_baseline.sh 9098503 r4     # round 4's shipped library = the kernels round 3 measured on an MI355X
set -e
cd "$(dirname "$0")/.."
rev=$1; name=${2:-$1}
src=build/src_$name
rm -rf "$src"; mkdir -p "$src"
git archive "$rev" pclsegmentation_amd/csrc include | tar -x -C "$src"
sha=$(cat "$src"/pclsegmentation_amd/csrc/pclseg_kernels.h "$src"/pclsegmentation_amd/csrc/pclseg_graph.h "$src"/pclsegmentation_amd/csrc/pclseg_api.hip "$src"/include/pclseg.h | sha256sum | cut -c1-16)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wall -Wno-unused-function \
  -DPCLSEG_SRC_SHA=\"$sha\" $EXTRA -o build/libpclseg_$name.so "$src"/pclsegmentation_amd/csrc/pclseg_api.hip
echo "built build/libpclseg_$name.so from $rev (source sha $sha)"
