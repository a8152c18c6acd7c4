#!/bin/bash
# builds ab_libs/libtnsx_<name>.so from the sources of a git revision (A/B against the working tree with tools/ab_libs.py).   usage: build_rev.sh <name> <rev> [-DFOO=1 ...]
set -e
cd "$(dirname "$0")/.."
name=$1; rev=$2; shift 2
src=ab_libs/src_$name; rm -rf $src; mkdir -p $src ab_libs/obj_$name
git archive $rev treensearch_amd/csrc include | tar -x -C $src
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -x hip -I$src/include"
objs=""
for f in tnsx_kernels.hip tnsx_build.hip tnsx_query.hip tnsx_engine.cpp tnsx_multi.cpp tnsx_slab.cpp; do
  /opt/rocm/bin/hipcc $FL "$@" -c $src/treensearch_amd/csrc/$f -o ab_libs/obj_$name/${f%.*}.o &
  objs="$objs ab_libs/obj_$name/${f%.*}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/libtnsx_$name.so $objs -ldl -lpthread
echo built ab_libs/libtnsx_$name.so
