#!/bin/bash
# builds gpurun_out/ab/libtnsx_<name>.so: the library with extra -D flags on tnsx_query.hip / tnsx_build.hip / tnsx_engine.cpp (A/B experiments,
# timed against each other in ONE process by tools/ab_libs.py).   usage: build_variant.sh <name> [-DFOO=1 ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -m treensearch_amd.build > /dev/null
mkdir -p ab_libs/obj_$name
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -x hip -Iinclude"
for f in tnsx_query.hip tnsx_build.hip tnsx_engine.cpp; do
  /opt/rocm/bin/hipcc $FL "$@" -c treensearch_amd/csrc/$f -o ab_libs/obj_$name/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/libtnsx_$name.so ab_libs/obj_$name/tnsx_query.o ab_libs/obj_$name/tnsx_build.o ab_libs/obj_$name/tnsx_engine.o treensearch_amd/lib/tnsx_kernels.o treensearch_amd/lib/tnsx_multi.o treensearch_amd/lib/tnsx_slab.o -ldl -lpthread
echo built ab_libs/libtnsx_$name.so
