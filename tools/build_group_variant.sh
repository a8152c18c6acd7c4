#!/bin/bash
# builds ab_libs/libtnsx_group.so: the library WITH the group formulation of round 3 (tools/ubench/tnsx_query_group.hip: MFMA tiles + exact re-test in the
# rounding band; measured 2.6 x slower than the cell kernels, profiles/r3_group_formulation.txt, docs/history).  A refutation kept reproducible, not part of the
# product: tools/test_group_formulation.py (its parity tests) and tools/group_probe.py / tools/pmc_group.sh (its timing and counters) load this variant.
set -e
cd "$(dirname "$0")/.."
python -m treensearch_amd.build > /dev/null
mkdir -p ab_libs/obj_group
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -x hip -Iinclude -Itreensearch_amd/csrc -DTNSX_WITH_GROUP_FORMULATION"
/opt/rocm/bin/hipcc $FL -c treensearch_amd/csrc/tnsx_query.hip -o ab_libs/obj_group/tnsx_query.o &
/opt/rocm/bin/hipcc $FL -c treensearch_amd/csrc/tnsx_engine.cpp -o ab_libs/obj_group/tnsx_engine.o &
/opt/rocm/bin/hipcc $FL -mllvm -amdgpu-mfma-vgpr-form -c tools/ubench/tnsx_query_group.hip -o ab_libs/obj_group/tnsx_query_group.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/libtnsx_group.so ab_libs/obj_group/tnsx_query.o ab_libs/obj_group/tnsx_engine.o ab_libs/obj_group/tnsx_query_group.o \
  treensearch_amd/lib/tnsx_build.o treensearch_amd/lib/tnsx_kernels.o treensearch_amd/lib/tnsx_multi.o treensearch_amd/lib/tnsx_slab.o -ldl -lpthread
echo built ab_libs/libtnsx_group.so
