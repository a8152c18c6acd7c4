# round 6: bounding-box cull in registers + LDS compaction for cells that fit the first tier's loop (tnsx_query.hip, cull_cell_to_stage): A/B of the threshold variants
set -x
ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6rc}
mkdir -p $O
LIBS="${LIBS:-ab_libs/libtnsx_nocull.so ab_libs/libtnsx_rc5.so ab_libs/libtnsx_rc6.so ab_libs/libtnsx_rc4.so}"
timeout 900 python tools/ab_libs.py $LIBS --check --zsort --move --rounds 6 --steps 15 > $O/ab_c2.txt 2>&1
grep -v amdgpu $O/ab_c2.txt
timeout 900 python tools/ab_libs.py $LIBS --check --workload c3 --rounds 5 --steps 10 > $O/ab_c3.txt 2>&1
grep -v amdgpu $O/ab_c3.txt
timeout 900 python tools/ab_libs.py $LIBS --check --workload c4 --points 10000000 --rounds 4 --steps 10 > $O/ab_c4.txt 2>&1
grep -v amdgpu $O/ab_c4.txt
