ulimit -c 0
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4f
mkdir -p $O
L="ab_libs/libtnsx_t1024i16.so ab_libs/libtnsx_t512i16.so ab_libs/libtnsx_t1024i8.so ab_libs/libtnsx_t256i16.so ab_libs/libtnsx_t512i8.so"
echo "== random order, moving points" > $O/ab_tiles.txt
timeout 600 python tools/ab_libs.py $L --rounds 4 --steps 10 --move >> $O/ab_tiles.txt 2>&1
echo "== z-order, moving points" >> $O/ab_tiles.txt
timeout 600 python tools/ab_libs.py $L --rounds 4 --steps 10 --move --zsort >> $O/ab_tiles.txt 2>&1
cat $O/ab_tiles.txt
