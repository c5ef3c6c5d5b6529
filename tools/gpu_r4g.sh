cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4g
mkdir -p $O
U2PL_LIB_PATH=$GRAFT_REPO_ROOT/u2pl_amd/lib/variants/libu2pl_hip_stamps.so timeout 200 python tools/ws_stamps.py < /dev/null > $O/stamps.txt 2> $O/stamps.err; echo "stamps rc=$?"
grep -v "blk1" $O/stamps.txt; tail -n 3 $O/stamps.err
