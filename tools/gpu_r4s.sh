cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4s
mkdir -p $O
ROUNDS=3 REPS=6 timeout 300 python tools/bench_igemm_ws.py 2>&1 | grep -v amdgpu.ids | tee $O/ab.log | cut -c1-400
