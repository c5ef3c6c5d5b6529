cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4r
mkdir -p $O
timeout 200 python tools/bench_ws_epilogues.py 2>&1 | grep -v amdgpu.ids | tee $O/epi.log
timeout 600 python -m pytest tests/test_gpu_igemm_ws.py tests/test_gpu_conv_stack.py -q -x < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"
tail -n 3 $O/tests.log | cut -c1-250
