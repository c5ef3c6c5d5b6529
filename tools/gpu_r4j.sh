cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4j
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv_stack.py -q -x -k "conv2d_fwd_bwd or winograd or large_pixel or full_size" < /dev/null > $O/tests_conv.log 2>&1; echo "conv tests rc=$?"
tail -n 5 $O/tests_conv.log
timeout 400 python tools/bench_wgrad.py < /dev/null > $O/bench_wgrad.jsonl 2> $O/bench_wgrad.err; echo "bench rc=$?"
cat $O/bench_wgrad.jsonl; tail -n 3 $O/bench_wgrad.err
