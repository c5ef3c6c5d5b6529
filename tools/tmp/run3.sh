timeout 600 python -m pytest tests/test_gpu_loss_path.py -x -q -k "contra or bank" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for pix in 256 128 64; do
U2PL_PROTO_PIX=$pix timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_loss_$pix -- python $GRAFT_REPO_ROOT/tools/bench_loss_path.py > $GRAFT_REPO_ROOT/gpurun_out/loss_path_$pix.log 2>&1
done
