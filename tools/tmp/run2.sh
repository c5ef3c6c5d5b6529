set -x
timeout 600 python -m pytest tests/test_gpu_loss_path.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_loss -- python $GRAFT_REPO_ROOT/tools/bench_loss_path.py > $GRAFT_REPO_ROOT/gpurun_out/loss_path.log 2>&1
tail -40 $GRAFT_REPO_ROOT/gpurun_out/loss_path.log
