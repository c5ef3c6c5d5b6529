cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6s; rm -rf /tmp/pf
U2PL_GRAPHS=0 U2PL_NO_SIDE_STREAM=1 U2PL_NO_WGRAD_STREAM=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-direct-leg --no-config5-leg > gpurun_out/r6s/bench.json 2> gpurun_out/r6s/bench.err
f=$(find /tmp/pf -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r6s/serial_kernel_stats.csv; echo rc $?
