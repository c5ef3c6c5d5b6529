#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c
mkdir -p $O
U2PL_GRAPH_DEBUG=1 timeout 500 python -m pytest tests/test_gpu_graphs.py -x -q -s > $O/graphs.log 2>&1; echo "graphs rc $?"
timeout 200 python tools/dbg_graphs.py > $O/dbg.log 2>&1; echo "dbg rc $?"
timeout 700 python bench.py --steps 10 --warmup 4 --no-cpu-baseline > $O/bench_graphs.json 2> $O/bench_graphs.err; echo "bench rc $?"
timeout 400 python bench.py --bf16 --crop 801 --steps 4 --warmup 3 --no-cpu-baseline --no-config5-leg > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bf16 rc $?"
tail -n 12 $O/graphs.log; tail -n 12 $O/dbg.log; tail -n 5 $O/bench_graphs.err $O/bench_bf16.err
