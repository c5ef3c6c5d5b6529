#!/bin/bash
# round 5, first GPU pass: the new tests + bench with / without HIP graphs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5a
mkdir -p $O
U2PL_GRAPH_DEBUG=1 timeout 700 python -m pytest tests/test_gpu_graphs.py -x -q -s > $O/graphs.log 2>&1; echo "graphs rc $?"
timeout 400 python -m pytest tests/test_gpu_igemm_ws.py -x -q -k "2gib or split_guard or nonfinite" > $O/ws.log 2>&1; echo "ws rc $?"
timeout 700 python -m pytest tests/test_gpu_miou_gate.py -x -q -s > $O/miou.log 2>&1; echo "miou rc $?"
timeout 700 python -m pytest tests/test_gpu_dist.py -x -q -k "eight" > $O/dist8.log 2>&1; echo "dist8 rc $?"
timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline > $O/bench_graphs.json 2> $O/bench_graphs.err; echo "bench rc $?"
U2PL_GRAPHS=0 timeout 500 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-config5-leg --no-direct-leg > $O/bench_eager.json 2> $O/bench_eager.err; echo "bench eager rc $?"
for f in graphs ws miou dist8; do tail -n 3 $O/$f.log; done
