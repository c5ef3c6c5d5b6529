#!/bin/bash
# round 5, second GPU pass: whole GPU suite with HIP graphs on by default + bench with graphs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5b
mkdir -p $O
U2PL_GRAPH_DEBUG=1 timeout 500 python -m pytest tests/test_gpu_graphs.py -x -q -s > $O/graphs.log 2>&1; echo "graphs rc $?"
timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline > $O/bench_graphs.json 2> $O/bench_graphs.err; echo "bench rc $?"
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_graphs.py > $O/tests.log 2>&1; echo "suite rc $?"
for f in graphs tests; do tail -n 15 $O/$f.log; done
tail -n 5 $O/bench_graphs.err
