#!/bin/bash
# Projection of the latency part of an N-rank step on the one-GPU box: a world of one on RCCL (U2PL_DIST_SINGLE=1) with a spin of
# L microseconds behind every synchronous collective (U2PL_EMULATE_COLL_US).  Writes gpurun_out/r06_latency_<L>.json.
mkdir -p gpurun_out
for L in ${LATS:-0 20 40}; do
  U2PL_DIST_SINGLE=1 U2PL_EMULATE_COLL_US=$L python bench.py --gpus 1 --steps 10 --warmup 3 2>gpurun_out/r06_latency_$L.err | tail -1 > gpurun_out/r06_latency_$L.json
done
python - <<'PY'
import json
import os
for L in [int(x) for x in os.environ.get("LATS", "0 20 40").split()]:
    d = json.loads(open(f"gpurun_out/r06_latency_{L}.json").read())
    print(L, d["ms_per_step"], d["value"], d.get("syncbn_collectives_per_step"), d.get("comm_exposed_ms"))
PY
