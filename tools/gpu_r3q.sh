#!/bin/bash
# block timelines of the phase-1 kernels + parity of the loss path
mkdir -p gpurun_out/r3q
python tools/bench_phase1_blocks.py > gpurun_out/r3q/phase1_blocks.json 2> gpurun_out/r3q/err.log
tail -5 gpurun_out/r3q/err.log
python - <<'P'
import json
d = json.load(open("gpurun_out/r3q/phase1_blocks.json"))
for k, v in d.items():
    print(k, {a: round(b, 2) for a, b in v.items() if not isinstance(b, list)}, {a: b for a, b in v.items() if isinstance(b, list)})
P
python tools/bench_loss_path.py 2>/dev/null | grep -E "contra|persistent"
timeout 600 python -m pytest tests/test_gpu_loss_path.py -x -q -m gpu 2>&1 | tail -5
