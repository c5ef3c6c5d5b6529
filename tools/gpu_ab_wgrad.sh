set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 400 python -m pytest tests/test_gpu_conv_stack.py tests/test_gpu_dist.py -x -q -m gpu < /dev/null > gpurun_out/r2b/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r2b/tests.log
U2PL_WGRAD_WAVES=4 timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 3 < /dev/null > gpurun_out/r2b/bench_w4.json 2> gpurun_out/r2b/bench_w4.err; echo "rc=$?"
timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 3 < /dev/null > gpurun_out/r2b/bench_w8.json 2> gpurun_out/r2b/bench_w8.err; echo "rc=$?"
python - <<'P'
import json
for n in ("w4","w8"):
    try:
        d=json.loads(open(f"gpurun_out/r2b/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline_wgrad"]["frac"], d["roofline_wgrad"]["ms_per_step"])
    except Exception as e: print(n, "ERR", e)
P
