cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_igemm_ws.py tests/test_gpu_conv_stack.py tests/test_gpu_train_step.py -q -x < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"
tail -n 3 $O/tests.log
timeout 400 python tools/bench_igemm_ws.py < /dev/null > $O/bench_igemm_ws.jsonl 2> $O/bench_igemm_ws.err; echo "bench rc=$?"
tail -n 1 $O/bench_igemm_ws.jsonl
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4n/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["ms_per_step"], "wgrad", d["roofline_wgrad"]["frac"], d["roofline_wgrad"]["ms_per_step"], "hbm", d["roofline_hbm"]["frac"])
P
