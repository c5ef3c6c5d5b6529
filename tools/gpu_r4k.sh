cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4k
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_stack.py tests/test_gpu_train_step.py tests/test_gpu_igemm_ws.py -q -x < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"
tail -n 4 $O/tests.log
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -n 3 $O/bench.err
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4k/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["ms_per_step"], "wgrad", d["roofline_wgrad"]["frac"], d["roofline_wgrad"]["ms_per_step"], "hbm", d["roofline_hbm"]["frac"])
print({k: d[k] for k in d if k.startswith("kernel_ms") or k in ("abi_calls_per_step","kernel_launches_per_step")})
P
timeout 300 python tools/host_overhead.py < /dev/null > $O/host_overhead.txt 2>&1; echo "host rc=$?"
head -45 $O/host_overhead.txt
