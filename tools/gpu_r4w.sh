cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4w
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_igemm_ws.py tests/test_gpu_conv_stack.py -q -x < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"
tail -n 5 $O/tests.log | cut -c1-250
ROUNDS=3 REPS=4 timeout 300 python tools/bench_igemm_ws.py 2>&1 | grep -v amdgpu.ids > $O/ab.log
python - <<'P'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4w/ab.log"):
    if not l.startswith("{"): print(l.strip()[:300]); continue
    d=json.loads(l)
    if "kind" in d: print(d["kind"], d["N"], d["Cin"], d["Cout"], d["k"], d["d"], "same" if d["bit_identical"] else "DIFFERENT", d["inloop"]["us"], d["ws"]["us"])
    else: print(d)
P
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4w/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["ms_per_step"], "wgrad", d["roofline_wgrad"]["frac"], d["roofline_wgrad"]["ms_per_step"], "hbm", d["roofline_hbm"]["frac"])
print({k: d[k] for k in d if k.startswith("kernel_ms") or k in ("abi_calls_per_step","kernel_launches_per_step","host_enqueue_ms","gpu_tail_after_last_enqueue_ms")})
P
