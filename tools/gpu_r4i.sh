cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4i
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"
tail -n 6 $O/tests.log
timeout 300 python bench.py --steps 8 --warmup 3 < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -n 3 $O/bench.err
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4i/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["ms_per_step"], "wgrad", d["roofline_wgrad"]["frac"], d["roofline_wgrad"]["ms_per_step"], "hbm", d["roofline_hbm"]["frac"])
print({k: d[k] for k in d if k.startswith("kernel_ms") or k in ("abi_calls_per_step","kernel_launches_per_step")})
P
U2PL_CONV_WS=0 timeout 300 python bench.py --steps 8 --warmup 3 < /dev/null > $O/bench_nows.json 2> $O/bench_nows.err; echo "bench nows rc=$?"
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4i/bench_nows.json").read().strip().splitlines()[-1])
print("WS=0:", d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["ms_per_step"])
P
