cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4l
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_igemm_ws.py -q -x < /dev/null > $O/tests_ws.log 2>&1; echo "ws tests rc=$?"
tail -n 12 $O/tests_ws.log
timeout 400 python tools/bench_igemm_ws.py < /dev/null > $O/bench_igemm_ws.jsonl 2> $O/bench_igemm_ws.err; echo "bench rc=$?"
python - <<'P'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4l/bench_igemm_ws.jsonl"):
    d=json.loads(l)
    if "kind" in d: print(d["kind"],d["N"],d["Cin"],d["Cout"],d["k"],d["d"],d["bit_identical"], *[f'{k}={d[k]["us"]}/{d[k]["frac"]}' for k in ("inloop","ws")])
    else: print(d)
P
tail -n 3 $O/bench_igemm_ws.err
