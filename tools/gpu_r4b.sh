cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_igemm_ws.py -q < /dev/null > $O/tests_ws.log 2>&1; echo "ws tests rc=$?"
tail -5 $O/tests_ws.log
U2PL_LIB_PATH=$GRAFT_REPO_ROOT/u2pl_amd/lib/variants/libu2pl_hip_abl.so timeout 300 python tools/bench_ws_ablate.py < /dev/null > $O/ablate.jsonl 2> $O/ablate.err; echo "ablate rc=$?"
cat $O/ablate.jsonl; tail -3 $O/ablate.err
timeout 400 python tools/bench_igemm_ws.py < /dev/null > $O/bench_igemm_ws.jsonl 2> $O/bench_igemm_ws.err; echo "bench rc=$?"
python - <<'P'
import json,os
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4b/bench_igemm_ws.jsonl"):
    d=json.loads(l)
    if "kind" in d: print(d["kind"],d["N"],d["Cin"],d["Cout"],d["k"],d["d"],d["bit_identical"], *[f'{k}={d[k]["us"]}/{d[k]["frac"]}' for k in ("inloop","ws_stagger","ws_pinned","ws_sgb")])
    else: print(d)
P
