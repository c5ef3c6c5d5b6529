# full GPU suite + smoke + A/B bench of the eval-BN epilogue fusion (run under gpurun)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 600 python -m pytest tests -q -m gpu < /dev/null > gpurun_out/r2c/tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r2c/tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null 2>&1 | tail -2
U2PL_NO_EVAL_BN_FUSION=1 timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 3 < /dev/null > gpurun_out/r2c/bench_nofuse.json 2> gpurun_out/r2c/bench_nofuse.err; echo "rc=$?"
timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 3 < /dev/null > gpurun_out/r2c/bench_fuse.json 2> gpurun_out/r2c/bench_fuse.err; echo "rc=$?"
python - <<'P'
import json
for n in ("nofuse","fuse"):
    try:
        d=json.loads(open(f"gpurun_out/r2c/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["launches_per_step"], d["roofline_wgrad"]["frac"], d["roofline_hbm"]["frac"])
    except Exception as e: print(n, "ERR", e)
P
