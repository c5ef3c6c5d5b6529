# round-3 validation (run under gpurun): full GPU suite, smoke(), the plain bench line with the CPU-baseline leg
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3z
mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/tests.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null 2>&1 | tail -2
timeout 300 python bench.py --steps 8 --warmup 3 < /dev/null > $O/bench_plain.json 2> $O/bench_plain.err; echo "plain rc=$?"
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3z/bench_plain.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["algorithmic_frac"], "wgrad", d["roofline_wgrad"]["frac"], "hbm", d["roofline_hbm"]["frac"], d["roofline_hbm"]["stages_us"], d.get("cpu_baseline",{}).get("value"))
P
cp gpurun_out/full_size_parity.json $O/ 2>/dev/null
