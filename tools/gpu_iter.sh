# one iteration of the kernel work: conv + loss-path parity tests, loss-path kernel stats, headline bench (run under gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2g
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_conv_stack.py tests/test_gpu_loss_path.py -q -x -m gpu < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
rm -rf /tmp/prof_lp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lp -- python tools/bench_loss_path.py < /dev/null > $O/lp.json 2> $O/lp.err; echo "lp rc=$?"
f=$(find /tmp/prof_lp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/loss_path_kernel_stats.csv
python - <<'P'
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r2g/loss_path_kernel_stats.csv")))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('k_reliability_fused','k_infonce','k_proto','k_contra','k_compact','k_bank','k_scatter_rows','k_zero_rows')):
        print(n[:50].ljust(50), r['Calls'], round(float(r['AverageNs'])/1e3,2))
P
timeout 300 python bench.py --no-cpu-baseline --steps 12 --warmup 3 < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r2g/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["ms_per_step"], "wgrad", d["roofline_wgrad"]["frac"], d["roofline_wgrad"]["ms_per_step"], "hbm", d["roofline_hbm"]["frac"], d["roofline_hbm"]["stages_us"])
P
