cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_loss_path.py -q -x -m gpu < /dev/null > $O/tests_loss.log 2>&1; echo "loss-path tests rc=$?"; tail -4 $O/tests_loss.log
rm -rf /tmp/prof_lp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lp -- python tools/bench_loss_path.py < /dev/null > $O/lp.json 2> $O/lp.err; echo "lp rc=$?"
f=$(find /tmp/prof_lp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/loss_path_kernel_stats.csv
python - <<'P'
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3d/loss_path_kernel_stats.csv")))
tot=0
for r in rows:
    n=r['Name']
    if any(k in n for k in ('k_reliability_fused','k_infonce','k_proto','k_contra','k_compact','k_bank','k_scatter_rows','k_zero_rows','k_phase1')):
        print(n[:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,2)); tot+=float(r['AverageNs'])/1e3
print("sum of averages", round(tot,1))
P
timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3d/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], "hbm", d["roofline_hbm"]["frac"], d["roofline_hbm"]["stages_us"])
print(d["roofline_hbm"].get("split_phases_us_block0"))
P
