# loss-path parity tests + kernel-only timings of the HBM-bound group (run under gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2i
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_loss_path.py -q -x -m gpu < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
rm -rf /tmp/prof_lp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lp -- python tools/bench_loss_path.py < /dev/null > $O/lp.json 2> $O/lp.err; echo "lp rc=$?"
f=$(find /tmp/prof_lp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/loss_path_kernel_stats.csv
python - <<'P'
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r2i/loss_path_kernel_stats.csv")))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('k_reliability_fused','k_infonce','k_proto','k_contra','k_compact','k_bank','k_scatter_rows','k_zero_rows')):
        print(n[:50].ljust(50), r['Calls'], round(float(r['AverageNs'])/1e3,2))
P
