# kernel-only times of the HBM-bound group at the 19-class upper bound (run under gpurun) -> gpurun_out/r3f/loss_path_kernel_stats.csv
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $O
rm -rf /tmp/prof_lp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lp -- python tools/bench_loss_path.py < /dev/null > $O/lp.json 2> $O/lp.err; echo "lp rc=$?"
f=$(find /tmp/prof_lp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/loss_path_kernel_stats.csv
python - <<'P'
import csv,os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3f/loss_path_kernel_stats.csv")))
tot=0
for r in rows:
    n=r['Name']
    if any(k in n for k in ('k_reliability_fused','k_infonce','k_proto','k_contra','k_compact','k_bank','k_scatter_rows','k_zero_rows','k_phase1')):
        print(n[:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,2), r['MinNs'], r['MaxNs']); tot+=float(r['AverageNs'])/1e3
print("sum of averages", round(tot,1))
P
cat $O/lp.json
