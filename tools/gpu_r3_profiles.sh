# round-3 measurement artefacts (run under gpurun): rocprofv3 kernel stats of the headline bench (side streams on /
# serialised), the loss-path group at the 19-class upper bound, and the two PMC passes for the HBM traffic.
# Outputs under gpurun_out/r3f/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $O
prof() {   # name, extra env...
  n=$1; shift
  rm -rf /tmp/prof_$n
  env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline < /dev/null > $O/bench_${n}_under_rocprof.json 2> $O/bench_${n}.err
  echo "prof $n rc=$?"
  f=$(find /tmp/prof_$n -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/bench_${n}_kernel_stats.csv
}
prof overlap U2PL_DUMMY=1
prof serial U2PL_NO_SIDE_STREAM=1 U2PL_NO_WGRAD_STREAM=1
rm -rf /tmp/prof_lp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lp -- python tools/bench_loss_path.py < /dev/null > $O/lp.json 2> $O/lp.err; echo "lp rc=$?"
f=$(find /tmp/prof_lp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/loss_path_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline < /dev/null > $O/bench_$c.json 2> $O/bench_$c.err
  echo "pmc $c rc=$?"
done
python tools/parse_pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/traffic.json 6 10
ls -la $O
