# round-end measurement artefacts (run under gpurun): rocprofv3 kernel stats of the headline bench with the side streams
# on and serialised, and the plain JSON line with the CPU baseline leg.  Outputs under gpurun_out/r2d/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2d
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
timeout 400 python bench.py --steps 8 --warmup 3 < /dev/null > $O/bench_plain.json 2> $O/bench_plain.err; echo "plain rc=$?"
tail -c 600 $O/bench_plain.json
ls -la $O
