# HBM traffic of the headline bench from the PMC counters, two separate passes (MI355X_MICROARCH.md): run under gpurun
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2e
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline < /dev/null > $O/bench_$c.json 2> $O/bench_$c.err
  echo "pmc $c rc=$?"
done
python tools/parse_pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/traffic.json 5 10
