cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4o
mkdir -p $O
for rep in 1 2; do
for p in 1 0; do
U2PL_WS_PERSIST=$p timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline < /dev/null > $O/bench_p$p.json 2> $O/bench_p$p.err
python - $p <<'P'
import json,os,sys
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4o/bench_p%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("persist", sys.argv[1], d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["ms_per_step"], "wgrad", d["roofline_wgrad"]["ms_per_step"])
P
done
done
