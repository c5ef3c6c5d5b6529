cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $O
U2PL_BENCH_SHAPES=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3e/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"], "\nwgrad", d["roofline_wgrad"])
print("kernel_ms_per_step", d["kernel_ms_per_step"])
for s in d["conv_shapes"]:
    print(s)
P
