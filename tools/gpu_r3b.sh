cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $O
timeout 100 python tools/bench_split.py 2>&1 | grep -v amdgpu.ids
timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3b/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], "hbm", d["roofline_hbm"]["frac"], d["roofline_hbm"]["stages_us"])
print(d["roofline_hbm"].get("split_phases_us_block0"))
print({k: d.get(k) for k in ("abi_calls_per_step", "kernel_launches_per_step", "ms_step_lr0.01", "split_launches_timed", "split_second_barrier_launches_timed")})
P
