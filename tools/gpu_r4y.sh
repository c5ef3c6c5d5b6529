cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4y
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_loss_path.py tests/test_gpu_train_step.py -q -x < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"
tail -n 3 $O/tests.log | cut -c1-250
timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4y/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["ms_per_step"], "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_per_step"), d["roofline"]["algorithmic_bytes_per_step"])
print(d["roofline"]["traffic_source"]); print(d["roofline_hbm"]["traffic"])
P
