# round-3 iteration script (run under gpurun): GPU suite, split / loss-path micro-benchmarks, the headline bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_loss_path.py -q -x -m gpu < /dev/null > $O/tests_loss.log 2>&1; echo "loss-path tests rc=$?"; tail -5 $O/tests_loss.log
timeout 60 python tools/bench_split.py < /dev/null > $O/split.log 2>&1; echo "split rc=$?"; cat $O/split.log
timeout 120 python tools/bench_loss_path.py < /dev/null > $O/lp.json 2> $O/lp.err; echo "lp rc=$?"; grep -E "persistent|contra_fwd|reliability" $O/lp.json
timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline < /dev/null > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json,os
try:
    d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3a/bench.json").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], "wgrad", d["roofline_wgrad"]["frac"], "hbm", d["roofline_hbm"]["frac"], d["roofline_hbm"]["stages_us"])
except Exception as e:
    print("bench parse failed", e)
P
timeout 900 python -m pytest tests -q -m gpu --ignore=tests/test_gpu_loss_path.py < /dev/null > $O/tests_rest.log 2>&1; echo "rest tests rc=$?"; tail -8 $O/tests_rest.log
