cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4x
mkdir -p $O
for mode in mix nomix mix2 nomix2; do
  if [ ${mode#no} != $mode ]; then export U2PL_WS_NARROW=2; else unset U2PL_WS_NARROW; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline < /dev/null > $O/bench_$mode.json 2> $O/bench_$mode.err; echo "bench $mode rc=$?"
done
python - <<'P'
import json,os
for m in ("mix","nomix","mix2","nomix2"):
    d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4x/bench_%s.json"%m).read().strip().splitlines()[-1])
    print(m, d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["ms_per_step"], "wgrad", d["roofline_wgrad"]["ms_per_step"], "launches", d["kernel_launches_per_step"])
P
