cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4z
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_igemm_ws.py tests/test_gpu_conv_stack.py -q -x < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"
tail -n 4 $O/tests.log | cut -c1-250
for mode in one two nomix one2 nomix2; do
  unset U2PL_WS_NARROW U2PL_WS_MIX2
  case $mode in two) export U2PL_WS_MIX2=1;; nomix*) export U2PL_WS_NARROW=2;; esac
  ROUNDS=3 REPS=4 timeout 300 python tools/bench_igemm_ws.py 2>&1 | grep -v amdgpu.ids > $O/ab_$mode.log
done
unset U2PL_WS_NARROW U2PL_WS_MIX2
python - <<'P'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4z/"
R={}
for m in ("one","two","nomix","one2","nomix2"):
    R[m]=[json.loads(l) for l in open(O+"ab_%s.log"%m) if l.startswith("{")]
for i,d in enumerate(R["one"][:-1]):
    print(d["kind"], d["N"], d["Cin"], d["Cout"], d["k"], "same" if d["bit_identical"] else "DIFFERENT", *[R[m][i]["ws"]["us"] for m in R])
print([R[m][-1]["total_us"]["ws"] for m in R])
P
