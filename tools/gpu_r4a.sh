# round 4, first lease: parity of the pre-split-weight GEMM, A/B timing against the in-loop split, MFMA-busy counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4a
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_igemm_ws.py -q -x < /dev/null > $O/tests_ws.log 2>&1; echo "ws tests rc=$?"
tail -15 $O/tests_ws.log
timeout 400 python tools/bench_igemm_ws.py < /dev/null > $O/bench_igemm_ws.jsonl 2> $O/bench_igemm_ws.err; echo "bench rc=$?"
cat $O/bench_igemm_ws.jsonl | cut -c1-400
tail -5 $O/bench_igemm_ws.err
rm -rf /tmp/pmc_ws
QUICK=1 ROUNDS=2 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_ws -- python tools/bench_igemm_ws.py < /dev/null > $O/pmc_ws.out 2> $O/pmc_ws.err; echo "pmc rc=$?"
python - <<'P'
import csv,glob,collections,json,os
f=glob.glob("/tmp/pmc_ws/*/*counter_collection.csv")
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4a"
if f:
    per=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "igemm" not in k: continue
        k=k+" grid="+r.get("Grid_Size","")
        per[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE": cnt[k]+=1
    out={}
    for k,v in per.items():
        busy,act=v.get("SQ_VALU_MFMA_BUSY_CYCLES",0),v.get("GRBM_GUI_ACTIVE",0)
        out[k]=dict(dispatches=cnt[k], mfma_pipe_util=round(busy/(act/8*1024),4) if act else None)
    json.dump(out, open(O+"/pmc_ws_mfma_busy.json","w"), indent=1)
    for k,v in sorted(out.items()): print(k[:90], v)
P
