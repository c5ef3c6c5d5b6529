# profiles for the MFMA roofline (run under gpurun): dense-replay kernel trace, MFMA-busy PMC pass
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p $O
rm -rf /tmp/prof_dense
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dense -- python tools/dense_replay.py < /dev/null > $O/dense_replay.json 2> $O/dense.err; echo "dense rc=$?"
f=$(find /tmp/prof_dense -name '*kernel_trace.csv' | head -1)
python tools/parse_dense_trace.py "$f" $O/dense_replay.json $O/dense_replay_frac.json
python - "$f" $O/dense_replay_kernel_trace_tail.csv <<'P'
import csv,sys,json,os
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_conv_igemm" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
m=json.loads([l for l in open(os.path.dirname(sys.argv[2])+"/dense_replay.json").read().splitlines() if l.startswith("{")][-1])
last=rows[-int(m["kernel_launches"]):]
w=csv.writer(open(sys.argv[2],"w"))
w.writerow(["Kernel_Name","Start_Timestamp","End_Timestamp","Grid_Size","Workgroup_Size"])
for r in last: w.writerow([r["Kernel_Name"].split("(")[0][:60],r["Start_Timestamp"],r["End_Timestamp"],r.get("Grid_Size",""),r.get("Workgroup_Size","")])
P
rm -rf /tmp/pmc_mfma
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python tools/dense_replay.py < /dev/null > $O/pmc_mfma.json 2> $O/pmc_mfma.err; echo "pmc rc=$?"
python - <<'P'
import csv,glob,collections,json,os
f=glob.glob("/tmp/pmc_mfma/*/*counter_collection.csv")
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3p"
if f:
    per=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "k_conv_igemm" not in k and "k_conv_wgrad" not in k: continue
        per[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE": cnt[k]+=1
    out={}
    for k,v in per.items():
        busy,act=v.get("SQ_VALU_MFMA_BUSY_CYCLES",0),v.get("GRBM_GUI_ACTIVE",0)
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs' MFMA pipes (4 per CU x 256 CUs); GRBM_GUI_ACTIVE = GPU-busy cycles
        out[k]=dict(dispatches=cnt[k], mfma_busy_cycles=busy, gui_active_cycles=act, mfma_pipe_util=busy/(act/8*1024) if act else None)
    json.dump(out, open(O+"/pmc_mfma_busy.json","w"), indent=1)
    for k,v in sorted(out.items(), key=lambda kv:-kv[1]["gui_active_cycles"])[:8]: print(k[:50], v)
P
