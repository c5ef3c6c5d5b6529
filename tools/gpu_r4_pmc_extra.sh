# extra PMC passes over the dense replay of the MFMA group (run under gpurun): where the waves of k_igemm_ws spend their cycles.
# Each counter group in its OWN pass (counters only: --pmc with --kernel-trace).  Output: gpurun_out/r4pmc/pmc_extra.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4pmc
rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" | sort -u > $O/counters.txt
wc -l $O/counters.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ok=""
  for c in $grp; do grep -qx "$c" $O/counters.txt && ok="$ok $c"; done
  [ -z "$ok" ] && continue
  rm -rf /tmp/pmcx_$i
  timeout 300 rocprofv3 --pmc $ok --kernel-trace --output-format csv -d /tmp/pmcx_$i -- python tools/dense_replay.py < /dev/null > $O/run_$i.json 2> $O/run_$i.err; echo "pass $i ($ok) rc=$?"
done
python - <<'P'
import csv,glob,collections,json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4pmc"
out=collections.defaultdict(dict)
for d in sorted(glob.glob("/tmp/pmcx_*")):
    f=glob.glob(d+"/*/*counter_collection.csv")
    if not f: continue
    per=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if not any(s in k for s in ("k_igemm_ws","k_wgrad_tr")): continue
        per[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
    for k,v in per.items():
        for c,x in v.items(): out[k][c]=dict(sum=x, dispatches=cnt[k][c])
stamp={}
try: stamp=json.load(open(O+"/../r4final/STAMP.json"))
except Exception: pass
from u2pl_amd.roofline import kernel_source_hash
json.dump(dict(kernel_sources_sha=kernel_source_hash(), note="rocprofv3 --pmc passes (one group per pass) of tools/dense_replay.py; sums over all dispatches of a kernel", kernels=out), open(O+"/pmc_extra.json","w"), indent=1)
for k,v in out.items():
    print(k[:44], {c:("%.3g"%x["sum"]) for c,x in v.items()})
P
