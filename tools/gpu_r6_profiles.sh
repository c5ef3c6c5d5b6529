# Round-6 measurement artefacts, ONE script run under gpurun at the final kernel commit (tools/run_r6_profiles.sh wraps it:
# writes the commit into gpurun_head.txt, calls gpurun, copies the results into profiles/r06_*).  Outputs: gpurun_out/r6final/.
#   1 full GPU suite (+ full_size_parity.json)   2 smoke()   3 plain bench line with the CPU-baseline leg
#   4 rocprofv3 kernel stats of the bench, side streams on / serialised   5 dense replay of the MFMA group: kernel trace
#   6 MFMA-busy PMC pass (own pass, counters only)   7 loss-path group kernel stats   8 FETCH_SIZE / WRITE_SIZE PMC passes
#   9 host overhead (cProfile of the enqueue)   10 the config-5 line (bf16 operands, 801^2)
# Round 5: HIP graphs are on by default (the `overlap` profile and the plain line replay the four static segments; the
# `serial` profile is eager: U2PL_GRAPHS=0, one stream); the profiled bench runs skip the direct-convolution and config-5 legs.
# Round 6: the number of full steps of a profiled bench run is read from its own JSON line (train_steps_in_process: priming +
# warm-up + timed + phase leg + roofline step + five lr-0.01 steps + four epoch-0 steps).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6final
rm -rf $O; mkdir -p $O
HEAD=$(cat gpurun_head.txt 2>/dev/null || echo unknown)
SHA=$(python -c "from u2pl_amd.roofline import kernel_source_hash as h; print(h())")
echo "{\"commit\": \"$HEAD\", \"kernel_sources_sha\": \"$SHA\"}" > $O/STAMP.json
cat $O/STAMP.json
# 8
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 420 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-direct-leg --no-config5-leg < /dev/null > $O/bench_$c.json 2> $O/bench_$c.err
  echo "pmc $c rc=$?"
done
NST=$(python -c "import json;print(json.loads(open('$O/bench_FETCH_SIZE.json').read().strip().splitlines()[-1])['train_steps_in_process'])")
echo "train steps in the profiled run: $NST"
python tools/parse_pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/traffic.json $NST 10 $HEAD
cp $O/traffic.json profiles/r06_traffic.json    # (so that the plain bench line below carries this build's traffic record)
# 1-3
timeout 1100 python -m pytest tests -q -m gpu < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"
tail -4 $O/tests.log | cut -c1-300
cp gpurun_out/full_size_parity.json gpurun_out/miou_gate.json $O/ 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 700 python bench.py --steps 10 --warmup 4 < /dev/null > $O/bench_plain.json 2> $O/bench_plain.err; echo "plain rc=$?"
python - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6final/bench_plain.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "igemm", d["roofline"]["frac"], d["roofline"]["ms_per_step"], "wgrad", d["roofline_wgrad"]["frac"], d["roofline_wgrad"]["ms_per_step"],
      "hbm", d["roofline_hbm"]["frac"], d["roofline_hbm"]["stages_us"], "host", d.get("host_enqueue_ms"), "cpu", d.get("cpu_baseline",{}).get("value"),
      "calls", d.get("abi_calls_per_step"), "direct", d.get("ms_per_step_direct"), "phases", d.get("phase_ms"), "cfg5", d.get("config5",{}).get("images_per_s"))
P
# 4
prof() {   # name, extra env...
  n=$1; shift
  rm -rf /tmp/prof_$n
  env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-direct-leg --no-config5-leg < /dev/null > $O/bench_${n}_under_rocprof.json 2> $O/bench_${n}.err
  echo "prof $n rc=$?"
  f=$(find /tmp/prof_$n -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/bench_${n}_kernel_stats.csv
}
prof overlap U2PL_DUMMY=1
prof serial U2PL_GRAPHS=0 U2PL_NO_SIDE_STREAM=1 U2PL_NO_WGRAD_STREAM=1
# 5
rm -rf /tmp/prof_dense
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dense -- python tools/dense_replay.py < /dev/null > $O/dense_replay.json 2> $O/dense.err; echo "dense rc=$?"
f=$(find /tmp/prof_dense -name '*kernel_trace.csv' | head -1)
python tools/parse_dense_trace.py "$f" $O/dense_replay.json $O/dense_replay_frac.json | tr -d '\n' | cut -c1-600; echo
python - "$f" $O/dense_replay_kernel_trace.csv <<'P'
import csv,sys,json,os
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_conv_igemm" in r["Kernel_Name"] or "k_igemm_ws" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
m=json.loads([l for l in open(os.path.dirname(sys.argv[2])+"/dense_replay.json").read().splitlines() if l.startswith("{")][-1])
last=rows[-int(m["kernel_launches"]):]
w=csv.writer(open(sys.argv[2],"w"))
w.writerow(["Kernel_Name","Start_Timestamp","End_Timestamp","Grid_Size","Workgroup_Size"])
for r in last: w.writerow([r["Kernel_Name"].split("(")[0][:60],r["Start_Timestamp"],r["End_Timestamp"],r.get("Grid_Size",""),r.get("Workgroup_Size","")])
P
# 6
rm -rf /tmp/pmc_mfma
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python tools/dense_replay.py < /dev/null > $O/pmc_mfma.json 2> $O/pmc_mfma.err; echo "pmc mfma rc=$?"
python - <<'P'
import csv,glob,collections,json,os
f=glob.glob("/tmp/pmc_mfma/*/*counter_collection.csv")
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6final"
if f:
    per=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if not any(s in k for s in ("k_conv_igemm","k_igemm_ws","k_conv_wgrad","k_wgrad_tr")): continue
        per[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE":
            cnt[k]+=1
            if r.get("End_Timestamp") and r.get("Start_Timestamp"): dur[k]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    out={}
    for k,v in per.items():
        busy,act=v.get("SQ_VALU_MFMA_BUSY_CYCLES",0),v.get("GRBM_GUI_ACTIVE",0)
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs' MFMA pipes (4 per CU x 256 CUs); GRBM_GUI_ACTIVE is summed over the 8 XCDs
        out[k]=dict(dispatches=cnt[k], mfma_busy_cycles=busy, gui_active_cycles=act, mfma_pipe_util=busy/(act/8*1024) if act else None,
                    sum_duration_ns=dur[k], shader_clock_GHz=(act/8)/dur[k] if dur[k] else None)
    json.dump(dict(json.load(open(O+"/STAMP.json")), kernels=out), open(O+"/pmc_mfma_busy.json","w"), indent=1)
    for k,v in sorted(out.items(), key=lambda kv:-kv[1]["gui_active_cycles"])[:8]: print(k[:60], {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
P
# 7
rm -rf /tmp/prof_lp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lp -- python tools/bench_loss_path.py < /dev/null > $O/lp.json 2> $O/lp.err; echo "lp rc=$?"
f=$(find /tmp/prof_lp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/loss_path_kernel_stats.csv
python - <<'P'
import csv,os
try:
    rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6final/loss_path_kernel_stats.csv")))
    tot=0
    for r in rows:
        n=r['Name']
        if any(k in n for k in ('k_reliability_fused','k_infonce','k_proto_stream','k_contra_classify_rows','k_bank_append_multi','k_scatter_rows_ordered','k_phase1_tail')):
            tot+=float(r['AverageNs'])/1e3
    print("loss-path group: sum of average kernel durations", round(tot,1), "us")
except Exception as e: print("lp parse failed", e)
P
# 9
timeout 200 python tools/host_overhead.py < /dev/null 2>&1 | grep -v amdgpu.ids > $O/host_overhead.txt; head -4 $O/host_overhead.txt
# 10
timeout 300 python bench.py --bf16 --crop 801 --steps 6 --warmup 3 --no-cpu-baseline --no-config5-leg < /dev/null > $O/bench_bf16_801.json 2> $O/bench_bf16_801.err; echo "bf16 rc=$?"
ls -la $O | head -40
