#!/bin/bash
# Run in the build container at a CLEAN tree: stamps the commit, runs tools/gpu_r6_profiles.sh on the GPU box, copies the
# artefacts into profiles/r06_* (every JSON carries / sits next to r06_STAMP.json = {commit, kernel_sources_sha}).
set -e
cd "$(dirname "$0")/.."
git rev-parse --short HEAD > gpurun_head.txt
/usr/local/graft/bin/gpurun --timeout ${1:-2100} -- 'bash tools/gpu_r6_profiles.sh' 2>&1 | tail -60
S=gpurun_out/r6final
# (a refused / transient gpurun call leaves the previous run's files in place: copy nothing unless the stamp is this commit's)
if ! grep -q "\"$(cat gpurun_head.txt)\"" $S/STAMP.json 2>/dev/null; then echo "no fresh results for $(cat gpurun_head.txt): nothing copied"; exit 3; fi
cp $S/STAMP.json profiles/r06_STAMP.json
for f in bench_plain.json bench_overlap_under_rocprof.json bench_serial_under_rocprof.json bench_overlap_kernel_stats.csv bench_serial_kernel_stats.csv \
         dense_replay.json dense_replay_frac.json dense_replay_kernel_trace.csv pmc_mfma_busy.json loss_path_kernel_stats.csv traffic.json \
         host_overhead.txt bench_bf16_801.json full_size_parity.json miou_gate.json; do
  [ -s $S/$f ] && cp $S/$f profiles/r06_$f
done
tail -5 $S/tests.log > profiles/r06_gpu_tests_tail.txt
ls -la profiles | grep r06_
