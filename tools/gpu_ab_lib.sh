# A/B of library build variants on the headline bench (run under gpurun): default, HBM kernels capped at 4 / 2 waves per SIMD
cd $GRAFT_REPO_ROOT
for v in "" hbm4 hbm2 ""; do
  if [ -n "$v" ]; then export U2PL_LIB_PATH=$GRAFT_REPO_ROOT/u2pl_amd/lib/variants/libu2pl_hip_$v.so; else unset U2PL_LIB_PATH; fi
  timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant=${v:-default}', d['ms_per_step'], d['value'], 'igemm', d['roofline']['frac'], 'hbm', d['roofline_hbm']['frac'], {k: d['kernel_ms_per_step'].get(k) for k in ('u2pl_bn_apply_f32','u2pl_bn_bwd_apply_f32','u2pl_wino_input_f32','u2pl_bn_bwd_sums_f32','u2pl_wino_output_f32')})"
done
