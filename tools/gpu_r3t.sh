# A/B of environment variants on the headline bench (run under gpurun): VARIANTS="A=1 B=2|C=3" -> one bench per |-separated env set
cd $GRAFT_REPO_ROOT
IFS='|' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  env $v timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['ms_per_step'], d['value'], 'igemm', d['roofline']['frac'], d['roofline']['ms_per_step'], 'wgrad', d['roofline_wgrad']['frac'], d['roofline_wgrad']['ms_per_step'], 'hbm', d['roofline_hbm']['frac'])"
done
