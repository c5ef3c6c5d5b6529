# split-fp32 convolution: accuracy + speed per shape, then the headline bench in both modes (run under gpurun)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3s
mkdir -p $O
timeout 600 python tools/bench_conv_split.py > $O/conv_split.jsonl 2> $O/conv_split.err; echo "rc=$?"; tail -3 $O/conv_split.err
cat $O/conv_split.jsonl
for m in 1 0; do
  U2PL_CONV_SPLIT=$m timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>$O/bench_$m.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$m', d['ms_per_step'], d['value'], 'igemm', d['roofline']['frac'], d['roofline']['ms_per_step'], 'wgrad', d['roofline_wgrad']['frac'], d['roofline_wgrad']['ms_per_step'], 'losses', d['losses_last_step'])"
done
