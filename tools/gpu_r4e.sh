cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4e
mkdir -p $O
timeout 300 python tools/bench_ws_pitch.py < /dev/null > $O/pitch.jsonl 2> $O/pitch.err; echo "rc=$?"
cat $O/pitch.jsonl; tail -n 3 $O/pitch.err
