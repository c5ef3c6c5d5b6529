cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4p
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu < /dev/null > $O/tests.log 2>&1; echo "tests rc=$?"
tail -n 30 $O/tests.log | cut -c1-300
grep -n "vs f64\|relative max error\|max error in eps32" $O/tests.log | head -40
