cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; echo "all rc=$? $(grep -c . $O/pytest_all.txt)" >> $O/summary.txt
grep -n "passed\|failed" $O/pytest_all.txt | tail -2
cat $O/summary.txt
