cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
