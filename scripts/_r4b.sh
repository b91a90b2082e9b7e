cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multiproc.py -x -q -m gpu > $O/pytest_multiproc.txt 2>&1
