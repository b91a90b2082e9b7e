cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 300 python /root/repo/scripts/_dbg_ft.py > $O/dbg.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_trial_loop.py -x -q -m gpu > $O/pytest.txt 2>&1
