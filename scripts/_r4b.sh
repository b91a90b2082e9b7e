cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_soak.py -x -q -m gpu -k "units_in_flight" > $O/pytest_a.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_soak.py::test_units_in_flight_never_capture_and_share_one_cohort -x --lf > $O/pytest_lf.txt 2>&1
