cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_fwd.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
Q="--steps 20 --warmup 5 --no-cpu-baseline --repeats 3 --no-other"
for i in 1 2 3 4 5; do
python bench.py $Q 2>/dev/null | tail -1 > $O/new_$i.json
done
