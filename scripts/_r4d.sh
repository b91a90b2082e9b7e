cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_production.py -x -q -m gpu -k "triplet or Triplet or cfg4 or random" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
Q="--steps 30 --warmup 10 --no-cpu-baseline --sweep-trials-per-gpu 0 --no-other --repeats 5 --config cfg4"
for rep in 1 2 3; do
python bench.py $Q 2>/dev/null | tail -1 > $O/cfg4_new_$rep.json
FX_GRAM_KB_WIDE=0 python bench.py $Q 2>/dev/null | tail -1 > $O/cfg4_old_$rep.json
done
