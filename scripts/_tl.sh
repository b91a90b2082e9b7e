cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3m
for c in cfg3 cfg4; do
python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline --sweep-trials-per-gpu 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], d['value'])"
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/r3m/tr$c -- python bench.py --config $c --steps 30 --warmup 6 --no-cpu-baseline --sweep-trials-per-gpu 0 > /dev/null 2>&1; DB=$(find gpurun_out/r3m/tr$c -name "*.db" | head -1); python scripts/rocpd_timeline.py $DB 20 > gpurun_out/r3m/timeline_$c.txt; rm -rf gpurun_out/r3m/tr$c
done
