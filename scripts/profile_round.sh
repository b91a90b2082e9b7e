set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
for c in cfg2 cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -- python bench.py --config $c --no-cpu-baseline --sweep-trials-per-gpu 0 > $O/bench_prof_$c.json 2> $O/bench_prof_$c.err
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --sweep-trials-per-gpu 0 --no-graph > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --sweep-trials-per-gpu 0 --no-graph > $O/pmc_write.json 2> $O/pmc_write.err
du -sh $O/*; find $O -name "*.csv" | head -30
