# Round-end evidence: bench line, rocprofv3 kernel-trace stats per config, PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes).
# usage (on the GPU box): bash scripts/profile_round.sh <tag>     -> gpurun_out/<tag>/
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06}; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
bash scripts/boxinfo.sh ${1:-r06}/box > /dev/null 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
Q="--no-cpu-baseline --sweep-trials-per-gpu 0 --no-other --repeats 0 --no-pmc"
for c in cfg2 cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -- python bench.py --config $c --steps 40 --warmup 5 $Q > $O/bench_prof_$c.json 2> $O/bench_prof_$c.err
done
for c in cfg2 cfg3 cfg4; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$c -- python bench.py --config $c --steps 6 --warmup 2 --no-graph $Q > $O/pmc_fetch_$c.json 2> $O/pmc_fetch_$c.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$c -- python bench.py --config $c --steps 6 --warmup 2 --no-graph $Q > $O/pmc_write_$c.json 2> $O/pmc_write_$c.err
  cp $(find $O/pmc_fetch_$c -name "*counter_collection.csv" | head -1) $O/pmc_fetch_$c.csv
  cp $(find $O/pmc_write_$c -name "*counter_collection.csv" | head -1) $O/pmc_write_$c.csv
  rm -rf $O/pmc_fetch_$c $O/pmc_write_$c
done
# the plain-bf16 throughput mode (round 6): kernel traces only
for c in cfg2 cfg3 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${c}_bf16 -- python bench.py --config $c --precision bf16 --steps 40 --warmup 5 $Q > $O/bench_prof_${c}_bf16.json 2> $O/bench_prof_${c}_bf16.err
done
for c in cfg2 cfg3 cfg4 cfg2_bf16 cfg3_bf16 cfg4_bf16; do
  cp $(find $O/trace_$c -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$c.csv
  cp $(find $O/trace_$c -name "*kernel_trace.csv" | head -1) $O/kernel_trace_$c.csv
  rm -rf $O/trace_$c
done
ls -la $O
# small, committable summaries (profiles/<tag>_*): top-kernel tables, one step's launch timeline, PMC traffic per launch
T=${1:-r06}
for c in cfg2 cfg3 cfg4 cfg2_bf16 cfg3_bf16 cfg4_bf16; do
  python scripts/stats_to_md.py $O/kernel_stats_$c.csv 14 > $O/${T}_a_kernel_stats_$c.md
  python scripts/trace_timeline.py $O/kernel_trace_$c.csv 30 > $O/${T}_timeline_$c.txt
done
for c in cfg2 cfg3 cfg4; do
  python scripts/pmc_to_json.py $O/pmc_fetch_$c.csv $O/pmc_write_$c.csv $O/${T}_pmc_traffic_$c.json > $O/pmc_$c.txt 2>&1
done
rm -f $O/kernel_trace_*.csv
ls -la $O
