# One config's step timeline + per-kernel averages under rocprofv3 (kernel trace only): the narrow-chain view of a change.
# usage (on the GPU box): bash scripts/chain_timeline.sh <tag> [config] [precision]   -> gpurun_out/<tag>/timeline_<config>.txt, stats_<config>.md
R=$GRAFT_REPO_ROOT; T=${1:-chain}; C=${2:-cfg2}; P=${3:-bf16x3}; O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
cd $R
Q="--no-cpu-baseline --sweep-trials-per-gpu 0 --no-other --repeats 0 --no-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$C -- python bench.py --config $C --precision $P --steps 40 --warmup 5 $Q > $O/bench_prof_$C.json 2> $O/bench_prof_$C.err
python scripts/stats_to_md.py $(find $O/trace_$C -name "*kernel_stats.csv" | head -1) 20 > $O/stats_$C.md
python scripts/trace_timeline.py $(find $O/trace_$C -name "*kernel_trace.csv" | head -1) 30 > $O/timeline_$C.txt
rm -rf $O/trace_$C
head -40 $O/timeline_$C.txt
