cd $GRAFT_REPO_ROOT
Q="--config cfg2 --steps 60 --repeats 8 --no-cpu-baseline --sweep-trials-per-gpu 0 --no-other --no-pmc"
run() { python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['repeat_stats']; print('$1', d['value'], r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'])"; }
for i in 1 2; do
run shipped
FX_FORK_DEP_BEGIN=0 FX_FORK_AT_MARK=2 run dep_mark2
FX_FORK_DEP_BEGIN=0 FX_FORK_AT_MARK=5 run dep_mark5
FX_FORK_DEP_BEGIN=0 FX_FORK_AT_MARK=3 run dep_mark3
done
