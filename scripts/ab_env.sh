# Step time of one bench config under an environment switch, one process per setting, alternating, twice:  bash scripts/ab_env.sh cfg4 FX_BLOCK_BWD_KB=0
cd $GRAFT_REPO_ROOT
C=$1; shift
Q="--config $C --steps 60 --repeats 8 --no-cpu-baseline --sweep-trials-per-gpu 0 --no-other --no-pmc"
run() { python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['repeat_stats']; print('$1', d['value'], r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'])"; }
for i in 1 2; do
  run shipped
  env "$@" bash -c "$(declare -f run); Q='$Q'; run '$*'"
done
