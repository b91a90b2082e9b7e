# Step time of one bench config under A/B builds of the library (scripts/build_variant.py), ONE process per build, alternating, twice:
#   bash scripts/ab_libs.sh cfg2 gs16 gs32 ...        ("shipped" = the in-tree library, always first)
cd $GRAFT_REPO_ROOT
C=$1; shift
Q="--config $C --steps 60 --repeats 8 --no-cpu-baseline --sweep-trials-per-gpu 0 --no-other --no-pmc"
run() { python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['repeat_stats']; print('$1', d['value'], r['ms_per_step_median'], r['ms_per_step_min'], r['ms_per_step_max'])"; }
for i in 1 2; do
  unset FXHIP_LIB; run shipped
  for v in "$@"; do FXHIP_LIB=build_tmp/libfxhip_$v.so run $v; done
done
