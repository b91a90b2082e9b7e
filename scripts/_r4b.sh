cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
timeout 300 python scripts/dom_context.py > gpurun_out/r4m/context_new.txt 2>&1
FX_FUSED_MAP=1 FX_FUSED_PRIO=1 FX_FUSED_RUNS=6 timeout 300 python scripts/dom_context.py > gpurun_out/r4m/context_old.txt 2>&1
