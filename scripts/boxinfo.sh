# Which kind of box is this?  (the pool's boxes differ by +-8 % in what the dominant kernel gets.)
# streaming patterns (column-fastest vs persistent runs), then the kernel itself: run plan x priority scheme x mapping.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-boxinfo}; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 scripts/adamprobe.hip -o /tmp/adamprobe 2>/dev/null && timeout 120 /tmp/adamprobe r 2>&1 | grep "64x128 (512 B segs), cols fastest, 2 WG\|64x128 (512 B segs), rows fastest, 2 WG\|persistent runs 64x128, S=6, interleaved tiles" > $O/adamprobe.txt
export DOM_QUICK=1
for rep in 1 2; do
for cfg in "bal:0" "bal:1" "runs6:0" "runs6:1"; do
  plan=${cfg%%:*}; pr=${cfg##*:}
  if [ $plan = runs6 ]; then export FX_FUSED_RUNS=6; else unset FX_FUSED_RUNS; fi
  FX_FUSED_PRIO=$pr timeout 200 python scripts/dom_diag.py 2>&1 | grep "operands random" | tail -1 > $O/diag_${plan}_prio${pr}_$rep.txt
done
done
unset FX_FUSED_RUNS
