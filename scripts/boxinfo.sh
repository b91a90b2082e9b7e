# Which kind of box is this?  (the pool's boxes differ by +-8 % in what the dominant kernel gets.)
# address-translation reach, streaming patterns (column-fastest vs persistent runs), then the kernel itself old / new mapping.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-boxinfo}; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 scripts/tlbprobe.hip -o /tmp/tlbprobe 2>/dev/null && timeout 120 /tmp/tlbprobe > $O/tlbprobe.txt 2>&1
hipcc --offload-arch=gfx950 -O3 scripts/adamprobe.hip -o /tmp/adamprobe 2>/dev/null && timeout 120 /tmp/adamprobe r 2>&1 | grep "64x128 (512 B segs), cols fastest, 2 WG\|64x128 (512 B segs), rows fastest, 2 WG\|persistent runs 64x128, S=6, interleaved\|4x2048 (8 KB segs), cols fastest, 2 WG" > $O/adamprobe.txt
DOM_QUICK=1 timeout 200 python scripts/dom_diag.py > $O/diag_new.txt 2>&1
DOM_QUICK=1 FX_FUSED_PRIO=1 FX_FUSED_RUNS=6 timeout 200 python scripts/dom_diag.py > $O/diag_old.txt 2>&1
rocm-smi --showtemp --showpower --showclocks > $O/smi.txt 2>&1
rocm-smi --showmeminfo vram --showmemvendor >> $O/smi.txt 2>&1
