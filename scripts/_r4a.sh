set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 scripts/tlbprobe.hip -o /tmp/tlbprobe && timeout 120 /tmp/tlbprobe > gpurun_out/r4a/tlbprobe.txt 2>&1
hipcc --offload-arch=gfx950 -O3 scripts/adamprobe.hip -o /tmp/adamprobe && timeout 120 /tmp/adamprobe r > gpurun_out/r4a/adamprobe.txt 2>&1
timeout 300 python scripts/dom_diag.py > gpurun_out/r4a/dom_diag.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench.txt 2>&1
FX_FUSED_MAP=3 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sweep-trials-per-gpu 0 > gpurun_out/r4a/bench_map3.txt 2>&1
rocm-smi --showclocks > gpurun_out/r4a/smi.txt 2>&1
cat /sys/module/amdgpu/parameters/vm_fragment_size /sys/module/amdgpu/parameters/vm_block_size > gpurun_out/r4a/vmparams.txt 2>&1
uname -r >> gpurun_out/r4a/vmparams.txt; cat /sys/module/amdgpu/version >> gpurun_out/r4a/vmparams.txt 2>&1
