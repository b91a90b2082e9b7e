# Is the dominant kernel POWER-limited?  Package power / clocks / temperatures while (a) libfxhip's 16-byte copy, (b, c) the dW + Adam traffic
# pattern without GEMM (column-fastest grid, persistent runs), (d) the dominant kernel itself run back to back for a few seconds each.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-clocks}; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 scripts/adamprobe.hip -o /tmp/adamprobe 2>/dev/null
sample() {   # $1 = tag; samples 4 times, 1 s apart, starting 3 s into the load
  sleep 3
  for i in 1 2 3 4; do
    echo "$1 $(rocm-smi --showpower --showtemp --showclocks 2>/dev/null | grep -i 'Package Power\|memory) (C)\|junction\|sclk\|fclk' | sed 's/.*: //' | tr '\n' ' ')" >> $O/samples.txt
    echo "$1 sclk per XCD: $(cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | grep '\*' | sed 's/.*: //;s/ \*//' | tr '\n' ' ')" >> $O/samples.txt
    sleep 1
  done
}
cat > /tmp/loop.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
what, secs = sys.argv[1], float(sys.argv[2])
n_out, k_in, B = 5000, 20000, 128
g = torch.Generator(device=dev); g.manual_seed(1)
if what == "copy":
    src = torch.randn(1 << 28, generator=g, device=dev); dst = torch.empty_like(src)
    fn = lambda: ops.stream_copy(ops.IMMEDIATE, dst, src)
    nbytes = 2 * src.numel() * 4
else:
    ldw = ops.pad32(k_in)
    ctrl = torch.zeros(64, device=dev); ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3)
    S = ops.dw_adam_fwd_slabs(n_out, k_in)
    slabs = torch.zeros(S, B, n_out, device=dev)
    zero = what == "kernel_zero_operands"
    dy = torch.zeros(B, n_out, device=dev) if zero else torch.randn(B, n_out, generator=g, device=dev) * 1e-2
    x = torch.zeros(B, k_in, device=dev) if zero else torch.randn(B, k_in, generator=g, device=dev)
    dyt, xt = ops.new_split(n_out, B, dev), ops.new_split(k_in, B, dev)
    ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
    xnh, xnl = ops.new_split_kb(B, k_in, dev); ops.split_bf16(ops.IMMEDIATE, xnh, xnl, x)
    W = torch.randn(n_out, ldw, generator=g, device=dev) / k_in ** 0.5
    m = torch.randn(n_out, ldw, generator=g, device=dev) * 1e-3
    v = torch.rand(n_out, ldw, generator=g, device=dev) * 1e-5
    fn = lambda: ops.linear_dw_adam_fwd_bf16x3(ops.IMMEDIATE, W[:, :k_in], m[:, :k_in], v[:, :k_in], dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, B, slabs)
    nbytes = 24 * n_out * k_in
t_end = time.time() + secs
n, t0 = 0, time.time()
while time.time() < t_end:
    for _ in range(50):
        fn()
    torch.cuda.synchronize(); n += 50
dt = (time.time() - t0) / n
print(f"{what}: {n} launches, {dt * 1e6:.1f} us per launch, {nbytes / dt / 1e12:.2f} TB/s")
PY
echo "idle $(rocm-smi --showpower --showtemp 2>/dev/null | grep -i 'Package Power\|memory) (C)\|junction' | sed 's/.*: //' | tr '\n' ' ')" > $O/samples.txt
python /tmp/loop.py copy 8 > $O/loop_copy.txt 2>&1 & sample copy; wait
/tmp/adamprobe r loop cols 8 > $O/loop_cols.txt 2>&1 & sample adam_cols_fastest; wait
/tmp/adamprobe r loop pers 8 > $O/loop_pers.txt 2>&1 & sample adam_persistent; wait
python /tmp/loop.py kernel 8 > $O/loop_kernel.txt 2>&1 & sample kernel; wait
if [ -f $PWD/build_tmp/libfxhip_exact.so ]; then
FXHIP_LIB=$PWD/build_tmp/libfxhip_exact.so python /tmp/loop.py kernel 8 > $O/loop_kernel_exact.txt 2>&1 & sample kernel_exact_adam; wait
fi
if [ -f $PWD/build_tmp/libfxhip_ieee.so ]; then
FXHIP_LIB=$PWD/build_tmp/libfxhip_ieee.so python /tmp/loop.py kernel 8 > $O/loop_kernel_ieee.txt 2>&1 & sample kernel_ieee_adam; wait
fi
cat $O/loop_*.txt | grep -v amdgpu > $O/rates.txt
