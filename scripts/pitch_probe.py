"""Does the ROW PITCH of a wide weight decide its stream rate?  The K = 384 dW + Adam launch of cfg4 ([7500, 30000], pitch 30016 floats) streams
4.6-5.0 TB/s where cfg2's [10000, 20000] (pitch 20000) streams 5.8.  For each pitch: fresh W / m / v arrays (earlier ones kept alive, so every
trial is another placement), the unfused dW + Adam kernel timed on them and the placement probe.   python scripts/pitch_probe.py [N K B]"""
import sys
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
N, K, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (7500, 30000, 384)
dy = torch.randn(B, N, device=dev) * 1e-2; x = torch.randn(B, K, device=dev)
dyt, xt = ops.new_split(N, B, dev), ops.new_split(K, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
ctrl = torch.zeros(64, device=dev); ctrl[0] = 9.0
ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3); ctrl[4] = 0.5
def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)
keep = []
base = (K + 31) // 32 * 32
for pitch in [base, base + 32, base + 64, base + 128, base + 256, base + 1024, (K + 1023) // 1024 * 1024]:
    rates, probes = [], []
    for trial in range(4):
        arrs = [torch.zeros(N, pitch, device=dev) for _ in range(3)]
        keep.append(arrs if trial % 2 else None)          # hold every other set: the next allocation lands elsewhere
        W, m, v = (a[:, :K] for a in arrs)
        W.normal_(0, 0.01)
        t = timeit(lambda: ops.linear_dw_adam_bf16x3(ops.IMMEDIATE, W, m, v, dyt[0], dyt[1], xt[0], xt[1], ctrl))
        rates.append(24.0 * N * K / t / 1e6)
        probes.append(24.0 * N * K / ops.placement_probe_us(W, m, v) / 1e6)
    print(f"pitch {pitch:6d} floats ({pitch * 4 % 4096:4d} mod 4096 B)   dW + Adam TB/s " + " ".join(f"{r:5.2f}" for r in rates) + "   probe TB/s " + " ".join(f"{r:5.2f}" for r in probes), flush=True)
    if len(keep) > 12:
        keep = keep[-8:]
