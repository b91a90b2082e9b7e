"""Where do the joules of the dominant kernel go?  The fused dW + Adam + forward launch (cfg2 shape, parity mode) and four ablated builds of it
(scripts/build_variant.py ablN fx_dw_adam_fwd.hip -DFT_ABL=N; results of the ablated builds are WRONG, only their time and power mean anything),
each looped for 5 s in a process of its own with rocm-smi sampled beside it: us per launch, package power, shader clock, and the launch's
energy = power x time.  Differences against build 4 (no GEMM work at all: LDS transposes + the W / m / v stream) price the operand LDS-DMA
(L2 -> LDS), the MFMAs with their fragment reads, and the `lo` fragment reads alone.
    for n in 3 4 5 6; do python scripts/build_variant.py abl$n fx_dw_adam_fwd.hip -DFT_ABL=$n; done      (on the build host)
    python scripts/fused_energy_ablation.py                                                              (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import subprocess, sys, threading, time
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
from flexynesis_amd.engine import PartitionArena
dev = torch.device("cuda:0")
ar = PartitionArena.get(dev)
N, K, B = 5000, 20000, 128
ld = ops.pad32(K)
(w, m, v), _ = ar.take3(N * ld)
W, m, v = [t.view(N, ld)[:, :K] for t in (w, m, v)]
W.normal_(0, 0.01)
dy = torch.randn(B, N, device=dev) * 1e-2; x = torch.randn(B, K, device=dev)
dyt, xt = ops.new_split(N, B, dev), ops.new_split(K, B, dev)
ops.split_bf16_t(ops.IMMEDIATE, dyt[0], dyt[1], dy); ops.split_bf16_t(ops.IMMEDIATE, xt[0], xt[1], x)
xnh, xnl = ops.new_split_kb(B, K, dev); ops.split_bf16(ops.IMMEDIATE, xnh, xnl, x)
slabs = torch.zeros(16, B, N, device=dev)
ctrl = torch.zeros(64, device=dev); ctrl[0] = 9.0
ops.step_begin(ops.IMMEDIATE, ctrl, 1e-3); ctrl[4] = 0.5
products = int(sys.argv[1])
rec = ops.TapeRecorder(products=products)
ops.linear_dw_adam_fwd_bf16x3(rec, W, m, v, dyt[0], dyt[1], xt[0], xt[1], ctrl, xnh, xnl, B, slabs)
fn = rec.run
samples, stop = [], [False]
def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = [l for l in out.splitlines() if "Power" in l and "W" in l]
            ck = [l for l in out.splitlines() if "sclk" in l]
            samples.append((float(pw[0].split(":")[-1].strip()), ck[0].split("(")[-1].split("M")[0]))
        except Exception:
            pass
        time.sleep(0.5)
th = threading.Thread(target=sampler); th.start()
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time(); n = 0; e0.record()
while time.time() - t0 < 5.0:
    for _ in range(20): fn()
    n += 20; torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop[0] = True; th.join()
us = e0.elapsed_time(e1) / n * 1e3
ws = [p for p, _ in samples[2:9]]
print(us, sum(ws) / max(len(ws), 1), samples[4][1] if len(samples) > 4 else "?")
'''
rows = []
for name, lib, products, what in (("shipped", None, 3, "everything (parity mode: three products)"),
                                  ("abl6", "abl6", 3, "MFMAs without the `lo` fragment reads"),
                                  ("abl7", "abl7", 3, "products ordered so that consecutive MFMAs share an operand (correct results)"),
                                  ("shipped", None, 3, "everything, again"),
                                  ("abl3", "abl3", 3, "no MFMAs / fragment reads; operand LDS-DMA stays"),
                                  ("abl5", "abl5", 3, "no operand LDS-DMA; MFMAs + fragment reads on stale LDS"),
                                  ("abl4", "abl4", 3, "no GEMM work: LDS transposes + W / m / v stream"),
                                  ("plain", None, 1, "plain-bf16 mode (one product), shipped build")):
    only = os.environ.get("ABL_ONLY")            # e.g. ABL_ONLY=shipped,abl7: just these rows (in the table's order)
    if only and name not in only.split(","):
        continue
    env = dict(os.environ)
    if lib:
        env["FXHIP_LIB"] = os.path.join(ROOT, "build_tmp", f"libfxhip_{lib}.so")
    r = subprocess.run([sys.executable, "-c", CHILD, str(products)], cwd=ROOT, env=env, capture_output=True, text=True)
    try:
        us, w, clk = r.stdout.strip().splitlines()[-1].split()
        us, w = float(us), float(w)
        rows.append((name, what, us, w, clk, us * 1e-6 * w))
        print(f"{name:8s} {us:7.1f} us  {w:7.1f} W  {clk:>5s} MHz  {us * 1e-6 * w:6.3f} J per launch   {what}", flush=True)
    except Exception as e:
        print(name, "FAILED", r.returncode, repr(e), r.stderr[-400:], flush=True)
base = {r[0]: r for r in rows}
if "abl4" in base:
    e4 = base["abl4"][5]
    for k, label in (("abl3", "operand LDS-DMA (L2 -> LDS: 1.6 GB per launch)"), ("abl5", "MFMAs + fragment reads (156 GFLOP)"), ("shipped", "both GEMM phases")):
        if k in base:
            print(f"  {label:55s} {base[k][5] - e4:+.3f} J  ({base[k][2] - base['abl4'][2]:+.1f} us)")
    if "abl6" in base and "shipped" in base:
        print(f"  {'the `lo` fragment reads alone (shipped - abl6)':55s} {base['shipped'][5] - base['abl6'][5]:+.3f} J  ({base['shipped'][2] - base['abl6'][2]:+.1f} us)")
