"""Is the stacked-rows forward (M = 384, [7500, 30000]) limited by its arithmetic or by the package power?  Same kernel, same shapes:
random operands, W = 0, X = 0 (operand values decide the switching power of the MFMA, not its cycle count), and -- with a rocm-smi
sampler beside a 6 s loop -- power and shader clock.   python scripts/fwd_mt_power.py"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
dev = torch.device("cuda:0")
M, K, N = 384, 30000, 7500
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.01; b = torch.zeros(N, device=dev)
Y = torch.empty(M, N, device=dev); ws = ops.Workspace(dev)
def split(x):
    s = ops.new_split_kb(M, K, dev); ops.split_bf16(ops.IMMEDIATE, s[0], s[1], x); return s
xs, x0 = split(X), split(torch.zeros_like(X))
W0 = torch.zeros_like(W)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (s, w) in (("random X, random W", (xs, W)), ("random X, W = 0", (xs, W0)), ("X = 0, random W", (x0, W)), ("X = 0, W = 0", (x0, W0))):
    print(f"{name:22s} {timeit(lambda: ops.linear_fwd_bf16x3(ops.IMMEDIATE, Y, s[0], s[1], w, b, ws)):7.1f} us")
samples = []
stop = [False]
def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            pw = [l for l in out.splitlines() if "Power" in l and "W" in l]
            ck = [l for l in out.splitlines() if "sclk" in l]
            samples.append((pw[0].split(":")[-1].strip() if pw else "?", ck[0].split(":")[-1].strip() if ck else "?"))
        except Exception as e:
            samples.append((repr(e), ""))
        time.sleep(0.5)
th = threading.Thread(target=sampler); th.start()
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(50): ops.linear_fwd_bf16x3(ops.IMMEDIATE, Y, xs[0], xs[1], W, b, ws)
    torch.cuda.synchronize()
stop[0] = True; th.join()
print("under load (random operands):", samples[2:10])
