"""What happens on ROCm 7.x / torch 2.10 when a torch.cuda.CUDAGraph object DIES (a) while another graph is being captured,
(b) right after its own replay was launched (no synchronisation), (c) while another graph is replaying -- the candidates for
the intermittent SIGSEGV inside hipGraphLaunch of round 2.  Each case runs in its own subprocess so a
crash is observed, not suffered.   python scripts/graph_lifetime_probe.py"""
import os
import subprocess
import sys

CASES = {
    "a_destroy_other_graph_during_capture": """
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=side):
    y.copy_(x * 2)
g1.replay(); torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph()
err = None
try:
    with torch.cuda.graph(g2, stream=side):
        y.copy_(x * 3)
        del g1                      # the last reference dies INSIDE the capture of g2 (what a cyclic-GC pass can do)
        y.add_(1)
except Exception as e:
    err = repr(e)[:200]
print("capture error:", err)
if err is None:
    for _ in range(50):
        g2.replay()
    torch.cuda.synchronize()
    print("replayed ok, y[0] =", float(y[0]))
""",
    "b_destroy_right_after_replay_no_sync": """
for it in range(300):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(20):
            big.mul_(1.0001)
    g.replay()
    del g                            # destroyed while its kernels may still be running
torch.cuda.synchronize()
print("ok", float(big[0]))
""",
    "c_destroy_while_other_graph_replays": """
gs = []
for k in range(4):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(20):
            big.mul_(1.0001)
    gs.append(g)
for it in range(200):
    gs[it % 3].replay()
    if it % 10 == 5:
        gs[3] = None                 # a replaying process destroys an unrelated graph and captures a new one
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(20):
                big.mul_(1.0001)
        gs[3] = g
torch.cuda.synchronize()
print("ok", float(big[0]))
""",
}
PRE = """
import torch, gc
dev = torch.device("cuda:0")
x = torch.ones(1 << 20, device=dev); y = torch.zeros(1 << 20, device=dev); big = torch.ones(1 << 26, device=dev)
side = torch.cuda.Stream()
torch.cuda.synchronize()
"""
if __name__ == "__main__":
    for name, body in CASES.items():
        r = subprocess.run([sys.executable, "-c", PRE + body], capture_output=True, text=True, timeout=300)
        tail = (r.stdout.strip().splitlines() or [""])[-2:]
        errt = [l for l in r.stderr.strip().splitlines() if "amdgpu.ids" not in l][-3:]
        print(f"[{name}] rc={r.returncode} stdout={tail} stderr_tail={errt}", flush=True)
