"""Round 6: the VAE step with the supervisor heads on a graph branch of their OWN (FX_VAE_HEADS_BRANCH=2: the schedule that made
hipGraphLaunch segfault under torch.cuda.CUDAGraph in round 4) against the shipped schedule (heads inside decoder 0's branch), on the
library's own hipGraph capture (FX_GRAPH_BACKEND=fx) -- cfg3 step time, one process per setting, alternating; then the two test
sequences that used to crash, with the extra branch, under both graph backends.     python scripts/vae_heads_branch_ab.py"""
import json, os, subprocess, sys
Q = ["--config", "cfg3", "--steps", "40", "--repeats", "8", "--no-cpu-baseline", "--sweep-trials-per-gpu", "0", "--no-other", "--no-pmc"]
for name, env in (("heads in decoder 0's branch (shipped)", {}), ("heads on their own branch", {"FX_VAE_HEADS_BRANCH": "2"}),
                  ("shipped again", {}), ("own branch again", {"FX_VAE_HEADS_BRANCH": "2"}),
                  ("own branch, bf16 mode", {"FX_VAE_HEADS_BRANCH": "2", "_P": "bf16"}), ("shipped, bf16 mode", {"_P": "bf16"})):
    env = dict(env)
    prec = env.pop("_P", "bf16x3")
    r = subprocess.run([sys.executable, "bench.py", "--precision", prec] + Q, capture_output=True, text=True, env=dict(os.environ, **env))
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        rs = d["repeat_stats"]
        print(f"{name:42s} {d['value']:9.1f} samples/s  median {rs['ms_per_step_median']} ms  min {rs['ms_per_step_min']}  max {rs['ms_per_step_max']}  "
              f"launches {d['config']['launches_per_step']}", flush=True)
    except Exception as e:
        print(name, "FAILED rc", r.returncode, repr(e), r.stderr[-400:], flush=True)
sel = ["tests/test_gpu_parity.py", "tests/test_gpu_api.py", "-k", "vae or svae or crossmodal or level1 or drop_in or training_step", "-q", "-x", "-m", "gpu"]
for backend in ("fx", "torch"):
    env = dict(os.environ, FX_VAE_HEADS_BRANCH="2", FX_GRAPH_BACKEND=backend)
    r = subprocess.run([sys.executable, "-m", "pytest"] + sel, capture_output=True, text=True, env=env)
    tail = (r.stdout.strip().splitlines() or ["?"])[-1]
    print(f"VAE + level-1 test sequence with the heads branch, FX_GRAPH_BACKEND={backend}: rc {r.returncode}  {tail}", flush=True)
