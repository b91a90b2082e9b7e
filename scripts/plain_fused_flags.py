"""Round 6: the fused dW + Adam + forward kernel in the plain-bf16 mode is not power-bound (profiles/r06_power_under_kernels.txt), so its
scheduling switches -- workgroup mapping (FX_FUSED_MAP), issue-priority scheme (FX_FUSED_PRIO), runs per row block (FX_FUSED_RUNS) -- are
re-measured there (in the parity mode they were within noise of each other).  One bench.py process per setting, cfg2.
    python scripts/plain_fused_flags.py"""
import json, os, subprocess, sys
Q = ["--config", "cfg2", "--precision", "bf16", "--steps", "40", "--repeats", "6", "--no-cpu-baseline", "--sweep-trials-per-gpu", "0", "--no-other", "--no-pmc"]
for env in ({}, {"FX_FUSED_PRIO": "1"}, {"FX_FUSED_PRIO": "2"}, {"FX_FUSED_PRIO": "3"}, {"FX_FUSED_MAP": "1"}, {"FX_FUSED_MAP": "2"}, {"FX_FUSED_MAP": "3"},
            {"FX_FUSED_RUNS": "5"}, {"FX_FUSED_RUNS": "7"}, {"FX_NT_ADAM": "0"}, {}):
    r = subprocess.run([sys.executable, "bench.py"] + Q, capture_output=True, text=True, env=dict(os.environ, **env))
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        rs = d["repeat_stats"]
        print(f"{str(env) or 'shipped':28s} {d['value']:9.1f} samples/s  median {rs['ms_per_step_median']} ms  min {rs['ms_per_step_min']}  dominant launch {d['roofline']['avg_launch_ms']} ms", flush=True)
    except Exception as e:
        print(env, "FAILED", r.returncode, repr(e), r.stderr[-300:], flush=True)
