"""Step time of one bench config under an environment switch, ONE process per setting, alternating, in both contraction modes:
    python scripts/bench_ab.py cfg4 FX_FORK_AFTER_WIDE=0 [MORE=SWITCHES ...]"""
import json, os, subprocess, sys
cfg = sys.argv[1]
sw = dict(a.split("=", 1) for a in sys.argv[2:])
tag = " ".join(f"{k}={v}" for k, v in sw.items())
for prec in ("bf16x3", "bf16"):
    for name, env in (("shipped", {}), (tag, sw), ("shipped again", {}), (tag + " again", sw)):
        r = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--precision", prec, "--steps", "40", "--repeats", "8", "--no-cpu-baseline",
                            "--sweep-trials-per-gpu", "0", "--no-other", "--no-pmc"], capture_output=True, text=True, env=dict(os.environ, **env))
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rs = d["repeat_stats"]
            print(f"{cfg} {prec:7s} {name:40s} {d['value']:9.1f} samples/s  median {rs['ms_per_step_median']} ms  min {rs['ms_per_step_min']}  max {rs['ms_per_step_max']}", flush=True)
        except Exception as e:
            print(cfg, prec, name, "FAILED", r.returncode, repr(e), r.stderr[-300:], flush=True)
