"""cfg3 step time under an environment switch, in ONE process per setting, alternating:
    python scripts/bench_cfg3_ab.py [SWITCH=VALUE ...]        (default: FX_VAE_PARTIAL_JOIN=0 against the shipped schedule, both modes)"""
import json, os, subprocess, sys
sw = dict(a.split("=", 1) for a in sys.argv[1:]) or {"FX_VAE_PARTIAL_JOIN": "0"}
tag = " ".join(f"{k}={v}" for k, v in sw.items())
for prec in ("bf16x3", "bf16"):
    for name, env in (("shipped", {}), (tag, sw), ("shipped again", {}), (tag + " again", sw)):
        r = subprocess.run([sys.executable, "bench.py", "--config", "cfg3", "--precision", prec, "--steps", "40", "--repeats", "8", "--no-cpu-baseline",
                            "--sweep-trials-per-gpu", "0", "--no-other", "--no-pmc"], capture_output=True, text=True, env=dict(os.environ, **env))
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rs = d["repeat_stats"]
            print(f"{prec:7s} {name:40s} {d['value']:9.1f} samples/s  median {rs['ms_per_step_median']} ms  min {rs['ms_per_step_min']}  max {rs['ms_per_step_max']}", flush=True)
        except Exception as e:
            print(prec, name, "FAILED", r.returncode, repr(e), r.stderr[-300:], flush=True)
