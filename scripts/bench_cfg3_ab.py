"""cfg3 step time under an environment switch, in ONE process per setting (placement search on): python scripts/bench_cfg3_ab.py"""
import json, os, subprocess, sys
for name, env in (("default", {}), ("FX_VAE_MMD_LATE=0", {"FX_VAE_MMD_LATE": "0"}), ("default again", {})):
    r = subprocess.run([sys.executable, "bench.py", "--config", "cfg3", "--steps", "40", "--repeats", "5", "--no-cpu-baseline", "--sweep-trials-per-gpu", "0",
                        "--no-other", "--no-pmc"], capture_output=True, text=True, env=dict(os.environ, **env))
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(name, d["value"], d["repeat_stats"])
