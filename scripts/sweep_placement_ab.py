"""cfg5 leg on one GPU (8 trials, 3 epochs) under different placement policies for the trials' wide weights, one process each:
python scripts/sweep_placement_ab.py"""
import json, os, subprocess, sys
code = ("import sys, json, torch; sys.path.insert(0, '.'); from flexynesis_amd.sweep import run_cfg5; dev = torch.device('cuda:0'); torch.cuda.set_device(0);"
        "run_cfg5(dev, n_trials=1, epochs=1, keep_winner=False); r = run_cfg5(dev, n_trials=8, epochs=3);"
        "print(json.dumps({k: r[k] for k in ('aggregate_samples_per_s', 'sweep_wall_s', 'busy_over_wall')}))")
for name, env in (("first placement (default)", {}), ("trial search: 2 x 8 candidates, 30 ms", {"FX_PLACEMENT_TRIES_TRIAL": "2", "FX_PLACEMENT_BUDGET_S": "0.03"}),
                  ("trial search: 4 x 8 candidates, 60 ms", {"FX_PLACEMENT_TRIES_TRIAL": "4", "FX_PLACEMENT_BUDGET_S": "0.06"}),
                  ("first placement (default) again", {})):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env))
    print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
