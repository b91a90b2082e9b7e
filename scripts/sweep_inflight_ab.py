"""cfg5 leg on one GPU (8 trials, 3 epochs) by the number of trials in flight: python scripts/sweep_inflight_ab.py"""
import json, os, subprocess, sys
for nf, graph in ((1, True), (1, False), (2, False), (3, False), (4, False), (2, False)):
    code = ("import sys, json, torch; sys.path.insert(0, '.'); from flexynesis_amd.sweep import run_cfg5; dev = torch.device('cuda:0'); torch.cuda.set_device(0);"
            f"run_cfg5(dev, n_trials=1, epochs=1, keep_winner=False, in_flight={nf}, use_graph={graph}); r = run_cfg5(dev, n_trials=8, epochs=3, in_flight={nf}, use_graph={graph});"
            "print(json.dumps({k: r[k] for k in ('aggregate_samples_per_s', 'sweep_wall_s', 'busy_over_wall')}))")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(f"in_flight {nf} graph {graph}:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
