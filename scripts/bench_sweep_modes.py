"""cfg5 leg on one GPU: trials one at a time with hipGraph replay (round 2), one at a time with eager launches, two / three in
flight on host threads (eager).   python scripts/bench_sweep_modes.py [n_trials]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexynesis_amd.sweep import run_cfg5
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
run_cfg5(dev, n_trials=2, epochs=1, in_flight=1)            # warm the process (code objects, torch's RNG kernels)
for rep in range(2):
    for infl, ug in ((1, True), (1, False), (2, False), (3, False)):
        out = run_cfg5(dev, n_trials=n, epochs=3, in_flight=infl, use_graph=ug, seed=0)
        print(f"in_flight {infl} graph {int(ug)}: {out['aggregate_samples_per_s']:9.1f} samples/s  wall {out['sweep_wall_s']:.3f} s  "
              f"best trial {out['best_trial']} val {out['best_val_loss']:.6f}  busy/wall {out['busy_over_wall']}", flush=True)
