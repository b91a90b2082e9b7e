"""The cfg5 sweep leg (8 DirectPred trials, 4 in flight) three times in one process, with the partition arena's state after each: pools
free, chunks grown, leases outstanding -- does the arena keep up with trials that come and go?     python scripts/sweep_arena_check.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from flexynesis_amd import ops
from flexynesis_amd.engine import PartitionArena, placement_memory
from flexynesis_amd.sweep import run_cfg5
dev = torch.device("cuda:0")
run_cfg5(dev, n_trials=1, epochs=1, features=20000, samples=2048, seed=1, keep_winner=False)
for rep in range(3):
    t0 = time.time()
    s = run_cfg5(dev, n_trials=8, epochs=3, features=20000, samples=2048, seed=0)
    ar = PartitionArena._arenas.get(0)
    print(f"rep {rep}: {s['aggregate_samples_per_s']} samples/s, wall {s['sweep_wall_s']} s, busy {s['rank_busy_s']}, leases outstanding {ops.LEASES.outstanding()}, "
          f"arena {placement_memory(dev)}, classes {ar.info.get('pool_B_classes') if ar else None}, grown {ar.info.get('grown') if ar else None}", flush=True)
