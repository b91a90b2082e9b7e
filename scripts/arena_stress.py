"""PartitionArena under churn: 80 stores of random wide shapes created and dropped in random order from four threads (as trials in flight), every store's
placement rated; afterwards every byte is back in the pools.   python scripts/arena_stress.py"""
import gc, random, sys, threading
import torch
sys.path.insert(0, ".")
from flexynesis_amd.arch import ArchSpec
from flexynesis_amd.engine import ParamStore, PartitionArena, placement_tries
dev = torch.device("cuda:0")
ar = PartitionArena.get(dev)
assert ar is not None, PartitionArena._arenas[0].info
free0 = ar.free_bytes()
rates, fallbacks, lock = [], [0], threading.Lock()
def worker(seed):
    rnd = random.Random(seed)
    torch.cuda.set_device(dev)
    alive = []
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        for i in range(20):
            F = rnd.choice([20000, 19873, 30000]); f = rnd.uniform(0.2, 0.5)
            spec = ArchSpec("DirectPred", [("gex", F)], rnd.randint(16, 128), f, 16, [("y", "numerical", 1)], None, None, True)
            with placement_tries(1):
                st = ParamStore(spec, dev, materialize_big_grads=False)
            info = st.placement.get("encoders.0.layer_1.weight")
            out, fin = st.eshapes["encoders.0.layer_1.weight"]
            with lock:
                if info and info.get("arena"):
                    rates.append(24.0 * out * fin / info["kept_us"] / 1e6)
                else:
                    fallbacks[0] += 1
            alive.append(st)
            if len(alive) > 2:
                alive.pop(rnd.randrange(len(alive)))
        del alive
ths = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]
gc.collect(); torch.cuda.synchronize()
print(f"{len(rates)} stores from the arena, {fallbacks[0]} fell back (pool full); rated {min(rates):.2f} .. {max(rates):.2f} TB/s (concurrent probes share the memory)")
print("free bytes before / after:", free0, ar.free_bytes())
assert ar.free_bytes() == free0
