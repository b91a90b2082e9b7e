"""the REFERENCE ITSELF (imported from /root/reference through oracle/ref_shim.py; build
container only) timed on this container's CPU cores with the section 8(d) synthetic generator: manual loop
zero_grad -> training_step -> backward -> clip_grad_norm_(1.0) -> Adam.step, matmul precision "highest".

    python scripts/time_reference_cpu.py [cfg2|cfg1] [threads ...]

Prints one JSON line per thread count; the numbers are recorded in BASELINE.md."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from oracle import ref_capture, ref_shim
from oracle import restate as O

CFG = {"cfg1": ([("gex", 5000)], 500), "cfg2": ([("gex", 20000), ("cnv", 20000)], 2048)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    threads = [int(t) for t in sys.argv[2:]] or [8, 1]
    layers, n = CFG[name]
    R = ref_shim.load()
    torch.set_float32_matmul_precision("highest")
    spec = O.Spec("DirectPred", layers, 64, 0.25, 16, [("y", "numerical", 1)])
    dat, ann = O.synthetic_cohort(layers, n, seed=1234)
    ds = ref_capture.make_dataset(R, dat, {"y": ann["y"]}, {"y": "numerical"})
    cfg = {"latent_dim": 64, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 16, "epochs": 1, "batch_size": 128}
    B = 128
    for th in threads:
        torch.set_num_threads(th)
        torch.manual_seed(0)
        model = ref_capture.build_reference_model(R, spec, ds, cfg)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        model.train()
        g = torch.Generator().manual_seed(1)
        steps, warm = (12, 2) if th > 1 else (6, 1)
        times = []
        for s in range(steps + warm):
            idx = torch.randperm(n, generator=g)[:B]
            batch = ({k: v[idx] for k, v in dat.items()}, {"y": ann["y"][idx]}, tuple(f"s{i}" for i in idx.tolist()))
            t0 = time.perf_counter()
            opt.zero_grad()
            loss = model.training_step(batch, s, log=False)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            dt = time.perf_counter() - t0
            if s >= warm:
                times.append(dt)
        ms = 1e3 * sum(times) / len(times)
        print(json.dumps({"config": name, "threads": th, "steps": len(times), "ms_per_step": round(ms, 1),
                          "samples_per_s": round(B / (ms * 1e-3), 2), "params": sum(p.numel() for p in model.parameters()),
                          "cpu": os.popen("grep -m1 'model name' /proc/cpuinfo").read().split(":")[-1].strip(),
                          "nproc": os.cpu_count(), "last_loss": float(loss.detach().reshape(-1)[0])}), flush=True)
        del model, opt


if __name__ == "__main__":
    main()
