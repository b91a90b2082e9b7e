#!/usr/bin/env python
"""Headline benchmark: training samples/sec of the DirectPred hot path on synthetic
2-omics x 20k-feature cohorts (BASELINE.json configs[1]) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus N ...      (no launcher: starts the N ranks itself under torch.distributed.run on 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one full optimisation step of the hot path on one batch of B=128 samples already resident in
HBM: cursor advance, on-device batch assembly (of the NEXT step's batch, double-buffered, as a parallel graph
branch), forward, losses, backward, global-norm clip, Adam -- all hand-written HIP kernels replayed from a hipGraph.  Multi-GPU: one process per GPU, each rank trains
its own independent trial on its own cohort replica (trial sharding, SURVEY.md section 8e: no data-path
collective); value = total samples of all ranks / max-over-ranks wall time ("weak" scaling).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel fx_linear_dw_adam_fwd_bf16x3 -- dW + clip + Adam of one wide
weight and the next step's forward through it --, HIP-event timed on the launch stream) and `cpu_baseline` (the oracle's CPU
training loop timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

CONFIGS = {
    # BASELINE.json configs[1]: DirectPred, 2 omics (gex+cnv, 20000 feat each), intermediate fusion
    "cfg2": dict(model="DirectPred", layers=[("gex", 20000), ("cnv", 20000)],
                 variables=[("y", "numerical", 1)], surv=(None, None), n_samples=2048),
    # configs[0] plumbing case (reference's CPU-runnable shape)
    "cfg1": dict(model="DirectPred", layers=[("gex", 5000)], variables=[("y", "numerical", 1)], surv=(None, None),
                 n_samples=500),
    "cfg3": dict(model="supervised_vae", layers=[("gex", 20000), ("cnv", 20000)],
                 variables=[("c", "categorical", 4), ("event", "numerical", 1)], surv=("event", "time"),
                 n_samples=2048),
    "cfg4": dict(model="MultiTripletNetwork", layers=[("gex", 30000), ("cnv", 30000), ("meth", 30000)],
                 variables=[("c", "categorical", 4)], surv=(None, None), n_samples=2048),
    # cfg2 with --fusion_type early (reference data.py:234-257): ONE layer of 40000 features, a single [10000, 40000] weight
    "cfg2_early": dict(model="DirectPred", layers=[("all", 40000)], variables=[("y", "numerical", 1)], surv=(None, None),
                       n_samples=2048),
    # cfg2 at a point the reference's search space really draws (config.py:7-15: latent_dim any integer in 16..128, hidden =
    # int(F * U[0.2, 0.5])) on a cohort whose feature counts are multiples of nothing: widths odd mod 4 everywhere.  Hidden sizes
    # 4968 / 5000, i.e. the headline's work within 0.3 %: the leg shows that the 14-launch schedule and the fused next-step forward
    # do not depend on the widths (ArchSpec.engine_shapes)
    "cfg2_odd": dict(model="DirectPred", layers=[("gex", 19873), ("cnv", 20001)], variables=[("y", "numerical", 1)], surv=(None, None),
                     n_samples=2048, latent=61, factor=0.25, sup=13),
}


def _spec_of(cfg):
    from flexynesis_amd.arch import ArchSpec
    return ArchSpec(cfg["model"], cfg["layers"], cfg.get("latent", 64), cfg.get("factor", 0.25), cfg.get("sup", 16), cfg["variables"],
                    cfg["surv"][0], cfg["surv"][1], True)
PMC_FILE = "r06_pmc_traffic_cfg2.json"      # rocprofv3 --pmc passes of this command, this round, this kernel (scripts/profile_round.sh)
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


class _stdout_to_stderr:
    """fd-level redirect: RCCL prints a version banner to STDOUT when its first communicator is created; rank 0's
    stdout must carry exactly one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:        # RCCL prints through C stdio, which is block-buffered on a pipe: flush it while fd 1 is still stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def _live_traffic(want_prefix, args):
    """HBM bytes per launch of the dominant kernel, measured NOW on this box: two more runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, --kernel-trace only: MI355X_MICROARCH.md, HBM section), a few eager
    steps each; FETCH_SIZE doubled (gfx950 counts the 128-byte requests of a wide streaming read as 64 bytes), as scripts/pmc_to_json.py
    does for profiles/.  Returns (bytes | None, note)."""
    import csv, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="fxpmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--config", args.config, "--batch", str(args.batch), "--precision", args.precision, "--steps", "4", "--warmup", "1", "--no-graph",
               "--no-cpu-baseline", "--sweep-trials-per-gpu", "0", "--no-other", "--repeats", "0", "--no-pmc"]
        try:
            # (the child is a plain single-process run: a launcher's rendezvous variables must not reach it)
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
                   and not k.startswith("TORCHELASTIC_")}
            env["TMPDIR"] = "/tmp"
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, check=True)
            path = None
            for root, _, files in os.walk(d):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(root, f)
            got = []
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"].replace("void ", "").split("(")[0]
                if r["Counter_Name"] == counter and name.startswith(want_prefix):
                    got.append(float(r["Counter_Value"]))
            if not got:
                return None, f"{counter}: no launch of {want_prefix} in the counter file"
            vals[counter] = (sum(got) / len(got), len(got))
        except Exception as e:
            return None, f"{counter} pass failed: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch_kb, write_kb = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    return int(round((2.0 * fetch_kb + write_kb) * 1024)), (f"measured in this run: rocprofv3 --pmc FETCH_SIZE ({vals['FETCH_SIZE'][1]} launches, x 2: gfx950 "
                                                          f"correction) + --pmc WRITE_SIZE ({vals['WRITE_SIZE'][1]} launches), separate passes of this script")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-graph", action="store_true", help="launch the recorded tapes eagerly instead of a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep-trials-per-gpu", type=int, default=8, help="cfg5 leg after the timed region: this many "
                    "DirectPred HPO trials per GPU, sharded over the ranks with the real collectives (0 = skip)")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU baseline steps (0 = auto, about 10-30 s)")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--features", type=int, default=0, help="override the per-layer feature count (shape experiments; "
                    "the headline metric is quoted on the default)")
    ap.add_argument("--repeats", type=int, default=25, help="after the timed region: this many more windows of --steps steps, "
                    "reported as median / min / max (0 = skip)")
    ap.add_argument("--no-other", action="store_true", help="skip the short legs of the other BASELINE configs / modes")
    ap.add_argument("--dry", action="store_true", help="multi-GPU day-one check instead of the benchmark: cohort broadcast, ONE one-epoch "
                    "trial per rank, all_gather of the records, winner broadcast -- prints the seconds of every phase and each rank's "
                    "placement of the cfg2 weights (one JSON line from rank 0)")
    ap.add_argument("--n1-sweep", type=float, default=0.0, help="the N = 1 sweep leg's aggregate samples/s to quote `sweep_scaling` against "
                    "(0 = the round's committed N = 1 line under profiles/)")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-run this script under rocprofv3 --pmc for roofline.traffic")
    ap.add_argument("--settle", type=int, default=40, help="untimed hipGraph replays before the --warmup steps (the first windows after "
                    "the captures run 3-5 %% slower while the clocks settle; reported as config.untimed_settle_steps)")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "f32", "bf16"],
                    help="wide-layer contraction: split-bf16 MFMA with fp32 accumulate (default: the parity mode), exact fp32 MFMA, or "
                         "plain bf16 (one product: the labelled THROUGHPUT mode, looser documented tolerance)")
    return ap.parse_args()


def _engine_leg(config, B, dev, precision, steps, warmup, lr, rank=0, features=0):
    """One engine configuration timed like the headline: hipGraph replay of PipelinedStep, per-epoch device reshuffle inside the
    timed region, `steps` steps after `warmup`.  Returns (record, pipe, store, reshuffle) -- the caller deletes what it does
    not keep."""
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.data import synthetic_cohort
    from flexynesis_amd.engine import ParamStore, PipelinedStep
    cfg = dict(CONFIGS[config])
    if features:
        cfg["layers"] = [(n, features) for n, _ in cfg["layers"]]
    spec = _spec_of(cfg)
    cohort = synthetic_cohort(cfg["layers"], cfg["n_samples"], dev, seed=1234 + rank)
    n_train = cfg["n_samples"] - int(cfg["n_samples"] * 0.2)
    rows_per_batch = B * (3 if cfg["model"] == "MultiTripletNetwork" else 1)
    n_batches = max(n_train // rows_per_batch, 1)
    torch.manual_seed(1000 + rank)
    store = ParamStore(spec, dev, materialize_big_grads=False)
    pipe = PipelinedStep(store, B, cohort=cohort, n_batches=n_batches, seed=17 + rank, precision=precision)
    from flexynesis_amd import ops as _ops
    gen = _ops.DeviceRng(4321 + rank)

    def reshuffle():
        perm = _ops.randperm(n_train, gen, dev)              # fx_randperm: Philox keys + bitonic sort, one launch
        pipe.idx.copy_(perm[:n_batches * rows_per_batch])

    reshuffle()
    pipe.prime()
    pipe.step(lr)
    pipe.capture(lr)

    def run(k):
        for _ in range(k):
            if pipe.epoch_end_next():
                reshuffle()
            pipe.replay()

    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    P = store.n_params()
    sumF = sum(F for _, F in cfg["layers"])
    k_reads = 3 if cfg["model"] == "MultiTripletNetwork" else 1
    bytes_step = 28.0 * P + 4.0 * k_reads * B * sumF
    ms = 1e3 * dt / steps
    losses = pipe.losses()
    rec = {"workload": f"{config}: {cfg['model']} {len(cfg['layers'])} x {cfg['layers'][0][1]} features, B={B}, {precision}, hipGraph replay",
           "samples_per_s": round(steps * B / dt, 1), "ms_per_step": round(ms, 4), "steps": steps,
           "step_hbm_frac_of_8TBs": round(bytes_step / (ms * 1e-3) / 8e12, 4), "params": P, "launches_per_step": pipe.n_launches(),
           "next_forward_fused": sorted(pipe.plans[0]._next_fwd), "schedule": dict(pipe.plans[0].path),
           "loss_finite": all(v == v and abs(v) != float("inf") for v in losses.values())}
    return rec, pipe, store, run


def _dropin_leg(dev, kind, steps, warmup, B=128):
    """The unmodified-caller path (INTEGRATION.md level 1): the model class under the Lightning protocol -- zero_grad, training_step,
    loss.backward(), configure_gradient_clipping, optimizer.step() -- at the cfg2 shape.  kind: 'fused' = FxAdam running the
    engine's clip + dW + Adam launches (what a Lightning Trainer gets automatically), 'fx' = FxAdam on materialised gradients,
    'torch' = torch.optim.Adam + torch's clip_grad_norm_."""
    from flexynesis_amd import models as M
    from flexynesis_amd.data import MultiOmicDataset
    n, F = 2048, 20000
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    dat = {k: torch.randn(n, F, generator=g, device=dev) for k in ("gex", "cnv")}
    ann = {"y": dat["gex"][:, :16].sum(1) / 4 + 0.1 * torch.randn(n, generator=g, device=dev)}
    feats = {k: [f"{k}_{j}" for j in range(F)] for k in dat}
    ds = MultiOmicDataset(dat, ann, {"y": "numerical"}, feats, [f"s{i}" for i in range(n)], {})
    cfg = {"latent_dim": 64, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 16, "epochs": 1, "batch_size": B}
    m = M.DirectPred(cfg, ds, ["y"], device_type="cuda")
    m.to(dev)
    m.train()
    m.fused_optimizer = kind == "fused"
    opt = m.configure_optimizers() if kind in ("fx", "fused") else torch.optim.Adam(m.parameters(), lr=1e-3)
    perm = torch.randperm(n, generator=g, device=dev)

    def step(i):
        o = (i * B) % (n - B)
        idx = perm[o:o + B]
        batch = ({k: v[idx] for k, v in dat.items()}, {"y": ann["y"][idx]}, None)
        opt.zero_grad()
        loss = m.training_step(batch, i, log=False)
        loss.backward()
        m.configure_gradient_clipping(opt, 1.0, "norm")
        opt.step()
        return loss

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = {"workload": f"cfg2 under the Lightning protocol (level-1 drop-in), optimizer {type(opt).__name__}" + (" fused" if kind == "fused" else ""),
           "samples_per_s": round(steps * B / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps,
           "loss_finite": bool(torch.isfinite(loss).all())}
    if hasattr(m, "close"):
        m.close()
    return rec


def _self_launch(n):
    """Re-exec this command under torch.distributed.run, one process per GPU (rendezvous on 127.0.0.1, a free port)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max((os.cpu_count() or n) // n, 1)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FX_BENCH_SHARE_GPU"):      # multi-process readiness on ONE GPU (tests): every rank uses device 0; implies gloo
        local = 0
        os.environ.setdefault("FX_BENCH_BACKEND", "gloo")
    backend = os.environ.get("FX_BENCH_BACKEND", "nccl")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N` (the form the driver uses at N = 1): start the N ranks ourselves, exactly as an external
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` would,
        # forward the arguments, let rank 0's one JSON line through, propagate the return code
        sys.exit(_self_launch(a.gpus))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_pg = world > 1 or bool(os.environ.get("FX_BENCH_FORCE_PG"))   # the env switch exercises the RCCL path on one GPU
    if use_pg:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with _stdout_to_stderr():
            if backend == "gloo":     # collectives through host memory: same code path above the transport (ranks may share a GPU)
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            dist.barrier()            # creates the communicator now (and its banner goes to stderr)
            torch.cuda.synchronize()

    from flexynesis_amd import ops
    from flexynesis_amd.arch import ArchSpec
    from flexynesis_amd.data import synthetic_cohort
    from flexynesis_amd.engine import ParamStore, PipelinedStep

    cfg = dict(CONFIGS[a.config])
    if a.features:
        cfg["layers"] = [(n, a.features) for n, _ in cfg["layers"]]
    B = a.batch
    spec = _spec_of(cfg)
    if a.dry:
        # every collective of the cfg5 path once, with the seconds of each phase, so that a first N-GPU run measures instead of debugging
        from flexynesis_amd.sweep import run_cfg5
        t0 = time.perf_counter()
        st = ParamStore(spec, dev, materialize_big_grads=False)          # this rank's placement of the headline weights (search on)
        mine = {"rank": rank, "device": torch.cuda.get_device_name(dev), "store_build_s": round(time.perf_counter() - t0, 3),
                "placement": {k: {kk: v.get(kk) for kk in ("arena", "probed", "kept_TBps", "kept_us", "search_s", "build_s", "spacer_GB")} for k, v in st.placement.items()}}
        del st
        with _stdout_to_stderr():
            try:
                sw = run_cfg5(dev, n_trials=world, epochs=1, features=cfg["layers"][0][1], samples=cfg["n_samples"], seed=1, keep_winner=True,
                              force_collectives=use_pg, in_flight=1)
            except Exception as e:
                sw = {"error": repr(e)}
            ranks = [mine]
            if use_pg:
                ranks = [None] * world
                dist.all_gather_object(ranks, mine)
        if rank == 0:
            print(json.dumps({"dry": True, "n_gpus": world, "backend": backend if use_pg else None,
                              "phases_s": {"cohort_generate": sw.get("cohort_generate_s"), "cohort_broadcast": sw.get("cohort_broadcast_s"),
                                           **(sw.get("rank0_phases_s") or {}), "sweep_wall": sw.get("sweep_wall_s")},
                              "trials_ok": sw.get("trials_ok"), "trial_val_losses": sw.get("trial_val_losses"),
                              "winner_state_tensors": sw.get("winner_state_tensors"), "error": sw.get("error"),
                              "failed_units_rank0": sw.get("failed_units_rank0"), "ranks": ranks}), flush=True)
        if use_pg:
            with _stdout_to_stderr():
                dist.destroy_process_group()
        return
    cohort = synthetic_cohort(cfg["layers"], cfg["n_samples"], dev, seed=1234 + rank)
    n_train = cfg["n_samples"] - int(cfg["n_samples"] * 0.2)           # 80/20 split, reference main.py:272-276
    rows_per_batch = B * (3 if cfg["model"] == "MultiTripletNetwork" else 1)
    n_batches = max(n_train // rows_per_batch, 1)                      # drop_last=True, reference main.py:294
    torch.manual_seed(1000 + rank)
    store = ParamStore(spec, dev, materialize_big_grads=False)
    pipe = PipelinedStep(store, B, cohort=cohort, n_batches=n_batches, seed=17 + rank, precision=a.precision)
    dominant = "fx_linear_dw_adam_bf16x3" if a.precision in ("bf16x3", "bf16") else "fx_linear_dw_adam_f32"
    if pipe.plans[0]._next_fwd:
        dominant = "fx_linear_dw_adam_fwd_bf16x3"       # the same optimiser step + the next step's wide forward
    gen = ops.DeviceRng(4321 + rank)

    def reshuffle():
        """shuffle=True: a fresh permutation of the training split per epoch, drawn on the device (fx_randperm: one launch)."""
        perm = ops.randperm(n_train, gen, dev)
        pipe.idx.copy_(perm[:n_batches * rows_per_batch])

    reshuffle()
    pipe.prime()                     # assemble batch 0; from here on every step assembles the batch of the next one
    use_graph = not a.no_graph
    pipe.step(a.lr)                  # one real eager step loads the code objects (counts as neither warm-up nor timed)
    if use_graph:
        pipe.capture(a.lr)
        step = pipe.replay
    else:
        step = lambda: pipe.step(a.lr)

    def run(k):
        for i in range(k):
            if pipe.epoch_end_next():
                reshuffle()          # the epoch's last step prefetches the first batch of the next permutation
            step()

    # The first windows after the eager step and the two captures run 3-5 % slower than the steady state (the clocks are still
    # settling; `repeat_stats` below shows it): a training run is thousands of steps, so the untimed part is long enough to get
    # there -- SETTLE replays before the W warm-up steps the caller asked for (reported in config.untimed_settle_steps).
    SETTLE = max(int(a.settle), 0) if use_graph else 0
    run(SETTLE)
    run(a.warmup)

    def window():
        """EXACTLY --steps steps between barrier + synchronize on both sides, the maximum over the ranks (seconds)."""
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(a.steps)
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_pg:
            t = torch.tensor([dt], device="cpu" if backend == "gloo" else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    first_window = window()
    losses = pipe.losses()
    finite = all(v == v and abs(v) != float("inf") for v in losses.values())

    # ---- the same window R more times, timed the same way.  `value` is the MEDIAN window (the first one included): one 20-60 step
    # window is 20-60 ms, within reach of a single clock / power-management event; the first window is printed beside it.
    wins = [first_window] + [window() for _ in range(max(a.repeats, 0))]
    srt = sorted(wins)
    elapsed = srt[len(srt) // 2]
    repeat_stats = {"windows": len(wins), "steps_per_window": a.steps, "first_window_ms_per_step": round(1e3 * first_window / a.steps, 4),
                    "ms_per_step_median": round(1e3 * elapsed / a.steps, 4), "ms_per_step_min": round(1e3 * srt[0] / a.steps, 4),
                    "ms_per_step_max": round(1e3 * srt[-1] / a.steps, 4),
                    "first_window_samples_per_s": round(world * a.steps * B / first_window, 1)}

    # ---- dominant-kernel timing with HIP events on the launch stream (eager re-issue of the same tapes)
    roof = None
    if rank == 0:
        sink = []
        for i in range(min(a.steps, 20)):
            if pipe.epoch_end_next():
                reshuffle()
            pipe.step(a.lr, timed=({dominant}, sink))
        torch.cuda.synchronize()
        if sink:
            ms = [e0.elapsed_time(e1) for _, e0, e1 in sink]
            avg_ms = sum(ms) / len(ms)
            big_elems = [store.big[k]["W"].numel() for k in store.big_keys]
            bytes_per_launch = 24.0 * sum(big_elems) / len(big_elems)   # read+write of W, m, v (fp32)
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), collected
            # in separate rocprofv3 --pmc passes of this same command and committed under profiles/.
            traffic = traffic_note = None
            want = "fx_dw_adam_fwd_kernel" if dominant.endswith("_fwd_bf16x3") else "fx_gemm_bf16x3_kernel<false, 1"
            if not a.no_pmc and world == 1 and not a.features:
                traffic, traffic_note = _live_traffic(want, a)
            if traffic is None:
                try:
                    # fallback: the PMC file committed under profiles/ (another box, labelled as such)
                    pm = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
                    if a.config == "cfg2" and B == 128 and a.precision == "bf16x3" and not a.features:
                        for kname, d in pm["kernels"].items():
                            if kname.startswith(want):
                                traffic = d["hbm_bytes_per_launch_corrected"]
                                traffic_note = (f"profiles/{PMC_FILE} (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on another "
                                                f"box of the pool; not re-measured in this run" + (f": {traffic_note}" if traffic_note else "") + ")")
                except Exception:
                    traffic = None
            # Boxes of the pool differ by +-10 % in what their HBM delivers: record this box's plain device-to-device
            # copy rate (1 GiB read + 1 GiB write, HIP events) next to the kernel's figure.  `frac` stays relative to
            # the 8 TB/s vendor peak.
            copy_gbs = copy_torch = None
            try:
                src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
                dst = torch.empty_like(src)

                def rate(fn):
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    return round(5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
                copy_gbs = rate(lambda: ops.stream_copy(ops.IMMEDIATE, dst, src))      # libfxhip's own 16 B / lane copy
                copy_torch = rate(lambda: dst.copy_(src))
                del src, dst
            except Exception:
                pass
            roof = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": traffic_note if traffic else None,
                    "avg_launch_ms": round(avg_ms, 4), "launches_timed": len(ms),
                    "algorithmic_bytes_per_launch": bytes_per_launch, "device_copy_GBps_this_box": copy_gbs,
                    "torch_copy_GBps_this_box": copy_torch}
    P = store.n_params()
    n_launch = pipe.n_launches()
    sumF = sum(F for _, F in cfg["layers"])
    k_reads = 3 if cfg["model"] == "MultiTripletNetwork" else 1
    bytes_step = 28.0 * P + 4.0 * k_reads * B * sumF                  # the roofline the step is priced against
    # what this schedule actually has to move: wide weights whose next forward rides on the dW + Adam launch cost 24 B/param
    fused_elems = sum(store.big[k]["W"].numel() for k in pipe.plans[0]._next_fwd) if pipe.plans[0]._next_fwd else 0
    bytes_moved = bytes_step - 4.0 * fused_elems
    ms_per_step = 1e3 * elapsed / a.steps
    value = world * a.steps * B / elapsed

    # ---- cfg5 leg (outside the timed region): the sharded HPO sweep with its real collectives -- cohort broadcast from
    # rank 0, LPT assignment, engine trials, all_gather of the records, winner state_dict broadcast (flexynesis_amd/sweep.py)
    # everything the legs below need from the headline objects has been read: release them (graphs first, at a known point)
    pipe.close()
    placement = dict(store.placement)
    try:                                  # how this process's partition arena was built (engine.PartitionArena), or why it was not
        from flexynesis_amd.engine import PartitionArena
        _ar = PartitionArena._arenas.get(dev.index)
        placement_arena = dict(_ar.info) if _ar is not None else None
        if placement_arena is not None:
            # what the arena costs: resident for the life of the process (pools), transient while it was built (spacers, back with the driver)
            from flexynesis_amd.engine import placement_memory
            pm = placement_memory(dev)
            placement_arena["resident_GB"] = round(pm["arena_resident_bytes"] / 2 ** 30, 2)
            placement_arena["transient_spacer_GB"] = pm["arena_build_spacer_GB"]
            placement_arena["placement_pool_GB"] = round(pm["pool_bytes"] / 2 ** 30, 2)
    except Exception:
        placement_arena = None
    del pipe, store, cohort
    torch.cuda.empty_cache()
    # every rank's device and partition arena (N > 1: gathered over the process group the timed region used)
    mine = {"rank": rank, "local_rank": local, "device": torch.cuda.get_device_name(dev), "backend": backend if use_pg else None,
            "placement_arena": placement_arena}
    rank_info = [mine]
    if use_pg:
        rank_info = [None] * world
        with _stdout_to_stderr():
            dist.all_gather_object(rank_info, mine)
    sweep = None
    if a.sweep_trials_per_gpu > 0:
        try:
            from flexynesis_amd.sweep import run_cfg5
            with _stdout_to_stderr():
                # one untimed single-epoch trial per rank first: a process's first model pays torch's RNG / elementwise kernel
                # loads and the host-loop's first-call costs (~0.25 s, profiles/r03_trial_breakdown.md "trial 0"), which a real
                # sweep of tens of trials per GPU amortises and an 8-trial leg would report as 10-15 % of its wall time
                run_cfg5(dev, n_trials=world, epochs=1, features=cfg["layers"][0][1] if a.features else 20000, samples=2048,
                         seed=1, keep_winner=False)
                sweep = run_cfg5(dev, n_trials=a.sweep_trials_per_gpu * world, epochs=3,
                                 features=cfg["layers"][0][1] if a.features else 20000, samples=2048, seed=0)
                sweep["untimed_warmup_trials_per_gpu"] = 1
        except Exception as e:     # reported, never fatal for the headline number
            sweep = {"error": repr(e)}

    # ---- the other BASELINE configs and modes, 20 steps each (rank 0, N = 1 only; after the headline so that nothing of them is
    # resident while it is timed): driver-visible counterparts of what profiles/ and DESIGN.md quote
    other = None
    if rank == 0 and world == 1 and not a.no_other and a.config == "cfg2" and not a.features:
        other = {}
        with _stdout_to_stderr():
            for name, (cfgname, prec) in (("cfg1", ("cfg1", "bf16x3")), ("cfg3", ("cfg3", "bf16x3")), ("cfg4", ("cfg4", "bf16x3")),
                                          ("cfg2_f32", ("cfg2", "f32")), ("cfg2_early_fusion", ("cfg2_early", "bf16x3")),
                                          ("cfg2_odd", ("cfg2_odd", "bf16x3")),
                                          # the plain-bf16 throughput mode (one MFMA product instead of three)
                                          ("cfg2_bf16", ("cfg2", "bf16")), ("cfg3_bf16", ("cfg3", "bf16")), ("cfg4_bf16", ("cfg4", "bf16"))):
                try:
                    rec, p_, s_, _ = _engine_leg(cfgname, B, dev, prec, 20, 5, a.lr)
                    p_.close()
                    del p_, s_
                    other[name] = rec
                except Exception as e:
                    other[name] = {"error": repr(e)}
                torch.cuda.empty_cache()
            for kind in ("fused", "fx", "torch"):
                try:
                    other["dropin_" + kind] = _dropin_leg(dev, kind, 20, 5, B)
                except Exception as e:
                    other["dropin_" + kind] = {"error": repr(e)}
                torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            from oracle import cpu_baseline, restate as O
            ospec = O.Spec(cfg["model"], cfg["layers"], cfg.get("latent", 64), cfg.get("factor", 0.25), cfg.get("sup", 16), cfg["variables"],
                           cfg["surv"][0], cfg["surv"][1], True)
            threads = min(os.cpu_count() or 1, 64)
            steps = a.cpu_steps or (20 if a.config == "cfg2" else 100 if a.config == "cfg1" else 8)   # ~15 s of CPU work
            r = cpu_baseline.time_training(ospec, cfg["n_samples"], B, steps=steps, warmup=1, threads=threads, lr=a.lr)
            cpu = {"value": round(r["samples_per_s"], 2), "unit": "samples/s", "cores": r["threads"], "kind": "port",
                   "sample": f"{r['steps']} optimisation steps (after 1 warm-up) of the same {a.config} workload, "
                             f"B={B}, torch-CPU fp32 'highest', {r['ms_per_step']:.0f} ms/step"}
        except Exception as e:  # the baseline is reported, never required
            cpu = {"value": None, "unit": "samples/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}

    if rank == 0:
        out = {
            "metric": "training samples/sec (2-omics x 20k-feat DirectPred)" if a.config == "cfg2"
            else f"training samples/sec ({a.config})",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16x3": "bf16x3 (fp32 split into 2 bf16 terms, 3 bf16-MFMA products, fp32 accumulate; fp32 master weights/Adam)",
                      "bf16": "bf16 (operands rounded to bf16, 1 bf16-MFMA product, fp32 accumulate; fp32 master weights/Adam) -- throughput mode",
                      "f32": "f32"}[a.precision], "data": "synthetic",
            "config": {"workload": f"{a.config}: {cfg['model']} {len(cfg['layers'])} omics x "
                                   f"{cfg['layers'][0][1]} features, N={cfg['n_samples']}, B={B}, latent 64, "
                                   f"hidden_dim_factor 0.25, lr {a.lr}, clip 1.0 + Adam, "
                                   f"{'hipGraph replay' if use_graph else 'eager tapes'}",
                       "params": P, "global_batch": B * world, "parallelism": f"{world} independent trials (trial sharding)",
                       "launches_per_step": n_launch, "untimed_settle_steps": SETTLE, "algorithmic_bytes_per_step": bytes_step,
                       "step_hbm_frac_of_8TBs": round(bytes_step / (ms_per_step * 1e-3) / 8e12, 4),
                       # the bytes THIS schedule must move (24 instead of 28 B/param where the next forward is fused)
                       "schedule_bytes_per_step": bytes_moved,
                       "schedule_hbm_frac_of_8TBs": round(bytes_moved / (ms_per_step * 1e-3) / 8e12, 4),
                       # wide weights: probe times (us per pass of the dW + Adam traffic pattern over ONE array) of the candidate placements
                       # ParamStore tried at allocation, the three it kept and their time together (FX_PLACEMENT_TRIES=1: first)
                       "placement": placement, "placement_arena": placement_arena,
                       "loss_finite": finite, "last_losses": {k: round(v, 6) for k, v in losses.items()}},
            "roofline": roof, "cpu_baseline": cpu, "repeat_stats": repeat_stats, "other": other, "sweep": sweep,
        }
        if use_pg:
            # north_star's N-GPU figure is the SHARDED SWEEP's aggregate (`value` above is N independent fits: weak scaling by
            # construction); `sweep_scaling` = that aggregate over the N = 1 sweep leg's (--n1-sweep, or this round's committed line)
            n1, n1_src = a.n1_sweep, "--n1-sweep"
            if not n1:
                for fn in ("r06_bench_line.json", "r05_bench_line.json"):
                    try:
                        n1 = float(json.load(open(os.path.join(ROOT, "profiles", fn)))["sweep"]["aggregate_samples_per_s"])
                        n1_src = f"profiles/{fn} (N = 1, another box)"
                        break
                    except Exception:
                        n1 = 0.0
            agg = (sweep or {}).get("aggregate_samples_per_s")
            out["sweep_scaling"] = {"n_gpus": world, "aggregate_samples_per_s": agg, "n1_aggregate_samples_per_s": n1 or None,
                                    "n1_source": n1_src if n1 else None, "ratio": round(agg / n1, 3) if agg and n1 else None,
                                    "busy_over_wall": (sweep or {}).get("busy_over_wall"), "tail_imbalance": (sweep or {}).get("tail_imbalance")}
            out["rccl_ranks_seen"] = len([r for r in rank_info if r]) if backend != "gloo" else 0
            out["ranks_seen"] = len([r for r in rank_info if r])
            out["ranks"] = rank_info
        print(json.dumps(out), flush=True)
    if use_pg:
        with _stdout_to_stderr():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
