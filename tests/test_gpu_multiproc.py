"""Multi-process readiness on the ONE GPU this suite has: two ranks share device 0 and talk over gloo
(RCCL refuses two ranks on one GPU), so everything above the transport -- the cohort broadcast to a real peer, the TCPStore
work queue across processes, all_gather of the records, the winner's state_dict broadcast, sharded fine-tuning and
``bench.py --gpus 2`` end to end -- runs with real engine fits on the device.  The first 8-GPU run then exercises nothing new
except the RCCL transport underneath torch.distributed."""
import copy
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SWEEP_KW = dict(n_trials=6, epochs=2, features=1500, samples=320, seed=3, in_flight=1)
FT_KW = dict(n_splits=2, batch_size=16, learning_rates=[3e-3, 3e-4], max_epoch=3, seed=1, device="cuda",
             freeze_configs=[{"encoders": True, "supervisors": False}, {"encoders": False, "supervisors": False}])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ft_inputs():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import flexynesis_amd.models as M
    from test_gpu_api import _synthetic_ds
    ds = _synthetic_ds(n=90)
    cfg = {"latent_dim": 16, "hidden_dim_factor": 0.5, "lr": 3e-3, "supervisor_hidden_dim": 8, "epochs": 3, "batch_size": 16}
    torch.manual_seed(1)
    return M.DirectPred(cfg, ds, ["y", "c"], device_type="cuda"), ds


def _rank_main(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)                                   # both ranks on device 0
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flexynesis_amd import trials
    from flexynesis_amd.fit import fine_tune
    from flexynesis_amd.sweep import run_cfg5
    # 1. cohort broadcast to a real peer: rank 1 receives what rank 0 generated, in rank 0's layer order, on the device
    dat = ann = None
    if rank == 0:
        g = torch.Generator(device=dev).manual_seed(7)
        dat = {"zeta": torch.randn(64, 300, generator=g, device=dev), "alpha": torch.randn(64, 200, generator=g, device=dev)}
        ann = {"y": torch.randn(64, generator=g, device=dev)}
    d2, a2 = trials.broadcast_cohort(dat, ann, dev)
    cohort = (list(d2), bool(d2["zeta"].is_cuda), float(d2["zeta"].double().sum()), float(a2["y"].double().sum()))
    # 2. the cfg5 sweep: units claimed from the shared counter, records gathered, winner broadcast
    out = run_cfg5(dev, **SWEEP_KW)
    queue_sweep = {k: out[k] for k in ("trial_val_losses", "best_trial", "best_val_loss", "winner_state_tensors", "trials_ok", "n_gpus")}
    static = run_cfg5(dev, schedule="static", **SWEEP_KW)
    # ranks that disagree about the sweep fail together instead of splitting the queue
    try:
        trials.run_units(3, lambda u: (1.0, 1, None), costs=[1.0, 2.0, 3.0 + rank], device=dev)
        mismatch = "no error"
    except RuntimeError as e:
        mismatch = str(e)
    # 3. sharded fine-tuning (run_experiments over lr x freeze x fold units)
    m, ds = _ft_inputs()
    final, best, results = fine_tune(copy.deepcopy(m), ds, sharded=True, **FT_KW)
    sd = {k: v.detach().cpu().numpy() for k, v in final.state_dict().items()}     # (numpy: a tensor in a Queue shares an fd its sender must outlive)
    q.put((rank, cohort, queue_sweep, static["trial_val_losses"], mismatch, best, results, sd, [round(b, 4) for b in out["rank_busy_s"]]))
    dist.destroy_process_group()


def test_two_ranks_sharing_gpu0_match_the_single_process_results():
    from flexynesis_amd.fit import fine_tune
    from flexynesis_amd.sweep import run_cfg5
    dev = torch.device("cuda", 0)
    assert not dist.is_initialized()
    single = run_cfg5(dev, **SWEEP_KW)
    m, ds = _ft_inputs()
    f_seq, b_seq, r_seq = fine_tune(copy.deepcopy(m), ds, **FT_KW)
    sd_seq = {k: v.detach().cpu() for k, v in f_seq.state_dict().items()}

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r0, r1 = res
    # the peer holds rank 0's cohort (order, device, contents)
    assert r0[1] == r1[1] and r0[1][0] == ["zeta", "alpha"] and r0[1][1]
    # identical table on both ranks, identical to the single-process sweep: which rank ran a trial does not change its result
    assert r0[2] == r1[2]
    assert r0[2]["n_gpus"] == 2 and r0[2]["trials_ok"] == SWEEP_KW["n_trials"]
    assert r0[2]["trial_val_losses"] == single["trial_val_losses"] and r0[2]["best_trial"] == single["best_trial"]
    assert r0[2]["winner_state_tensors"] == single["winner_state_tensors"] > 10
    assert r0[3] == r1[3] == single["trial_val_losses"]          # the static LPT schedule too
    assert all(b > 0 for b in r0[8]) and len(r0[8]) == 2          # both ranks actually trained (work queue across processes)
    for r in (r0, r1):
        assert "do not agree" in r[4], r[4]
    # sharded fine-tuning == the sequential driver
    assert r0[5] == r1[5] == b_seq and r0[6] == r1[6] == r_seq
    for k in sd_seq:
        assert np.array_equal(r0[7][k], sd_seq[k].numpy()) and np.array_equal(r1[7][k], sd_seq[k].numpy()), k


def test_bench_two_ranks_on_one_gpu_end_to_end():
    """``bench.py --gpus 2`` as the driver launches it (torch.distributed.run, one process per rank), both ranks on GPU 0 with the
    collectives on gloo (FX_BENCH_SHARE_GPU): barrier + max-over-ranks timing, whole-job value, the cfg5 leg sharded over the two
    processes, one JSON line from rank 0."""
    env = dict(os.environ, FX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--features", "2000", "--sweep-trials-per-gpu", "2", "--repeats", "0", "--no-other"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak" and out["config"]["loss_finite"]
    assert out["value"] > 0 and abs(out["value"] - 2 * 4 * 128 / (out["ms_per_step"] * 4e-3)) < 1e-3 * out["value"]
    sw = out["sweep"]
    assert sw["n_gpus"] == 2 and sw["trials"] == 4 and sw["trials_ok"] == 4 and len(sw["rank_busy_s"]) == 2
    assert np.isfinite(sw["best_val_loss"]) and sw["winner_state_tensors"] > 10


def test_bench_self_launch_without_a_launcher():
    """Plain ``python bench.py --gpus 2`` (no torch.distributed.run in front: the form the driver uses at N = 1): bench.py starts its
    two ranks itself, rank 0 prints the one JSON line, the return code is the launcher's; the N > 1 line carries `sweep_scaling`
    (the sharded sweep's aggregate over the N = 1 leg's) and every rank's device / arena."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--features", "2000",
           "--sweep-trials-per-gpu", "2", "--repeats", "0", "--no-other", "--n1-sweep", "1000"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["loss_finite"]
    sc = out["sweep_scaling"]
    assert sc["n_gpus"] == 2 and sc["aggregate_samples_per_s"] == out["sweep"]["aggregate_samples_per_s"] > 0
    assert sc["n1_aggregate_samples_per_s"] == 1000 and abs(sc["ratio"] - sc["aggregate_samples_per_s"] / 1000) < 1e-2
    assert out["ranks_seen"] == 2 and [r["rank"] for r in out["ranks"]] == [0, 1]


def test_bench_dry_self_launch():
    """``python bench.py --gpus 2 --dry`` without a launcher: the day-one check of every collective, one JSON line, rc 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--features", "2000"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["dry"] and out["n_gpus"] == 2 and out["trials_ok"] == 2 and len(out["ranks"]) == 2 and not out["error"]
