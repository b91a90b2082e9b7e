"""Body of tests/test_gpu_api.py::test_level1_tape_graphs_equal_eager_launches (own process, FX_LEVEL1_GRAPHS=1)."""
import copy

import torch

DEV = torch.device("cuda:0")


def _synthetic_ds(n, F, seed):
    from flexynesis_amd.data import MultiOmicDataset
    g = torch.Generator().manual_seed(seed)
    dat = {"gex": torch.randn(n, F[0], generator=g), "cnv": torch.randn(n, F[1], generator=g)}
    w = torch.randn(F[0], generator=g) / F[0] ** 0.5
    y = dat["gex"] @ w + 0.05 * torch.randn(n, generator=g)
    c = (dat["cnv"][:, :3].sum(1) > 0).float() + (dat["gex"][:, 0] > 1).float()
    ann = {"y": y, "c": c}
    feats = {k: [f"{k}{i}" for i in range(v.shape[1])] for k, v in dat.items()}
    return MultiOmicDataset(dat, ann, {"y": "numerical", "c": "categorical"}, feats, [f"s{i}" for i in range(n)], {})


def check(fused):
    import flexynesis_amd.models as M
    torch.manual_seed(9)
    ds = _synthetic_ds(n=256, F=(8192, 4100), seed=6)
    cfg = {"latent_dim": 32, "hidden_dim_factor": 0.25, "lr": 1e-3, "supervisor_hidden_dim": 8, "epochs": 1, "batch_size": 64}
    m = M.DirectPred(cfg, ds, ["y", "c"], device_type="cuda")
    m.to(DEV)
    finals = []
    for graphs in (True, False):
        mm = copy.deepcopy(m)
        mm.fused_optimizer = fused
        oo = mm.configure_optimizers()
        losses = []
        for it in range(6):
            idx = torch.arange(it * 16, it * 16 + 64) % 256
            batch = ({k: v[idx].to(DEV) for k, v in ds.dat.items()}, {k: torch.as_tensor(v)[idx].to(DEV) for k, v in ds.ann.items()}, None)
            mm.train()
            oo.zero_grad()
            loss = mm.training_step(batch, it, log=False)
            plan = mm._plans[(64, True, fused)]
            assert it > 0 or plan.tape_graphs             # FX_LEVEL1_GRAPHS=1 in this process
            plan.tape_graphs = graphs
            loss.backward()
            mm.configure_gradient_clipping(oo, 1.0, "norm")
            oo.step()
            losses.append(float(loss.detach()))
        if graphs:
            assert set(plan._tape_graph) == ({"fwd", "bwd", "opt"} if fused else {"fwd", "bwd"})
        finals.append((losses, {k: v.clone() for k, v in mm.state_dict().items()}))
    assert finals[0][0] == finals[1][0]
    for k in finals[0][1]:
        assert torch.equal(finals[0][1][k], finals[1][1][k]), k


if __name__ == "__main__":
    n = 0
    for fused in (False, True):
        check(fused)
        n += 1
    print("LEVEL1_GRAPHS_OK", n)
