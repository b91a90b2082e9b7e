"""flexynesis_amd.adapters on the GPU: stand-ins carrying exactly the attributes the reference's HyperparameterTuning / FineTuner
set (main.py:76-128, 493-528; the reference itself does not travel to the GPU box) are driven through the adapters with the
reference's call pattern -- perform_tuning's ``objective(params, current_step=i + 1, total_steps=n_iter)`` (main.py:352-368), the
final ``objective(best, 0, 1, full_train=True)`` of a cross-validated search (:403-414), ``run_experiments()`` -- and return
what the reference's methods return."""
import types

import numpy as np
import pytest
import torch

from test_gpu_api import _synthetic_ds

pytestmark = pytest.mark.gpu


def _tuner(model_class, ds, use_cv=False, **kw):
    t = types.SimpleNamespace(dataset=ds, model_class=model_class, target_variables=["y", "c"], batch_variables=None,
                              surv_event_var=None, surv_time_var=None, use_loss_weighting=True, device_type="cuda", val_size=0.2,
                              use_cv=use_cv, n_splits=3, early_stop_patience=3, gnn_conv_type=None, input_layers=None, output_layers=None)
    t.__dict__.update(kw)
    return t


def test_objective_adapter_runs_the_trial_loop_with_the_references_call_pattern():
    import flexynesis_amd.models as M
    from flexynesis_amd import adapters
    from flexynesis_amd.fit import run_trial
    ds = _synthetic_ds(n=400)
    params = {"latent_dim": 16, "hidden_dim_factor": 0.25, "lr": 3e-3, "supervisor_hidden_dim": 8, "epochs": 6, "batch_size": 32}
    # the orchestration may hold the REFERENCE's class: the adapter dispatches on its name, as main.py:159 / 241 / 243 do
    ref_like = type("DirectPred", (), {})
    torch.manual_seed(21)
    tuner = _tuner(ref_like, ds)
    losses = []
    for i in range(2):                                                     # perform_tuning's loop body (main.py:352-368)
        loss, avg_epochs, model = adapters.objective(tuner, dict(params), current_step=i + 1, total_steps=2)
        assert np.isfinite(loss) and isinstance(avg_epochs, int) and 1 <= avg_epochs <= params["epochs"]
        assert isinstance(model, M.DirectPred) and model.config["latent_dim"] == 16
        losses.append(loss)
    assert losses[0] != losses[1]                                          # different trials draw different splits / inits
    # the same trial through fit.run_trial with the adapter's seed: the adapter adds nothing to the numbers
    torch.manual_seed(21)
    val, ep, _, _ = run_trial(M.DirectPred, dict(params), ds, ["y", "c"], val_size=0.2, early_stop_patience=3,
                              seed=adapters._seed(1), device="cuda")
    assert val == losses[0]
    # cross-validated search: mean over folds, then the final model on all samples (main.py:403-414)
    cv = _tuner(M.DirectPred, ds, use_cv=True)
    loss, avg_epochs, _ = adapters.objective(cv, dict(params), 1, 1)
    assert np.isfinite(loss)
    best = dict(params, epochs=max(avg_epochs, 1))
    final = adapters.objective(cv, best, current_step=0, total_steps=1, full_train=True)
    assert isinstance(final, M.DirectPred)
    pred = final.predict(ds)
    assert set(pred) == {"y", "c"} and np.isfinite(pred["y"]).all()
    # early_stop_patience <= 0 (the class default, main.py:92): no early stopping, the trial runs its epochs
    nostop = _tuner(M.DirectPred, ds, early_stop_patience=-1)
    _, ep2, _ = adapters.objective(nostop, dict(params), 1, 1)
    assert ep2 == params["epochs"]
    # a CPU / MPS orchestration is refused loudly (there is no CPU path), not run somewhere else
    with pytest.raises(RuntimeError):
        adapters.objective(_tuner(M.DirectPred, ds, device_type="cpu"), dict(params), 1, 1)


def test_run_experiments_adapter_leaves_the_final_model_and_the_records():
    import flexynesis_amd.models as M
    from flexynesis_amd import adapters
    ds = _synthetic_ds(n=240)
    cfg = {"latent_dim": 16, "hidden_dim_factor": 0.5, "lr": 3e-3, "supervisor_hidden_dim": 8, "epochs": 3, "batch_size": 32}
    torch.manual_seed(2)
    model = M.DirectPred(cfg, ds, ["y", "c"], device_type="cuda")
    model.to("cuda:0")
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ft = types.SimpleNamespace(original_model=model, dataset=ds, n_splits=2, batch_size=32, learning_rates=[3e-3, 3e-4], max_epoch=3,
                               freeze_configs=[{"encoders": True, "supervisors": False}, {"encoders": False, "supervisors": False}])
    out = adapters.run_experiments(ft)
    assert ft.model is out and isinstance(out, M.DirectPred) and out is not model
    assert len(ft.val_loss_results) == 4 and set(ft.val_loss_results[0]) == {"learning_rate", "average_val_loss", "freeze", "epochs"}
    assert ft.best_config == min(ft.val_loss_results, key=lambda r: r["average_val_loss"]) and ft.learning_rate == ft.best_config["learning_rate"]
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k                                 # the original model is not touched (deep copies, main.py:586-588)
    # a reference (non-engine) model is refused with the way out
    with pytest.raises(TypeError):
        adapters.run_experiments(types.SimpleNamespace(original_model=torch.nn.Linear(2, 2)))
