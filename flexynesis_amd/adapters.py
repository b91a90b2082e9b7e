"""Level-2 drop-in with the REFERENCE'S OWN SIGNATURES: the engine loop behind ``HyperparameterTuning.objective`` and
``FineTuner.run_experiments``, so that the unmodified HPO / CLI orchestration (reference ``flexynesis/main.py``,
``__main__.py``) reaches the fast path -- hipGraph-replayed steps, on-device batch assembly, the next step's wide forward
fused into the weight update -- without an edit of the reference:

    from flexynesis_amd.adapters import install
    install()                       # HyperparameterTuning.objective / FineTuner.run_experiments now run on the engine

``objective(self, params, current_step, total_steps, full_train=False)`` reads exactly the attributes the reference's
method reads (main.py:228-333): ``self.dataset, self.model_class, self.target_variables, self.batch_variables,
self.surv_event_var, self.surv_time_var, self.use_loss_weighting, self.device_type, self.val_size, self.use_cv, self.n_splits,
self.early_stop_patience, self.gnn_conv_type, self.input_layers, self.output_layers`` and returns what it returns:
``(avg_val_loss, avg_epochs, model)``, or the trained model when ``full_train``.  ``run_experiments(self)`` reads
``self.original_model, self.dataset, self.n_splits, self.batch_size, self.learning_rates, self.max_epoch,
self.freeze_configs`` (main.py:493-528) and leaves ``self.model`` = the final model, like the reference's (:646-659), plus the
records in ``self.val_loss_results`` / ``self.best_config``.

What differs, by construction and documented: shuffles, splits and dropout draws come from device Philox streams seeded from
torch's global seed and the trial number (the reference draws from the host's global generators); a trial whose fit
fails reports ``+inf`` instead of raising (a sharded sweep must not hang on a bad configuration); there is no progress bar /
live plot (``setup_trainer``'s callbacks are a Lightning ``Trainer``'s, and no ``Trainer`` runs here)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import models as _models
from .fit import fine_tune, run_trial
from .fit import full_train as _full_train


def engine_class(model_class):
    """The flexynesis_amd class for a model class given by the reference's orchestration: its own classes pass through, a
    reference class (``flexynesis.models.DirectPred`` ...) maps by NAME -- the orchestration dispatches on
    ``model_class.__name__`` itself (main.py:159, 241, 243)."""
    if isinstance(model_class, type) and issubclass(model_class, _models.direct_pred.FxModel):
        return model_class
    name = getattr(model_class, "__name__", str(model_class))
    cls = getattr(_models, name, None)
    if cls is None:
        raise TypeError(f"flexynesis_amd has no engine class for model_class {name!r} "
                        f"(built: {', '.join(_models.__all__)})")
    return cls


def _dataset_of(dataset):
    """The MultiOmicDataset behind what the orchestration holds: FineTuner wraps it in TripletMultiOmicDataset for the
    triplet network (main.py:526-528); the engine samples triplets on the device from the plain dataset."""
    return getattr(dataset, "dataset", dataset) if type(dataset).__name__ == "TripletMultiOmicDataset" else dataset


def _seed(step) -> int:
    return int((torch.initial_seed() + 7919 * int(step)) % (2 ** 31))


def objective(self, params, current_step, total_steps, full_train=False):
    """``HyperparameterTuning.objective`` (reference main.py:228-333) on the engine loop."""
    model_kwargs = {}
    name = getattr(self.model_class, "__name__", "")
    if name == "GNN":                                                   # main.py:241-242
        model_kwargs["gnn_conv_type"] = self.gnn_conv_type
    if name == "CrossModalPred":                                        # main.py:243-245
        model_kwargs["input_layers"] = self.input_layers
        model_kwargs["output_layers"] = self.output_layers
    cls = engine_class(self.model_class)
    _models.base.resolve_device(self.device_type)                      # "cpu" / "mps": refused here, loudly (a failed fit would only report +inf)
    common = dict(batch_variables=self.batch_variables, surv_event_var=self.surv_event_var, surv_time_var=self.surv_time_var,
                  use_loss_weighting=self.use_loss_weighting, seed=_seed(current_step), device=self.device_type)
    if full_train:                                                      # main.py:247-262: all samples, no validation, no early stopping
        model, _ = _full_train(cls, params, self.dataset, self.target_variables, **common, **model_kwargs)
        return model
    patience = int(self.early_stop_patience) if int(self.early_stop_patience) > 0 else 0     # main.py:207-209
    val, epochs, model, info = run_trial(cls, params, self.dataset, self.target_variables, val_size=self.val_size,
                                         early_stop_patience=patience, use_cv=self.use_cv, n_splits=self.n_splits,
                                         **common, **model_kwargs)
    if "error" in info:
        print(f"[INFO] hpo config:{params} failed on the engine: {info['error']}")
    return val, int(epochs), model


def run_experiments(self):
    """``FineTuner.run_experiments`` (reference main.py:575-659) on the engine loop; ``self.model`` ends as the final model."""
    model = self.original_model
    if not isinstance(model, _models.direct_pred.FxModel):
        raise TypeError("FineTuner on the engine needs a flexynesis_amd model (train it through the adapter's objective, or "
                        "load the reference model's state_dict into the flexynesis_amd class of the same name)")
    final, best, results = fine_tune(model, _dataset_of(self.dataset), n_splits=int(self.n_splits), batch_size=int(self.batch_size),
                                     learning_rates=list(self.learning_rates), max_epoch=int(self.max_epoch),
                                     freeze_configs=list(self.freeze_configs), seed=_seed(0) % 1000, verbose=True,
                                     device=str(next(model.parameters()).device) if next(model.parameters()).is_cuda else None)
    print(f"Best learning rate: {best['learning_rate']} and freeze {best['freeze']}",
          f"with average validation loss: {best['average_val_loss']} and average epochs: {best['epochs']}")
    self.val_loss_results, self.best_config = results, best
    self.learning_rate = best["learning_rate"]
    self.model = final
    return final


def install(main_module=None, models: bool = True):
    """Point the reference's orchestration at the engine: ``HyperparameterTuning.objective`` and ``FineTuner.run_experiments``
    of ``flexynesis.main`` (or the module given) become the functions above, and -- with ``models`` -- the model classes the CLI
    looks up by name (``flexynesis.main.DirectPred`` ..., __main__.py's ``available_models``) become the engine's.  Returns the
    names it replaced."""
    if main_module is None:
        import importlib
        main_module = importlib.import_module("flexynesis.main")
    done = []
    hp = getattr(main_module, "HyperparameterTuning", None)
    if hp is not None:
        hp.objective = objective
        done.append("HyperparameterTuning.objective")
    ft = getattr(main_module, "FineTuner", None)
    if ft is not None:
        ft.run_experiments = run_experiments
        done.append("FineTuner.run_experiments")
    if models:
        for name in _models.__all__:
            if hasattr(main_module, name):
                setattr(main_module, name, getattr(_models, name))
                done.append(name)
    return done
