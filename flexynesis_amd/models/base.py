"""Shared machinery of the three hot-path model classes.

The classes keep the reference's LightningModule-style API surface (constructor kwargs, attributes,
``forward`` / ``training_step`` / ``validation_step`` / ``configure_optimizers`` / ``predict`` /
``transform``; reference models/direct_pred.py:30-415) but every step runs as a
recorded tape of HIP kernels (engine.StepPlan).  Parameters are ordinary ``nn.Parameter``s held by
torch ``nn.Linear`` / ``nn.BatchNorm1d`` containers with the reference's names; once the model is on a
GPU they are re-pointed into the engine's flat arenas (``ParamStore``), so ``state_dict`` /
``load_state_dict`` / ``parameters()`` / ``requires_grad`` keep working while the kernels see contiguous
arenas.
"""
from __future__ import annotations

import copy
import weakref
from typing import Dict, List, Optional

import numpy as np
import os

import torch
from torch import nn

from .. import ops
from ..arch import ArchSpec, is_buffer_key, spec_from_dataset
from ..engine import ParamStore, StepPlan

try:  # subclass the real LightningModule when the package exists (it is absent in the build image)
    import lightning as _pl
    _Base = _pl.LightningModule
except Exception:  # pragma: no cover - exercised in this image
    class _Base(nn.Module):
        """Duck-typed stand-in for lightning.LightningModule: nn.Module + log_dict/log + .device."""

        def log_dict(self, d, *a, **k):
            self._logged = {k_: (float(v.detach().reshape(-1)[0]) if torch.is_tensor(v) else float(v))
                            for k_, v in d.items()}

        def log(self, name, value, *a, **k):
            self._logged = getattr(self, "_logged", {})
            self._logged[name] = float(value)

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")


def resolve_device(device_type) -> torch.device:
    """'gpu' / 'cuda' / 'cuda:N' / None / 'auto' -> a HIP device (ROCm reports as torch.cuda, same strings as
    reference utils.py:2198-2238).  There is no CPU execution path."""
    if device_type in (None, "auto", "gpu", "cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("flexynesis_amd needs an AMD GPU visible to PyTorch-ROCm (no CPU fallback)")
        return torch.device("cuda", torch.cuda.current_device())
    d = torch.device(device_type)
    if d.type != "cuda":
        raise RuntimeError(f"flexynesis_amd has no '{d.type}' execution path; use the reference for CPU/MPS runs")
    return d


def _rows(mat, idx):
    """Rows ``idx`` of one omics layer as a [len(idx), F] tensor: one indexing op for tensors / arrays (the reference's
    DataLoader stacks sample by sample; 2048 samples x 2 layers took ~20 ms of host time per predict() that way)."""
    if isinstance(mat, torch.Tensor):
        return mat[torch.as_tensor(idx, dtype=torch.long, device=mat.device)]
    if isinstance(mat, np.ndarray):
        return torch.as_tensor(mat[np.asarray(idx)])
    return torch.stack([torch.as_tensor(mat[i]) for i in idx])


class _PlanLoss(torch.autograd.Function):
    """Connects a recorded StepPlan to autograd: forward has already run the forward tape; backward runs the backward
    tape (hand-written HIP kernels, gradients materialised in the arenas) and hands autograd VIEWS of those arenas, so
    ``loss.backward()`` / Lightning / torch optimisers see ``param.grad`` without another pass over the 0.8 GB of
    gradients.  The upstream gradient (1 for a plain ``loss.backward()``) is applied by a device-side scaling pass that
    reads the scalar on the GPU and returns at once when it is 1 -- no host synchronisation."""

    @staticmethod
    def forward(ctx, model, plan, *params):
        ctx.model, ctx.plan = model, plan
        n = len(plan.spec.loss_names())
        return plan.loss_vec[n:n + 1].clone()

    @staticmethod
    def backward(ctx, gout):
        model, plan = ctx.model, ctx.plan
        st = model._store
        model._unalias_grads()             # (normally done by training_step, before the forward tape touched the arena)
        plan.backward()
        scale = gout.reshape(-1)[:1].to(torch.float32).contiguous()
        ops.scale_by(ops.IMMEDIATE, st.G, scale)
        if plan.fused:
            # fused optimiser mode: the wide weights' gradients are never formed (FxAdam.step() runs the engine's clip +
            # dW+Adam launches on the saved dY / X operands); their .grad stays None.  An upstream gradient other than 1
            # cannot be applied to a gradient that does not exist: checked one step late, without a host sync.
            model._fused_check_scale(scale)
            model._fused_ready = plan
        else:
            for k in st.big_keys:
                ops.scale_by(ops.IMMEDIATE, st.big[k]["_G"].view(-1), scale)
        grads = []
        weighted = plan.spec.weighted
        for key, p in model._param_items():
            if not p.requires_grad:
                grads.append(None)
                continue
            if key.startswith("log_vars.") and not weighted:
                grads.append(None)          # reference: log_vars get no grad with a single loss term
                continue
            if plan.fused and key in st.big:
                grads.append(None)
                continue
            grads.append(st.g(key))         # a view into the gradient arena (overwritten by the next backward)
        return (None, None, *grads)


class FxAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(model.parameters(), lr)`` (reference models/direct_pred.py:135-144) for a model whose
    parameters and gradients live in the engine's arenas: ``step()`` is one ``fx_adam_flat`` launch over the
    small-parameter arena plus one per wide weight, instead of torch's multi-tensor passes.  Same update rule and
    defaults (betas 0.9 / 0.999, eps 1e-8, no weight decay); parameters whose ``.grad`` is None are skipped, like torch's.
    Gradients are whatever ``param.grad`` holds when ``step()`` is called -- i.e. after Lightning's
    ``clip_grad_norm_`` -- because ``param.grad`` IS the arena."""

    def __init__(self, model, lr, fused=False):
        self.model = model
        super().__init__(list(model.parameters()), dict(lr=float(lr)))
        # fused=True: the engine's own optimiser step -- gradient norm from the Gram identity, clip coefficient and Adam
        # for the small parameters in one launch, dW + clip + Adam per wide weight in one launch each (24 B/param, dW is
        # never stored).  Clipping then happens INSIDE step(): max_norm is set by FxModel.configure_gradient_clipping
        # (the hook Lightning's Trainer calls with its gradient_clip_val) or by hand.  Needs exactly one
        # loss.backward() per step() with unit upstream gradient (no gradient accumulation, no loss scaling).
        self.fused = bool(fused)
        self.max_norm = None
        self._mask_sig = None
        self._mask = None
        self._ctrl = None          # this optimiser's own step-control block: t counts optimizer.step() calls (the model's
                                   # block counts training_step calls for the dropout streams; they differ under
                                   # gradient accumulation)

    # -- checkpointing: the Adam moments and the step count live in the engine's arenas, not in torch's per-parameter
    # ``state``; without these overrides ``optimizer.state_dict()`` (and a Lightning checkpoint) would hold no moments and a
    # resumed run would silently restart Adam, unlike the reference's torch.optim.Adam.
    def _step_count(self, st) -> int:
        ctrl = st.ctrl if self.fused else self._ctrl
        if ctrl is None:
            return 0
        return int(ctrl[ops.CTRL_STEP]) + (int(ctrl[ops.CTRL_STEP_HI]) << 24)

    def state_dict(self):
        m = self.model
        st = m._bind()
        sd = super().state_dict()
        keys = [k for k, p in m._param_items()]
        sd["fx"] = {"step": self._step_count(st), "fused": self.fused,
                    "exp_avg": {k: st.m(k).detach().cpu().clone(memory_format=torch.contiguous_format) for k in keys},
                    "exp_avg_sq": {k: st.v(k).detach().cpu().clone(memory_format=torch.contiguous_format) for k in keys}}
        return sd

    def load_state_dict(self, state_dict):
        fx = state_dict.get("fx")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "fx"})
        if fx is None:
            return
        m = self.model
        st = m._bind()
        with torch.no_grad():
            for k, v in fx["exp_avg"].items():
                st.m(k).copy_(torch.as_tensor(v).to(torch.float32))
            for k, v in fx["exp_avg_sq"].items():
                st.v(k).copy_(torch.as_tensor(v).to(torch.float32))
            t = int(fx["step"])
            if self.fused:
                ctrl = st.ctrl
            else:
                if self._ctrl is None or self._ctrl.device != st.device:
                    self._ctrl = torch.zeros(ops.CTRL_FLOATS, dtype=torch.float32, device=st.device)
                ctrl = self._ctrl
            ctrl[ops.CTRL_STEP] = float(t % (1 << 24))
            ctrl[ops.CTRL_STEP_HI] = float(t >> 24)
        self._mask_sig = self._mask = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:               # Lightning's automatic optimisation: training_step + backward (+ clipping) run in here
            with torch.enable_grad():
                loss = closure()
        m = self.model
        st = m._bind()
        lr = float(self.param_groups[0]["lr"])
        if self.fused:
            plan = m._fused_ready
            if plan is None:                       # no backward since the last step: nothing to apply (torch skips None grads)
                return loss
            with torch.cuda.device(st.device):
                plan.set_clip(self.max_norm)
                if lr != m._ctrl_lr:               # training_step put config['lr'] into the control block; a scheduler may differ
                    st.ctrl[ops.CTRL_LR:ops.CTRL_LR + 1].fill_(lr)
                plan.run_optimizer_tape()
            m._fused_ready = None
            return loss
        with torch.cuda.device(st.device):
            items = m._param_items()
            # a gradient that is not the arena view (e.g. autograd had to clone a row-padded wide gradient, or a user set
            # it) is copied into its arena slot first
            live = []
            for key, p in items:
                has = p.grad is not None
                live.append(has)
                if has and p.grad.data_ptr() != st.g(key).data_ptr():
                    st.g(key).copy_(p.grad)
            sig = (id(st), str(st.device)) + tuple(live)      # (the mask lives in THIS store's arena layout and device)
            if sig != self._mask_sig:
                mask = torch.zeros_like(st.P)
                for (key, p), has in zip(items, live):
                    if has and key not in st.big:
                        st._view(mask, key).fill_(1.0)
                self._mask, self._mask_sig = mask, sig
            if self._ctrl is None or self._ctrl.device != st.device:
                self._ctrl = torch.zeros(ops.CTRL_FLOATS, dtype=torch.float32, device=st.device)
            ctrl = self._ctrl
            ops.step_begin(ops.IMMEDIATE, ctrl, lr, 0)               # t += 1, bias corrections; clip coefficient = 1
            ops.adam_flat(ops.IMMEDIATE, st.P, st.G, st.M, st.V, ctrl, None if all(live) else self._mask)
            for (key, p), has in zip(items, live):
                if has and key in st.big:
                    d = st.big[key]
                    ops.adam_flat(ops.IMMEDIATE, d["_W"].view(-1), d["_G"].view(-1), d["_M"].view(-1), d["_V"].view(-1), ctrl)
        return loss


class FxModel(_Base):
    MODEL = None  # "DirectPred" | "supervised_vae" | "MultiTripletNetwork" | "CrossModalPred"
    # run-time state that belongs to one bound instance: never copied, pickled or kept across .to()
    _RUNTIME = {"_store": None, "_plans": dict, "_fused_ready": None, "_fused_scale_host": None, "_fused_scale_event": None,
                "_fx_optimizer": None, "_fx_param_cache": None, "_fx_nbt_cache": None, "_fx_nbt_seen": dict}

    def _reset_runtime(self, target=None):
        d = self.__dict__ if target is None else target
        if target is None:                                 # (a pickled / copied state dict only loses its references)
            for plan in (d.get("_plans") or {}).values():  # release their hipGraphs at a known point (StepPlan.close)
                plan.close()
        for k, v in self._RUNTIME.items():
            d[k] = v() if callable(v) else v

    def close(self):
        """Release the engine state of this model (plans, their hipGraphs, the arena binding) now instead of whenever the
        object is collected.  The parameters keep their current values (they become ordinary tensors again on the next
        ``state_dict`` / ``to``); the model can be used again and re-binds lazily."""
        store = self.__dict__.get("_store")
        if store is not None:
            torch.cuda.synchronize(store.device)
        self._reset_runtime()

    def __init__(self, config, dataset, target_variables, batch_variables=None, surv_event_var=None,
                 surv_time_var=None, use_loss_weighting=True, device_type=None, **spec_kw):
        super().__init__()
        self.config = config
        self.target_variables = list(target_variables)
        self.surv_event_var = surv_event_var
        self.surv_time_var = surv_time_var
        if surv_event_var is not None and surv_time_var is not None:
            self.target_variables = self.target_variables + [surv_event_var]
        self.batch_variables = batch_variables
        self.variables = self.target_variables + list(batch_variables) if batch_variables else self.target_variables
        self.feature_importances = {}
        self.use_loss_weighting = use_loss_weighting
        self.device_type = device_type
        self.variable_types = dataset.variable_types
        self.ann = dataset.ann
        self.layers = list(dataset.dat.keys())
        self.input_dims = [len(dataset.features[l]) for l in self.layers]
        self.spec: ArchSpec = spec_from_dataset(self.MODEL, config, dataset, target_variables, batch_variables,
                                                surv_event_var, surv_time_var, use_loss_weighting, **spec_kw)
        if self.use_loss_weighting:
            self.log_vars = nn.ParameterDict({n: nn.Parameter(torch.zeros(1)) for n in self.spec.logvar_names()})
        # Initialise the parameters directly on the GPU when one is requested: drawing 2 x 100 M weights on the host
        # and copying them costs 0.6-0.9 s per model, more than a short HPO trial's whole fit (scripts/trial_setup_time.py).
        # Without a device request (or without a GPU, e.g. unpickling / CPU-side tests) construction stays on the host,
        # like the reference's.
        on_gpu = device_type is not None and str(device_type) != "cpu" and torch.cuda.is_available()
        if on_gpu:
            with torch.device(resolve_device(device_type)):
                self._build_modules()
                if self.use_loss_weighting:
                    self.log_vars = nn.ParameterDict({n: nn.Parameter(torch.zeros(1)) for n in self.spec.logvar_names()})
        else:
            self._build_modules()
        self._store: Optional[ParamStore] = None
        self._plans: Dict[tuple, StepPlan] = {}
        self._seed = int(torch.initial_seed() % (2 ** 31))
        # Level-1 fast path (see FxAdam): FX_LEVEL1_FUSED=1 / model.fused_optimizer = True before configure_optimizers();
        # None = decide in configure_optimizers() (on when a Lightning Trainer without accumulation / mixed precision drives
        # the model -- it clips through the configure_gradient_clipping hook; FX_LEVEL1_FUSED=0 turns that off)
        env = os.environ.get("FX_LEVEL1_FUSED")
        self.fused_optimizer = None if env is None else env == "1"
        self._fused_ready: Optional[StepPlan] = None
        self._fused_scale_host = None
        self._fused_scale_event = None
        self._ctrl_lr = None
        keys = [k for k, _ in self.state_dict().items()]
        want = list(self.spec.state_shapes().keys())
        assert sorted(keys) == sorted(want), (set(keys) ^ set(want))

    # -- module construction is subclass-specific -------------------------------------------------------
    def _build_modules(self):
        raise NotImplementedError

    # -- arena binding ------------------------------------------------------------------------------------
    def _param_items(self):
        # (walking the module tree costs 0.15 ms per call and the level-1 step calls it twice: cached until the parameters move, _apply)
        c = self.__dict__.get("_fx_param_cache")
        if c is None:
            c = self.__dict__["_fx_param_cache"] = list(self.named_parameters())
        return c

    def _nbt_buffers(self):
        c = self.__dict__.get("_fx_nbt_cache")
        if c is None:
            c = self.__dict__["_fx_nbt_cache"] = [(k, b) for k, b in self.named_buffers() if k.endswith("num_batches_tracked")]
        return c

    def _bind(self, device=None) -> ParamStore:
        """Move parameters/buffers into the engine arenas on ``device`` (idempotent)."""
        dev = resolve_device(device if device is not None else self.device_type)
        if self._store is not None and self._store.device == dev:
            return self._store
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        store = ParamStore(self.spec, dev, materialize_big_grads=False)
        store.load_state(sd)
        for k, p in self.named_parameters():
            p.data = store.p(k)
        for k, b in self.named_buffers():
            if not k.endswith("num_batches_tracked"):
                b.data = store.b(k)
            else:
                b.data = b.data.to(dev)
        self._store, self._plans = store, {}
        return store

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        # .to()/.cuda()/.cpu() replace parameter storage: the arenas are stale after any such move (the optimiser stays:
        # it holds the same Parameter objects)
        opt = self.__dict__.get("_fx_optimizer")
        self._reset_runtime()
        self.__dict__["_fx_optimizer"] = opt
        return out

    def _check_param_cache(self):
        """Parameters registered / replaced behind the model's back (register_parameter, ParameterDict edits, torch.__future__'s
        overwrite-on-conversion): the cached walk is compared with a fresh one whenever a plan is BUILT (not per step) and dropped,
        together with the binding, when they differ."""
        c = self.__dict__.get("_fx_param_cache")
        if c is None:
            return
        fresh = list(self.named_parameters())
        if len(fresh) != len(c) or any(a[1] is not b[1] for a, b in zip(fresh, c)):
            opt = self.__dict__.get("_fx_optimizer")
            self._reset_runtime()
            self.__dict__["_fx_optimizer"] = opt

    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in self._RUNTIME:
                continue
            setattr(new, k, copy.deepcopy(v, memo))
        new._reset_runtime()
        # cloned parameters may still alias one cloned arena; give each its own storage
        for p in new.parameters():
            p.data = p.data.clone()
        for b in new.buffers():
            b.data = b.data.clone()
        return new

    def __getstate__(self):
        st = dict(self.__dict__)
        self._reset_runtime(st)
        return st

    def _sync_nbt(self):
        if self._store is None:
            return
        seen = self.__dict__.setdefault("_fx_nbt_seen", {})
        for k, b in self._nbt_buffers():
            b.fill_(self._store.nbt[k])
            seen[k] = (b._version, b.data_ptr())

    def _plan(self, B: int, train: bool, fused: bool = False) -> StepPlan:
        store = self._bind()
        # Pick up externally loaded counters (load_state_dict after the binding, a checkpoint restore) -- but only when a counter HAS been
        # written by someone else since our last write: int(b) on a device tensor is a host synchronisation, and three of them per
        # training_step kept the host from ever running ahead of the GPU (level-1 drop-in 1.41 -> 1.3 ms per step at cfg2).
        seen = self.__dict__.setdefault("_fx_nbt_seen", {})
        for k, b in self._nbt_buffers():
            tag = (b._version, b.data_ptr())
            if seen.get(k) != tag:
                store.nbt[k] = int(b)
                seen[k] = tag
        key = (int(B), bool(train), bool(fused))
        if key not in self._plans:
            self._check_param_cache()
            store = self._bind()
            self._plans[key] = StepPlan(store, B, train=train, fused=fused, supplied_draws=False,
                                        seed=self._seed + len(self._plans), forward_alone=True)
            # Training plans of the level-1 path (driven tape by tape from an external loop) replay their tapes as hipGraphs
            # from the third use on (fused drop-in 70.9 -> 77.6 k samples/s at cfg2); FX_LEVEL1_GRAPHS=0 launches eagerly.
            # (Opt-in in round 2, when the test suite crashed intermittently with it: the cause was graph objects dying during
            # another graph's capture -- ops.retire_graph -- not the tape graphs themselves; tests/test_gpu_soak.py.)
            self._plans[key].tape_graphs = bool(train) and os.environ.get("FX_LEVEL1_GRAPHS", "1") != "0"
        return self._plans[key]

    # -- batch plumbing -------------------------------------------------------------------------------------
    def _feed(self, plan: StepPlan, batch):
        dev = plan.dev
        if self.MODEL == "MultiTripletNetwork":
            anchor, pos, neg, y = batch[0], batch[1], batch[2], batch[3]
            parts = [[torch.as_tensor(d[l]).to(dev, torch.float32) for l in self.layers] for d in (anchor, pos, neg)]
            plan.set_batch(parts=parts, y={k: torch.as_tensor(v).to(dev) for k, v in y.items() if k in plan.y})
        else:
            dat, y = batch[0], batch[1]
            xs = [torch.as_tensor(dat[l]).to(dev, torch.float32) for l in dat.keys()]
            plan.set_batch(x_list=xs, y={k: torch.as_tensor(v).to(dev) for k, v in y.items() if k in plan.y})

    @staticmethod
    def _batch_size(batch):
        first = batch[0]
        return int(next(iter(first.values())).shape[0])

    def _named_losses(self, plan, total_name):
        vals = plan.loss_vec.detach().clone()
        names = self.spec.loss_names()
        d = {n: vals[i] for i, n in enumerate(names)}
        d[total_name] = vals[len(names)]
        return d

    # -- LightningModule protocol ------------------------------------------------------------------------------
    def configure_optimizers(self):
        """Adam(lr) as the reference (models/direct_pred.py:135-144), as a ``torch.optim.Optimizer`` whose ``step()``
        runs on the arenas (``FxAdam``).  ``torch.optim.Adam(model.parameters(), lr)`` works too (it updates the arena
        views in place); pass ``FX_TORCH_ADAM=1`` to get it from here."""
        import os
        if os.environ.get("FX_TORCH_ADAM", "0") == "1" or not torch.cuda.is_available():
            return torch.optim.Adam(self.parameters(), lr=self.config["lr"])
        if self.fused_optimizer is None:
            self.fused_optimizer = self._trainer_allows_fused()
        opt = FxAdam(self, self.config["lr"], fused=self.fused_optimizer)
        self._fx_optimizer = weakref.ref(opt)       # (weak: the optimiser holds the model; a cycle would leave the arenas to the GC)
        return opt

    def _trainer_allows_fused(self) -> bool:
        """True when a Lightning Trainer is attached whose loop satisfies the fused optimiser's contract: automatic
        optimisation, one backward per step (no gradient accumulation), full precision (no loss scaling), norm clipping
        (the Trainer then routes its gradient_clip_val through configure_gradient_clipping).  Hand-written loops keep the
        default, materialised gradients, unless they opt in."""
        tr = self.__dict__.get("_trainer", None) or getattr(self, "_fx_trainer_probe", None)
        if tr is None or not getattr(self, "automatic_optimization", True):
            return False
        try:
            if int(getattr(tr, "accumulate_grad_batches", 1)) != 1:
                return False
            if str(getattr(tr, "precision", "32")).lower() not in ("32", "32-true", "64", "64-true"):
                return False
            return str(getattr(tr, "gradient_clip_algorithm", None) or "norm").lower() == "norm"
        except Exception:
            return False

    def configure_gradient_clipping(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
        """The hook Lightning's Trainer calls between backward and optimizer.step() with its ``gradient_clip_val``
        (reference main.py:216: 1.0).  With a fused FxAdam the norm clip is part of step() itself (the wide gradients it
        would have to read do not exist); otherwise the usual torch clipping."""
        opt = getattr(optimizer, "optimizer", optimizer)          # Lightning wraps optimisers in LightningOptimizer
        if isinstance(opt, FxAdam) and opt.fused:
            if gradient_clip_algorithm not in (None, "norm"):
                raise ValueError("the fused FxAdam clips by global norm only (gradient_clip_algorithm='norm')")
            opt.max_norm = float(gradient_clip_val) if gradient_clip_val else None
            return
        if not gradient_clip_val:
            return
        if gradient_clip_algorithm == "value":
            torch.nn.utils.clip_grad_value_(self.parameters(), gradient_clip_val)
        else:
            torch.nn.utils.clip_grad_norm_(self.parameters(), gradient_clip_val)

    def _fused_check_scale(self, scale):
        """Fused optimiser mode supports only loss.backward() with upstream gradient 1.  The value is copied to pinned host
        memory asynchronously and looked at when the NEXT backward arrives (no host synchronisation in the step)."""
        if self._fused_scale_host is None:
            self._fused_scale_host = torch.ones(1, dtype=torch.float32).pin_memory()
            self._fused_scale_event = torch.cuda.Event()
        elif self._fused_scale_event.query() and float(self._fused_scale_host[0]) != 1.0:
            raise RuntimeError("fused FxAdam: loss.backward() was called with an upstream gradient of "
                               f"{float(self._fused_scale_host[0])} (loss scaling / gradient accumulation): set "
                               "model.fused_optimizer = False for such loops")
        self._fused_scale_host.copy_(scale, non_blocking=True)
        self._fused_scale_event.record()

    def _unalias_grads(self):
        """Gradient accumulation (a second training_step + loss.backward() before zero_grad: Lightning's
        accumulate_grad_batches, hand-written loops): a parameter whose .grad is still the zero-copy view of the gradient
        arena from the previous backward would see the arena overwritten by this step's tapes (the supervisor heads'
        gradients are produced by the forward tape already) and then be added to ITSELF by AccumulateGrad -- 2 g2 instead
        of g1 + g2.  Such gradients are moved out of the arena before the arena is written; the first backward after
        zero_grad(set_to_none) stays zero-copy."""
        st = self._store
        if st is None:
            return
        for key, p in self._param_items():
            g = p.grad
            if g is not None and key not in st.big and g.data_ptr() == st.g(key).data_ptr():
                p.grad = g.clone()

    def training_step(self, train_batch, batch_idx, log=True):
        self._unalias_grads()
        ref = self.__dict__.get("_fx_optimizer")
        opt = ref() if ref is not None else None
        fused = bool(self.fused_optimizer) and isinstance(opt, FxAdam) and opt.fused
        plan = self._plan(self._batch_size(train_batch), train=True, fused=fused)
        self._feed(plan, train_batch)
        self._ctrl_lr = float(self.config.get("lr", 0.0))
        # the in-kernel dropout / eps / prior draws are keyed on the step counter of the control block: advance it once per
        # training_step (an external optimiser never touches it, and the same masks would be drawn every step)
        ops.step_begin(ops.IMMEDIATE, plan.store.ctrl, float(self.config.get("lr", 0.0)), 0)
        plan.forward()
        plan.bump_nbt()
        self._sync_nbt()
        params = [p for _, p in self._param_items()]
        total = _PlanLoss.apply(self, plan, *params)
        if not self.spec.weighted:
            total = total.reshape(())        # the reference's unweighted total is 0-dim, the weighted one is [1]
        if log:
            self.log_dict(self._named_losses(plan, "train_loss"), on_step=False, on_epoch=True, prog_bar=True)
        return total

    def validation_step(self, val_batch, batch_idx, log=True):
        was_training = self.training
        plan = self._plan(self._batch_size(val_batch), train=False)
        self._feed(plan, val_batch)
        plan.forward()
        losses = self._named_losses(plan, "val_loss")
        if log:
            self.log_dict(losses, on_step=False, on_epoch=True, prog_bar=True)
        if was_training:
            self.train()
        return losses["val_loss"]

    # -- inference helpers ----------------------------------------------------------------------------------------
    def _eval_batches(self, dataset, batch_size=64):
        n = len(dataset)
        for s in range(0, n, batch_size):
            idx = list(range(s, min(s + batch_size, n)))
            dat = {l: _rows(dataset.dat[l], idx) for l in dataset.dat.keys()}
            yield idx, dat

    def _run_eval(self, dat):
        B = int(next(iter(dat.values())).shape[0])
        plan = self._plan(B, train=False)
        if self.MODEL == "MultiTripletNetwork":
            xs = [torch.as_tensor(dat[l]).to(plan.dev, torch.float32) for l in self.layers]
            plan.set_batch(parts=[xs, xs, xs], y=None)
        else:
            plan.set_batch(x_list=[torch.as_tensor(dat[l]).to(plan.dev, torch.float32) for l in dat.keys()], y=None)
        for t in plan.y.values():          # labels are irrelevant for predictions: mark them missing
            t.fill_(float("nan"))
        plan.forward()
        return plan

    def predict(self, dataset):
        """{var: np.ndarray} -- softmax probabilities for categorical heads, raw outputs for numerical
        (reference models/direct_pred.py:296-351), batches of 64, eval mode."""
        self.eval()
        preds = {v: [] for v in self.variables}
        for _, dat in self._eval_batches(dataset, 64):
            plan = self._run_eval(dat)
            for v in self.variables:
                o = plan.buf[f"MLPs.{v}/out"].detach()
                if dataset.variable_types[v] == "categorical":
                    pr = torch.empty_like(o)
                    ops.softmax_rows(ops.IMMEDIATE, pr, o)
                    o = pr
                preds[v].append(o if dataset.variable_types[v] == "categorical" else o.clone())   # (the plan's buffer is overwritten by the next batch)
        # one read-back at the end: a .cpu() per 64-row batch was a host synchronisation per batch
        return {v: (torch.cat(a, 0).cpu().numpy() if a else np.array([])) for v, a in preds.items()}

    def transform(self, dataset):
        """Latent embeddings as a DataFrame E0..E{L-1} indexed by sample (reference direct_pred.py:353-415)."""
        import pandas as pd
        self.eval()
        chunks = []
        for _, dat in self._eval_batches(dataset, 64):
            plan = self._run_eval(dat)
            chunks.append(plan.embeddings.detach().clone())          # (on the device: one read-back at the end)
        emb = torch.cat(chunks, 0).cpu().numpy()
        return pd.DataFrame(emb, index=list(dataset.samples), columns=[f"E{i}" for i in range(emb.shape[1])])

    # -- attributions (reference models/direct_pred.py:418-590, called by the CLI at __main__.py:1385-1400) ---------------
    @staticmethod
    def _ig_quadrature(n_steps: int):
        """Captum's default IntegratedGradients rule ("gausslegendre"): nodes / weights of the n-point Gauss-Legendre
        rule mapped from [-1, 1] to [0, 1]."""
        xs, ws = np.polynomial.legendre.leggauss(int(n_steps))
        return (0.5 * (1.0 + xs)).tolist(), (0.5 * ws).tolist()

    def compute_feature_importance(self, dataset, target_var, method="IntegratedGradients", steps_or_samples=5,
                                   batch_size=512, alphas=None, eps=None):
        """Mean absolute attribution of every input feature for ``target_var`` (one row set per class of a categorical
        target), as the reference computes it through Captum with all-zero baselines:

          IntegratedGradients  attr = x * sum_i w_i * dF(alpha_i x)/dx   Gauss-Legendre nodes alpha_i, weights w_i
          GradientShap         attr = x * mean_i dF(alpha_i x)/dx         alpha_i ~ U(0, 1), one draw per sample block

        where F is the eval-mode head output (logit of the class / the regression output).  The forward passes and the
        input gradients run on the HIP kernels (eval plans with input-gradient tapes); the result has the reference's
        DataFrame layout and is stored in ``self.feature_importances[target_var]``.  The VAE family differentiates through
        the SAMPLED latent z = mean + log_var * eps like the reference (a fresh eps per forward; CrossModalPred attributes
        its input layers; the GNN attributes its node features and reports them per omics layer and node).  ``alphas`` overrides the GradientShap draws and ``eps(batch, chunk_start, draw, rows)`` the
        reparameterisation draws (tests).  Captum is not installed in this image: the quadrature / sampling rule is restated
        from its documentation (parity unpinned for that part; oracle/attribution.py)."""
        import pandas as pd
        if method not in ("IntegratedGradients", "GradientShap"):
            raise ValueError(f"Unsupported method '{method}'. Choose 'IntegratedGradients' or 'GradientShap'.")
        if target_var not in self.variables:
            raise KeyError(target_var)
        n_draws = int(steps_or_samples)
        if dataset.variable_types[target_var] == "numerical":
            num_class = 1
        else:
            num_class = len(np.unique(np.asarray(dataset.ann[target_var])))
        self.eval()
        all_layers = list(dataset.dat.keys())
        vae = self.spec.is_vae
        # the layers that are differentiated: every layer, or the VAE family's encoder inputs (CrossModalPred: input_layers)
        layers = [all_layers[i] for i in self.spec.enc_idx] if vae else all_layers
        store = self._bind()
        dev = store.device
        gen = torch.Generator().manual_seed(self._seed)
        n = len(dataset)
        sums = [[torch.zeros(len(dataset.features[l]), dtype=torch.float64, device=dev) for l in layers] for _ in range(num_class)]
        CH = 128                                              # rows per launch chain (the eval kernels' register-resident limit)
        for s0 in range(0, n, int(batch_size)):               # the reference's DataLoader batches (one alpha set per batch)
            rows = list(range(s0, min(s0 + int(batch_size), n)))
            if method == "IntegratedGradients":
                al, wt = self._ig_quadrature(n_draws)
            else:
                al = list(alphas) if alphas is not None else torch.rand(n_draws, generator=gen).tolist()
                wt = [1.0 / n_draws] * n_draws
            for c0 in range(0, len(rows), CH):
                idx = rows[c0:c0 + CH]
                B = len(idx)
                key = (B, "attr", eps is not None)
                if key not in self._plans:
                    for k, b in self.named_buffers():
                        if k.endswith("num_batches_tracked"):
                            store.nbt[k] = int(b)
                    self._plans[key] = StepPlan(store, B, train=False, attribution=True, seed=self._seed + 4242,
                                                supplied_draws=eps is not None)
                plan = self._plans[key]
                xall = [_rows(dataset.dat[l], idx).to(dev, torch.float32) for l in all_layers]
                xs = [xall[all_layers.index(l)] for l in layers]
                for t in plan.y.values():
                    t.fill_(float("nan"))
                acc = [[torch.zeros_like(x) for x in xs] for _ in range(num_class)]
                for di, (a_i, w_i) in enumerate(zip(al, wt)):
                    # the interpolation point: the differentiated layers scaled by alpha (the others are not read by the heads)
                    scaled = [x * float(a_i) if l in layers else x for l, x in zip(all_layers, xall)]
                    if self.MODEL == "MultiTripletNetwork":
                        plan.set_batch(parts=[scaled, scaled, scaled], y=None)
                    else:
                        plan.set_batch(x_list=scaled, y=None)
                    if eps is not None:                       # parity tests: the reparameterisation draw of this forward
                        plan.set_draws({"eps": torch.as_tensor(eps(s0 // int(batch_size), c0, di, B)).to(dev)})
                    plan.forward()
                    for c in range(num_class):
                        do = plan.attr_dout[target_var]
                        do.zero_()
                        do[:, c if num_class > 1 else 0] = 1.0
                        plan.input_gradient(target_var)
                        for j in range(len(layers)):
                            acc[c][j].add_(plan.dX[j][:, :xs[j].shape[1]], alpha=float(w_i))     # (the engine's buffer may be wider: zero pad columns)
                for c in range(num_class):
                    for j in range(len(layers)):
                        sums[c][j] += (acc[c][j] * xs[j]).abs().sum(0).double()
        df_list = []
        if self.MODEL == "GNN":
            # reference gnn_early.py:599-631: one row set per omics layer of the underlying dataset, names = the graph's nodes,
            # column layer_idx of the [nodes, node_features] importances -- the reference enumerates multiomic_dataset.dat in
            # its own key order while the node features are stacked in SORTED layer order (data.py:1219); reproduced as is
            omics = list(getattr(dataset, "multiomic_dataset", dataset).dat.keys())
            F = int(self.spec.gnn["node_features"])
            for c in range(num_class):
                label = dataset.label_mappings[target_var].get(c) if target_var in getattr(dataset, "label_mappings", {}) else ""
                imp = (sums[c][0] / n).float().cpu().numpy().reshape(-1, F)
                for li, lname in enumerate(omics):
                    col = imp[:, 0] if F == 1 else imp[:, li]
                    df_list.append(pd.DataFrame({"target_variable": target_var, "target_class": c, "target_class_label": label,
                                                 "layer": lname, "name": dataset.common_features, "importance": col}))
            df_imp = pd.concat(df_list, ignore_index=True)
            self.feature_importances[target_var] = df_imp
            return df_imp
        for c in range(num_class):
            for j, l in enumerate(layers):
                label = dataset.label_mappings[target_var].get(c) if target_var in getattr(dataset, "label_mappings", {}) else ""
                df_list.append(pd.DataFrame({"target_variable": target_var, "target_class": c, "target_class_label": label,
                                             "layer": l, "name": dataset.features[l],
                                             "importance": (sums[c][j] / n).float().cpu().numpy()}))
        df_imp = pd.concat(df_list, ignore_index=True)
        self.feature_importances[target_var] = df_imp
        return df_imp

    def load_state_dict(self, state_dict, strict=True, **kw):
        """``assign=True`` REPLACES the Parameter objects (torch >= 2.1): the engine's arenas, the cached parameter walk and the optimiser's
        references would go on using the old ones -- the binding is dropped and rebuilt lazily from the new parameters."""
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        if kw.get("assign"):
            opt = self.__dict__.get("_fx_optimizer")
            self._reset_runtime()
            self.__dict__["_fx_optimizer"] = opt
        if self._store is not None:
            for k, b in self.named_buffers():
                if k.endswith("num_batches_tracked"):
                    self._store.nbt[k] = int(b)
        return out
