"""MI355X training engine for the flexynesis hot path.

``ParamStore``  owns one model instance's parameters on one GPU: every "small" parameter (biases,
                BatchNorm affine, latent/fusion/head weights, log_vars) lives in ONE flat fp32 arena
                with matching grad / Adam-m / Adam-v arenas, so the optimiser is a single launch; each
                wide omics weight (>= ``big_threshold`` elements, e.g. layer_1.weight [5000,20000])
                has its own W/m/v (and, only when gradients must be materialised, dW).
``StepPlan``    the hand-derived forward + backward + optimiser schedule of one model for one batch
                size, recorded once as a tape of C-ABI launches (ops.TapeRecorder) and then re-issued
                (or hipGraph-replayed) every step.  Replaces, per step, the reference's
                training_step -> zero_grad -> backward -> clip_grad_norm_(1.0) -> Adam.step
                (models/direct_pred.py:225-260, main.py:212-225) and validation_step (:262-294).
``Engine``      epoch loop over a device-resident cohort: seeded on-device shuffling, drop_last,
                validation, early stopping -- the counterpart of trainer.fit/validate inside
                HyperparameterTuning.objective (main.py:228-333).

Two gradient modes:
  fused=True   (engine fast path) wide-layer dW is never written: its squared norm comes from the Gram
               identity |dY^T X|_F^2 = <X X^T, dY dY^T>, then dW tiles are produced on the MFMA and
               consumed by Adam in registers (fx_linear_dw_adam_f32): 24 B/param of HBM traffic.
  fused=False  (drop-in path: loss.backward() semantics) every gradient is materialised in the grad
               arenas so an external optimiser / Lightning loop can consume ``param.grad``.
"""
from __future__ import annotations

import math
import os
import threading
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .arch import ArchSpec, is_buffer_key, is_tail_key, pad4
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU, TapeRecorder, Workspace

DROPOUT_P = 0.1          # nn.Dropout(p=0.1), reference modules.py:132
GNN_DROPOUT_P = 0.2      # flexGCN(dropout_rate=0.2), reference modules.py:204
MMD_PRIOR = 200          # torch.randn(200, latent_dim), reference supervised_vae.py:545
CLIP_MAX_NORM = 1.0      # gradient_clip_val=1.0, reference main.py:216
TRIPLET_MARGIN = 1.0


def default_precision() -> str:
    """Contraction mode of the wide layers for plans built without an explicit ``precision``: FX_PRECISION = bf16x3 (default: the parity
    mode, split bf16 with three products), bf16 (the throughput mode: one product) or f32 (exact-fp32 MFMA)."""
    return os.environ.get("FX_PRECISION", "bf16x3")


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


_PLACEMENT = threading.local()


class placement_tries:
    """``with placement_tries(n): ...``: ParamStores created by this thread inside the block try at most n placements per wide weight
    (ParamStore._place_big).  HPO trials use a small n: a fit of a few dozen steps does not earn back a long search."""

    def __init__(self, n: int):
        self.n = int(n)

    def __enter__(self):
        self._old = getattr(_PLACEMENT, "tries", None)
        _PLACEMENT.tries = self.n
        return self

    def __exit__(self, *exc):
        _PLACEMENT.tries = self._old
        return False


def _wait_events(device, events, unsynced=False):
    """Order the caller's current stream behind the streams that last used a range (the events its previous owner recorded when it
    let go); a range released while nothing could be recorded (during a hipGraph capture) costs one device synchronisation."""
    if unsynced:
        torch.cuda.synchronize(device)
        return
    st = torch.cuda.current_stream(device)
    for ev in events:
        st.wait_event(ev)


class PlacementPool:
    """Rated arrays for wide weights, kept for the life of the process (per device and SHAPE).  ``take`` hands out fast arrays of exactly
    the shape asked for; an array comes back by itself when the LAST view of the memory has gone (ops.LEASES: the ParamStore that took it,
    every nn.Parameter .data and state_dict tensor that aliased it) -- only arrays rated at least ParamStore.PLACE_GOOD_TBS are kept, at most
    FX_PLACEMENT_POOL_GB (default 64) per device; the rest go back to the allocator.  A rating does NOT travel to another shape: the
    first half of a fast [10000, 20000] array, used as [5000, 20000], probes at 4.7-4.9 TB/s, slower than most fresh arrays of that shape
    (profiles/r05_placement_pool.txt) -- so the pool serves series of models of ONE shape: the k folds of a cross-validated trial, the
    lr x freeze x fold grid of the FineTuner (45 fits of one model), repeated fits in one process.  Thread-safe (trials in flight on
    several host threads); an array handed back carries the events of the streams that last used it, and the taker's stream waits for
    them."""

    def __init__(self):
        self._lock = threading.Lock()
        self._free: Dict[int, list] = {}
        self.stats = {"given": 0, "taken": 0, "dropped": 0}

    def _cap_bytes(self) -> int:
        return int(float(os.environ.get("FX_PLACEMENT_POOL_GB", "64")) * (1 << 30))

    def take(self, device, shape: Tuple[int, int], count: int):
        ops.LEASES.drain()
        out = []
        with self._lock:
            free = self._free.get(device.index, [])
            rest = []
            for e in free:             # (by position: list.remove() would compare the entries' tensors element by element)
                if len(out) < count and e[3] == tuple(shape):
                    out.append(e)
                else:
                    rest.append(e)
            free[:] = rest
            self.stats["taken"] += len(out)
        for e in out:
            _wait_events(device, e[2], e[4])
        return [(e[0], e[1]) for e in out]

    def give(self, device, shape: Tuple[int, int], arrays, events=(), unsynced: bool = False):
        """``arrays``: [(flat array, TB/s or None)] that NOTHING views any more (ParamStore hands its arrays back through their leases)."""
        good = float(os.environ.get("FX_PLACEMENT_GOOD_TBS", ParamStore.PLACE_GOOD_TBS))
        keep = [(flat, tbs) for flat, tbs in arrays if tbs is not None and tbs >= good]
        if not keep:
            return
        with self._lock:
            free = self._free.setdefault(device.index, [])
            held = sum(e[0].numel() * 4 for e in free)
            for flat, tbs in keep:
                if held + flat.numel() * 4 > self._cap_bytes():
                    self.stats["dropped"] += 1
                    continue
                free.append((flat, tbs, list(events), tuple(shape), bool(unsynced)))
                held += flat.numel() * 4
                self.stats["given"] += 1

    def held(self, device) -> List[Tuple[Tuple[int, int], float]]:
        ops.LEASES.drain()
        with self._lock:
            return [(e[3], e[1]) for e in self._free.get(device.index, [])]

    def clear(self, device=None):
        with self._lock:
            if device is None:
                self._free.clear()
            else:
                self._free.pop(device.index, None)


POOL = PlacementPool()


class PartitionArena:
    """Where W and (m, v) of every wide weight live so that the dominant kernel streams them at the FAST rate, by construction.

    What rounds 2-5 called the placement lottery is this (profiles/r05_partitions.txt): the GPU's memory consists of a few PARTITIONS,
    contiguous physical ranges tens of GB long, each of which delivers ~4.9 TB/s to the fused kernel's access pattern whatever lies in it;
    arrays in two different partitions are streamed at 6.1 TB/s together.  A fresh process allocates from one partition, so W, m and v
    of a weight usually share it (the "slow placement"); the "fast" ones were arrays that happened to straddle a boundary or to land
    on different sides of one.  Two arrays are in different partitions exactly when fx_placement_probe over the PAIR runs fast.

    The arena is built once per process and device, at the first wide weight: pool A is allocated where the process stands; spacer
    blocks are allocated (and kept) in steps until a test array behind them rates fast against pool A -- the allocator has crossed a
    boundary --, pool B is allocated there, the spacers go back to the driver.  Every wide weight then takes W from pool A and m, v
    from pool B (a third / two thirds of the kernel's traffic).  No per-weight search, nothing to rate per shape, HPO trials included.
    Falls back to ParamStore's search (returns None) when disabled (FX_PARTITION_ARENA=0), when the device has no room for the
    spacers, when no boundary is found, or when a pool is exhausted."""

    GRAN = 1 << 21                      # sub-allocation granularity (bytes)
    REF = (5000, 20000)                 # shape of the arrays the boundary search rates
    FAST_TBS = 5.6                      # pair rate that means "two partitions" (one: 4.9-5.0, two: 6.0-6.1)
    _arenas: Dict[int, "PartitionArena"] = {}
    _glock = threading.Lock()

    def __init__(self, device):
        self.device = device
        self.lock = threading.Lock()
        self.chunks: List[torch.Tensor] = []                         # uint8 tensors: chunk 0 is pool A, the rest make up pool B
        self.kind: List[int] = []                                    # 0: pool A (W), 1: pool B (m, v)
        self.free: List[List[Tuple[int, int]]] = []                  # per chunk: (offset, bytes) free ranges, sorted
        self.pending: List[list] = []                                # [chunk, offset, bytes, events, unsynced] of ranges that came back: whoever
        #                                                              takes memory overlapping one waits for ITS events (one entry per hand-back)
        self.info: dict = {}
        self.ok = False

    @classmethod
    def get(cls, device) -> Optional["PartitionArena"]:
        if os.environ.get("FX_PARTITION_ARENA", "1") == "0" or ops.capturing():
            return None
        with cls._glock:
            a = cls._arenas.get(device.index)
            if a is None:
                a = cls._arenas[device.index] = PartitionArena(device)
                try:
                    a._build()
                except Exception as e:             # (out of memory while stepping, a runtime error of the probe: the legacy path takes over)
                    a.info["error"] = repr(e)[:200]
                    a.chunks, a.kind, a.free = [], [], []
                    a.ok = False
                    torch.cuda.empty_cache()
            return a if a.ok else None

    @classmethod
    def reset(cls, device=None):
        with cls._glock:
            if device is None:
                cls._arenas.clear()
            else:
                cls._arenas.pop(device.index, None)

    def _ref_view(self, buf: torch.Tensor, off: int = 0) -> torch.Tensor:
        r, c = self.REF
        return buf[off:off + r * c * 4].view(torch.float32).view(r, c)

    def _pair(self, a: torch.Tensor, b: torch.Tensor) -> float:
        r, c = self.REF
        return 16.0 * r * c / ops.placement_probe_us(a, b, None, launches=2) / 1e6

    def _build(self):
        import time
        t0 = time.perf_counter()
        dev = self.device
        gb = 1 << 30
        a_bytes = int(float(os.environ.get("FX_ARENA_A_GB", "8")) * gb)
        b_bytes = int(float(os.environ.get("FX_ARENA_B_GB", "16")) * gb)
        chunk = int(float(os.environ.get("FX_ARENA_CHUNK_GB", "4")) * gb)
        step = int(float(os.environ.get("FX_ARENA_STEP_GB", "8")) * gb)
        max_spacer = int(float(os.environ.get("FX_ARENA_MAX_SPACER_GB", "112")) * gb)    # (boundaries seen so far: 0-64 GB along)
        ref_bytes = self.REF[0] * self.REF[1] * 4
        last = lambda n: (n - ref_bytes) // self.GRAN * self.GRAN
        with torch.cuda.device(dev):
            free_b, _total = torch.cuda.mem_get_info(dev)
            budget = min(max_spacer, int(free_b * 0.45) - a_bytes - b_bytes)      # (two processes may be doing this on one GPU)
            if budget < 2 * step:
                self.info["skipped"] = f"{free_b / gb:.0f} GB free: no room to look for a partition boundary"
                return
            # Allocation is what the build costs, and its price differs 60-fold between boxes (4 ms per GB on most, 250 on one: 96 GB of
            # pools + spacers took 23.7 s there): every further allocation is priced at the rate measured so far and the build stops
            # (-> the bounded search) when FX_ARENA_BUDGET_S (5) would be exceeded.
            budget_s = float(os.environ.get("FX_ARENA_BUDGET_S", "5"))
            torch.cuda.synchronize(dev)
            ta = time.perf_counter()
            pool_a = torch.empty(a_bytes, dtype=torch.uint8, device=dev)
            pool_a[::1 << 21].zero_()
            torch.cuda.synchronize(dev)
            alloc_bytes, alloc_s = a_bytes, max(time.perf_counter() - ta, 1e-4)
            self.info["alloc_ms_per_GB"] = round(1e3 * alloc_s / (a_bytes / gb), 1)

            def affordable(nbytes):
                return (time.perf_counter() - t0) + nbytes * (alloc_s / alloc_bytes) <= budget_s

            def timed_empty(nbytes):
                nonlocal alloc_bytes, alloc_s
                t1 = time.perf_counter()
                buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                alloc_bytes += nbytes
                alloc_s += time.perf_counter() - t1
                return buf
            a0, a1 = self._ref_view(pool_a), self._ref_view(pool_a, last(a_bytes))
            self.info["pool_A_ends_TBps"] = round(self._pair(a0, a1), 2)       # (slow: the pool lies in one partition)
            spacers, spent, rates, good = [], 0, [], []
            # candidates of pool B, one chunk at a time: a chunk is kept when BOTH its ends rate fast against BOTH ends of pool A (it lies in
            # other partitions than pool A over its whole length); a chunk that does not stays allocated as a spacer, with a further spacer
            # behind it, so that the next candidate comes from further along
            # FX_ARENA_PARTITIONS=3 (A/B, not the default: the third partition can lie > 100 GB = 2 s of allocations away) keeps stepping
            # until half of pool B lies in a partition other than pool A's AND the first chunk's
            want3 = os.environ.get("FX_ARENA_PARTITIONS", "2") == "3"
            n2 = 0
            while (spent + chunk <= budget and affordable(chunk)
                   and (sum(c.numel() for c in good) < b_bytes or (want3 and 2 * n2 * chunk < b_bytes))):
                cand = timed_empty(chunk)
                c0, c1 = self._ref_view(cand), self._ref_view(cand, last(chunk))
                r = min(self._pair(a0, c0), self._pair(a0, c1), self._pair(a1, c0), self._pair(a1, c1))
                rates.append(round(r, 2))
                other = bool(good) and self._pair(self._ref_view(good[0]), c0) >= self.FAST_TBS
                full = sum(c.numel() for c in good) >= b_bytes
                if r >= self.FAST_TBS and (not full or other):
                    good.append(cand)
                    n2 += int(other)
                else:
                    spacers.append(cand)
                    spent += chunk
                    if spent + step <= budget and affordable(step + chunk):
                        spacers.append(timed_empty(step))
                        spent += step
            self.info.update(spacer_GB=round(spent / gb, 1), candidate_TBps=rates)
            del spacers
            if sum(c.numel() for c in good) < min(b_bytes, 2 * chunk):
                del good, pool_a, a0, a1
                torch.cuda.empty_cache()
                self.info["skipped"] = "no partition boundary within the spacer / time budget"
                self.info["build_s"] = round(time.perf_counter() - t0, 3)
                return
            torch.cuda.empty_cache()                       # the spacers go back to the driver; the pools stay where they are
            # pool B's chunks may themselves lie in two partitions (the stepping often passes more than one boundary): those that rate
            # fast against the first one form class 2 -- m is then taken from class 1 and v from class 2, three partitions for nothing
            # (the real kernel: 456-460 us against 463-469 with two, profiles/r05_partitions.txt item 4)
            g0 = self._ref_view(good[0])
            cls = [1] + [2 if self._pair(g0, self._ref_view(c)) >= self.FAST_TBS else 1 for c in good[1:]]
            self.info["pool_B_classes"] = cls
            self.chunks = [pool_a] + good
            self.kind = [0] + cls
            self.free = [[(0, c.numel())] for c in self.chunks]
            self.ok = True
            self.info.update(pool_GB=[a_bytes / gb, sum(c.numel() for c in good) / gb], build_s=round(time.perf_counter() - t0, 3))

    def _alloc(self, kind: int, nbytes: int) -> Optional[Tuple[int, int, int]]:
        size = (nbytes + self.GRAN - 1) // self.GRAN * self.GRAN
        for ci, fl in enumerate(self.free):
            if self.kind[ci] != kind:
                continue
            for i, (off, sz) in enumerate(fl):
                if sz >= size:
                    if sz == size:
                        fl.pop(i)
                    else:
                        fl[i] = (off + size, sz - size)
                    return ci, off, size
        return None

    def _release(self, ci: int, off: int, size: int):
        fl = self.free[ci]
        fl.append((off, size))
        fl.sort()
        merged = []
        for o, z in fl:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1] = (merged[-1][0], merged[-1][1] + z)
            else:
                merged.append((o, z))
        fl[:] = merged

    def free_bytes(self) -> Tuple[int, int]:
        ops.LEASES.drain()
        with self.lock:
            return tuple(sum(z for ci, fl in enumerate(self.free) if (self.kind[ci] > 0) == bool(k) for _, z in fl) for k in (0, 1))

    def _grow(self, want_a: bool, nbytes: int) -> bool:
        """One more chunk for pool A (``want_a``) or pool B, at least ``nbytes`` long: a weight larger than what is free in a pool (early
        fusion at hidden_dim_factor 2: 12.8 GB per array against the 8 GB pool A is built with) used to fall back to the allocator's
        first placement -- the slow one.  The new chunk joins the pool whose partition it lies in: both of its ends are rated against
        both ends of pool A's first chunk (fast pair = other partition: pool B; slow = pool A's).  A chunk that lies across a boundary
        is given back.  Bounded by FX_ARENA_GROW_GB (default 64) of growth per device and by a third of what is free now.
        In practice this grows pool A: a fresh allocation comes from where the process stands, i.e. pool A's partition (a chunk that
        does is kept for pool A whichever pool asked); the partition of pool B lay up to 100 GB of spacers away when the arena was
        built, so a model whose m / v outgrow pool B needs FX_ARENA_B_GB set before the first wide weight -- or takes the bounded
        per-weight search, as before."""
        gb = 1 << 30
        size = max((nbytes + self.GRAN - 1) // self.GRAN * self.GRAN, int(float(os.environ.get("FX_ARENA_CHUNK_GB", "4")) * gb))
        cap = int(float(os.environ.get("FX_ARENA_GROW_GB", "64")) * gb)
        ref_bytes = self.REF[0] * self.REF[1] * 4
        if ops.capturing() or size < 2 * ref_bytes:
            return False
        with torch.cuda.device(self.device):
            for _ in range(1):
                free_b = torch.cuda.mem_get_info(self.device)[0]
                with self.lock:
                    grown = self.info.get("grown_GB", 0.0) * gb
                if grown + size > cap or size > free_b // 3:
                    return False
                try:
                    cand = torch.empty(size, dtype=torch.uint8, device=self.device)
                except torch.OutOfMemoryError:
                    return False
                last = (size - ref_bytes) // self.GRAN * self.GRAN
                a = self.chunks[0]
                a0, a1 = self._ref_view(a), self._ref_view(a, (a.numel() - ref_bytes) // self.GRAN * self.GRAN)
                c0, c1 = self._ref_view(cand), self._ref_view(cand, last)
                rates = [self._pair(a0, c0), self._pair(a0, c1), self._pair(a1, c0), self._pair(a1, c1)]
                in_b, in_a = min(rates) >= self.FAST_TBS, max(rates) < self.FAST_TBS
                if not (in_a or in_b):
                    del cand, c0, c1
                    torch.cuda.empty_cache()
                    continue
                kind = 0 if in_a else 1
                with self.lock:
                    self.chunks.append(cand)
                    self.kind.append(kind)
                    self.free.append([(0, size)])
                    self.info["grown_GB"] = round((grown + size) / gb, 1)
                    self.info.setdefault("grown", []).append(["A" if kind == 0 else "B", round(size / gb, 1), round(min(rates), 2), round(max(rates), 2)])
                if (kind == 0) == want_a:
                    return True
        return False

    def _events_for(self, ci: int, off: int, size: int):
        """(events, unsynced) of the hand-backs whose ranges overlap [off, off + size) of chunk ci; entries whose events have all
        completed are dropped on the way (called under the lock)."""
        evs, unsynced, keep = [], False, []
        for e in self.pending:
            done = not e[4] and all(ev.query() for ev in e[3])
            if done:
                continue
            if e[0] == ci and e[1] < off + size and off < e[1] + e[2]:
                evs.extend(e[3])
                unsynced = unsynced or e[4]
                if e[1] >= off and e[1] + e[2] <= off + size:
                    continue                       # wholly inside what is being handed out: the taker's stream now orders it
            keep.append(e)
        self.pending = keep
        return evs, unsynced

    def take3(self, need_elems: int):
        """(W, m, v) as flat fp32 tensors of need_elems elements -- W from pool A, m and v from pool B -- zero-filled, each a LEASE
        (ops.LEASES) on its range: the range returns to its pool when the last view of that tensor's storage has gone, wherever
        that happens, and the next taker waits for the events recorded by ``ParamStore.release_big`` on the streams that used it.
        Returns (tensors, lease ids), or None when a pool has no room."""
        import gc
        nbytes = need_elems * 4
        for attempt in (0, 1, 2, 3):
            ops.LEASES.drain()
            short_a = None
            with self.lock:
                token = []
                for kinds in ((0,), (1, 2), (2, 1)):          # W | m | v; m and v share a class when the other has no room (or does not exist)
                    r = None
                    for kind in kinds:
                        r = self._alloc(kind, nbytes)
                        if r is not None:
                            break
                    if r is None:
                        short_a = kinds == (0,)
                        for t in token:
                            self._release(*t)
                        token = None
                        break
                    token.append(r)
                waits = [self._events_for(*t) for t in token] if token else None
            if token is not None:
                break
            if attempt == 0 and not ops.capturing():
                gc.collect()                   # ranges of dropped models that sit in reference cycles come back now
            elif attempt == 3 or not self._grow(bool(short_a), nbytes):     # one more chunk for the pool that ran short (W, then m / v)
                break
        if token is None:
            return None
        for evs, unsynced in waits:
            _wait_events(self.device, evs, unsynced)
        views, ids = [], []
        for ci, off, size in token:
            t, lid = ops.LEASES.wrap(self.chunks[ci].data_ptr() + off, need_elems, self.device,
                                     (lambda evs, unsynced, r=(ci, off, size), keep=self.chunks[ci]: self._returned(r, evs, unsynced)))
            # (``keep``: the lease does not own the pool's memory -- the chunk must outlive every view of it, also across reset())
            t.zero_()
            views.append(t)
            ids.append(lid)
        return views, ids

    def _returned(self, r, events, unsynced):
        """A lease's storage has gone (ops.LEASES.drain): the range is free again, with the events its last owner recorded."""
        with self.lock:
            if not self.ok or r[0] >= len(self.free):
                return
            self._release(*r)
            if events or unsynced:
                self.pending.append([r[0], r[1], r[2], list(events), bool(unsynced)])


def _cuda_device(device=None) -> torch.device:
    device = torch.device("cuda" if device is None else device)
    return torch.device("cuda", torch.cuda.current_device()) if device.index is None else device


def placement_memory(device=None) -> dict:
    """What this process holds on ``device`` for the placement of wide weights, outside torch's per-tensor accounting: the partition
    arena's pools (resident for the life of the process unless released) and the rated arrays of the placement pool."""
    dev = _cuda_device(device)
    ar = PartitionArena._arenas.get(dev.index)
    res = sum(c.numel() for c in ar.chunks) if (ar is not None and ar.ok) else 0
    free = sum(ar.free_bytes()) if (ar is not None and ar.ok) else 0
    pooled = sum(int(np.prod(shp)) * 4 for shp, _ in POOL.held(dev))
    return {"arena_resident_bytes": int(res), "arena_free_bytes": int(free), "pool_bytes": int(pooled),
            "arena_build_spacer_GB": (ar.info.get("spacer_GB") if ar is not None else None)}


def release_placement_memory(device=None):
    """Give the partition arena's pools and the placement pool's arrays of ``device`` (default: every device) back to the allocator
    and the driver -- for a process that is done training and wants its 24 GB back, or that shares the GPU.  Ranges still viewed by
    live models stay valid (their leases keep the pool's memory alive) and are freed with them; the next wide weight builds a new
    arena."""
    dev = None if device is None else _cuda_device(device)
    ops.LEASES.drain()
    PartitionArena.reset(dev)
    POOL.clear(dev)
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def search_arrays(device, out: int, fin: int, want: int, tries: int, seed: int = 20240):
    """Up to ``want`` flat arrays of out x pad32(fin) floats that stream the dW + Adam traffic pattern at >= PLACE_GOOD_TBS, found among
    at most 8 x tries candidates allocated behind spacers of varying size.  Returns ([(array, TB/s)], info); when the fast group is
    out of reach (budget, memory, a shape whose fast placements do not reach the absolute rate) the best arrays seen fill up the
    list.  Memory discipline: the spacer is capped by FX_PLACEMENT_SPACER_GB and by a quarter of what is free NOW, an allocation
    failure drops the spacer and tries once more, and with less than ~6 arrays of free memory there is no search at all."""
    import time
    ld = (fin + 31) // 32 * 32
    need = out * ld
    good_tbs = float(os.environ.get("FX_PLACEMENT_GOOD_TBS", ParamStore.PLACE_GOOD_TBS))
    budget = float(os.environ.get("FX_PLACEMENT_BUDGET_S", "0.6"))
    cap_gb = float(os.environ.get("FX_PLACEMENT_SPACER_GB", "32"))
    rs = np.random.RandomState(seed)
    fixed_mb = [0, 6, 3, 254, 1201, 5, 777, 2403]
    kept, spare, probes = [], [], []
    s_per_mb = 0.0
    t0 = time.perf_counter()

    def tbs_of(us):
        return 8.0 * out * fin / us / 1e6

    def alloc():
        try:
            return torch.zeros(need, dtype=torch.float32, device=device)
        except torch.OutOfMemoryError:
            return None
    with torch.cuda.device(device):
        for t in range(8 * tries):
            free_b = torch.cuda.mem_get_info(device)[0]
            if free_b < 6 * need * 4:
                break
            sp = None
            t_it = time.perf_counter()
            sp_mb = 0
            if t:
                # Allocating a spacer beyond ~2 GB costs time in proportion to its size (measured 50-65 ms per GB: 0.26 s for 4 GB, 0.82 s
                # for 16 GB; below 2 GB it is free): the cap follows what is left of the budget at that rate, or at the rate observed
                # in this search if that is worse.
                left = max(budget - (t_it - t0), 0.0)
                time_mb = 2048 + int(0.5 * left / max(s_per_mb, 65e-6))
                cap_mb = max(16, min(int(0.25 * free_b) >> 20, int(cap_gb * 1024), max(time_mb, 16)))
                mb = fixed_mb[t] if t < len(fixed_mb) else int(np.exp(rs.uniform(np.log(4.0), np.log(float(cap_mb)))))
                sp_mb = min(mb, cap_mb)
                try:           # (a spacer is a means, not a need)
                    sp = torch.empty(sp_mb << 20, dtype=torch.uint8, device=device)
                except torch.OutOfMemoryError:
                    sp = None
            a = alloc()
            if a is None and sp is not None:          # the spacer was in the way: without it
                del sp
                sp = None
                torch.cuda.empty_cache()
                a = alloc()
            if a is None:
                break
            us = ops.placement_probe_us(a.view(out, ld)[:, :fin])
            probes.append(round(tbs_of(us), 2))
            (kept if tbs_of(us) >= good_tbs else spare).append((us, a))
            spare.sort(key=lambda c: c[0])
            del a, sp
            del spare[max(want - len(kept), 0):]       # only as many rejects as could still be needed stay alive
            if t:
                torch.cuda.empty_cache()               # spacer and dropped rejects: back to the driver
            if sp_mb > 2048:
                s_per_mb = max(s_per_mb, (time.perf_counter() - t_it) / (sp_mb - 2048))
            if len(kept) >= want:
                break
            if t + 1 >= tries:
                # a shape whose fast placements do not reach the absolute rate: after `tries` candidates, arrays within 3 % of the
                # fastest one seen are as good as it gets (if that one is itself out of the slow group)
                top = sorted(kept + spare, key=lambda c: c[0])[:want]
                if len(top) == want and top[-1][0] <= 1.03 * top[0][0] and tbs_of(top[0][0]) >= 5.3:
                    break
            if time.perf_counter() - t0 > budget:
                break
    chosen = sorted(kept + spare, key=lambda c: c[0])[:want]
    info = {"probed": len(probes), "probed_TBps": probes, "search_candidates_s": round(time.perf_counter() - t0, 3)}
    return [(a, tbs_of(us)) for us, a in chosen], info


def prewarm_placement(device, out: int, fin: int, count: int, tries: Optional[int] = None) -> dict:
    """Put ``count`` rated fast arrays of out x pad32(fin) floats into the process-level pool BEFORE a series of short fits of models
    with a wide weight of exactly that shape (a fine-tuning grid, the folds of a cross-validated trial): every model of the series
    then takes them at no cost of its own, instead of the first placement the allocator offers."""
    import time
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t0 = time.perf_counter()
    have = sum(1 for shp, _ in POOL.held(device) if shp == (out, fin))
    if tries is None:
        tries = int(os.environ.get("FX_PLACEMENT_TRIES", "12"))
    found, info = ([], {"probed": 0})
    if count > have and out * fin >= ParamStore.PLACE_MIN_ELEMS and fin % 4 == 0 and tries > 1:
        found, info = search_arrays(device, out, fin, count - have, max(tries, (count - have) * 4), seed=777)
        POOL.give(device, (out, fin), found)
    return {"shape": [out, fin], "wanted": count, "already_pooled": have, "probed": info.get("probed", 0),
            "pooled_TBps": [round(r, 2) for _, r in found if r >= float(os.environ.get("FX_PLACEMENT_GOOD_TBS", ParamStore.PLACE_GOOD_TBS))],
            "seconds": round(time.perf_counter() - t0, 3)}


class ParamStore:
    def __init__(self, spec: ArchSpec, device, big_threshold: int = 1 << 20, materialize_big_grads: bool = True,
                 big_min_dim: Optional[int] = None):
        self.spec = spec
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("flexynesis_amd.ParamStore needs a GPU device (no CPU path)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        # ``shapes``: the reference's state_dict ABI; ``eshapes``: what the engine computes with (hidden widths and, for the MLP family,
        # encoder input widths rounded up to 4: ArchSpec.engine_shapes).  p / g / m / v / b return the LOGICAL tensors (views of the
        # top-left corner), ep / eg / em / ev / eb the engine's; everything outside the logical part is zero and stays zero (arch.py).
        self.shapes = spec.state_shapes()
        self.eshapes = spec.engine_shapes()
        self.padded = any(tuple(self.shapes[k]) != tuple(self.eshapes[k]) for k in self.shapes)
        self.param_keys = [k for k in self.shapes if not is_buffer_key(k)]
        self.buffer_keys = [k for k in self.shapes if is_buffer_key(k) and not k.endswith("num_batches_tracked")]
        self.nbt_keys = [k for k in self.shapes if k.endswith("num_batches_tracked")]
        # "wide" weights take the split-bf16 MFMA kernels with the fused dW+clip+Adam epilogue and the Gram-identity norm.
        # That pays when BOTH dimensions are large: for a short-fat weight such as the GNN's fc [latent <= 128,
        # nodes * C] the Gram factors cost more than materialising dW (measured at [64, 128000]: 1.36 vs 1.18 ms/step),
        # so such weights stay in the small-parameter arena.  An explicit big_threshold (tests) disables the rule.
        if big_min_dim is None:
            big_min_dim = 256 if big_threshold == (1 << 20) else 1
        self.big_keys = [k for k in self.param_keys if len(self.eshapes[k]) == 2
                         and int(np.prod(self.eshapes[k])) >= big_threshold and min(self.eshapes[k]) >= big_min_dim]
        self.small_keys = [k for k in self.param_keys if k not in self.big_keys]
        # small arena
        self.off: Dict[str, Tuple[int, int]] = {}
        o = 0
        for k in self.small_keys:
            n = int(np.prod(self.eshapes[k])) if self.eshapes[k] else 1
            self.off[k] = (o, n)
            # (the small Linears behind an encoder's BatchNorm block get their ROWS allocated up to a multiple of 4 -- zero rows that
            # the grouped tail kernel reads as further outputs, so that its partial products have an aligned pitch for any latent size)
            alloc = pad4(self.eshapes[k][0]) * int(np.prod(self.eshapes[k][1:])) if (is_tail_key(k) and self.eshapes[k]) else n
            o += _align4(alloc)
        self.n_small = max(o, 4)
        f = dict(dtype=torch.float32, device=self.device)
        self.P = torch.zeros(self.n_small, **f)
        self.G = torch.zeros(self.n_small, **f)
        self.M = torch.zeros(self.n_small, **f)
        self.V = torch.zeros(self.n_small, **f)
        # buffers arena (running_mean / running_var)
        self.boff: Dict[str, Tuple[int, int]] = {}
        o = 0
        for k in self.buffer_keys:
            n = int(np.prod(self.eshapes[k]))
            self.boff[k] = (o, n)
            o += _align4(n)
        self.Bf = torch.zeros(max(o, 4), **f)
        self.nbt: Dict[str, int] = {k: 0 for k in self.nbt_keys}
        # wide weights
        self.big: Dict[str, Dict[str, Optional[torch.Tensor]]] = {}
        self._streams: Dict[int, "torch.cuda.Stream"] = {}
        self.note_stream()
        # Rows are padded to a multiple of 32 floats: with an arbitrary feature count the rows of W / m / v start at
        # arbitrary offsets inside a 128-byte line, every 512-byte row segment of the dW+Adam kernel then straddles a
        # fifth line (reads) and partial lines (writes) -- the decoders' [20000, 5000] weights took 510-630 us per launch
        # instead of ~430.  "W"/"M"/"V"/"G" are the [out, in] VIEWS (state_dict layout), "_W".. the padded buffers.
        self.placement: Dict[str, dict] = {}          # per wide weight: probe times of the candidate placements (us), the kept one
        for k in self.big_keys:
            self.big[k] = {}
            self._place_big(k)
            self.big[k]["G"] = None
            if materialize_big_grads:
                self._big_alloc(k, "G")
        self.ctrl = torch.zeros(ops.CTRL_FLOATS, **f)
        self.ctrl[ops.CTRL_CLIP_COEF] = 1.0
        self.reset_parameters()

    # -- views ------------------------------------------------------------------------------------
    def _view(self, arena, key):
        o, n = self.off[key]
        return arena[o:o + n].view(self.eshapes[key])

    # engine tensors (what the kernels are given)
    def ep(self, key) -> torch.Tensor:
        return self.big[key]["W"] if key in self.big else self._view(self.P, key)

    def eg(self, key) -> Optional[torch.Tensor]:
        return self.big[key]["G"] if key in self.big else self._view(self.G, key)

    def em(self, key):
        return self.big[key]["M"] if key in self.big else self._view(self.M, key)

    def ev(self, key):
        return self.big[key]["V"] if key in self.big else self._view(self.V, key)

    def eb(self, key) -> torch.Tensor:
        o, n = self.boff[key]
        return self.Bf[o:o + n].view(self.eshapes[key])

    # logical views: the reference's shapes (state_dict ABI, nn.Parameter data of the level-1 classes, parameter initialisation)
    def _logical(self, t: torch.Tensor, key) -> torch.Tensor:
        ls = self.shapes[key]
        if tuple(t.shape) == tuple(ls):
            return t
        return t[tuple(slice(0, n) for n in ls)]

    def p(self, key) -> torch.Tensor:
        return self._logical(self.ep(key), key)

    def g(self, key) -> Optional[torch.Tensor]:
        g = self.eg(key)
        return None if g is None else self._logical(g, key)

    def m(self, key):
        return self._logical(self.em(key), key)

    def v(self, key):
        return self._logical(self.ev(key), key)

    def b(self, key) -> torch.Tensor:
        return self._logical(self.eb(key), key)

    def rows4(self, key) -> torch.Tensor:
        """A tail Linear's weight [L, H] (or bias [L]) with its rows up to the next multiple of 4 (zeros): see __init__."""
        o, n = self.off[key]
        shp = self.eshapes[key]
        rows = pad4(shp[0])
        return self.P[o:o + rows * int(np.prod(shp[1:]))].view((rows,) + tuple(shp[1:]))

    def _big_alloc(self, key, name):
        out, fin = self.eshapes[key]
        buf = torch.zeros(out, (fin + 31) // 32 * 32, dtype=torch.float32, device=self.device)
        self.big[key]["_" + name] = buf
        self.big[key][name] = buf[:, :fin]

    # Placement of a wide weight's three arrays.  The dW + Adam kernels stream W, m and v of a 64 x 128 tile together, ~500 tiles at a
    # time; how fast that goes depends on which physical pages the three allocations happen to get -- 400 to 494 us for the same
    # kernel on the same [5000, 20000] arrays of the same MI355X, by the allocation history of the process alone (DESIGN.md section
    # 3.10, profiles/r04_placement.txt).  The quality of a placement is, to a good approximation, a property of each ARRAY on its own
    # (scripts/placement_single.py: arrays probed alone fall into a fast group, 5.7-5.9 TB/s of their 8 B per element, and a slow one,
    # 4.8-5.0; the three fastest of 14 together probe 397-400 us, the three slowest 486-490) -- of the array AT ITS SHAPE: the same
    # memory viewed with another shape rates differently (profiles/r05_placement_pool.txt).  So
    #   * arrays are rated one by one with a timed pass of that traffic pattern (fx_placement_probe: contents untouched);
    #   * a ParamStore takes its arrays from a PROCESS-LEVEL POOL of rated arrays of its shape first (PlacementPool: what earlier
    #     models of this process -- folds, fine-tuning fits, repeated fits -- gave back when they were dropped, or prewarm_placement());
    #   * what the pool cannot serve is searched for: candidate arrays behind spacer allocations of varying size (they decide which
    #     physical blocks the driver hands out next), fast ones kept, the rest back to the DRIVER (torch's cache would hand the same
    #     blocks out again) -- bounded by FX_PLACEMENT_TRIES candidates (default 12 x 8; 1 = no search: first placement, as rounds
    #     1-3), FX_PLACEMENT_BUDGET_S seconds (default 0.6 per weight) and FX_PLACEMENT_SPACER_GB (default 32, and never more than a
    #     quarter of what is free at that moment).
    PLACE_MIN_ELEMS = 1 << 24           # 64 MB per array: smaller weights are a few tiles per workgroup, placement is in the noise
    PLACE_GOOD_TBS = 5.75               # an array read + written (8 B per element) per probe pass: arrays at this rate are kept

    def _place_big(self, key):
        import time
        out, fin = self.eshapes[key]
        ld = (fin + 31) // 32 * 32
        need = out * ld
        tries = getattr(_PLACEMENT, "tries", None)
        if tries is None or "FX_PLACEMENT_TRIES" in os.environ:
            tries = int(os.environ.get("FX_PLACEMENT_TRIES", "12"))
        eligible = out * fin >= self.PLACE_MIN_ELEMS and fin % 4 == 0
        got: List[Tuple[torch.Tensor, Optional[float]]] = []          # (flat array of >= need elements, its rate in TB/s or None)
        t0 = time.perf_counter()
        info = {}
        with torch.cuda.device(self.device):
            # (FX_PLACEMENT_TRIES=1 in the environment is the A/B "take what the allocator gives"; a short fit's thread-local tries = 1 only
            # switches the per-weight SEARCH off -- the arena costs a trial nothing)
            arena = PartitionArena.get(self.device) if (eligible and os.environ.get("FX_PLACEMENT_TRIES") != "1") else None
            if arena is not None:
                taken = arena.take3(need)
                if taken is not None:
                    flats, lease_ids = taken
                    views = []
                    for name, flat in zip(("W", "M", "V"), flats):
                        buf = flat.view(out, ld)
                        self.big[key]["_" + name] = buf
                        self.big[key][name] = buf[:, :fin]
                        views.append(buf[:, :fin])
                    self.big[key]["_leases"] = list(lease_ids)
                    self.placement[key] = dict(arena=True, kept_us=round(ops.placement_probe_us(*views), 1),
                                               search_s=round(time.perf_counter() - t0, 3), **{k: v for k, v in arena.info.items() if k in ("build_s", "spacer_GB")})
                    return
                info["arena"] = "pool exhausted"
            if eligible:
                got = POOL.take(self.device, (out, fin), 3)
                info["from_pool"] = len(got)
            if eligible and tries > 1 and len(got) < 3:
                found, sinfo = search_arrays(self.device, out, fin, 3 - len(got), tries, seed=20240 + len(self.big))
                got += found
                info.update(sinfo)
            while len(got) < 3:
                got.append((torch.zeros(need, dtype=torch.float32, device=self.device), None))
            views = []
            lease_ids = []
            for name, (flat, tbs) in zip(("W", "M", "V"), got):
                if tbs is not None:
                    # a rated array: used through a lease, so that it returns to the process-level pool exactly when the last view
                    # of it (this store, an nn.Parameter's .data, a state_dict tensor) has gone
                    dev = self.device
                    lt, lid = ops.LEASES.wrap(flat.data_ptr(), need, dev,
                                              (lambda evs, unsynced, a=flat, r=tbs: POOL.give(dev, (out, fin), [(a, r)], evs, unsynced)))
                    lease_ids.append(lid)
                    buf = lt.view(out, ld)
                else:
                    buf = flat[:need].view(out, ld)
                if info.get("from_pool"):
                    buf.zero_()
                self.big[key]["_" + name] = buf
                self.big[key][name] = buf[:, :fin]
                views.append(buf[:, :fin])
            self.big[key]["_leases"] = lease_ids
            if eligible and (info.get("from_pool") or info.get("probed")):
                info["kept_TBps"] = [None if r is None else round(r, 2) for _, r in got]
                info["kept_us"] = round(ops.placement_probe_us(*views), 1)
                info["search_s"] = round(time.perf_counter() - t0, 3)
                self.placement[key] = info

    def note_stream(self, stream=None):
        """Remember a stream that launches work on this store's arrays (plans call it when they are built): ``release_big`` records
        one event on each, and whoever is given the memory next waits for them."""
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._streams[st.cuda_stream] = st

    def release_big(self):
        """Let go of the wide weights' arrays; the store is unusable afterwards.  Arena ranges and rated arrays are LEASES
        (ops.LEASES): the memory itself returns to its pool when the last view of it has gone -- parameters of a model that has been
        closed or moved keep their values for as long as they exist -- and carries the events recorded here, one per stream that
        ran this store's plans, for the next taker to wait on.  Called by fit() / FxModel.close() at a known point, or when the
        store is dropped."""
        ids = [lid for d in self.big.values() for lid in (d.pop("_leases", None) or [])]
        if ids:
            events, unsynced = [], False
            if ops.capturing():            # (a store collected while a hipGraph capture is in progress: nothing may be recorded now)
                unsynced = True
            else:
                try:
                    for st in self._streams.values():
                        ev = torch.cuda.Event()
                        ev.record(st)
                        events.append(ev)
                except Exception:
                    unsynced = True
            for lid in ids:
                ops.LEASES.add_events(lid, events, unsynced)
        self.big = {}

    def __del__(self):
        try:
            if self.big:
                self.release_big()
        except Exception:
            pass

    def ensure_big_grads(self):
        for k, d in self.big.items():
            if d["G"] is None:
                self._big_alloc(k, "G")

    # -- init / (de)serialisation ------------------------------------------------------------------
    @torch.no_grad()
    def reset_parameters(self, seed: Optional[int] = None):
        """Same init *distributions* as the reference modules (modules.py:26-41,76-89,125-130):
        nn.Linear default (kaiming_uniform a=sqrt(5) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and
        bias) for MLP / fusion layers, xavier_uniform weights for Encoder/Decoder Linear layers,
        BatchNorm weight 1 / bias 0 / running stats 0,1, log_vars 0.  Drawn on the device."""
        gen = torch.Generator(device=self.device)
        gen.manual_seed(int(seed) if seed is not None else int(torch.initial_seed() % (2 ** 31)))
        if self.padded:                   # whatever lies outside the logical tensors is zero (and stays zero: arch.py)
            self.P.zero_(); self.Bf.zero_()
            for d in self.big.values():
                d["_W"].zero_()
        for k in self.param_keys:
            t, shp = self.p(k), self.shapes[k]
            if k.startswith("log_vars."):
                t.zero_()
            elif ".batchnorm." in k or ".hidden_layers.2." in k or ".bns." in k:
                t.fill_(1.0) if k.endswith("weight") else t.zero_()
            elif ".convs." in k and k.endswith(".bias"):
                t.zero_()                                    # torch_geometric initialises conv biases to zero
            elif k.endswith(".weight"):
                fan_out, fan_in = shp
                xavier = (k.startswith(("encoders.", "decoders.")) and
                          (".hidden_layers." in k or ".FC_" in k))
                bound = math.sqrt(6.0 / (fan_in + fan_out)) if xavier else 1.0 / math.sqrt(fan_in)
                t.uniform_(-bound, bound, generator=gen)
            else:  # Linear bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) with the fan_in of its weight
                fan_in = self.shapes[k[:-4] + "weight"][1]
                bound = 1.0 / math.sqrt(fan_in)
                t.uniform_(-bound, bound, generator=gen)
        for k in self.buffer_keys:
            self.b(k).fill_(1.0) if k.endswith("running_var") else self.b(k).zero_()
        for k in self.nbt:
            self.nbt[k] = 0
        self.reset_optimizer()

    @torch.no_grad()
    def reset_optimizer(self):
        self.M.zero_(); self.V.zero_(); self.G.zero_()
        for d in self.big.values():
            d["M"].zero_(); d["V"].zero_()
            if d["G"] is not None:
                d["G"].zero_()
        self.ctrl.zero_()
        self.ctrl[ops.CTRL_CLIP_COEF] = 1.0

    @torch.no_grad()
    def load_state(self, state: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in self.shapes if k not in state]
        extra = [k for k in state if k not in self.shapes]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing}, unexpected {extra}")
        for k, val in state.items():
            if k not in self.shapes:
                continue
            val = torch.as_tensor(val)
            if tuple(val.shape) != tuple(self.shapes[k]):
                raise ValueError(f"{k}: shape {tuple(val.shape)} != {self.shapes[k]}")
            if k in self.nbt:
                self.nbt[k] = int(val)
            elif is_buffer_key(k):
                self.b(k).copy_(val.to(torch.float32))
            else:
                self.p(k).copy_(val.to(torch.float32))

    @torch.no_grad()
    def state_dict(self, device="cpu") -> Dict[str, torch.Tensor]:
        out = {}
        for k in self.shapes:
            if k in self.nbt:
                out[k] = torch.tensor(self.nbt[k], dtype=torch.int64)
            elif is_buffer_key(k):
                out[k] = self.b(k).detach().to(device).clone(memory_format=torch.contiguous_format)
            else:
                out[k] = self.p(k).detach().to(device).clone(memory_format=torch.contiguous_format)
        return out

    @torch.no_grad()
    def load_optimizer(self, t: int, m: Dict[str, torch.Tensor], v: Dict[str, torch.Tensor]):
        self.ctrl[ops.CTRL_STEP] = float(int(t) % (1 << 24))          # t = hi * 2^24 + lo, both exact in fp32 (fx_common.h)
        self.ctrl[ops.CTRL_STEP_HI] = float(int(t) >> 24)
        for k, val in m.items():
            self.m(k).copy_(torch.as_tensor(val).to(torch.float32))
        for k, val in v.items():
            self.v(k).copy_(torch.as_tensor(val).to(torch.float32))

    def n_params(self) -> int:
        return sum(int(np.prod(self.shapes[k])) if self.shapes[k] else 1 for k in self.param_keys)


class StepPlan:
    """Forward/backward/optimiser tapes for one (model, batch size)."""

    @ops.device_guard
    def __init__(self, store: ParamStore, B: int, train: bool = True, fused: bool = True, clip: bool = True,
                 supplied_draws: bool = False, seed: int = 0, cohort=None, n_batches: int = 0,
                 epoch_acc: bool = False, precision: Optional[str] = None, branches: bool = True, share: "StepPlan" = None,
                 fuse_heads: bool = True, frozen: Tuple[str, ...] = (), fuse_next_fwd: bool = False,
                 attribution: bool = False, clip_norm: float = CLIP_MAX_NORM, forward_alone: bool = False):
        self.store, self.spec, self.B, self.train = store, store.spec, int(B), train
        store.note_stream()                     # (the stream this plan's launches will be ordered on: ParamStore.release_big)
        # the forward tape may be run on its own and must leave every loss (the total included) behind: the level-1 path, whose
        # caller reads the loss between ``forward()`` and ``backward()``.  Otherwise loss bookkeeping may ride in the backward tape.
        self.forward_alone = bool(forward_alone)
        self.fused = bool(fused) and train
        self.clip = clip
        self.clip_norm = float(clip_norm)       # gradient_clip_val of the caller's Trainer (the reference uses 1.0)
        # FineTuner (reference main.py:530-539): state_dict key prefixes with requires_grad=False, e.g. ("encoders.",)
        # or ("MLPs.",).  Frozen parameters get no gradient work and are skipped by Adam; BatchNorm buffers of frozen
        # blocks still update.  The reference never clips while fine-tuning (its Trainer has no gradient_clip_val).
        self.frozen = tuple(frozen)
        if self.frozen and clip and train:
            raise ValueError("frozen parameter groups are the FineTuner's configuration, which trains without gradient "
                             "clipping (reference main.py:591-600): pass clip=False")
        self.supplied = supplied_draws
        self.seed = int(seed)
        self.dev = store.device
        self._ws = [Workspace(self.dev)]
        self._branch = 0
        self.branches = bool(branches) and os.environ.get("FX_BRANCHES", "1") != "0"   # per-modality chains as parallel hipGraph branches
        # FX_BN_SLABS=1 folds the reduction of the wide layer's partial sums (+ bias) into the BatchNorm kernel (one launch
        # less per wide layer).  Measured twice, no gain: 1.289 vs 1.276 ms/step at cfg2 with the fused next-step forward
        # (the BatchNorm kernel then re-reads 6 slabs instead of one array), so it stays an A/B switch.
        self.bn_slabs = os.environ.get("FX_BN_SLABS", "0") == "1"
        # whole encoder-tail backward in one launch (fx_block_bwd); FX_BLOCK_BWD=0 is an A/B switch for benchmarks
        self.block_bwd = os.environ.get("FX_BLOCK_BWD", "1") != "0"
        self.block_bwd_passes = os.environ.get("FX_BLOCK_BWD_PASSES", "1") != "0"
        self.gram_kb_wide = os.environ.get("FX_GRAM_KB_WIDE", "1") != "0"          # split-bf16 Gram kernel for wide activations / output gradients of <= 128 rows (A/B)
        self.gram_bf16x3 = os.environ.get("FX_GRAM_BF16X3", "1") != "0"              # X X^T of stacked rows on the split-bf16 kernel (A/B)    # ... also per pass of stacked rows (A/B)
        # all supervisor heads in one launch each way (fx_heads_fwd/bwd); FX_FUSE_HEADS=0 is an A/B switch for benchmarks
        self.fuse_heads = bool(fuse_heads) and os.environ.get("FX_FUSE_HEADS", "1") != "0"
        self.small_linear = os.environ.get("FX_SMALL_LINEAR", "1") != "0"
        # encoder tails of all modalities in one launch + row-parallel reduce / fusion layer (fx_enc_tail.hip); FX_FUSE_TAIL=0: A/B
        self.fuse_tail = os.environ.get("FX_FUSE_TAIL", "1") != "0"
        self.group_bwd = os.environ.get("FX_GROUP_BWD", "1") != "0"       # encoder-tail backward of all modalities in one launch (A/B)
        self._bb_group = None
        # heads forward + losses + total + heads backward in one launch (fx_heads_step); FX_HEADS_STEP=0: A/B
        self.fuse_heads_step = self.fuse_heads and os.environ.get("FX_HEADS_STEP", "1") != "0"
        self._gram_x: Dict[int, tuple] = {}
        self._jobs: Dict[str, tuple] = {}
        self._slot_o = 0
        if precision is None:                   # the process-wide default: FX_PRECISION (f32 | bf16x3 | bf16), else the parity mode
            precision = default_precision()
        if precision not in ("f32", "bf16x3", "bf16"):
            raise ValueError(f"precision must be 'f32', 'bf16x3' or 'bf16', got {precision!r}")
        # "bf16": the THROUGHPUT mode -- the wide kernels' contractions in plain bf16 (hi . hi, fp32 accumulate: torch's "medium"
        # matmul precision, which the reference itself trains under, main.py:24) instead of three split products; same kernels, same
        # schedule, same operand buffers (the tapes pass NULL for the `lo` halves); master weights, Adam moments, the Gram norm and
        # everything narrow stay as in "bf16x3".  Looser, documented tolerance; every parity gate runs bf16x3.
        self.plain_bf16 = precision == "bf16"
        self.precision = "bf16x3" if self.plain_bf16 else precision
        self._split_cache: Dict[tuple, tuple] = {}
        self.cohort = cohort
        self.n_batches = int(n_batches)
        spec = self.spec
        if not self.fused and train:
            store.ensure_big_grads()
        self.passes = 3 if spec.model == "MultiTripletNetwork" else 1
        self.R = self.B * self.passes                     # rows through the encoders
        f = dict(dtype=torch.float32, device=self.dev)
        self._f = f
        self.draws: Dict[str, torch.Tensor] = {}
        self.buf: Dict[str, torch.Tensor] = {}
        n_terms = len(spec.loss_names())
        self.loss_vec = torch.zeros(n_terms + 1, **f)     # raw named losses..., total
        if share is not None:      # second half of a PipelinedStep: same index table and epoch accumulators
            self.epoch_acc, self.idx = share.epoch_acc, share.idx
        else:
            self.epoch_acc = torch.zeros(n_terms + 2, **f) if epoch_acc else None
            self.idx = torch.zeros(max(self.R, 1) * max(self.n_batches, 1), dtype=torch.int64, device=self.dev)
        self.X = [torch.zeros(self.R, spec.engine_features(i), **f) for i in range(len(spec.layers))]   # (engine width: pad columns stay zero)
        # reconstruction targets of the VAE family: the batch at the layer's own width, contiguous -- X itself unless the encoder input is padded
        self.Xt = list(self.X)
        if spec.is_vae:
            for i in spec.dec_idx:
                if self.X[i].shape[1] != spec.layers[i][1]:
                    self.Xt[i] = torch.zeros(self.R, spec.layers[i][1], **f)
        self.y: Dict[str, torch.Tensor] = {}
        for (v, _, _) in spec.variables:
            self.y[v] = torch.zeros(self.B, **f)
        if spec.surv_time_var:
            self.y[spec.surv_time_var] = torch.zeros(self.B, **f)
        self._rng_ctr = 0
        self.big_jobs: List[Tuple[str, torch.Tensor, torch.Tensor]] = []   # (weight key, dY, X) for fused dW+Adam
        self.t_gather, self.t_fwd, self.t_bwd, self.t_opt = self._tape(), self._tape(), self._tape(), self._tape()
        # Next-step forward fused into the optimiser kernel (fx_linear_dw_adam_fwd_bf16x3): this plan's wide forward is
        # then only the ordered reduce of partial sums that the PREVIOUS step's dW+Adam launch left behind (it had this
        # plan's batch, assembled one step ahead, and the freshly updated weight tile in registers).  Needs a partner plan
        # (PipelinedStep links the two); t_boot computes the partial sums stand-alone for the first step / after any
        # weight change made outside the pipeline.
        self.fuse_next = bool(fuse_next_fwd) and self.fused and self.precision == "bf16x3" and cohort is not None
        self._next_fwd: Dict[str, tuple] = {}
        self.path: Dict[str, bool] = {}       # which of the fused schedules this plan records (tests / bench: "did the fast path engage")
        self.t_boot = self._tape()
        # attribution (eval plans): input-gradient tapes d head_output / d X for IntegratedGradients / GradientShap
        self.attribution = bool(attribution) and not train
        self.t_attr_head: Dict[str, TapeRecorder] = {}
        self.t_attr_common = self._tape()
        self.attr_dout: Dict[str, torch.Tensor] = {}
        self.dX: List[torch.Tensor] = []
        self._build()
        self.graph = None

    # ---------------------------------------------------------------------------------------------
    def _new(self, name, *shape):
        t = torch.zeros(*shape, **self._f)
        self.buf[name] = t
        return t

    def _rng(self):
        self._rng_ctr += 1
        return self.seed, (self._rng_ctr << 32)

    def _draw(self, name, *shape, logical=None):
        """Supplied-randomness slot (parity mode): a static buffer the caller fills.  ``logical``: the reference's shape when the
        engine's buffer is wider (padded hidden width); the caller sees that view, the rest stays 1."""
        t = torch.ones(*shape, **self._f)
        self.draws[name] = t if logical is None or tuple(logical) == tuple(shape) else t[tuple(slice(0, n) for n in logical)]
        return t

    def _logvar(self, name):
        return self.store.ep("log_vars." + name).view(-1) if (self.train and self.spec.weighted) else None

    # ---- blocks ----------------------------------------------------------------------------------
    def _mlp_fwd(self, rec, prefix, x, out, rows, passes, tag_names):
        """Linear -> BN -> ReLU -> Dropout -> Linear   (reference modules.py:145-149)."""
        st, H = self.store, self.store.eshapes[prefix + ".layer_1.weight"][0]
        y1 = self._new(prefix + "/y1", rows, H)
        a1 = self._new(prefix + "/a1", rows, H)
        sm = self._new(prefix + "/save_mean", passes, H)
        si = self._new(prefix + "/save_invstd", passes, H)
        Bp = rows // passes
        slabs = self._lin_fwd(rec, y1, x, prefix + ".layer_1.weight", prefix + ".layer_1.bias",
                              want_slabs=self.bn_slabs and Bp <= 128)
        for p in range(passes):
            sl = slice(p * Bp, (p + 1) * Bp)
            mask = self._draw(tag_names[p], Bp, H, logical=(Bp, st.shapes[prefix + ".layer_1.weight"][0])) if (self.supplied and self.train) else None
            seed, off = self._rng()
            bn = (st.ep(prefix + ".batchnorm.weight"), st.ep(prefix + ".batchnorm.bias"),
                  st.eb(prefix + ".batchnorm.running_mean"), st.eb(prefix + ".batchnorm.running_var"), sm[p], si[p],
                  ACT_NONE, ACT_RELU, self.train, DROPOUT_P if self.train else 0.0)
            if slabs is not None:      # split-K reduction + bias folded into the BatchNorm pass
                sbuf, ns = slabs
                ops.bn_act_fwd_slabs(rec, a1[sl], y1[sl], sbuf.view(-1)[p * Bp * H:], ns, rows * H,
                                     st.ep(prefix + ".layer_1.bias"), *bn, mask=mask, seed=seed, offset=off, ctrl=st.ctrl)
            else:
                ops.bn_act_fwd(rec, a1[sl], y1[sl], *bn, mask=mask, seed=seed, offset=off, ctrl=st.ctrl)
        bias_key = prefix + ".layer_out.bias"
        ops.linear_fwd(rec, out, a1, st.ep(prefix + ".layer_out.weight"),
                       st.ep(bias_key) if bias_key in st.eshapes else None, self.ws)

    def _mlp_bwd(self, rec, prefix, x, dout, rows, passes, dx=None, dx_accumulate=False):
        st, H = self.store, self.store.eshapes[prefix + ".layer_1.weight"][0]
        y1, a1 = self.buf[prefix + "/y1"], self.buf[prefix + "/a1"]
        sm, si = self.buf[prefix + "/save_mean"], self.buf[prefix + "/save_invstd"]
        if dx is None and self._block_ok(rows, passes) and not self._is_frozen(prefix + ".layer_out.weight"):
            bias_key = prefix + ".layer_out.bias"
            self._tail_bwd(rec, [(dout, prefix + ".layer_out.weight", bias_key if bias_key in st.eshapes else None)], x, y1, a1,
                           (prefix + ".batchnorm", prefix), prefix + ".layer_1.bias", prefix + ".layer_1.weight",
                           ACT_NONE, ACT_RELU, DROPOUT_P)
            return
        da1 = self._new(prefix + "/da1", rows, H)           # also holds dy1 (BN backward runs in place)
        Bp = rows // passes
        if (dx is None and self.block_bwd and self.train and passes > 1 and Bp <= 128 and self.block_bwd_passes
                and not self._is_frozen(prefix + ".layer_out.weight")):
            # stacked BatchNorm passes (triplet: anchor / positive / negative rows): one fx_block_bwd launch per pass, the
            # parameter gradients accumulated from the second on -- instead of seven dependent per-layer launches
            bias_key = prefix + ".layer_out.bias"
            wk, w1 = prefix + ".layer_out.weight", prefix + ".layer_1.weight"
            # the wide layer's fused optimiser wants dY as a transposed split [H, rows]: each pass writes its own columns
            want_t = (self.fused and w1 in st.big and not self._is_frozen(w1) and self.precision == "bf16x3" and Bp % 32 == 0)
            dyt = ops.new_split(H, rows, self.dev) if want_t else None
            # ... and the Gram norm's dY dY^T product (stacked rows: _weight_grad) wants dY's K-blocked split: a workgroup's 32 columns
            # are one K-block, so each pass writes its rows of it as well (FX_BLOCK_BWD_KB=0: a launch of fx_split_bf16 over dY behind
            # the last pass -- 47 us on the chain between the encoder tails' backward and the clip coefficient at cfg4)
            dkb = None
            if (want_t and self.clip and self._stacked_gram_kb_ok(rows, H) and Bp % 8 == 0
                    and os.environ.get("FX_BLOCK_BWD_KB", "1") != "0"):
                dkb = ops.new_split_kb(rows, H, self.dev)
            for p in range(passes):
                sl = slice(p * Bp, (p + 1) * Bp)
                ops.block_bwd(rec, [(dout[sl], st.ep(wk), st.eg(wk), st.eg(bias_key) if bias_key in st.eshapes else None)], y1[sl], a1[sl],
                              st.ep(prefix + ".batchnorm.weight"), sm[p], si[p], st.eg(prefix + ".batchnorm.weight"),
                              st.eg(prefix + ".batchnorm.bias"), st.eg(prefix + ".layer_1.bias"), ACT_NONE, ACT_RELU, DROPOUT_P,
                              dy=da1[sl], dyT=(dyt[0][:, p * Bp:], dyt[1][:, p * Bp:]) if want_t else None, accumulate=p > 0,
                              dy_kb=(dkb[0], dkb[1], p * Bp) if dkb is not None else None)
            self._weight_grad(rec, w1, da1, x, dyt=dyt, dy_kb=dkb)
            return
        self._weight_grad(rec, prefix + ".layer_out.weight", dout, a1)
        if prefix + ".layer_out.bias" in st.eshapes:
            ops.colsum(rec, st.eg(prefix + ".layer_out.bias"), dout)
        ops.linear_bwd_x(rec, da1, dout, st.ep(prefix + ".layer_out.weight"), self.ws)
        for p in range(passes):
            sl = slice(p * Bp, (p + 1) * Bp)
            ops.bn_act_bwd(rec, da1[sl], st.eg(prefix + ".batchnorm.weight"), st.eg(prefix + ".batchnorm.bias"),
                           st.eg(prefix + ".layer_1.bias"), da1[sl], y1[sl], a1[sl], st.ep(prefix + ".batchnorm.weight"),
                           sm[p], si[p], ACT_NONE, ACT_RELU, DROPOUT_P, accumulate=p > 0)
        self._weight_grad(rec, prefix + ".layer_1.weight", da1, x)
        if dx is not None:
            ops.linear_bwd_x(rec, dx, da1, st.ep(prefix + ".layer_1.weight"), self.ws, accumulate=dx_accumulate)

    def _tails_groupable(self, n, L) -> bool:
        """fx_enc_tail_fwd + fx_fusion_fwd cover one BatchNorm pass of <= 128 rows, <= 4 modalities, latent <= 128 (the reference's
        whole search space, config.py:7-15); block widths are multiples of 4 by construction (ArchSpec.engine_shapes), any latent
        size goes (the tail Linears' rows are allocated up to a multiple of 4, ParamStore.rows4)."""
        st = self.store
        if not self.fuse_tail or self.passes != 1 or self.R > 128 or n > 4 or L > 128 or n * L > 512:
            return False
        wide = "layer_1" if "encoders.0.layer_1.weight" in st.eshapes else "hidden_layers.0"      # MLP / VAE encoder
        if not all(st.eshapes[f"encoders.{i}.{wide}.weight"][0] % 4 == 0 for i in range(n)):
            return False
        # the Linear layers behind the block (layer_out; FC_mean / FC_var) are read as plain [L, H] arena tensors by the grouped
        # kernel: a non-default big_threshold / big_min_dim that turned one of them into a padded "wide" weight, or an arena
        # offset that is not 16-byte aligned, falls back to the per-layer path instead of failing the plan build
        for i in range(n):
            for nm in (("layer_out",) if wide == "layer_1" else ("FC_mean", "FC_var")):
                k = f"encoders.{i}.{nm}.weight"
                if k in st.big:
                    return False
                if k in st.eshapes:
                    t = st.ep(k)
                    if not t.is_contiguous() or t.data_ptr() % 16:
                        return False
        return True

    def _vae_tails_fwd(self, rec, enc, L, mcat, vcat, mean, logv):
        """All VAE encoders' tails in one launch -- slab sum + bias, LeakyReLU, BatchNorm and the column blocks' shares of
        FC_mean and FC_var (modules.py:25-41) -- then mcat / vcat (ordered sums of the shares + biases) with the top-level
        FC_mean / FC_log_var (supervised_vae.py:104-107,172-176) row-parallel: 3 launches on one stream instead of 6 per
        modality on parallel graph branches + join + 2.  Returns the encoders' block outputs."""
        st, R = self.store, self.R
        descs, parts_m, parts_v, hs = [], [], [], []
        for i, xi in enumerate(enc):
            p = f"encoders.{i}"
            H = st.eshapes[p + ".hidden_layers.0.weight"][0]
            y, h = self._new(p + "/y", R, H), self._new(p + "/h", R, H)
            sm, si = self._new(p + "/save_mean", 1, H), self._new(p + "/save_invstd", 1, H)
            slabs = self._lin_fwd(rec, y, self.X[xi], p + ".hidden_layers.0.weight", p + ".hidden_layers.0.bias", want_slabs=True)
            nb = ops.enc_tail_blocks(H)
            pm, pv = self._new(p + "/mean_parts", nb, R, pad4(L)), self._new(p + "/var_parts", nb, R, pad4(L))
            descs.append(ops.enc_tail_desc(
                slabs=slabs[0] if slabs is not None else None, n_slabs=slabs[1] if slabs is not None else 0, slab_stride=R * H,
                lin_bias=st.ep(p + ".hidden_layers.0.bias") if slabs is not None else None, x=y, out=h,
                gamma=st.ep(p + ".hidden_layers.2.weight"), beta=st.ep(p + ".hidden_layers.2.bias"),
                running_mean=st.eb(p + ".hidden_layers.2.running_mean"), running_var=st.eb(p + ".hidden_layers.2.running_var"),
                save_mean=sm[0], save_invstd=si[0], mask=None,
                ups=[(st.rows4(p + ".FC_mean.weight"), pm), (st.rows4(p + ".FC_var.weight"), pv)], seed=0, offset=0))
            parts_m.append((pm, nb))
            parts_v.append((pv, nb))
            hs.append(h)
        ops.enc_tail_fwd(rec, descs, R, ACT_LEAKY, ACT_NONE, self.train, 0.0, ctrl=st.ctrl)
        n = len(enc)
        bm, bv = [st.ep(f"encoders.{i}.FC_mean.bias") for i in range(n)], [st.ep(f"encoders.{i}.FC_var.bias") for i in range(n)]
        if os.environ.get("FX_VAE_FUSION_PAIR", "1") != "0":        # mean and log_var in one launch (A/B: two launches back to back)
            ops.fusion_fwd_pair(rec, (mean, logv), (mcat, vcat), (parts_m, parts_v), (bm, bv),
                                (st.ep("FC_mean.weight"), st.ep("FC_log_var.weight")), (st.ep("FC_mean.bias"), st.ep("FC_log_var.bias")), width=L)
        else:
            ops.fusion_fwd(rec, mean, mcat, parts_m, bm, st.ep("FC_mean.weight"), st.ep("FC_mean.bias"), width=L)
            ops.fusion_fwd(rec, logv, vcat, parts_v, bv, st.ep("FC_log_var.weight"), st.ep("FC_log_var.bias"), width=L)
        return hs

    def _mlp_tails_fwd(self, rec, n, L, ecat):
        """All MLP encoders' tails in one launch -- slab sum + bias, BatchNorm, ReLU, Dropout and the column blocks' shares of
        layer_out (modules.py:145-149) -- then ecat (ordered sum of the shares + layer_out bias) and the fusion Linear
        (direct_pred.py:118-124) row-parallel in a second one: 2 launches on one stream instead of 4 per modality on parallel
        graph branches + join + fusion."""
        st, R = self.store, self.R
        descs, parts, biases = [], [], []
        for i in range(n):
            prefix = f"encoders.{i}"
            H = st.eshapes[prefix + ".layer_1.weight"][0]
            y1 = self._new(prefix + "/y1", R, H)
            a1 = self._new(prefix + "/a1", R, H)
            sm = self._new(prefix + "/save_mean", 1, H)
            si = self._new(prefix + "/save_invstd", 1, H)
            slabs = self._lin_fwd(rec, y1, self.X[i], prefix + ".layer_1.weight", prefix + ".layer_1.bias", want_slabs=True)
            mask = self._draw(prefix, R, H, logical=(R, st.shapes[prefix + ".layer_1.weight"][0])) if (self.supplied and self.train) else None
            seed, off = self._rng()
            nb = ops.enc_tail_blocks(H)
            part = self._new(prefix + "/layer_out_parts", nb, R, pad4(L))
            descs.append(ops.enc_tail_desc(
                slabs=slabs[0] if slabs is not None else None, n_slabs=slabs[1] if slabs is not None else 0, slab_stride=R * H,
                lin_bias=st.ep(prefix + ".layer_1.bias") if slabs is not None else None, x=y1, out=a1,
                gamma=st.ep(prefix + ".batchnorm.weight"), beta=st.ep(prefix + ".batchnorm.bias"),
                running_mean=st.eb(prefix + ".batchnorm.running_mean"), running_var=st.eb(prefix + ".batchnorm.running_var"),
                save_mean=sm[0], save_invstd=si[0], mask=mask, ups=[(st.rows4(prefix + ".layer_out.weight"), part)], seed=seed, offset=off))
            parts.append((part, nb))
            bias_key = prefix + ".layer_out.bias"
            biases.append(st.ep(bias_key) if bias_key in st.eshapes else None)
        ops.enc_tail_fwd(rec, descs, R, ACT_NONE, ACT_RELU, self.train, DROPOUT_P if self.train else 0.0, ctrl=st.ctrl)
        # From here to the encoder-tail backward the chain is a few workgroups wide (16 for the fusion layer, one per head):
        # the next batch's assembly (PipelinedStep) forks HERE, under that part, instead of beside the wide backward kernels.
        rec.mark("fork_assembly")
        if n > 1:
            emb = self._new("emb", R, L)
            ops.fusion_fwd(rec, emb, ecat, parts, biases, st.ep("fusion_block.weight"), st.ep("fusion_block.bias"), width=L)
            rec.mark("fork_issue_1")
            return emb
        ops.fusion_fwd(rec, None, ecat, parts, biases, width=L)
        rec.mark("fork_issue_1")
        return ecat

    def _hidden_fwd(self, rec, prefix, x, rows):
        """Linear -> LeakyReLU(0.2) -> BN   (reference modules.py:25-34 / :75-84)."""
        st, H = self.store, self.store.eshapes[prefix + ".hidden_layers.0.weight"][0]
        y = self._new(prefix + "/y", rows, H)
        h = self._new(prefix + "/h", rows, H)
        sm = self._new(prefix + "/save_mean", 1, H)
        si = self._new(prefix + "/save_invstd", 1, H)
        slabs = self._lin_fwd(rec, y, x, prefix + ".hidden_layers.0.weight", prefix + ".hidden_layers.0.bias",
                              want_slabs=self.bn_slabs and rows <= 128)
        bn = (st.ep(prefix + ".hidden_layers.2.weight"), st.ep(prefix + ".hidden_layers.2.bias"),
              st.eb(prefix + ".hidden_layers.2.running_mean"), st.eb(prefix + ".hidden_layers.2.running_var"),
              sm[0], si[0], ACT_LEAKY, ACT_NONE, self.train, 0.0)
        if slabs is not None:
            ops.bn_act_fwd_slabs(rec, h, y, slabs[0], slabs[1], rows * H, st.ep(prefix + ".hidden_layers.0.bias"), *bn)
        else:
            ops.bn_act_fwd(rec, h, y, *bn)
        return h

    def _hidden_bwd(self, rec, prefix, x, dh, dx=None, dx_accumulate=False):
        st = self.store
        y = self.buf[prefix + "/y"]
        sm, si = self.buf[prefix + "/save_mean"], self.buf[prefix + "/save_invstd"]
        ops.bn_act_bwd(rec, dh, st.eg(prefix + ".hidden_layers.2.weight"), st.eg(prefix + ".hidden_layers.2.bias"),
                       st.eg(prefix + ".hidden_layers.0.bias"), dh, y, None, st.ep(prefix + ".hidden_layers.2.weight"),
                       sm[0], si[0], ACT_LEAKY, ACT_NONE, 0.0)
        self._weight_grad(rec, prefix + ".hidden_layers.0.weight", dh, x)
        if dx is not None:
            ops.linear_bwd_x(rec, dx, dh, st.ep(prefix + ".hidden_layers.0.weight"), self.ws, accumulate=dx_accumulate)

    def _tape(self) -> TapeRecorder:
        return TapeRecorder(products=1 if self.plain_bf16 else 3)

    def _lin_fwd(self, rec, y, x, wkey, bkey, want_slabs=False, raw_slabs=False, gram_after=False):
        """nn.Linear forward; wide weights take the split-bf16 MFMA path when precision == 'bf16x3'.
        With ``want_slabs`` the wide path leaves its split-K partial sums unreduced and returns
        (slab buffer, n_slabs) so that the following BatchNorm kernel folds reduction + bias into its own pass
        (``y`` is then written by that kernel).  ``raw_slabs``: the stand-alone split-bf16 product only, its split-K slabs left for
        the caller's epilogue kernel (returns (slabs, n) -- or None with nothing recorded when that path does not apply).
        ``gram_after``: the caller records the batch-only Gram factor of ``x`` (``_want_gram``) itself, behind the product."""
        st = self.store
        if raw_slabs and not (self.precision == "bf16x3" and wkey in st.big and not self._can_fuse_next(wkey, x)):
            return None
        if self.precision == "bf16x3" and wkey in st.big:
            sp = self._split_cache.get(("fwd", x.data_ptr()))
            if sp is None:
                sp = ops.new_split_kb(x.shape[0], x.shape[1], self.dev)
                self._split_cache[("fwd", x.data_ptr())] = sp
                ops.split_bf16(rec, sp[0], sp[1], x)
            if not gram_after:
                self._want_gram(rec, x, wkey, x.shape[0], self.passes if x.shape[0] == self.R else 1)
            if self._can_fuse_next(wkey, x):
                M, N = y.shape
                S = ops.dw_adam_fwd_slabs(N, x.shape[1], ops.pad32(x.shape[0]))
                slabs = self._new(f"yslabs/{wkey}", S, M, N)
                self._next_fwd[wkey] = (slabs, S, sp)
                if not want_slabs:
                    ops.reduce_slabs(rec, y, slabs, st.ep(bkey), S)      # the whole wide forward of this step
                ops.fill(self.t_boot, slabs, 0.0)                       # stand-alone: full forward (no bias) into slab 0
                ops.linear_fwd_bf16x3(self.t_boot, slabs[0], sp[0], sp[1], st.ep(wkey), None, self._ws[0])
                return (slabs.view(S, M * N), S) if want_slabs else None
            # Stagger the HBM-bound wide kernels of the parallel modality branches: two of them side by side
            # take as long as back to back, but back to back lets modality i's narrow post-chain (reduce, BN,
            # layer_out) run underneath modality i+1's wide kernel instead of after both.
            stagger = self.branches and isinstance(rec, TapeRecorder) and len(rec.segments[-1]) > 1
            if stagger and getattr(self, "_last_wide_ev", None) is not None:
                rec.wait_event(self._last_wide_ev)
            if raw_slabs or (want_slabs and os.environ.get("FX_BN_SLABS_UNFUSED", "0") == "1"):      # (BN: no gain with the stand-alone forward)
                M, N = y.shape
                # FX_FWD_TILE256=1 (round 6, A/B; off): a tall weight behind at most 128 rows (the VAE decoders' FC_output) on the 128 x 256
                # output tile -- half the activation re-reads from the L2 per byte of W, bit-identical slabs; 3-9 % faster stand-alone
                # (profiles/r06_fwd128_variants.txt), nothing inside the cfg3 step (2.511 vs 2.505 ms: profiles/r06_tile256.txt)
                wc = 8 if (M <= 128 and N >= 4096 and os.environ.get("FX_FWD_TILE256", "0") == "1") else 0
                ns = ops.fwd_slabs_splitk(M, N, x.shape[1], wc)
                sbuf = self._new(f"slabs/{wkey}", ns, M * N)
                ops.linear_fwd_bf16x3_slabs(rec, sbuf, sp[0], sp[1], st.ep(wkey), M, wave_cols=wc)
                if stagger:              # (right behind the product: the next branch's weight read does not wait for this one's epilogue)
                    self._last_wide_ev = torch.cuda.Event()
                    rec.record_event(self._last_wide_ev)
                return sbuf, ns
            ops.linear_fwd_bf16x3(rec, y, sp[0], sp[1], st.ep(wkey), st.ep(bkey), self.ws)
            if stagger:
                self._last_wide_ev = torch.cuda.Event()
                rec.record_event(self._last_wide_ev)
        elif (self.precision == "bf16x3" and x.shape[1] >= 8192 and x.shape[1] % 8 == 0 and st.ep(wkey).data_ptr() % 16 == 0
              and x.is_contiguous()):
            # long contraction through a short-fat arena weight (the GNN's fc [latent, nodes * C]): the split-bf16 MFMA
            # forward streams it at ~2x the rate of the exact-fp32 GEMM's 1 x 1 output tiling (100 -> ~50 us at
            # [64, 128000], B = 32); gradients and Adam stay on the arena path
            sp = self._split_cache.get(("fwd", x.data_ptr()))
            if sp is None:
                sp = ops.new_split_kb(x.shape[0], x.shape[1], self.dev)
                self._split_cache[("fwd", x.data_ptr())] = sp
                ops.split_bf16(rec, sp[0], sp[1], x)
            ops.linear_fwd_bf16x3(rec, y, sp[0], sp[1], st.ep(wkey), st.ep(bkey), self.ws)
        else:
            ops.linear_fwd(rec, y, x, st.ep(wkey), st.ep(bkey), self.ws)
        return None

    def _can_fuse_next(self, wkey, x) -> bool:
        """The wide forward can ride on the previous step's dW+Adam launch when its input is a batch operand that the
        gather assembles one step ahead (not an activation of this step), of at most 384 rows (three 128-row M-tiles: the
        triplet network's stacked anchor / positive / negative rows at B <= 128), and the weight is trained."""
        if not (self.fuse_next and self.train and wkey in self.store.big) or self._is_frozen(wkey):
            return False
        # The kernel handles up to three 128-row M-tiles of the next batch (the triplet network's 3 B stacked rows), but beyond
        # one tile it is slower than the separate kernels: at cfg4 (K = 3 B = 384, 3 M-tiles) 1.92 ms per weight against
        # 1.18 + 0.59 ms -- 24 dependent LDS-DMA steps per tile in 64 KB of LDS leave it latency-bound (FX_FUSE_NEXT_MT=1
        # enables it for measurements).
        max_rows = 384 if os.environ.get("FX_FUSE_NEXT_MT", "0") == "1" else 128
        if x.shape[0] != self.R or x.shape[0] > max_rows or x.shape[1] % 4 != 0:
            return False
        return any(x.data_ptr() == X.data_ptr() for X in self.X) and ("fwd", x.data_ptr()) in self._split_cache

    def _lin_bwd_x(self, rec, dx, dy, wkey):
        """dX = dY . W.  Through a WIDE weight this is a second full read of W (4 B/param on top of the forward's):
        it takes the split-bf16 MFMA path like the forward instead of the exact-fp32 one (293 -> ~115 us at
        [20000, 5000])."""
        W = self.store.ep(wkey)
        if self.precision == "bf16x3" and wkey in self.store.big:
            if f"dy_kb/{wkey}" in self.buf:        # (the producer of dy wrote its split as well: fx_recon_sigmoid_slabs)
                sp = self.buf[f"dy_kb/{wkey}"], self.buf[f"dy_kb_lo/{wkey}"]
            else:
                sp = ops.new_split_kb(dy.shape[0], dy.shape[1], self.dev)
                self.buf[f"dy_kb/{wkey}"], self.buf[f"dy_kb_lo/{wkey}"] = sp
                ops.split_bf16(rec, sp[0], sp[1], dy)
            ops.linear_bwd_x_bf16x3(rec, dx, sp[0], sp[1], W, self.ws)
        else:
            ops.linear_bwd_x(rec, dx, dy, W, self.ws)

    def _small_fwd(self, rec, y, x, wkey, bkey):
        """A small dense layer on the critical chain (fusion layer, VAE FC_mean / FC_log_var): one latency-lean launch
        (fx_small_linear_fwd) instead of the tiled MFMA GEMM (FX_SMALL_LINEAR=0: A/B switch)."""
        st = self.store
        W = st.ep(wkey)
        if self.small_linear and wkey not in st.big and ops.small_linear_ok(x, W):
            ops.small_linear_fwd(rec, y, x, W, st.ep(bkey))
        else:
            ops.linear_fwd(rec, y, x, W, st.ep(bkey), self.ws)

    def _small_bwd(self, rec, dx, dy, x, wkey, bkey, need_dx=True):
        """All three gradients of such a layer in one launch (weight, bias, data) instead of three."""
        st = self.store
        W = st.ep(wkey)
        if self.small_linear and wkey not in st.big and not self._is_frozen(wkey) and ops.small_linear_ok(x, W):
            ops.small_linear_bwd(rec, dx if need_dx else None, st.eg(wkey), st.eg(bkey), dy, x, W)
            return
        self._weight_grad(rec, wkey, dy, x)
        ops.colsum(rec, st.eg(bkey), dy)
        if need_dx:
            ops.linear_bwd_x(rec, dx, dy, W, self.ws)

    def _is_frozen(self, key: str) -> bool:
        return bool(self.frozen) and key.startswith(self.frozen)

    def _stacked_gram_kb_ok(self, R, cols) -> bool:
        """dY dY^T of stacked rows runs on the split-bf16 forward kernel (dY's K-blocked split against dY itself): _weight_grad."""
        return (self.precision == "bf16x3" and R > 128 and cols >= 4096 and cols % 4 == 0 and self.gram_bf16x3 and self.gram_kb_wide)

    def _weight_grad(self, rec, key, dy, x, dyt=None, dy_kb=None):
        """dW = dY^T X: materialised, or deferred to the fused dW+clip+Adam kernel for wide layers.  ``dy_kb``: dY's K-blocked split,
        when the producer of dY wrote it (fx_block_bwd_ex)."""
        if self._is_frozen(key):
            return                                       # requires_grad=False: no gradient, not in the optimiser
        if self.fused and key in self.store.big:
            # wide layer on the engine path: nothing is materialised.  Here (inside the modality's backward
            # branch) only the pieces the optimiser tape needs are prepared:
            #   |dW|_F^2 = <X X^T, dY dY^T>  -> partial sums into the norm slots (Gram identity)
            #   split-bf16 transposed operands dY^T, X^T for the fused dW+clip+Adam kernel
            R = dy.shape[0]
            if self.clip:                                # the norm is only needed for the clip coefficient
                gx = self._gram_x_for(rec, x)
                if self._gram_kb_ok(dy):
                    # dY dY^T of a wide output gradient (the decoders' FC_output: [B, 20000]) on the split-bf16 Gram kernel the
                    # batch assembly uses for X X^T, instead of an exact-fp32 split-K GEMM (73-99 us beside the HBM-bound
                    # data-gradient product, at the package's power limit): one K-blocked split + one launch, slabs consumed un-reduced
                    if f"dy_kb/{key}" in self.buf:            # (made for the data-gradient product, or by the producer of dy)
                        sd = self.buf[f"dy_kb/{key}"], self.buf[f"dy_kb_lo/{key}"]
                    else:
                        sd = ops.new_split_kb(R, dy.shape[1], self.dev)
                        ops.split_bf16(rec, sd[0], sd[1], dy)
                    nd = ops.gram_kb_slices(dy.shape[1])
                    gd = self._new(f"gram_dy/{key}", nd, R * R)
                    ops.gram_kb_group(rec, [sd], [gd], [dy.shape[1]], R)
                    self.buf[f"gram_dy_split/{key}"], self.buf[f"gram_dy_split_lo/{key}"] = sd
                elif self._stacked_gram_kb_ok(R, dy.shape[1]) and dy.is_contiguous():
                    # stacked rows (the triplet network's 3 B = 384): dY dY^T like X X^T in _gram_x_for -- dY's K-blocked split against dY
                    # itself as the fp32 "weight" on the split-bf16 forward kernel, instead of the exact-fp32 GEMM (100 us per modality
                    # on the chain between the encoder tails' backward and the clip coefficient)
                    if dy_kb is not None:
                        sd = dy_kb
                    else:
                        sd = ops.new_split_kb(R, dy.shape[1], self.dev)
                        ops.split_bf16(rec, sd[0], sd[1], dy)
                    self.buf[f"gram_dy_split/{key}"], self.buf[f"gram_dy_split_lo/{key}"] = sd
                    nd = 1
                    gd = self._new(f"gram_dy/{key}", 1, R * R)
                    ops.linear_fwd_bf16x3(rec, gd.view(R, R), sd[0], sd[1], dy, None, self.ws)
                else:
                    nd = int(ops.lib.fx_gemm_splitk(R, R, dy.shape[1]))
                    gd = self._new(f"gram_dy/{key}", nd, R * R)
                    ops.gemm_slabs(rec, ops.GEMM_NT, gd, dy, dy, R, R)
                nb = ops.gram_hadamard_blocks(R * R)
                ops.gram_hadamard(rec, self.slots[self._slot_o:self._slot_o + nb], gx[0], gx[1], gd, nd, R * R)
                self._slot_o += nb
            xt = None
            if self.precision != "bf16x3":
                dyt = None
            if self.precision == "bf16x3":
                xt = self._split_cache.get(("T", x.data_ptr()))
                if xt is None:
                    xt = ops.new_split(x.shape[1], x.shape[0], self.dev)
                    self._split_cache[("T", x.data_ptr())] = xt
                    ops.split_bf16_t(rec, xt[0], xt[1], x)
                if dyt is None:                          # (fx_block_bwd may already have written the transposed split)
                    dyt = ops.new_split(dy.shape[1], dy.shape[0], self.dev)
                    ops.split_bf16_t(rec, dyt[0], dyt[1], dy)
                self.buf[f"dyT/{key}"], self.buf[f"dyT_lo/{key}"] = dyt
            self._jobs[key] = (dy, x, dyt, xt)
        else:
            ops.linear_bwd_w(rec, self.store.eg(key), dy, x, self.ws)

    def _weight_grad_prep_x(self, rec, key, x):
        """The pieces of _weight_grad that depend on the layer's INPUT only (X X^T slabs for the Gram norm, the transposed operand
        split): a caller that has the input long before the output gradient emits them early, where they cost nothing -- the
        later _weight_grad finds them in the caches."""
        if self._is_frozen(key) or not (self.fused and key in self.store.big):
            return
        if self.clip:
            self._gram_x_for(rec, x)
        if self.precision == "bf16x3" and ("T", x.data_ptr()) not in self._split_cache:
            xt = ops.new_split(x.shape[1], x.shape[0], self.dev)
            self._split_cache[("T", x.data_ptr())] = xt
            ops.split_bf16_t(rec, xt[0], xt[1], x)

    def _block_ok(self, rows, passes) -> bool:
        """fx_block_bwd covers one BatchNorm pass of at most 128 rows (everything but the triplet network's stacked passes)."""
        return self.block_bwd and self.train and passes == 1 and rows <= 128

    def _gram_full(self, rec, x):
        """X X^T [R, R] (reduced): the batch-only factor of the Gram norm that fx_block_bwd consumes."""
        gx = self._gram_x.get(("full", x.data_ptr()))
        if gx is None:
            R = x.shape[0]
            gx = self._new(f"gram_x_full/{x.data_ptr()}", R, R)
            sp = self._split_cache.get(("fwd", x.data_ptr())) if self.precision == "bf16x3" else None
            if sp is not None and (R * R) % 4 == 0 and os.environ.get("FX_GRAM_KB", "1") != "0":
                # the K-blocked split of x exists already (the wide forward made it): partial sums on the bf16 MFMA + one ordered reduce,
                # as the engine loop's batch assembly does -- 16 instead of 25 us per modality on the level-1 forward tape's chain
                F = x.shape[1]
                k = ops.gram_kb_slices(F)
                slabs = self._new(f"gram_x_slabs1/{x.data_ptr()}", k, R * R)
                ops.gram_kb_group(rec, [sp], [slabs], [F], R)
                ops.reduce_group(rec, [(gx, slabs, k, None)])
            else:
                ops.gemm(rec, ops.GEMM_NT, gx, x, x, None, self.ws)
            self._gram_x[("full", x.data_ptr())] = gx
        return gx

    def _want_gram(self, rec, x, wkey, rows, passes):
        """Emit the batch-only Gram factor for wide weight ``wkey`` in the form its backward will consume."""
        if not (self.fused and self.train and self.clip) or self._is_frozen(wkey) or wkey not in self.store.big:
            return
        if self._block_ok(rows, passes):
            self._gram_full(rec, x)
        else:
            self._gram_x_for(rec, x)

    def _tail_bwd(self, rec, ups, x_in, y, out, bn_prefix, bias_key, wkey, pre_act, post_act, drop_p):
        """Encoder tail backward through fx_block_bwd, then hand the wide layer ``wkey`` to the optimiser.
        ups = [(dE, weight key, bias key | None)]; y / out = saved wide-Linear output / block output."""
        st = self.store
        B, H = y.shape
        big = self.fused and wkey in st.big and not self._is_frozen(wkey)
        want_t = big and self.precision == "bf16x3"
        dyT = ops.new_split(H, B, self.dev) if want_t else None
        dy = None if want_t else self._new(f"dy/{wkey}", B, H)
        gx = slots = None
        if big and self.clip:
            gx = self._gram_full(rec, x_in)
            nb = ops.block_bwd_blocks(H)
            slots = self.slots[self._slot_o:self._slot_o + nb]
            self._slot_o += nb
        sm, si = self.buf[bn_prefix[1] + "/save_mean"], self.buf[bn_prefix[1] + "/save_invstd"]
        bp = bn_prefix[0]
        ups_t = [(dE, st.ep(wk), st.eg(wk), st.eg(bk) if bk else None) for (dE, wk, bk) in ups]
        grp = getattr(self, "_bb_group", None)
        if grp is not None:              # several modalities' tails in one launch: the caller emits fx_block_bwd_group
            grp[0].append(ops.block_bwd_desc(ups_t, y, out, st.ep(bp + ".weight"), sm[0], si[0], st.eg(bp + ".weight"),
                                             st.eg(bp + ".bias"), st.eg(bias_key), dy=dy, dyT=dyT, gram_x=gx, slots=slots))
            if not big and not self._is_frozen(wkey):
                grp[1].append(lambda: ops.linear_bwd_w(rec, st.eg(wkey), dy, x_in, self.ws))
        else:
            ops.block_bwd(rec, ups_t, y, out, st.ep(bp + ".weight"), sm[0], si[0], st.eg(bp + ".weight"), st.eg(bp + ".bias"),
                          st.eg(bias_key), pre_act, post_act, drop_p, dy=dy, dyT=dyT, gram_x=gx, slots=slots)
        if big:
            xt = None
            if want_t:
                self.buf[f"dyT/{wkey}"], self.buf[f"dyT_lo/{wkey}"] = dyT
                xt = self._split_cache.get(("T", x_in.data_ptr()))
                if xt is None:
                    xt = ops.new_split(x_in.shape[1], x_in.shape[0], self.dev)
                    self._split_cache[("T", x_in.data_ptr())] = xt
                    ops.split_bf16_t(rec, xt[0], xt[1], x_in)
            self._jobs[wkey] = (dy, x_in, dyT, xt)
        elif not self._is_frozen(wkey) and grp is None:
            ops.linear_bwd_w(rec, st.eg(wkey), dy, x_in, self.ws)

    def _gram_kb_ok(self, t) -> bool:
        """The split-bf16 Gram kernel (fx_gram_kb_group) applies: one M-tile of rows, a wide contiguous operand, bf16x3 precision."""
        return (self.precision == "bf16x3" and self.gram_kb_wide and t.shape[0] <= 128 and t.shape[1] >= 2048 and t.is_contiguous())

    def _gram_x_for(self, rec, x):
        """X X^T split-K slabs (depends on the batch only): emitted once per operand, in whatever branch asks
        first -- the forward branch of the modality when possible, so it hides under the other modality's
        HBM-bound forward kernel."""
        gx = self._gram_x.get(x.data_ptr())
        if gx is None:
            R = x.shape[0]
            sp = self._split_cache.get(("fwd", x.data_ptr()))
            if (self.precision == "bf16x3" and sp is not None and R > 128 and x.shape[1] >= 4096 and self.gram_bf16x3
                    and x.is_contiguous()):
                # stacked rows (triplet, 3 B = 384): X X^T on the split-bf16 forward kernel -- X's K-blocked split against X
                # itself as the fp32 "weight" -- instead of the exact-fp32 GEMM: ~200 us per modality of work that ran
                # beside the backward chain and starved its single-workgroup kernels (heads backward 52 -> 222 us)
                # (reduced here, in the batch-assembly branch: fx_gram_hadamard on the critical chain then reads one slab, not 85)
                gx = (self._new(f"gram_x/{x.data_ptr()}", 1, R * R), 1)
                ops.linear_fwd_bf16x3(rec, gx[0].view(R, R), sp[0], sp[1], x, None, self.ws)
            elif sp is not None and self._gram_kb_ok(x):
                # a wide hidden activation whose K-blocked split the forward already made (the decoders' [B, 5000] input of FC_output)
                nx = ops.gram_kb_slices(x.shape[1])
                gx = (self._new(f"gram_x/{x.data_ptr()}", nx, R * R), nx)
                ops.gram_kb_group(rec, [sp], [gx[0]], [x.shape[1]], R)
            else:
                nx = int(ops.lib.fx_gemm_splitk(R, R, x.shape[1]))
                gx = (self._new(f"gram_x/{x.data_ptr()}", nx, R * R), nx)
                ops.gemm_slabs(rec, ops.GEMM_NT, gx[0], x, x, R, R)
            self._gram_x[x.data_ptr()] = gx
        return gx

    def _head_losses(self, rec_f, emb):
        """Supervisor heads + their losses (value and output-gradient in one kernel each)."""
        spec, st, B = self.spec, self.store, self.B
        names = spec.loss_names()
        if self._heads_fusable(emb):
            self._heads_fwd_fused(rec_f, emb)
        else:
            for (v, kind, C) in spec.variables:
                o = self._new(f"MLPs.{v}/out", B, C)
                self._mlp_fwd(rec_f, "MLPs." + v, emb, o, B, 1, ["MLPs." + v])
        for (v, kind, C) in spec.variables:
            o = self.buf[f"MLPs.{v}/out"]
            do = self._new(f"MLPs.{v}/dout", B, C)
            li = self.loss_vec[names.index(v):names.index(v) + 1]
            lv = self._logvar(v)
            if v == spec.surv_event_var:
                ops.cox_ph(rec_f, li, do, o, self.y[spec.surv_time_var], self.y[v], lv)
            elif kind == "numerical":
                ops.mse_masked(rec_f, li, do, o, self.y[v], lv)
            else:
                ops.ce_masked(rec_f, li, do, o, self.y[v], lv)

    def _heads_step(self, rec, emb, demb, first_accumulate=False, with_total=True) -> bool:
        """Training plans: every supervisor head forward, its loss, its backward, the summed embedding gradient and (with
        ``with_total``) the model's total loss in ONE launch (fx_heads_step) instead of heads_fwd -> one loss kernel per head
        -> total_loss -> heads_bwd.  Returns False when the shapes are outside the kernel's range (the caller then records the
        separate launches)."""
        spec, st, B = self.spec, self.store, self.B
        if not (self.train and self.fuse_heads_step and spec.variables and self._heads_fusable(emb)):
            return False
        names = spec.loss_names()
        self._heads_fwd_fused(None, emb)                # descriptors (forward part); nothing is emitted
        kinds, labels, durs, lvs, losses = [], [], [], [], []
        for d, (v, kind, C) in zip(self._head_descs, spec.variables):
            pre = "MLPs." + v
            bias_key = pre + ".layer_out.bias"
            do = self._new(f"MLPs.{v}/dout", B, C)
            for field, t in (("dout", do), ("gW1", st.eg(pre + ".layer_1.weight")), ("gb1", st.eg(pre + ".layer_1.bias")),
                             ("ggamma", st.eg(pre + ".batchnorm.weight")), ("gbeta", st.eg(pre + ".batchnorm.bias")),
                             ("gW2", st.eg(pre + ".layer_out.weight")), ("gb2", st.eg(bias_key) if bias_key in st.eshapes else None)):
                setattr(d, field, t.data_ptr() if t is not None else None)
            if v == spec.surv_event_var:
                kinds.append(ops.LOSS_COX); durs.append(self.y[spec.surv_time_var])
            else:
                kinds.append(ops.LOSS_MSE if kind == "numerical" else ops.LOSS_CE); durs.append(None)
            labels.append(self.y[v])
            lvs.append(self._logvar(v))
            losses.append(self.loss_vec[names.index(v):names.index(v) + 1])
        scratch = self.buf["heads/dx_scratch"] = ops.heads_bwd_scratch(len(self._head_descs), B, emb.shape[1], self.dev)
        weighted = self.train and spec.weighted
        tl = [self.loss_vec[i:i + 1] for i in range(len(names))] if with_total else []
        tv = [st.ep("log_vars." + n).view(-1) for n in names] if (with_total and weighted) else []
        td = [st.eg("log_vars." + n).view(-1) for n in names] if (with_total and weighted) else []
        ops.heads_step(rec, self._head_descs, kinds, labels, durs, lvs, losses, emb, demb, B, emb.shape[1], DROPOUT_P, st.ctrl,
                       scratch, tl, tv, td, weighted, self.loss_vec[len(names):], self.epoch_acc if with_total else None,
                       dx_accumulate=first_accumulate)
        rec.mark("fork_issue_2")
        return True

    def _heads_fusable(self, emb) -> bool:
        """fx_heads_fwd / fx_heads_bwd cover every search-space shape of the reference (config.py:7-15: latent <= 128,
        supervisor hidden <= 32, batch <= 128); anything larger goes through the per-layer kernels."""
        m, st = ops.HEADS_MAX, self.store
        if not self.fuse_heads or self.B > m["B"] or emb.shape[1] > m["L"] or not (0 < len(self.spec.variables) <= m["heads"]):
            return False
        return all(st.eshapes[f"MLPs.{v}.layer_1.weight"][0] <= m["hidden"] and C <= m["n_out"]
                   for (v, _, C) in self.spec.variables)

    def _heads_fwd_fused(self, rec_f, emb):
        st, B = self.store, self.B
        descs = []
        for (v, kind, C) in self.spec.variables:
            pre = "MLPs." + v
            S = st.eshapes[pre + ".layer_1.weight"][0]
            bias_key = pre + ".layer_out.bias"
            mask = self._draw(pre, B, S) if (self.supplied and self.train) else None
            seed, off = self._rng()
            descs.append(ops.head_desc(
                W1=st.ep(pre + ".layer_1.weight"), b1=st.ep(pre + ".layer_1.bias"), gamma=st.ep(pre + ".batchnorm.weight"),
                beta=st.ep(pre + ".batchnorm.bias"), running_mean=st.eb(pre + ".batchnorm.running_mean"),
                running_var=st.eb(pre + ".batchnorm.running_var"), W2=st.ep(pre + ".layer_out.weight"),
                b2=st.ep(bias_key) if bias_key in st.eshapes else None, y1=self._new(pre + "/y1", B, S),
                a1=self._new(pre + "/a1", B, S), save_mean=self._new(pre + "/save_mean", 1, S),
                save_invstd=self._new(pre + "/save_invstd", 1, S), out=self._new(f"MLPs.{v}/out", B, C), mask=mask,
                seed=seed, offset=off, hidden=S, n_out=C))
        self._head_descs = descs
        if rec_f is not None:
            ops.heads_fwd(rec_f, descs, emb, B, emb.shape[1], self.train, DROPOUT_P if self.train else 0.0, ctrl=st.ctrl)

    def _head_bwd(self, rec_b, emb, demb, first_accumulate=False):
        if getattr(self, "_head_descs", None):
            st = self.store
            for d, (v, kind, C) in zip(self._head_descs, self.spec.variables):
                pre = "MLPs." + v
                bias_key = pre + ".layer_out.bias"
                for field, t in (("dout", self.buf[f"MLPs.{v}/dout"]), ("gW1", st.eg(pre + ".layer_1.weight")),
                                 ("gb1", st.eg(pre + ".layer_1.bias")), ("ggamma", st.eg(pre + ".batchnorm.weight")),
                                 ("gbeta", st.eg(pre + ".batchnorm.bias")), ("gW2", st.eg(pre + ".layer_out.weight")),
                                 ("gb2", st.eg(bias_key) if bias_key in st.eshapes else None)):
                    setattr(d, field, t.data_ptr() if t is not None else None)
            scratch = None
            if len(self._head_descs) > 1 and demb is not None and os.environ.get("FX_HEADS_SPLIT_DX", "1") != "0":
                scratch = self.buf.get("heads/dx_scratch")
                if scratch is None:
                    scratch = self.buf["heads/dx_scratch"] = ops.heads_bwd_scratch(len(self._head_descs), self.B, emb.shape[1], self.dev)
            ops.heads_bwd(rec_b, self._head_descs, emb, demb, self.B, emb.shape[1], DROPOUT_P, dx_accumulate=first_accumulate,
                          scratch=scratch)
            return
        acc = first_accumulate
        for (v, kind, C) in self.spec.variables:
            self._mlp_bwd(rec_b, "MLPs." + v, emb, self.buf[f"MLPs.{v}/dout"], self.B, 1, dx=demb, dx_accumulate=acc)
            acc = True

    def _total(self, rec_f):
        spec, st = self.spec, self.store
        names = spec.loss_names()
        weighted = self.train and spec.weighted
        losses = [self.loss_vec[i:i + 1] for i in range(len(names))]
        lvs = [st.ep("log_vars." + n).view(-1) for n in names] if weighted else []
        dls = [st.eg("log_vars." + n).view(-1) for n in names] if weighted else []
        ops.total_loss(rec_f, self.loss_vec[len(names):], losses, lvs, dls, weighted, self.epoch_acc)

    # ---- model schedules ---------------------------------------------------------------------------
    def _build(self):
        spec = self.spec
        rg = self.t_gather
        if self.cohort is not None:
            cur = self.store.ctrl if self.n_batches > 0 else None
            first_w = "encoders.{}.hidden_layers.0.weight" if spec.is_vae else "encoders.{}.layer_1.weight"
            enc_pos = {li: j for j, li in enumerate(spec.enc_idx)} if spec.is_vae else {i: i for i in range(len(spec.layers))}
            # FX_ASSEMBLY_BRANCHES=0: the whole batch assembly on ONE side stream (with the main chain that is two hardware queues:
            # more parallel branches alias onto the runtime's 4 queues and end up behind each other, profiles/r03_*)
            gbr = self.branches and os.environ.get("FX_ASSEMBLY_BRANCHES", "0") != "0"
            grouped = self._assembly_groupable(first_w, enc_pos)
            self.assembly_grouped = grouped
            gpar = rg.parallel(len(spec.layers) if (gbr and not grouped) else 1)
            gpar.__enter__()
            if grouped:                      # (inside the tape's single segment: PipelinedStep issues it as one detached fork)
                gpar.branch(0)
                self._build_assembly_grouped(rg, cur, first_w, enc_pos)
            for i, (name, _) in enumerate(spec.layers if not grouped else []):
                F = spec.engine_features(i)
                gpar.branch(i if gbr else 0)
                wk = first_w.format(enc_pos[i]) if i in enc_pos else None      # layers that are only reconstructed have no encoder
                if spec.model == "GNN":
                    wk = None                                                    # node features feed a graph conv, not a wide Linear
                if self.precision == "bf16x3" and wk in self.store.big:
                    # one pass: gather + fp32 copy + the bf16 splits the wide-layer kernels consume
                    sp, spt = ops.new_split_kb(self.R, F, self.dev), ops.new_split(F, self.R, self.dev)
                    self._split_cache[("fwd", self.X[i].data_ptr())] = sp
                    self._split_cache[("T", self.X[i].data_ptr())] = spt
                    ops.gather_split(rg, self.X[i], sp[0], sp[1], spt[0], spt[1], self.cohort.source(name, F), self.idx, cur, self.R,
                                     n_rows=self.R)
                else:
                    ops.gather_rows(rg, self.X[i], self.cohort.source(name, F), self.idx, cur, self.R)
                if wk is not None:
                    self._branch = i if gbr else 0
                    while len(self._ws) <= self._branch:
                        self._ws.append(Workspace(self.dev))
                    self._want_gram(rg, self.X[i], wk, self.R, self.passes)   # batch-only half of the Gram norm: part of batch assembly
                    self._branch = 0
            gpar.branch(0)
            for i, (name, F) in enumerate(spec.layers):
                if self.Xt[i] is not self.X[i]:          # a padded encoder input that is a reconstruction target as well
                    ops.gather_rows(rg, self.Xt[i], self.cohort.source(name, F), self.idx, cur, self.R)
            for k, t in self.y.items():      # labels of the anchors = first B indices of each batch row block
                ops.gather_rows(rg, t, self.cohort.ann[k], self.idx, cur, self.R)
            gpar.__exit__(None, None, None)
        if self.train:
            self._alloc_slots()
        if spec.is_vae:
            self._build_svae()
        elif spec.model == "GNN":
            self._build_gnn()
        else:
            self._build_mlp_family()
        if self.train and not self._next_fwd:
            self._build_optimizer()          # (with fused next-step forwards: built by link_next, once the partner exists)

    def _assembly_groupable(self, first_w, enc_pos) -> bool:
        """Every cohort layer feeds a wide, trained first Linear on the split-bf16 path with one BatchNorm pass of <= 128 rows:
        the whole assembly is then 3 launches (fx_gather_split_group, fx_gram_kb_group, fx_reduce_group) + the label gather."""
        spec, st = self.spec, self.store
        if (os.environ.get("FX_GROUP_ASSEMBLY", "1") == "0" or spec.model == "GNN" or self.precision != "bf16x3"
                or not (1 <= len(spec.layers) <= 4) or self.R > 128 or self.passes != 1):
            return False
        for i in range(len(spec.layers)):
            wk = first_w.format(enc_pos[i]) if i in enc_pos else None
            if wk not in st.big:
                return False
        # the Gram factor is produced in the reduced [R, R] form that fx_block_bwd consumes
        return self._block_ok(self.R, self.passes) or not (self.fused and self.train and self.clip)

    def _build_assembly_grouped(self, rg, cur, first_w, enc_pos):
        spec, st, R = self.spec, self.store, self.R
        items, splits, want = [], [], []
        for i, (name, _) in enumerate(spec.layers):
            F = spec.engine_features(i)
            wk = first_w.format(enc_pos[i])
            sp, spt = ops.new_split_kb(R, F, self.dev), ops.new_split(F, R, self.dev)
            self._split_cache[("fwd", self.X[i].data_ptr())] = sp
            self._split_cache[("T", self.X[i].data_ptr())] = spt
            items.append((self.X[i], sp[0], sp[1], spt[0], spt[1], self.cohort.source(name, F)))
            splits.append(sp)
            want.append(self.fused and self.train and self.clip and not self._is_frozen(wk))
        ops.gather_split_group(rg, items, self.idx, cur, R, R)
        if any(want):
            # X X^T [R, R] per modality (the batch-only factor of the Gram norm that fx_block_bwd consumes): partial sums on the
            # bf16 MFMA from the K-blocked splits just written, then one ordered reduce for all modalities
            sel = [i for i, w_ in enumerate(want) if w_]
            slabs = [self._new(f"gram_x_slabs/{i}", ops.gram_kb_slices(spec.engine_features(i)), R * R) for i in sel]
            ops.gram_kb_group(rg, [splits[i] for i in sel], slabs, [spec.engine_features(i) for i in sel], R)
            jobs = []
            for i, sl in zip(sel, slabs):
                gx = self._new(f"gram_x_full/{self.X[i].data_ptr()}", R, R)
                self._gram_x[("full", self.X[i].data_ptr())] = gx
                jobs.append((gx, sl, ops.gram_kb_slices(spec.engine_features(i)), None))
            if (R * R) % 4 == 0:
                ops.reduce_group(rg, jobs)
            else:
                for (gx, sl, k, _) in jobs:
                    ops.reduce_slabs(rg, gx, sl.view(k, R, R), None, k)

    def link_next(self, nxt: "StepPlan"):
        """Record the optimiser tape against the plan that holds the NEXT batch (PipelinedStep's other half): the fused
        dW+Adam launches multiply every updated weight tile into that plan's split input and leave the partial sums in
        its slab buffers."""
        if not self.train:
            return
        if set(self._next_fwd) != set(nxt._next_fwd):
            raise RuntimeError("link_next: the two plans fuse different layers")
        self.t_opt = self._tape()
        self._build_optimizer(nxt)

    def set_clip(self, max_norm: Optional[float]):
        """Re-record the optimiser tape for another gradient_clip_val (None / 0 = no clipping).  For plans driven by an
        external loop (FxAdam under the Lightning protocol), where the clip value is only known when the trainer calls
        configure_gradient_clipping."""
        clip = bool(max_norm) and float(max_norm) > 0.0
        norm = float(max_norm) if clip else self.clip_norm
        if not self.train or (clip == self.clip and norm == self.clip_norm):
            return
        if self._next_fwd:
            raise RuntimeError("set_clip: this plan's optimiser tape belongs to a PipelinedStep")
        if self.frozen and clip:
            raise ValueError("frozen parameter groups train without gradient clipping")
        self.clip, self.clip_norm = clip, norm
        self.t_opt = self._tape()
        self._build_optimizer()
        self.graph = None
        self.__dict__.get("_tape_graph", {}).pop("opt", None)
        self.__dict__.get("_tape_calls", {}).pop("opt", None)

    def _build_mlp_family(self):
        """DirectPred (direct_pred.py:107-133, :225-260) and MultiTripletNetwork
        (triplet_encoder.py:125-166, :276-330; anchor/positive/negative stacked as 3B rows)."""
        spec, st, B, R, n, L = self.spec, self.store, self.B, self.R, self.spec.n_layers, self.spec.latent_dim
        rf, rb = self.t_fwd, self.t_bwd
        trip = spec.model == "MultiTripletNetwork"
        tags = ["@a", "@p", "@n"] if trip else [""]
        ecat = self._new("ecat", R, n * L)
        self.path["grouped_tails"] = self._tails_groupable(n, L)
        if self.path["grouped_tails"]:
            emb = self._mlp_tails_fwd(rf, n, L, ecat)
        else:
            with rf.parallel(n if self.branches else 1) as par:      # one graph branch per modality
                for i in range(n):
                    self._enter_branch(par, i)
                    self._mlp_fwd(rf, f"encoders.{i}", self.X[i], ecat[:, i * L:(i + 1) * L], R, self.passes,
                                  [f"encoders.{i}{t}" for t in tags])
            self._branch = 0
            # FX_FORK_AFTER_WIDE=1 (A/B, round 6; off): the next batch's assembly (PipelinedStep: for the triplet network three gathers +
            # three X X^T products, ~0.4 ms of work) forks behind the wide forwards' join instead of behind the whole forward tape,
            # 110 us earlier -- its tail used to end after the Gram norm's last Hadamard sum (profiles/r06_timeline_cfg4.txt: 1966 vs
            # 1915 us).  Measured SLOWER: 5.31-5.35 vs 5.19-5.21 ms (bf16 mode 4.19 vs 4.11): beside the triplet / heads / fusion-backward
            # launches the assembly slows the critical chain by more than its tail cost (profiles/r06_cfg4_fork.txt).
            early = self.train and os.environ.get("FX_FORK_AFTER_WIDE", "0") == "1"
            if early:
                rf.mark("fork_assembly")
            if n > 1:
                emb = self._new("emb", R, L)
                self._small_fwd(rf, emb, ecat, "fusion_block.weight", "fusion_block.bias")
            else:
                emb = ecat
            if early:
                rf.mark("fork_issue_1")
        self.embeddings = emb[:B]
        demb = self._new("demb", R, L)
        if trip:
            names = spec.loss_names()
            ops.triplet(rf, self.loss_vec[0:1], demb[:B], demb[B:2 * B], demb[2 * B:], emb[:B], emb[B:2 * B],
                        emb[2 * B:], TRIPLET_MARGIN, self._logvar("triplet_loss"))
        stepped = self._heads_step(rf, emb[:B], demb[:B], first_accumulate=trip)
        self.path["heads_step"] = stepped
        if not stepped:
            self._head_losses(rf, emb[:B])
            self._total(rf)
        if not self.train:
            if self.attribution:
                self._build_mlp_attr(n, L)
            return
        # ---- backward
        if not stepped:
            self._head_bwd(rb, emb[:B], demb[:B], first_accumulate=trip)
        enc_frozen = self._is_frozen("encoders.0.layer_1.weight")
        if n > 1:
            decat = self._new("decat", R, n * L)
            self._small_bwd(rb, decat, demb, ecat, "fusion_block.weight", "fusion_block.bias", need_dx=not enc_frozen)
        else:
            decat = demb
        if enc_frozen:
            return          # FineTuner "encoders": True -- nothing upstream of the fusion layer needs a gradient
        self.path["grouped_bwd"] = bool(self.group_bwd and 1 < n <= 4 and self._block_ok(R, self.passes)
                                        and not any(self._is_frozen(f"encoders.{i}.layer_out.weight") for i in range(n)))
        if self.path["grouped_bwd"]:
            # every modality's encoder-tail backward in ONE launch (fx_block_bwd_group): no graph fork / join on the chain
            self._bb_group = ([], [])
            for i in range(n):
                self._mlp_bwd(rb, f"encoders.{i}", self.X[i], decat[:, i * L:(i + 1) * L], R, self.passes)
            descs, post = self._bb_group
            self._bb_group = None
            ops.block_bwd_group(rb, descs, R, ACT_NONE, ACT_RELU, DROPOUT_P)
            for f in post:
                f()
            return
        with rb.parallel(n if self.branches else 1) as par:
            for i in range(n):
                self._enter_branch(par, i)
                self._mlp_bwd(rb, f"encoders.{i}", self.X[i], decat[:, i * L:(i + 1) * L], R, self.passes)
        self._branch = 0

    def _build_mlp_attr(self, n, L):
        """Eval-mode input-gradient tapes of the MLP family: d out_v[:, c] / d X_i for a head v, the upstream gradient
        ``attr_dout[v]`` [B, C_v] being set by the caller (a one-hot column selects class c).  This is the gradient that
        Captum's IntegratedGradients / GradientShap evaluate on ``forward_target`` (reference direct_pred.py:418-431,
        :476-555) with the model in eval mode: BatchNorm = affine map of the running statistics, no dropout, ReLU gates
        from the forward's saved outputs.  Only the anchor rows [:B] carry a gradient (triplet: heads see emb[:B])."""
        spec, st, B = self.spec, self.store, self.B
        demb = self._new("attr/demb", B, L)
        for (v, kind, C) in spec.variables:
            ra = self.t_attr_head[v] = self._tape()
            pre = "MLPs." + v
            S = st.eshapes[pre + ".layer_1.weight"][0]
            do = self.attr_dout[v] = self._new(f"attr/dout.{v}", B, C)
            da1 = self._new(f"attr/{pre}/da1", B, S)
            ops.linear_bwd_x(ra, da1, do, st.ep(pre + ".layer_out.weight"), self._ws[0])
            ops.bn_eval_bwd(ra, da1, da1, None, self.buf[pre + "/a1"][:B], st.ep(pre + ".batchnorm.weight"),
                            st.eb(pre + ".batchnorm.running_var"), ACT_NONE, ACT_RELU)
            ops.linear_bwd_x(ra, demb, da1, st.ep(pre + ".layer_1.weight"), self._ws[0])
        rc = self.t_attr_common
        if n > 1:
            decat = self._new("attr/decat", B, n * L)
            ops.linear_bwd_x(rc, decat, demb, st.ep("fusion_block.weight"), self._ws[0])
        else:
            decat = demb
        for i in range(n):
            p = f"encoders.{i}"
            H = st.eshapes[p + ".layer_1.weight"][0]
            da1 = self._new(f"attr/{p}/da1", B, H)
            ops.linear_bwd_x(rc, da1, decat[:, i * L:(i + 1) * L], st.ep(p + ".layer_out.weight"), self._ws[0])
            ops.bn_eval_bwd(rc, da1, da1, None, self.buf[p + "/a1"][:B], st.ep(p + ".batchnorm.weight"),
                            st.eb(p + ".batchnorm.running_var"), ACT_NONE, ACT_RELU)
            dx = self._new(f"attr/dX.{i}", B, spec.engine_features(i))      # (engine width; the caller reads the logical columns)
            self._branch = 0
            self._lin_bwd_x(rc, dx, da1, p + ".layer_1.weight")
            self.dX.append(dx)

    def _build_svae_attr(self, enc, hs, mcat, vcat, eps_used, L):
        """Eval-mode input-gradient tapes of the VAE family: d out_v[:, c] / d X_i through z = mean + log_var * eps (the
        reference's forward_target differentiates its full forward, supervised_vae.py:553-563, :187-200: the latent is the
        SAMPLED z even in eval mode), the top-level FC_mean / FC_log_var, every encoder's FC_mean / FC_var and its
        Linear -> LeakyReLU -> BatchNorm block.  dX[j] belongs to input layer enc[j] (CrossModalPred: its input_layers)."""
        spec, st, B, n = self.spec, self.store, self.B, len(enc)
        dz = self._new("attr/dz", B, L)
        for (v, kind, C) in spec.variables:
            ra = self.t_attr_head[v] = self._tape()
            pre = "MLPs." + v
            S = st.eshapes[pre + ".layer_1.weight"][0]
            do = self.attr_dout[v] = self._new(f"attr/dout.{v}", B, C)
            da1 = self._new(f"attr/{pre}/da1", B, S)
            ops.linear_bwd_x(ra, da1, do, st.ep(pre + ".layer_out.weight"), self._ws[0])
            ops.bn_eval_bwd(ra, da1, da1, None, self.buf[pre + "/a1"][:B], st.ep(pre + ".batchnorm.weight"),
                            st.eb(pre + ".batchnorm.running_var"), ACT_NONE, ACT_RELU)
            ops.linear_bwd_x(ra, dz, da1, st.ep(pre + ".layer_1.weight"), self._ws[0])
        rc = self.t_attr_common
        dlv = self._new("attr/dlog_var", B, L)
        ops.mul(rc, dlv, dz, eps_used)                                    # z = mean + log_var * eps
        dmcat, dvcat = self._new("attr/dmcat", B, n * L), self._new("attr/dvcat", B, n * L)
        ops.linear_bwd_x(rc, dmcat, dz, st.ep("FC_mean.weight"), self._ws[0])
        ops.linear_bwd_x(rc, dvcat, dlv, st.ep("FC_log_var.weight"), self._ws[0])
        for i in range(n):
            p = f"encoders.{i}"
            H = st.eshapes[p + ".hidden_layers.0.weight"][0]
            dh = self._new(f"attr/{p}/dh", B, H)
            ops.linear_bwd_x(rc, dh, dmcat[:, i * L:(i + 1) * L], st.ep(p + ".FC_mean.weight"), self._ws[0])
            ops.linear_bwd_x(rc, dh, dvcat[:, i * L:(i + 1) * L], st.ep(p + ".FC_var.weight"), self._ws[0], accumulate=True)
            ops.bn_eval_bwd(rc, dh, dh, self.buf[p + "/y"][:B], None, st.ep(p + ".hidden_layers.2.weight"),
                            st.eb(p + ".hidden_layers.2.running_var"), ACT_LEAKY, ACT_NONE)
            dx = self._new(f"attr/dX.{i}", B, spec.engine_features(enc[i]))
            self._branch = 0
            self._lin_bwd_x(rc, dx, dh, p + ".hidden_layers.0.weight")
            self.dX.append(dx)

    def _build_gnn_attr(self, layers, h_last, gop, g, L):
        """Eval-mode input-gradient tapes of the GNN (reference gnn_early.py:427-438 under Captum): heads -> embedding -> fc ->
        for every conv layer, last to first: BatchNorm as the affine map of its running statistics with the ReLU gate of the
        saved output, then dL/dh_in = A^T (dY Wa) + dY Wr -- the training backward's message-passing chain, here carried
        through the FIRST layer too (training never needs the gradient of the node features)."""
        spec, st, B = self.spec, self.store, self.B
        if g["act"] != "relu":
            raise NotImplementedError("GNN attributions are implemented for act='relu' (the reference's default)")
        nodes, C = int(g["nodes"]), int(g["embedding_dim"])
        demb = self._new("attr/demb", B, L)
        for (v, kind, Cv) in spec.variables:
            ra = self.t_attr_head[v] = self._tape()
            pre = "MLPs." + v
            S = st.eshapes[pre + ".layer_1.weight"][0]
            do = self.attr_dout[v] = self._new(f"attr/dout.{v}", B, Cv)
            da1 = self._new(f"attr/{pre}/da1", B, S)
            ops.linear_bwd_x(ra, da1, do, st.ep(pre + ".layer_out.weight"), self._ws[0])
            ops.bn_eval_bwd(ra, da1, da1, None, self.buf[pre + "/a1"][:B], st.ep(pre + ".batchnorm.weight"),
                            st.eb(pre + ".batchnorm.running_var"), ACT_NONE, ACT_RELU)
            ops.linear_bwd_x(ra, demb, da1, st.ep(pre + ".layer_1.weight"), self._ws[0])
        rc = self.t_attr_common
        dh = self._new("attr/gnn/dh", B, nodes * C)
        self._branch = 0
        self._lin_bwd_x(rc, dh, demb, "encoders.0.fc.weight")
        da = dh.view(B, nodes, C)
        outs = [lay[0] for lay in layers[1:]] + [h_last]           # layer k's output = layer k + 1's input
        for k in reversed(range(len(layers))):
            h_in, y, sm, si, mask, seed, off, wa, ba, wr, bnp, agg = layers[k]
            d2, a2 = da.view(B * nodes, C), outs[k].view(B * nodes, C)
            ops.bn_eval_bwd(rc, d2, d2, None, a2, st.ep(bnp + ".weight"), st.eb(bnp + ".running_var"), ACT_NONE, ACT_RELU)
            cin = h_in.shape[2]
            t = self._new(f"attr/gnn/t{k}", B, nodes, cin)
            ops.rowlin2(rc, t, da, st.ep(wa), trans=True)
            dx = self._new(f"attr/gnn/dx{k}", B, nodes, cin)
            ops.spmm_rows(rc, dx, t, gop.s_rowptr, gop.s_idx, gop.s_w)
            if wr:
                ops.rowlin2(rc, dx, da, st.ep(wr), trans=True, accumulate=True)
            da = dx
        self.dX.append(da.view(B, nodes * da.shape[2]))

    def input_gradient(self, var: str):
        """Run the input-gradient tapes for head ``var`` (after forward(); attr_dout[var] set): fills self.dX."""
        if not self.attribution:
            raise RuntimeError("build the plan with attribution=True")
        self.t_attr_head[var].run()
        self.t_attr_common.run()

    def _build_gnn(self):
        """GNN (models/gnn_early.py:142-198): flexGCN encoder (modules.py:251-262: per layer conv -> BatchNorm1d over the
        batch*nodes rows -> act -> Dropout(0.2); flatten; fc) -> supervisor heads, losses as DirectPred.
        A conv layer is out = (A h) Wa^T + [h Wr^T] + b with A the graph's weighted adjacency (graph.py); its backward
        is dWa = dOut^T (A h) (the forward's aggregate), dWr = dOut^T h, db = colsum(dOut), dh = A^T (dOut Wa) + dOut Wr."""
        from . import graph as G
        from .arch import gnn_conv_keys
        spec, st, B, L = self.spec, self.store, self.B, self.spec.latent_dim
        g = spec.gnn
        nodes, C, K = int(g["nodes"]), int(g["embedding_dim"]), int(g["num_convs"])
        rf, rb = self.t_fwd, self.t_bwd
        # the graph operator is a property of the model, not of the plan: build the two CSR matrices once per (spec, device)
        cache = spec.__dict__.setdefault("_graph_ops", {})
        gop = cache.get(str(self.dev))
        if gop is None:
            gop = cache[str(self.dev)] = G.build(g["edge_index"], nodes, g["conv"], self.dev)
        self.buf["graph"] = gop
        act = ops.GACT[g["act"]]
        scratch = ops.gnn_scratch(B * nodes, 32, self.dev)
        self.buf["gnn/scratch"] = scratch
        drop = GNN_DROPOUT_P if self.train else 0.0
        h = self.X[0].view(B, nodes, int(g["node_features"]))
        layers = []
        for k in range(K):
            p, bnp = f"encoders.0.convs.{k}", f"encoders.0.bns.{k}"
            wa, ba, wr = gnn_conv_keys(p, g["conv"])
            cin = h.shape[2]
            agg = self._new(p + "/agg", B, nodes, cin)
            ops.spmm_rows(rf, agg, h, gop.t_rowptr, gop.t_idx, gop.t_w)
            y = self._new(p + "/y", B, nodes, C)
            ops.rowlin2(rf, y, agg, st.ep(wa), h if wr else None, st.ep(wr) if wr else None, st.ep(ba))
            a = self._new(p + "/a", B, nodes, C)
            sm, si = self._new(bnp + "/save_mean", C), self._new(bnp + "/save_invstd", C)
            mask = self._draw(f"encoders.0.drop.{k}", B, nodes, C) if (self.supplied and self.train) else None
            seed, off = self._rng()
            ops.bn_rows_fwd(rf, a, y, st.ep(bnp + ".weight"), st.ep(bnp + ".bias"), st.eb(bnp + ".running_mean"),
                            st.eb(bnp + ".running_var"), sm if self.train else None, si if self.train else None, act,
                            self.train, drop, scratch, mask=mask, seed=seed, offset=off, ctrl=st.ctrl)
            layers.append((h, y, sm, si, mask, seed, off, wa, ba, wr, bnp, agg))
            h = a
        hflat = h.view(B, nodes * C)
        emb = self._new("emb", B, L)
        self._lin_fwd(rf, emb, hflat, "encoders.0.fc.weight", "encoders.0.fc.bias")
        self.embeddings = emb
        demb = self._new("demb", B, L) if self.train else None
        stepped = self._heads_step(rf, emb, demb)
        if not stepped:
            self._head_losses(rf, emb)
            self._total(rf)
        if not self.train:
            if self.attribution:
                self._build_gnn_attr(layers, h, gop, g, L)
            return
        # ---- backward
        if not stepped:
            self._head_bwd(rb, emb, demb)
        if self._is_frozen("encoders.0.fc.weight"):
            return              # FineTuner "encoders": True
        self._weight_grad(rb, "encoders.0.fc.weight", demb, hflat)
        ops.colsum(rb, st.eg("encoders.0.fc.bias"), demb)
        dh = self._new("gnn/dh", B, nodes * C)
        self._lin_bwd_x(rb, dh, demb, "encoders.0.fc.weight")
        da = dh.view(B, nodes, C)
        # Two graph branches: the chain that carries dL/dh down the layers, and the weight-gradient reductions, which only
        # consume (u_k, dL/dy_k, h_k) and are off the critical path (FX_GNN_BRANCHES=0: one chain, A/B switch)
        two = self.branches and os.environ.get("FX_GNN_BRANCHES", "1") != "0"
        scratch_w = ops.gnn_scratch(B * nodes, 32, self.dev) if two else scratch
        self.buf["gnn/scratch_w"] = scratch_w
        jobs = []
        with rb.parallel(2 if two else 1) as par:
            par.branch(0)
            for k in reversed(range(K)):
                h_in, y, sm, si, mask, seed, off, wa, ba, wr, bnp, agg = layers[k]
                ops.bn_rows_bwd(rb, da, st.eg(bnp + ".weight"), st.eg(bnp + ".bias"), y, st.ep(bnp + ".weight"),
                                st.ep(bnp + ".bias"), sm, si, act, drop, scratch, mask=mask, seed=seed, offset=off,
                                ctrl=st.ctrl)                                                     # da <- dL/dy
                ev = torch.cuda.Event() if two else None
                if two:
                    rb.record_event(ev)
                jobs.append((ev, da, agg, h_in, wa, ba, wr))
                if k > 0:
                    # dL/dh_in = A^T (dY Wa) + dY Wr: the transposed message passing runs at the layer's INPUT width, and
                    # dWa = dY^T (A h_in) reuses the forward's aggregate -- no SpMM for the weight gradient, none at all
                    # in the first layer
                    cin = h_in.shape[2]
                    t = self._new(f"encoders.0.convs.{k}/t", B, nodes, cin)
                    ops.rowlin2(rb, t, da, st.ep(wa), trans=True)
                    dx = self._new(f"encoders.0.convs.{k}/dx", B, nodes, cin)
                    ops.spmm_rows(rb, dx, t, gop.s_rowptr, gop.s_idx, gop.s_w)
                    if wr:
                        ops.rowlin2(rb, dx, da, st.ep(wr), trans=True, accumulate=True)
                    da = dx
            par.branch(1 if two else 0)
            for ev, dy, agg, h_in, wa, ba, wr in jobs:
                if two:
                    rb.wait_event(ev)
                ops.rowlin_wgrad(rb, st.eg(wa), None, dy, agg, scratch_w)
                ops.rowlin_wgrad(rb, st.eg(wr) if wr else None, st.eg(ba), dy, h_in, scratch_w)
        self.buf["gnn/events"] = [j[0] for j in jobs]

    def _enter_branch(self, par, i):
        """Route subsequent emits to graph branch i (with its own split-K scratch)."""
        b = i if self.branches else 0
        par.branch(b)
        self._branch = b
        if i == 0:
            self._last_wide_ev = None
        while len(self._ws) <= b:
            self._ws.append(Workspace(self.dev))

    @property
    def ws(self):
        return self._ws[self._branch]

    def _build_svae(self):
        """supervised_vae (supervised_vae.py:132-200, :291-336, :494-550) and CrossModalPred, the same network with
        encoders over ``input_layers`` and decoders over ``output_layers`` (crossmodal_pred.py:79-117, :293-351)."""
        spec, st, B, L = self.spec, self.store, self.B, self.spec.latent_dim
        enc, dec = spec.enc_idx, spec.dec_idx       # encoders.j <- X[enc[j]] ; decoders.j -> X[dec[j]]
        n, nd = len(enc), len(dec)
        rf, rb = self.t_fwd, self.t_bwd
        mcat, vcat = self._new("mcat", B, n * L), self._new("vcat", B, n * L)
        hs = []
        vae_par = self.branches and os.environ.get("FX_VAE_BRANCHES", "1") != "0"
        mean, logv, z = self._new("mean", B, L), self._new("log_var", B, L), self._new("z", B, L)
        vae_group = self._tails_groupable(n, L) and os.environ.get("FX_VAE_GROUP", "1") != "0"
        self.path["grouped_tails"] = vae_group
        if vae_group:
            hs = self._vae_tails_fwd(rf, enc, L, mcat, vcat, mean, logv)
        else:
            with rf.parallel(n if vae_par else 1) as par:       # one graph branch per encoder (wide kernels staggered)
                for i in range(n):
                    if vae_par:
                        self._enter_branch(par, i)
                    p = f"encoders.{i}"
                    h = self._hidden_fwd(rf, p, self.X[enc[i]], B)
                    hs.append(h)
                    ops.linear_fwd(rf, mcat[:, i * L:(i + 1) * L], h, st.ep(p + ".FC_mean.weight"), st.ep(p + ".FC_mean.bias"), self.ws)
                    ops.linear_fwd(rf, vcat[:, i * L:(i + 1) * L], h, st.ep(p + ".FC_var.weight"), st.ep(p + ".FC_var.bias"), self.ws)
            self._branch = 0
            self._small_fwd(rf, mean, mcat, "FC_mean.weight", "FC_mean.bias")
            self._small_fwd(rf, logv, vcat, "FC_log_var.weight", "FC_log_var.bias")
        eps = self._draw("eps", B, L) if self.supplied else None
        eps_used = self._new("eps_used", B, L)
        seed, off = self._rng()
        ops.reparam(rf, z, mean, logv, eps=eps, eps_out=eps_used, seed=seed, offset=off, ctrl=st.ctrl)
        self.embeddings = z
        self.mean = mean
        # dz arrives in 1 + nd shares added in this order at the start of the backward tape: the heads' (written by the heads
        # launch, or zero-filled), then each decoder's MMD term (supervised_vae.py:309-313) computed inside that decoder's branch
        # ... and, with the fused latent backward, each decoder's split-K partial sums of dh . W_hidden behind them (written in the
        # decoder's own backward branch), all consumed by ONE ordered reduce after the branches join (FX_VAE_LATENT_FUSED=0: one
        # accumulating product + reduce per decoder on the main chain, then mul, then one launch each for FC_mean / FC_log_var)
        lat_fused = bool(self.train and os.environ.get("FX_VAE_LATENT_FUSED", "1") != "0")
        dz_ns = [int(ops.lib.fx_gemm_splitk(B, L, st.eshapes[f"decoders.{i}.hidden_layers.0.weight"][0])) if lat_fused else 0
                 for i in range(nd)]
        dzs = self._new("dz_shares", 1 + nd + sum(dz_ns), B * L)
        dz = dzs[0].view(B, L)
        lv_mmd = self._logvar("mmd_loss")
        hd, logits = [], []
        # dz must start from zero each step: the first head's data-grad GEMM overwrites it (accumulate=False),
        # so heads go FIRST in the backward tape and the MMD rows kernel (+=) is emitted after them; without heads an
        # explicit zero-fill takes their place.
        self.xhat = [self._new(f"xhat.{i}", B, spec.layers[dec[i]][1]) for i in range(nd)] if not self.train else None
        priors, rec_parts, row_sums = [], [], []
        # Schedule of a training plan with decoder branches (the engine's own step: forward and backward tapes always run together).
        # The chain is  z -> decoder 0's hidden layer -> FC_output products of decoder 0, 1, .. back to back (HBM-bound: one after the
        # other) -> last reconstruction epilogue -> data-gradient products of decoder 0, 1, .. -> latent backward;  everything else
        # is placed where nothing on that chain waits for it, WITHOUT further graph branches (one per decoder, as since round 2):
        #   * the supervisor heads need z only: behind decoder 0's reconstruction epilogue in ITS branch, under decoder 1's product
        #     (forward, losses, backward and their share of dz in one launch) instead of 68 us behind the last decoder's join;
        #   * the MMD terms (prior draw, kernel rows, dz share: 36 us each) need z only as well: decoder 0's moves to the head of the
        #     LAST decoder's branch, which waits for the earlier products anyway; they start on an idle chip, before decoder 0's
        #     product fills it;
        #   * the loss bookkeeping (MMD finalize, total) needs every reconstruction sum: the tail of decoder 0's BACKWARD branch;
        #   * what the optimiser needs of a decoder's FC_output (Gram norm share, transposed operand splits, bias gradient): input
        #     side before its product (decoders > 0: under decoder 0's), output-gradient side before its data-gradient product
        #     (decoders > 0) or behind it (decoder 0).
        # A branch of their own for the heads / the bookkeeping measured 1-2 % faster still, but hipGraphLaunch of the step graph then
        # segfaulted in some test sequences (ROCm 7.x; never with one branch per decoder) -- DESIGN.md section 4.1.
        # FX_VAE_HEADS_BRANCH=0 restores round 3's order (MMD rows first in each decoder branch; heads, MMD finalize and total on the
        # main chain after the decoders' join).  Level-1 plans (forward_alone) keep that order: their caller reads the total loss
        # between the tapes.
        heads_aside = bool(self.train and vae_par and nd > 1 and not self.forward_alone
                           and os.environ.get("FX_VAE_HEADS_BRANCH", "1") != "0")
        self._losses_in_bwd = heads_aside        # (losses() between forward() and backward() would read a stale MMD term / total)
        # FX_VAE_HEADS_BRANCH=2 (round 6, A/B): the heads on a graph branch of their OWN next to the decoder branches -- the schedule that
        # made torch's CUDAGraph layer segfault in round 4; with the library's own capture (ops.FxGraph) it can be measured again
        heads_branch = bool(heads_aside and self.branches and os.environ.get("FX_VAE_HEADS_BRANCH", "1") == "2")
        # FX_VAE_MMD_BRANCH=1 (round 6, A/B; off): every decoder's MMD term (prior draw, kernel rows, dz share: they need z only) on ONE
        # graph branch of their own that starts with the decoder branches, instead of inside the decoder branches -- where a later
        # decoder's term stands in front of its FC_output product (which starts 30 us after decoder 0's has ended:
        # profiles/r06_timeline_cfg3.txt) and decoder 0's deferred term is the last launch of the forward tape.  Measured SLOWER: 2.500
        # vs 2.470 ms (bf16 mode 2.395 vs 2.348; profiles/r06_mmd_branch.txt) -- like the heads branch, a further fork / join and two
        # more narrow kernels beside decoder 0's product cost more than the gaps they close.
        mmd_branch = bool(heads_aside and self.branches and not heads_branch and os.environ.get("FX_VAE_MMD_BRANCH", "0") == "1")
        self.path["mmd_branch"] = mmd_branch
        deferred_mmd = []
        n_fbr = nd + (1 if (heads_branch or mmd_branch) else 0)
        with rf.parallel(n_fbr if vae_par else 1) as par:      # one graph branch per decoder (+ the heads' or the MMD terms')
            for i in range(nd):
                if vae_par:
                    self._enter_branch(par, i)
                p = f"decoders.{i}"
                F = spec.layers[dec[i]][1]
                # The decoder's MMD prior sample (supervised_vae.py:420-426) and, behind FC_output, its reconstruction term
                # with dlogits overwriting the logits (:301-313) ride in the decoder's own branch
                prng = None
                if self.supplied:
                    pr = self._draw(f"prior.{i}", MMD_PRIOR, L)
                else:
                    pr = self._new(f"prior.{i}", MMD_PRIOR, L)
                    prng = self._rng()
                priors.append(pr)
                rs = self._new(f"mmd_rows.{i}", 2 * (MMD_PRIOR + B))
                row_sums.append(rs)
                dzm = dzs[1 + i].view(B, L)

                def mmd_term(rec, pr=pr, rs=rs, dzm=dzm, prng=prng):
                    if prng is not None:
                        ops.fill_normal(rec, pr, prng[0], prng[1], ctrl=st.ctrl)
                    ops.mmd_rows(rec, rs, dzm if self.train else None, pr, z, lv_mmd, 1.0 / nd, overwrite=True)
                # FX_VAE_MMD_LATE (default on, round 5): a later decoder's branch FIRST computes its hidden layer (so that its FC_output
                # product is ready the moment decoder 0's ends: the MMD rows kernels, 60 us each beside a product, used to stand in
                # front of it and the second product started 50 us late, profiles/r05_timeline_cfg3.txt), then its own MMD term under
                # decoder 0's product; decoder 0's term goes to the END of the last branch, behind the reconstruction epilogue.
                mmd_late = heads_aside and os.environ.get("FX_VAE_MMD_LATE", "1") != "0"
                if mmd_branch:
                    deferred_mmd.append(mmd_term)
                elif heads_aside and i == 0:
                    deferred_mmd.append(mmd_term)
                elif not mmd_late:
                    mmd_term(rf)
                    if heads_aside and i == nd - 1:
                        for term in deferred_mmd:
                            term(rf)
                h = self._hidden_fwd(rf, p, z, B)
                hd.append(h)
                if self.train and vae_par and i > 0 and os.environ.get("FX_VAE_PREP_X", "1") != "0":
                    # (beside decoder 0's FC_output product; decoder 0 itself heads the critical chain and prepares in the backward)
                    self._weight_grad_prep_x(rf, p + ".FC_output.weight", h)
                if mmd_late and i > 0 and not mmd_branch:
                    mmd_term(rf)
                lg = self._new(p + "/logits", B, F)
                logits.append(lg)
                wkey = p + ".FC_output.weight"
                # training: the reconstruction term, dlogits and their bf16 split as ONE epilogue pass over the product's split-K slabs
                # (FX_RECON_EPILOGUE=0: reduce -> recon -> split, three launches on the chain between the forward and the data gradient).
                # gram_after: the batch-only Gram factor FC_output's optimiser step needs is the un-reduced one _weight_grad asks for
                # (_gram_x_for), not the reduced [B, B] one fx_block_bwd consumes -- _want_gram made the latter here, unused.
                epi = None
                if (self.train and F % 4 == 0 and self.Xt[dec[i]].is_contiguous() and not self._is_frozen(wkey)
                        and os.environ.get("FX_RECON_EPILOGUE", "1") != "0"):
                    epi = self._lin_fwd(rf, lg, h, wkey, p + ".FC_output.bias", raw_slabs=True, gram_after=True)
                if epi is not None:
                    dsp = ops.new_split_kb(B, F, self.dev)
                    self.buf[f"dy_kb/{wkey}"], self.buf[f"dy_kb_lo/{wkey}"] = dsp
                    nblk = ops.recon_sigmoid_slabs_blocks(B, F)
                    rp = self._new(f"recon_part.{i}", nblk)
                    ops.recon_sigmoid_slabs(rf, rp, lg, dsp, epi[0], epi[1], st.ep(p + ".FC_output.bias"), self.Xt[dec[i]], lv_mmd, 1.0 / nd)
                else:
                    self._lin_fwd(rf, lg, h, wkey, p + ".FC_output.bias", gram_after=True)
                    nblk = int(ops.lib.fx_recon_blocks(B * F))
                    rp = self._new(f"recon_part.{i}", 1024)
                    ops.recon_sigmoid(rf, rp, lg if self.train else None, self.xhat[i] if self.xhat else None, lg, self.Xt[dec[i]], lv_mmd,
                                      1.0 / nd)
                rec_parts.append((rp, nblk))
                if mmd_late and i == nd - 1 and not mmd_branch:
                    for term in deferred_mmd:
                        term(rf)
                if heads_aside and i == 0 and not heads_branch:
                    self._svae_heads(rf, z, dz)
            if heads_branch:
                self._enter_branch(par, nd)
                self._svae_heads(rf, z, dz)
            if mmd_branch:
                self._enter_branch(par, nd)
                for term in deferred_mmd:
                    term(rf)
        self._branch = 0
        if not heads_aside:
            self._svae_heads(rf, z, dz)

        def bookkeeping(rec):
            for i in range(nd):       # mmd_loss = mean over the reconstructed layers (supervised_vae.py:309-313)
                F = spec.layers[dec[i]][1]
                rp, nblk = rec_parts[i]
                ops.mmd_finalize(rec, self.loss_vec[0:1], row_sums[i], MMD_PRIOR, B, rp, nblk, float(B * F), 1.0 / nd, i > 0)
            self._total(rec)
        if not heads_aside:
            bookkeeping(rf)
        if not self.train:
            if self.attribution:
                self._build_svae_attr(enc, hs, mcat, vcat, eps_used, L)
            return
        # ---- backward through decoders (dlogits live in logits[i]) -> dz, then latent, then encoders.  One graph branch per
        # decoder (the narrow launches around one decoder's FC_output products run beside the other's); their shares of dz
        # are added on the main chain afterwards, in decoder order.
        dz_sum = self._new("dz", B, L)
        if not lat_fused:
            ops.reduce_slabs(rb, dz_sum, dzs, None, 1 + nd)          # heads + MMD terms, in share order
        dz = dz_sum
        dhs = []
        # One graph branch per decoder.  Each holds the data-gradient product through FC_output (a full read of the weight:
        # HBM-bound, two of them side by side take as long as one after the other) with the hidden layer's backward behind it, and
        # the narrow launches that only the optimiser tape needs of this weight (Gram norm share = two B x B products + their
        # Hadamard sum, transposed operand splits, bias gradient).  Decoder 0 reads its weight FIRST and prepares afterwards, the
        # others prepare first: every branch's narrow work runs under another branch's weight read.  (As extra branches the
        # preparation cost more in fork / join edges than it hid: 2.93 vs 2.89 ms.)
        # PARTIAL JOIN (round 6, FX_VAE_PARTIAL_JOIN=0: A/B): the latent / encoder backward needs the decoders' shares of dz only, not
        # what the optimiser tape needs of decoder 0's FC_output (Gram norm share, transposed splits, bias gradient) nor the loss
        # bookkeeping, which follow decoder 0's dz share in ITS branch (~90 us of narrow launches, profiles/r05_timeline_cfg3.txt:
        # 626 -> 717 us).  So the chain continues INSIDE the last decoder's branch -- the last one issued, hence the one that may wait
        # for events of the others (TapeRecorder.record_event) -- as soon as every earlier branch has produced its share, and the
        # branches join behind it: decoder 0's tail runs beside the latent / encoder backward instead of in front of it.
        late = bool(vae_par and nd > 1 and lat_fused and self.branches and os.environ.get("FX_VAE_PARTIAL_JOIN", "1") != "0")
        self.path["partial_join"] = late
        dz_evs = []

        def latent_chain():
            if lat_fused:
                n_all = 1 + nd + sum(dz_ns)                  # heads, MMD terms, decoder 0's partial sums, decoder 1's, ...
                if L % 4 == 0 and n_all >= 16:
                    ops.reduce_slabs_par(rb, dz_sum, dzs, None, n_all)
                else:
                    ops.reduce_slabs(rb, dz_sum, dzs, None, n_all)
            self._svae_bwd_latent(rb, enc, hs, mcat, vcat, dz, dhs, eps_used, n, nd, L, B, dz_complete=lat_fused)

        with rb.parallel(nd if vae_par else 1) as par:
            for i in range(nd):
                p = f"decoders.{i}"
                dh = self._new(p + "/dh", B, st.eshapes[p + ".hidden_layers.0.weight"][0])
                dhs.append(dh)
                if vae_par:
                    self._enter_branch(par, i)
                prep_first = vae_par and i > 0
                if prep_first:
                    self._weight_grad(rb, p + ".FC_output.weight", logits[i], hd[i])
                    ops.colsum(rb, st.eg(p + ".FC_output.bias"), logits[i])
                self._lin_bwd_x(rb, dh, logits[i], p + ".FC_output.weight")
                self._hidden_bwd(rb, p, z, dh)
                if lat_fused:
                    o = 1 + nd + sum(dz_ns[:i])
                    ops.gemm_slabs(rb, ops.GEMM_NN, dzs[o:o + dz_ns[i]], dh, st.ep(p + ".hidden_layers.0.weight"), B, L)
                if late and i < nd - 1:
                    ev = torch.cuda.Event()
                    rb.record_event(ev)                      # this decoder's share of dz is complete
                    dz_evs.append(ev)
                if not prep_first:
                    self._weight_grad(rb, p + ".FC_output.weight", logits[i], hd[i])
                    ops.colsum(rb, st.eg(p + ".FC_output.bias"), logits[i])
                if heads_aside and i == 0:
                    bookkeeping(rb)
                if late and i == nd - 1:
                    for ev in dz_evs:
                        rb.wait_event(ev)
                    latent_chain()
        self._branch = 0
        if not late:
            latent_chain()

    def _svae_heads(self, rec, z, dz):
        """The supervisor heads of a VAE plan: losses and, when training, their backward with dz's first share (the total needs the
        MMD terms and is recorded separately)."""
        stepped = self._heads_step(rec, z, dz, with_total=False)
        if not stepped:
            self._head_losses(rec, z)
        if self.train and not stepped:
            if self.spec.variables:
                self._head_bwd(rec, z, dz, first_accumulate=False)  # emitted into the forward tape: dz's first share
            else:
                # unsupervised run (reference __main__.py:997 accepts supervised_vae / CrossModalPred without target
                # variables): no head overwrites dz, and every later contribution accumulates into it
                ops.fill(rec, dz, 0.0)

    def _svae_bwd_latent(self, rb, enc, hs, mcat, vcat, dz, dhs, eps_used, n, nd, L, B, dz_complete=False):
        """dz shares of the decoders (in decoder order), the reparameterisation, the top-level FC_mean / FC_log_var and the
        encoder tails (supervised_vae.py:172-200 backwards)."""
        st = self.store
        if not dz_complete:
            for i in range(nd):
                ops.linear_bwd_x(rb, dz, dhs[i], st.ep(f"decoders.{i}.hidden_layers.0.weight"), self.ws, accumulate=True)
        dmcat, dvcat = self._new("dmcat", B, n * L), self._new("dvcat", B, n * L)
        enc_frozen = self._is_frozen("encoders.0.FC_mean.weight")    # FineTuner "encoders": True -- the encoders need no gradient
        tops = ("FC_mean.weight", "FC_log_var.weight")
        if (dz_complete and self.small_linear and not any(k in st.big or self._is_frozen(k) for k in tops)
                and ops.small_linear_ok(mcat, st.ep(tops[0])) and ops.small_linear_ok(vcat, st.ep(tops[1]))):
            # z = mean + log_var * eps: d mean = dz, d log_var = dz * eps -- both layers' three gradients in one launch, the product
            # with eps taken where the kernel reads its upstream gradient
            ops.small_linear_bwd_group(rb, [
                dict(dx=None if enc_frozen else dmcat, gW=st.eg("FC_mean.weight"), gb=st.eg("FC_mean.bias"), dy=dz, x=mcat,
                     W=st.ep("FC_mean.weight")),
                dict(dx=None if enc_frozen else dvcat, gW=st.eg("FC_log_var.weight"), gb=st.eg("FC_log_var.bias"), dy=dz, dy_mul=eps_used,
                     x=vcat, W=st.ep("FC_log_var.weight"))])
        else:
            dlv = self._new("dlog_var", B, L)
            ops.mul(rb, dlv, dz, eps_used)
            self._small_bwd(rb, dmcat, dz, mcat, "FC_mean.weight", "FC_mean.bias", need_dx=not enc_frozen)
            self._small_bwd(rb, dvcat, dlv, vcat, "FC_log_var.weight", "FC_log_var.bias", need_dx=not enc_frozen)
        if enc_frozen:
            return
        if not self._block_ok(B, 1):
            for i in range(n):
                p = f"encoders.{i}"
                dm, dv = dmcat[:, i * L:(i + 1) * L], dvcat[:, i * L:(i + 1) * L]
                dh = self._new(p + "/dh", B, st.eshapes[p + ".hidden_layers.0.weight"][0])
                self._weight_grad(rb, p + ".FC_mean.weight", dm, hs[i])
                ops.colsum(rb, st.eg(p + ".FC_mean.bias"), dm)
                self._weight_grad(rb, p + ".FC_var.weight", dv, hs[i])
                ops.colsum(rb, st.eg(p + ".FC_var.bias"), dv)
                ops.linear_bwd_x(rb, dh, dm, st.ep(p + ".FC_mean.weight"), self.ws)
                ops.linear_bwd_x(rb, dh, dv, st.ep(p + ".FC_var.weight"), self.ws, accumulate=True)
                self._hidden_bwd(rb, p, self.X[enc[i]], dh)
            return
        # every encoder's tail backward in ONE launch (fx_block_bwd_group, two upstream Linears each); a single encoder: fx_block_bwd
        group = self.group_bwd and 1 < n <= 4
        if group:
            self._bb_group = ([], [])
        for i in range(n):
            p = f"encoders.{i}"
            dm, dv = dmcat[:, i * L:(i + 1) * L], dvcat[:, i * L:(i + 1) * L]
            self._tail_bwd(rb, [(dm, p + ".FC_mean.weight", p + ".FC_mean.bias"), (dv, p + ".FC_var.weight", p + ".FC_var.bias")],
                           self.X[enc[i]], self.buf[p + "/y"], hs[i], (p + ".hidden_layers.2", p), p + ".hidden_layers.0.bias",
                           p + ".hidden_layers.0.weight", ACT_LEAKY, ACT_NONE, 0.0)
        if group:
            descs, post = self._bb_group
            self._bb_group = None
            ops.block_bwd_group(rb, descs, B, ACT_LEAKY, ACT_NONE, 0.0)
            for f in post:
                f()

    def _build_optimizer(self, nxt: Optional["StepPlan"] = None):
        """clip_grad_norm_(1.0) + Adam over every parameter (main.py:216-217, direct_pred.py:143)."""
        st, ro = self.store, self.t_opt
        if self._next_fwd and nxt is None:
            raise RuntimeError("this plan's wide forward is fused into the partner's optimiser step: use PipelinedStep")
        o = self._slot_o
        if not self.fused:
            for k in st.big_keys:
                gbuf = st.big[k]["_G"]
                ops.sumsq(ro, self.slots[o:], gbuf.view(-1))      # padded buffer: the padding is zero
                o += ops.sumsq_blocks(gbuf.numel())
        ops.sumsq(ro, self.slots[o:], st.G)
        o += ops.sumsq_blocks(st.n_small)
        assert o <= self.slots.numel(), (o, self.slots.numel())
        trainable = None
        if self.frozen:             # 0/1 mask over the small-parameter arena: frozen tensors are skipped by Adam
            trainable = torch.ones_like(st.P)
            for k in st.small_keys:
                if self._is_frozen(k):
                    st._view(trainable, k).zero_()
            self.buf["trainable_mask"] = trainable
        if os.environ.get("FX_ADAM_CLIP_FUSED", "1") != "0":        # norm -> clip coefficient -> flat Adam in one launch
            ops.adam_flat_clip(ro, st.P, st.G, st.M, st.V, st.ctrl, self.slots, self.slots.numel(),
                               self.clip_norm if self.clip else 0.0, trainable)
        else:
            ops.clip_finalize(ro, st.ctrl, self.slots, self.slots.numel(), self.clip_norm if self.clip else 0.0)
            ops.adam_flat(ro, st.P, st.G, st.M, st.V, st.ctrl, trainable)
        for k in st.big_keys:
            if self._is_frozen(k):
                continue
            d = st.big[k]
            if self.fused and self.precision == "bf16x3":
                dy, x, dyt, xt = self._jobs[k]
                if k in self._next_fwd:
                    nslabs, _, nsp = nxt._next_fwd[k]
                    ops.linear_dw_adam_fwd_bf16x3(ro, d["W"], d["M"], d["V"], dyt[0], dyt[1], xt[0], xt[1], st.ctrl, nsp[0], nsp[1],
                                                  nxt.R, nslabs, nt=not ops.TUNE["adam_plain"], mapping=ops.TUNE["fused_map"])
                else:
                    ops.linear_dw_adam_bf16x3(ro, d["W"], d["M"], d["V"], dyt[0], dyt[1], xt[0], xt[1], st.ctrl)
            elif self.fused:
                dy, x, _, _ = self._jobs[k]
                ops.linear_dw_adam(ro, d["W"], d["M"], d["V"], dy, x, st.ctrl)
            else:
                ops.adam_flat(ro, d["_W"].view(-1), d["_G"].view(-1), d["_M"].view(-1), d["_V"].view(-1), st.ctrl)   # padding stays 0

    def _alloc_slots(self):
        """fp64 partial sums of the squared grad norm: Gram blocks per wide weight (fused) or Sum g^2 blocks of the
        materialised wide grads, plus the small-arena blocks.  Unused slots stay 0."""
        st = self.store
        n = ops.sumsq_blocks(st.n_small)
        for k in st.big_keys:
            if self.fused:       # Gram-hadamard blocks, or one slot per 32 columns when fx_block_bwd produces the norm share
                n += max(ops.gram_hadamard_blocks(self.R * self.R), ops.block_bwd_blocks(st.big[k]["W"].shape[0]))
            else:
                n += ops.sumsq_blocks(st.big[k]["_W"].numel())
        self.slots = torch.zeros(n, dtype=torch.float64, device=self.dev)

    # ---- execution ----------------------------------------------------------------------------------
    def set_batch(self, x_list=None, y=None, parts=None):
        """Direct mode: copy a collated batch (reference DataLoader output) into the static buffers."""
        B = self.B
        if parts is not None:      # triplet: (anchor, positive, negative) lists
            for j, xs in enumerate(parts):
                for i, x in enumerate(xs):
                    self.X[i][j * B:(j + 1) * B, :x.shape[1]].copy_(x, non_blocking=True)
        elif x_list is not None:
            for i, x in enumerate(x_list):
                self.X[i][:, :x.shape[1]].copy_(x, non_blocking=True)      # (the engine's buffer may be wider: zero pad columns)
                if self.Xt[i] is not self.X[i]:
                    self.Xt[i].copy_(x, non_blocking=True)
        if y is not None:
            for k, t in self.y.items():
                t.copy_(torch.as_tensor(y[k]).to(torch.float32), non_blocking=True)

    def set_draws(self, draws: Dict[str, torch.Tensor]):
        for k, t in self.draws.items():
            if k in draws:
                t.copy_(draws[k].to(torch.float32), non_blocking=True)

    def bump_nbt(self):
        for k in self.store.nbt:
            self.store.nbt[k] += self.passes if k.startswith("encoders.") and self.passes > 1 else 1

    input_gradient = ops.device_guard(input_gradient)

    def _run_tape(self, name: str):
        """Issue one tape; with ``tape_graphs`` (plans driven call by call from an external loop: the Lightning protocol) the
        second use captures it into a hipGraph and later uses replay that -- ~50 eager launches per step become three graph
        launches."""
        tape = getattr(self, "t_" + name)
        self.store.note_stream()
        if not getattr(self, "tape_graphs", False):
            tape.run()
            return
        calls = self.__dict__.setdefault("_tape_calls", {})
        graphs = self.__dict__.setdefault("_tape_graph", {})
        n = calls.get(name, 0)
        calls[name] = n + 1
        g = graphs.get(name)
        if g is None and n >= 1 and not ops.capturing():
            torch.cuda.synchronize()
            g = ops.new_graph()
            with ops.graph_capture(g):
                tape.run()
            graphs[name] = g
        if g is not None:
            g.replay()
        else:
            tape.run()

    @ops.device_guard
    def forward(self):
        """The forward tape alone.  A training plan of the VAE family built without ``forward_alone=True`` finishes its loss
        bookkeeping (MMD term, total) in the BACKWARD tape: ``losses()`` raises until ``backward()`` has run."""
        self._run_tape("fwd")
        self._fwd_pending = True

    @ops.device_guard
    def backward(self):
        self._run_tape("bwd")
        self._fwd_pending = False

    @ops.device_guard
    def run_optimizer_tape(self):
        """The recorded clip + Adam launches (the control block must already hold this step's counters: step_begin)."""
        self._run_tape("opt")

    @ops.device_guard
    def optimizer_step(self, lr: float):
        ops.step_begin(ops.IMMEDIATE, self.store.ctrl, lr, 0)
        self.t_opt.run()

    @ops.device_guard
    def train_step(self, lr: float, gather: bool = False):
        """One full optimisation step (eager launch of the recorded tapes)."""
        ops.step_begin(ops.IMMEDIATE, self.store.ctrl, lr, self.n_batches)
        if gather:
            self.t_gather.run()
        self.t_fwd.run()
        self.t_bwd.run()
        self.t_opt.run()
        self.bump_nbt()

    @ops.device_guard
    def capture(self, lr: float, gather: bool = True, warmup: bool = True):
        """Capture the whole step (cursor advance, gather, fwd, bwd, clip, Adam) into one hipGraph.
        ``warmup=True`` first runs one REAL step eagerly on a side stream (loads the code objects);
        pass ``warmup=False`` when an eager step has already been run (fit() does: its first step is eager)."""
        torch.cuda.synchronize()
        if warmup:
            s = ops.capture_stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._step_for_capture(lr, gather)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self.bump_nbt()
        g = ops.new_graph()
        with ops.graph_capture(g):
            self._step_for_capture(lr, gather)
        self.graph = g
        return g

    def _step_for_capture(self, lr, gather):
        rec = ops.IMMEDIATE
        ops.step_begin(rec, self.store.ctrl, lr, self.n_batches)
        if gather:
            self.t_gather.run()
        self.t_fwd.run()
        self.t_bwd.run()
        self.t_opt.run()

    @ops.device_guard
    def replay(self):
        self.graph.replay()
        self.bump_nbt()

    def n_launches(self):
        return len(self.t_gather) + len(self.t_fwd) + len(self.t_bwd) + len(self.t_opt) + 1

    def close(self):
        """Release every hipGraph of this plan NOW (idempotent).  Owners call it at a point where no capture is in progress
        and the plan's last replay has been waited for -- fit() when it returns, models when they drop a plan -- so that no
        graph is left to die wherever the last reference happens to go (ops.retire_graph: a graph dying during another
        capture terminates the process on ROCm)."""
        g, self.graph = self.graph, None
        ops.retire_graph(g)
        for name in ("_tape_graph", "_eval_graph"):
            d = self.__dict__.pop(name, None)
            if isinstance(d, dict):
                for v in d.values():
                    ops.retire_graph(v)
            else:
                ops.retire_graph(d)
        self.__dict__.pop("_tape_calls", None)
        self.__dict__.pop("_eval_calls", None)

    @ops.device_guard
    def eval_step(self, use_graph: bool = True):
        """Gather (from the index table the caller filled) + eval-mode forward of this plan.  The second call captures the
        two tapes into one hipGraph, later calls replay it: per-epoch validation runs ~25 launches per chunk."""
        if self.train:
            raise RuntimeError("eval_step is for eval plans")
        n = self.__dict__.get("_eval_calls", 0)
        self.__dict__["_eval_calls"] = n + 1
        g = self.__dict__.get("_eval_graph")
        if use_graph and g is None and n >= 1 and not ops.capturing():
            torch.cuda.synchronize()
            g = ops.new_graph()
            with ops.graph_capture(g):
                self.t_gather.run()
                self.t_fwd.run()
            self.__dict__["_eval_graph"] = g
        if use_graph and g is not None:
            g.replay()
        else:
            self.t_gather.run()
            self.t_fwd.run()

    def losses(self) -> Dict[str, float]:
        if getattr(self, "_fwd_pending", False) and getattr(self, "_losses_in_bwd", False):
            raise RuntimeError("this plan completes its MMD term and total loss in the backward tape: call backward() first, or build the "
                               "plan with forward_alone=True to read the losses between forward() and backward()")
        vals = self.loss_vec.detach().cpu().tolist()
        names = self.spec.loss_names()
        out = {n: vals[i] for i, n in enumerate(names)}
        out["total"] = vals[len(names)]
        return out


class PipelinedStep:
    """Double-buffered optimisation steps over one device-resident cohort.

    Two StepPlans share the ParamStore, the batch index table and the epoch accumulators but own separate
    batch buffers.  Step t computes from plan t%2; concurrently with its latency-bound head/backward chain
    (where the chip is nearly idle) the batch of step t+1 -- row gather, bf16 splits, X X^T Gram slabs -- is
    assembled into plan (t+1)%2 on a side stream, which under hipGraph capture is a parallel graph branch.
    The device cursor therefore runs ONE table row ahead of the step: the caller must have written the
    index table of the NEXT epoch before it launches the last step of an epoch (`epoch_end_next()`).

    This is the DataLoader-prefetch of the reference's loop (main.py:289-298, num_workers) moved onto the GPU."""

    def __init__(self, store: ParamStore, B: int, *, cohort, n_batches: int, seed: int = 0,
                 precision: Optional[str] = None, epoch_acc: bool = True, clip: bool = True, frozen: Tuple[str, ...] = (),
                 fuse_next_fwd: Optional[bool] = None, supplied_draws: bool = False):
        if fuse_next_fwd is None:            # FX_FUSE_NEXT_FWD=0: A/B switch (separate forward kernel, 28 B/param/step)
            fuse_next_fwd = os.environ.get("FX_FUSE_NEXT_FWD", "1") != "0"
        # supplied_draws: parity mode -- dropout masks / eps / priors are static buffers the caller fills before each step
        # (``pending.set_draws``) instead of in-kernel Philox draws; the schedule is otherwise the production one
        kw = dict(train=True, fused=True, supplied_draws=bool(supplied_draws), seed=seed, cohort=cohort, n_batches=n_batches,
                  epoch_acc=epoch_acc, precision=precision, clip=clip, frozen=frozen, fuse_next_fwd=fuse_next_fwd)
        a = StepPlan(store, B, **kw)
        self.plans = [a, StepPlan(store, B, share=a, **kw)]
        if a._next_fwd:
            a.link_next(self.plans[1])
            self.plans[1].link_next(a)
        self.store, self.n_batches, self.dev = store, int(n_batches), store.device
        self.idx, self.epoch_acc = a.idx, a.epoch_acc
        # FX_EARLY_GATHER=1 forks the batch assembly of step t+1 at the start of step t instead of after the losses.
        # Measured slower (1.313 vs 1.276 ms/step at cfg2): the gather / Gram kernels then compete with the forward chain.
        self.early_gather = bool(a._next_fwd) and os.environ.get("FX_EARLY_GATHER", "0") == "1"
        self.fork_at_mark = int(os.environ.get("FX_FORK_AT_MARK", "2"))       # A/B (see _issue); 0 = after the forward tape (round 2)
        # The assembly of the next batch depends on fx_step_begin only (the cursor), i.e. it starts beside the first chain
        # launch -- many workgroups, throughput-bound -- and is over before the one-workgroup-per-head launch, which its memory
        # traffic slowed from 32 to 43 us: cfg2 -5..9 us per step, cfg3 -60..90, cfg1 -10.  Not for the stacked-rows plans
        # (triplet): their assembly is 0.4 ms of X X^T products that would run against the wide forwards (5.78 vs 5.75 ms).
        # (only for plans whose assembly is the grouped one -- four launches without split-K scratch, the configurations the
        # full-size parity tests and the bench run; every other plan keeps the long-tested fork behind the forward tape)
        # NOT for the VAE family either (re-measured at the end of round 6, profiles/r06_chain.txt): with the dependency on fx_step_begin alone
        # the runtime places the assembly's launches BEHIND the backward chain's on their hardware queue (cfg3: it ran at 728-782 us of the step,
        # after the chain had finished at 752, and the clip coefficient waited for it); forked behind the forward tape it runs beside the first
        # data-gradient product, which hides it: 2.513 -> 2.479 ms per step
        vae = a.spec.model in ("supervised_vae", "CrossModalPred")
        dep_default = "1" if (a.passes == 1 and getattr(a, "assembly_grouped", False) and not vae) else "0"
        self.fork_dep_begin = os.environ.get("FX_FORK_DEP_BEGIN", dep_default) == "1"
        self.k = 0                       # plan holding the batch of the next step
        self.done = 0                    # steps issued since prime()
        self.graphs = [None, None]

    @ops.device_guard
    def prime(self):
        """Assemble table row 0 into plan 0 and point the cursor one row ahead of the step counter."""
        self.store.ctrl[ops.CTRL_CURSOR] = 0.0      # fx_step_begin advances it first: step t assembles row (t + 1) mod n_batches
        self.plans[0].t_gather.run()
        self.k, self.done = 0, 0
        self.refresh()

    @ops.device_guard
    def refresh(self):
        """(Re)compute the wide-forward partial sums of the pending batch with the CURRENT weights: at the start, and
        after any weight change made outside the pipeline (a partial-batch step, load_state, ...)."""
        self.plans[self.k].t_boot.run()

    def epoch_end_next(self) -> bool:
        """True when the NEXT step is the last of its epoch, i.e. its prefetch reads row 0 of the next epoch's
        table: write that table before issuing the step."""
        return (self.done + 1) % self.n_batches == 0

    def _issue(self, k, lr, timed=None):
        cur, nxt = self.plans[k], self.plans[1 - k]
        ops.step_begin(ops.IMMEDIATE, self.store.ctrl, lr, self.n_batches)
        main = torch.cuda.current_stream()
        used, point = [], []
        mode = self.fork_at_mark       # 0: after the forward tape; 1: at the plan's mark; 2 / 3: DEPEND on the mark, but issue after the
                                       # next one / two main-chain launches (the chain's own launches are queued first)
        if self.fork_dep_begin and mode >= 2:      # depend on fx_step_begin only (the cursor); still issued behind the chain's launches
            ev = torch.cuda.Event()
            ev.record(main)
            point.append(ev)

        def fork(name=None):
            if used:
                return
            if name is None:
                used.extend(nxt.t_gather.fork_from(main, after=point[0] if point else None))
            elif name == "fork_assembly" and mode == 1:
                used.extend(nxt.t_gather.fork_from(main))  # fork: batch assembly of step t+1 ...
            elif name == "fork_assembly" and 2 <= mode <= 4 and not point:
                ev = torch.cuda.Event()
                ev.record(main)
                point.append(ev)
            elif (name == "fork_issue_%d" % (mode - 1) or (mode == 4 and name == "fork_assembly")) and point:
                used.extend(nxt.t_gather.fork_from(main, after=point[0]))
            elif mode == 5 and name == "fork_issue_1":       # depend on the fusion layer's launch: the assembly starts under the
                ev = torch.cuda.Event()                      # one-workgroup-per-head kernel, not beside the 16-workgroup fusion launch
                ev.record(main)
                point.append(ev)
            elif mode == 5 and name == "fork_issue_2" and point:
                used.extend(nxt.t_gather.fork_from(main, after=point[0]))
        if self.early_gather:
            fork()
        cur.t_fwd.run(on_mark=fork)                       # (a plan may name its own fork point: StepPlan._mlp_tails_fwd)
        fork()
        cur.t_bwd.run()                                   # ... overlaps the head / backward chain of step t
        for st in used:
            main.wait_stream(st)                          # join before the HBM-saturating dW+Adam launches
        if timed is None:
            cur.t_opt.run()
        else:
            cur.t_opt.run_timed(*timed)

    @ops.device_guard
    def step(self, lr: float, timed=None):
        """One optimisation step, eager launch."""
        self._issue(self.k, lr, timed)
        self._advance()

    @ops.device_guard
    def capture(self, lr: float):
        """Capture both parities of the step into hipGraphs (no work is executed)."""
        torch.cuda.synchronize()
        for k in (0, 1):
            g = ops.new_graph()
            with ops.graph_capture(g):
                self._issue(k, lr)
            self.graphs[k] = g

    @ops.device_guard
    def replay(self):
        self.graphs[self.k].replay()
        self._advance()

    def close(self):
        """Release the captured step graphs of both parities and the plans' own graphs (see StepPlan.close)."""
        gs, self.graphs = self.graphs, [None, None]
        for g in gs:
            ops.retire_graph(g)
        del gs
        for p in self.plans:
            p.close()

    def _advance(self):
        self.plans[self.k].bump_nbt()
        self.k ^= 1
        self.done += 1

    @property
    def pending(self) -> StepPlan:
        """The plan the NEXT step computes from (its batch is already assembled)."""
        return self.plans[self.k]

    @property
    def last(self) -> StepPlan:
        """The plan the most recent step computed from."""
        return self.plans[self.k ^ 1]

    def losses(self):
        return self.last.losses()

    def n_launches(self):
        p = self.plans[0]
        return len(p.t_gather) + len(p.t_fwd) + len(p.t_bwd) + len(p.t_opt) + 1
