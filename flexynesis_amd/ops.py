"""Thin tensor-level wrappers over the libfxhip C ABI.

Every op is *emitted* to a ``Recorder``: ``ImmediateRecorder`` launches right away on torch's current
HIP stream; ``TapeRecorder`` stores the resolved (function, raw-pointer arguments) pairs so that a whole
training step can be re-issued with almost no host work -- or captured once into a hipGraph
(engine.py).  Tensors must be fp32 CUDA(=HIP) tensors with unit inner stride; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch

from . import _lib
from ._lib import lib, FxError

HIP_RUNTIME = _lib._check_runtime()        # (major, minor, patch) of the runtime the process runs on; warns once if unvalidated
ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
CTRL_FLOATS = 64
CTRL_STEP, CTRL_LR, CTRL_CLIP_COEF, CTRL_GNORM, CTRL_CURSOR, CTRL_STEP_HI = 0, 1, 4, 5, 8, 10


def _env_int(name: str, default: int) -> int:
    import os
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


# A/B switches of the wide kernels (benchmark experiments; every variant computes the same thing).  They are read HERE,
# once at import, and handed to the library as explicit arguments of its *_ex entry points -- libfxhip itself reads no
# environment variables.  FX_ADAM_XCD: 0 linear tile order, 1 auto (default), 2 always XCD-partitioned.
TUNE = {
    "fwd_splitk": _env_int("FX_SPLITK", 0), "fwd_wn": _env_int("FX_FWD_WN", 0), "fwd_no_mt": {0: 1, 2: 2, 3: 3, 4: 4}.get(_env_int("FX_FWD_MT", 1), 0),   # FX_FWD_MT (batches of more than 128 rows): 1 = stacked-rows kernel, X fragments straight into registers (default); 0 = one workgroup per M tile; 2 = the first stacked-rows kernel (operands staged through registers); 3 = the second (X by LDS-DMA)
    "fwd_nt": _env_int("FX_NT_FWD", 0), "adam_order": {0: 1, 1: 0, 2: 2}.get(_env_int("FX_ADAM_XCD", 1), 0),
    "adam_wn": _env_int("FX_ADAM_WN", 0), "adam_plain": int(_env_int("FX_NT_ADAM", 1) == 0),
    "fused_runs": _env_int("FX_FUSED_RUNS", 0),   # runs (= partial-sum slabs) per row block; 0 = the library's choice
    "fused_map": _env_int("FX_FUSED_MAP", 0),     # fx_linear_dw_adam_fwd_bf16x3 workgroup mapping: 0 auto, 1 plain, 2 XCD-grouped, 3 XCD-contiguous row blocks
    "fused_prio": _env_int("FX_FUSED_PRIO", 0),   # 1 = the two workgroups of a CU alternate s_setprio per tile (experiment)
    "mmd_tiled": _env_int("FX_MMD_TILED", 1),     # fx_mmd_rows_ex: 16-byte row loads + overwrite flag (0: fx_mmd_rows behind a zero-fill)
}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def device_guard(fn):
    """Method decorator: run with ``self.dev`` as the current HIP device.  Launch streams, the side-stream pool,
    events and graph capture are all keyed on torch's CURRENT device while buffers live on the device the caller
    asked for; without the guard ``fit(..., device='cuda:1')`` from a process whose current device is 0 would launch
    GPU-0 streams on GPU-1 pointers."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        dev = getattr(self, "dev", None)
        if dev is None:                      # constructors: the device is the ParamStore's (first argument)
            dev = (a[0] if a else k["store"]).device
        if dev.index is None or torch.cuda.current_device() == dev.index:
            return fn(self, *a, **k)
        with torch.cuda.device(dev):
            return fn(self, *a, **k)
    return wrapped


def _chk2d(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise FxError(f"{name}: expected an fp32 tensor on the GPU, got {t.dtype} on {t.device} "
                      "(flexynesis_amd has no CPU fallback)")
    if t.dim() != 2 or (t.stride(1) != 1 and t.shape[1] != 1):
        raise FxError(f"{name}: expected a 2-D row-major view, got shape {tuple(t.shape)} strides {t.stride()}")


def _ld(t: torch.Tensor) -> int:
    """Leading dimension (elements) of a 2-D row-major view; size-1 dims carry arbitrary strides in torch."""
    if t.dim() == 1:
        return 1
    return t.stride(0) if (t.shape[0] > 1 and t.stride(0) >= t.shape[1]) else max(t.shape[1], 1)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---- process-wide side-stream pool ---------------------------------------------------------------------------
# torch hands out torch.cuda.Stream() objects round-robin from 32 pre-created HIP streams per device.  Creating fresh
# Stream objects per tape / plan wraps that pool after a few model fits, and a "new" side stream then ALIASES another
# live one -- including the graph-capture stream itself, which crashed hipGraph replay (4th fit() in one process).
# All tapes therefore share one small set of streams that are checked to be pairwise distinct: group 0 = branches of a
# tape segment, group 1 = the detached prefetch fork of PipelinedStep, group 2 = graph capture.
_POOLS: dict = {}          # (device, thread id) -> {(group, slot): stream}
_GROUP = 8
_POOL_LOCK = __import__("threading").Lock()

# Host threads that run fits CONCURRENTLY on one GPU (trials.run_units(in_flight > 1)) mark themselves here.  Graph capture is a
# process-wide affair on ROCm -- a device synchronisation, a graph destructor or another capture on ANY thread while one stream
# captures terminates the process -- so such a thread must launch eagerly: fit() consults
# ``in_flight_thread()`` and downgrades ``use_graph`` there, StepPlan / PipelinedStep.capture refuse outright.
_TLS = __import__("threading").local()


def in_flight_thread() -> bool:
    return bool(getattr(_TLS, "in_flight", False))


class in_flight_scope:
    """``with ops.in_flight_scope(): ...`` in a worker thread of trials.run_units: no hipGraph capture on this thread, and the
    thread's side-stream pool is released when it leaves."""

    def __init__(self, stream=None):
        self.stream = stream            # the worker's own launch stream: no other thread's side-stream pool may hand it out

    def __enter__(self):
        _TLS.in_flight = True
        if self.stream is not None:
            with _POOL_LOCK:
                _WORKER_STREAMS.add(self.stream.cuda_stream)
        return self

    def __exit__(self, *exc):
        _TLS.in_flight = False
        if self.stream is not None:
            with _POOL_LOCK:
                _WORKER_STREAMS.discard(self.stream.cuda_stream)
        release_thread_streams()
        return False


_WORKER_STREAMS: set = set()


def release_thread_streams():
    """Drop the calling thread's side-stream pools (all devices): a finished worker thread must not leave its entries behind."""
    import threading
    me = threading.get_ident()
    with _POOL_LOCK:
        for key in [k for k in _POOLS if k[1] == me]:
            del _POOLS[key]


def side_streams(n: int, group: int = 0) -> List["torch.cuda.Stream"]:
    import threading
    # one pool per (device, host thread): trials in flight on several threads (trials.run_units(in_flight=...)) must not share
    # side streams -- a shared stream would order one trial's batch assembly behind the other's.  Streams are created on demand,
    # slot by slot: torch hands out 32 HIP streams per device round-robin, and every slot of every live thread must be a different
    # one (main thread, autograd's backward thread, worker threads: a handful of slots each)
    dev = (torch.cuda.current_device(), threading.get_ident())
    if n > _GROUP:
        raise FxError(f"at most {_GROUP} parallel branches per tape segment (got {n})")
    with _POOL_LOCK:
        pool = _POOLS.setdefault(dev, {})
        out = []
        for i in range(n):
            st = pool.get((group, i))
            tries = 0
            while st is None:
                cand = torch.cuda.Stream()
                tries += 1
                taken = {p.cuda_stream for key, pl in _POOLS.items() if key[0] == dev[0] for p in pl.values()}
                taken |= {torch.cuda.current_stream().cuda_stream, torch.cuda.default_stream().cuda_stream} | _WORKER_STREAMS
                if cand.cuda_stream not in taken:
                    st = pool[(group, i)] = cand
                elif tries > 96:
                    raise FxError("could not obtain enough distinct HIP streams from torch's stream pool "
                                  f"({len(taken)} in use by this process's threads on device {dev[0]})")
            out.append(st)
        return out


def capture_stream() -> "torch.cuda.Stream":
    return side_streams(1, group=2)[0]


# ---- hipGraph lifetime ------------------------------------------------------------------------------------------
# On ROCm torch's ~CUDAGraph calls hipDeviceSynchronize (ATen/hip/HIPGraph.cpp: hipGraphExecDestroy defers its frees to the
# next sync point, so torch forces one).  A device synchronisation is "operation not permitted when stream is capturing":
# a graph object that dies while ANOTHER graph is being captured terminates the process (scripts/graph_lifetime_probe.py
# case a: rc -6; the intermittent crashes of round 2 were graphs of dead models released by the cyclic collector in the
# middle of a later capture).  Three rules follow: (1) the collector is off while capturing; (2) owners release their
# graphs explicitly at a known point (StepPlan.close / PipelinedStep.close, called by fit()) instead of whenever the last
# reference happens to go; (3) a release requested DURING a capture is parked and executed when the capture has ended.
_CAPTURING = [0]
_GRAVEYARD: List[object] = []


def capturing() -> bool:
    return _CAPTURING[0] > 0


def retire_graph(g):
    """Give up a CUDAGraph (or any object owning one).  Outside a capture the caller's reference was the last one and the
    object dies on return; during a capture it is parked until the capture is over."""
    if g is not None and _CAPTURING[0] > 0:
        _GRAVEYARD.append(g)
    elif isinstance(g, FxGraph):
        g.close()                      # (the library's own graphs: destroyed here, deterministically)


class FxGraph:
    """A hipGraphExec captured and replayed through libfxhip's own entry points (fx_graph_begin / _end / _launch / _destroy:
    csrc/fx_runtime.hip) -- the default backend of ``graph_capture``.  Same surface as the part of torch.cuda.CUDAGraph the engine
    uses: ``replay()`` on the current stream; released explicitly (``close``, via ``retire_graph``) or with the object."""

    def __init__(self):
        self._exec = None
        self.n_nodes = 0
        self._streams = {}            # streams this graph has been launched on (cuda_stream -> torch stream)

    def replay(self):
        st = torch.cuda.current_stream()
        rc = lib.fx_graph_launch(self._exec, st.cuda_stream)
        if rc != 0:
            raise FxError(f"fx_graph_launch failed (rc={rc}): {_lib.last_error()}")
        self._streams[st.cuda_stream] = st

    def close(self):
        """Destroy the executable graph.  hipGraphExecDestroy does not wait for launches of it that are still in flight (torch's
        ~CUDAGraph synchronised the whole DEVICE for that reason -- the call that is illegal during another capture): the streams this
        graph was launched on are waited for first -- stream synchronisations, legal beside a relaxed-mode capture on another stream."""
        x, self._exec = self._exec, None
        if x is not None:
            try:
                for st in self._streams.values():
                    st.synchronize()
            except Exception:
                pass
            self._streams = {}
            lib.fx_graph_destroy(x)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


GRAPH_BACKEND = os.environ.get("FX_GRAPH_BACKEND", "fx")       # "fx": the library's own capture; "torch": torch.cuda.CUDAGraph (A/B)


def new_graph():
    """The graph object to hand to ``graph_capture`` (backend: FX_GRAPH_BACKEND, default the library's own)."""
    return FxGraph() if GRAPH_BACKEND != "torch" else torch.cuda.CUDAGraph()


class graph_capture:
    """``with ops.graph_capture(g): ...`` captures what the block launches into ``g`` (from ``new_graph()``).

    FxGraph: the block runs on the capture stream of the shared pool, ordered behind the caller's stream, between fx_graph_begin
    (relaxed mode: the engine owns every buffer the captured launches touch and allocates nothing inside) and fx_graph_end.
    torch.cuda.CUDAGraph (FX_GRAPH_BACKEND=torch): ``torch.cuda.graph(g, stream=capture_stream())``.
    Either way Python's cyclic collector is off for the duration and graph releases are deferred to the end (see above)."""

    def __init__(self, g):
        self._g = g
        self._fx = isinstance(g, FxGraph)
        self._ctx = None if self._fx else torch.cuda.graph(g, stream=capture_stream())
        self._gc = False

    def __enter__(self):
        import gc
        if in_flight_thread():
            raise FxError("hipGraph capture on a thread that runs fits concurrently with others (trials.run_units(in_flight > 1)): "
                          "a capture is process-wide on ROCm; launch eagerly there (fit(use_graph=False))")
        if self._fx:
            self._cs = capture_stream()
            self._caller = torch.cuda.current_stream()
            self._cs.wait_stream(self._caller)
            self._sctx = torch.cuda.stream(self._cs)
            self._sctx.__enter__()
            rc = lib.fx_graph_begin(self._cs.cuda_stream, 2)
            if rc != 0:
                self._sctx.__exit__(None, None, None)
                raise FxError(f"fx_graph_begin failed (rc={rc}): {_lib.last_error()}")
        else:
            self._ctx.__enter__()              # (torch collects garbage and empties the cache before capture_begin)
        self._gc = gc.isenabled()
        gc.disable()
        _CAPTURING[0] += 1
        return self

    def __exit__(self, *exc):
        import gc
        try:
            if not self._fx:
                return self._ctx.__exit__(*exc)
            try:
                if exc and exc[0] is not None:
                    lib.fx_graph_abort(self._cs.cuda_stream)
                else:
                    x, n = C.c_void_p(), C.c_int(0)
                    rc = lib.fx_graph_end(self._cs.cuda_stream, C.byref(x), C.byref(n))
                    if rc != 0 or not x.value:
                        raise FxError(f"fx_graph_end failed (rc={rc}): {_lib.last_error()}")
                    self._g._exec, self._g.n_nodes = x.value, int(n.value)
            finally:
                self._sctx.__exit__(None, None, None)
                self._caller.wait_stream(self._cs)
            return False
        finally:
            _CAPTURING[0] -= 1
            if _CAPTURING[0] == 0:
                _GRAVEYARD.clear()         # parked graphs die now: the device synchronisation of their destructor is legal again
            if self._gc:
                gc.enable()


# ---- memory leases (csrc/fx_runtime.hip) ------------------------------------------------------------------------------------
class LeaseRegistry:
    """Ranges of long-lived device memory (the partition arena's pools, rated arrays of the placement pool) handed out as torch
    tensors whose STORAGE reports back when its last view has gone.  ``wrap`` imports a DLPack tensor made by fx_lease_wrap: when
    torch drops the storage -- the ParamStore, every nn.Parameter .data, every state_dict tensor that viewed it are gone -- the
    library's deleter queues the lease id; ``drain`` (called by the owners before they allocate) runs ``on_release(events)`` for
    every queued id, with the HIP events recorded when the owning store let go (``add_events``): the next taker's stream waits for
    them.  A range is therefore never reused while anything can still read or write it (a model's parameters survived
    their ParamStore as views of memory the next trial was given)."""

    def __init__(self):
        import threading
        self._q = lib.fx_lease_queue_create()
        self._lock = threading.Lock()
        self._next = 1
        self._live: dict = {}           # id -> [on_release, events, unsynced]
        C.pythonapi.PyCapsule_New.restype = C.py_object
        C.pythonapi.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]

    _NAME = b"dltensor"

    def wrap(self, ptr: int, n: int, device, on_release):
        """(flat fp32 tensor of n elements at ptr on ``device``, lease id)."""
        device = torch.device(device)
        with self._lock:
            lid = self._next
            self._next += 1
            self._live[lid] = [on_release, [], False]
        m = lib.fx_lease_wrap(self._q, ptr, n, 10 if device.type == "cuda" else 1, device.index or 0, lid)
        if not m:
            with self._lock:
                self._live.pop(lid, None)
            raise FxError(f"fx_lease_wrap failed: {_lib.last_error()}")
        try:
            cap = C.pythonapi.PyCapsule_New(m, self._NAME, None)
            t = torch.utils.dlpack.from_dlpack(cap)
        except Exception:
            with self._lock:
                self._live.pop(lid, None)
            lib.fx_lease_discard(m)
            self.drain()
            raise
        return t, lid

    def add_events(self, lid: int, events, unsynced: bool = False):
        with self._lock:
            e = self._live.get(lid)
            if e is not None:
                e[1].extend(events)
                e[2] = e[2] or unsynced

    def drain(self) -> int:
        ids = (C.c_longlong * 64)()
        done = 0
        while True:
            n = lib.fx_lease_drain(self._q, ids, 64)
            if n <= 0:
                return done
            for i in range(n):
                with self._lock:
                    e = self._live.pop(int(ids[i]), None)
                if e is not None and e[0] is not None:
                    e[0](e[1], e[2])
                done += 1

    def outstanding(self) -> int:
        return int(lib.fx_lease_outstanding(self._q))


LEASES = LeaseRegistry()


class ImmediateRecorder:
    """Launch each op immediately on the current stream."""
    products = 3

    def emit(self, name: str, *args):
        rc = getattr(lib, name)(*args, _stream())
        if rc != 0:
            raise FxError(f"{name} failed (rc={rc}): {_lib.last_error()}")


class _Branches:
    """Context helper returned by TapeRecorder.parallel(n)."""

    def __init__(self, tape, n):
        self.tape, self.n = tape, n

    def __enter__(self):
        self.tape.segments.append([[] for _ in range(self.n)])
        self.tape._cur = 0
        return self

    def __exit__(self, *exc):
        self.tape.segments.append([[]])
        self.tape._cur = 0
        return False

    def branch(self, i):
        self.tape._cur = i
        return self


class TapeRecorder:
    """Record ops; ``run()`` re-issues them.  A tape is a list of SEGMENTS; a segment holds one or more
    independent BRANCHES (``with tape.parallel(n) as par: par.branch(i); ...``).  Branches of a segment are issued
    on separate HIP streams (fork/join with stream waits), so under hipGraph capture they become parallel
    graph branches: the narrow per-modality chains overlap each other and the other modality's HBM-bound
    wide-layer kernel instead of queueing behind it."""

    def __init__(self, products: int = 3):
        self.segments: List[List[List[tuple]]] = [[[]]]
        self._cur = 0
        self.keepalive: List[object] = []
        # 3: the wide kernels recorded on this tape contract in split bf16 (hi hi + hi lo + lo hi); 1: plain bf16 -- their `lo`
        # operands are passed as NULL, which the library reads as "one product" (include/fxhip.h)
        self.products = int(products)

    @property
    def calls(self):
        return [c for seg in self.segments for br in seg for c in br if c[0] is not None]

    def emit(self, name: str, *args):
        self.segments[-1][self._cur].append((getattr(lib, name), name, args))

    def parallel(self, n: int) -> _Branches:
        return _Branches(self, n)

    def record_event(self, ev):
        """Mark a point in the current branch that a later-issued branch of the same segment can wait for."""
        self.segments[-1][self._cur].append((None, "__record__", (ev,)))

    def wait_event(self, ev):
        self.segments[-1][self._cur].append((None, "__wait__", (ev,)))

    def keep(self, *objs):
        self.keepalive.extend(objs)

    def mark(self, name: str):
        """A named point in the tape: ``run(on_mark=f)`` calls ``f(name)`` when the launches before it have been issued
        (PipelinedStep forks the next batch's assembly there).  No-op for plain ``run()``."""
        self.segments[-1][self._cur].append((None, "__mark__", (name,)))

    @staticmethod
    def _issue(branch, hook=None, on_mark=None):
        s = _stream()
        for fn, name, args in branch:
            if fn is None:      # cross-branch ordering inside a parallel segment (HIP events; graph edges under capture)
                if name == "__mark__":
                    if on_mark is not None:
                        on_mark(args[0])
                    continue
                ev = args[0]
                if name == "__record__":
                    ev.record(torch.cuda.current_stream())
                else:
                    torch.cuda.current_stream().wait_event(ev)
                continue
            if hook is not None and hook(name):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*args, s)
                e1.record()
                hook.sink.append((name, e0, e1))
            else:
                rc = fn(*args, s)
            if rc != 0:
                raise FxError(f"{name} failed (rc={rc}): {_lib.last_error()}")

    def _run(self, hook=None, on_mark=None):
        for seg in self.segments:
            live = [br for br in seg if br]
            if not live:
                continue
            if len(live) == 1:
                self._issue(live[0], hook, on_mark)
                continue
            main = torch.cuda.current_stream()
            side = side_streams(len(live) - 1, group=0)
            for br, st in zip(live[1:], side):
                st.wait_stream(main)          # fork point: recorded BEFORE any branch work is queued on main
            self._issue(live[0], hook)
            for br, st in zip(live[1:], side):
                with torch.cuda.stream(st):
                    self._issue(br, hook)
            for st in side:
                main.wait_stream(st)          # join

    def run(self, on_mark=None):
        self._run(None, on_mark)

    def fork_from(self, main, pool=None, after=None):
        """Issue the whole tape off the critical path: every branch goes to its own stream (group 1 of the shared
        pool), each forked DIRECTLY from ``main``; returns the streams the caller must join (``main.wait_stream``).
        (A fork made from an already-forked stream -- i.e. ``run()`` under ``torch.cuda.stream(side)`` --
        crashes hipGraph capture in ROCm 7.2's hipStreamEndCapture, so nested forks are never generated.)"""
        live = [[br for br in seg if br] for seg in self.segments]
        live = [seg for seg in live if seg]
        if len(live) != 1:
            raise FxError("fork_from: only tapes with a single (parallel) segment can run as a detached fork")
        used = side_streams(len(live[0]), group=1)
        for st in used:
            if after is not None:
                st.wait_event(after)      # depend on an EARLIER point of ``main`` than the one the fork is issued at
            else:
                st.wait_stream(main)
        for br, st in zip(live[0], used):
            with torch.cuda.stream(st):
                self._issue(br)
        return used

    def run_timed(self, names, sink):
        """Like run(), but brackets every launch whose entry-point name is in ``names`` with a pair of
        HIP events recorded on the launch stream; (name, start, end) tuples are appended to ``sink``."""
        class _Hook:
            def __call__(self, n):
                return n in names
        h = _Hook()
        h.sink = sink
        self._run(h)

    def __len__(self):
        return len(self.calls)


IMMEDIATE = ImmediateRecorder()


class Workspace:
    """Split-K scratch shared by all GEMMs of one stream-ordered plan."""

    def __init__(self, device, nbytes: int = 0):
        self.device = device
        self.buf = torch.empty(max(nbytes, 16) // 4 + 4, dtype=torch.float32, device=device)

    def reserve(self, nbytes: int):
        if nbytes > self.buf.numel() * 4:
            # tapes recorded earlier hold raw pointers into the previous buffer: keep it alive
            self._retired = getattr(self, "_retired", []) + [self.buf]
            self.buf = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=self.device)

    @property
    def nbytes(self):
        return self.buf.numel() * 4


def gemm_ws_bytes(M, N, K) -> int:
    return int(lib.fx_gemm_workspace_bytes(M, N, K))


def gemm(rec, layout: int, Cm: torch.Tensor, A: torch.Tensor, Bm: torch.Tensor, bias: Optional[torch.Tensor],
         ws: Optional[Workspace], accumulate: bool = False):
    """layout NT: C[M,N] = A[M,K] B[N,K]^T ; NN: C = A[M,K] B[K,N] ; TN: C = A[K,M]^T B[K,N]."""
    for t, n in ((Cm, "C"), (A, "A"), (Bm, "B")):
        _chk2d(t, "gemm." + n)
    M, N = Cm.shape
    if layout == GEMM_NT:
        K = A.shape[1]
        ok = A.shape == (M, K) and Bm.shape == (N, K)
    elif layout == GEMM_NN:
        K = A.shape[1]
        ok = A.shape == (M, K) and Bm.shape == (K, N)
    else:
        K = A.shape[0]
        ok = A.shape == (K, M) and Bm.shape == (K, N)
    if not ok:
        raise FxError(f"gemm layout {layout}: shape mismatch C{tuple(Cm.shape)} A{tuple(A.shape)} B{tuple(Bm.shape)}")
    need = gemm_ws_bytes(M, N, K)
    wptr, wbytes = None, 0
    if need and ws is not None:
        ws.reserve(need)
        wptr, wbytes = ws.buf.data_ptr(), ws.nbytes
    rec.emit("fx_gemm_f32", layout, Cm.data_ptr(), A.data_ptr(), Bm.data_ptr(), _ptr(bias), M, N, K,
             _ld(A), _ld(Bm), _ld(Cm), int(accumulate), wptr, wbytes)


def linear_fwd(rec, y, x, W, b, ws):
    gemm(rec, GEMM_NT, y, x, W, b, ws)


def linear_bwd_x(rec, dx, dy, W, ws, accumulate=False):
    gemm(rec, GEMM_NN, dx, dy, W, None, ws, accumulate)


def linear_bwd_w(rec, dW, dy, x, ws, accumulate=False):
    gemm(rec, GEMM_TN, dW, dy, x, None, ws, accumulate)


SMALL_LINEAR_MAX = 4096


def small_linear_ok(x, W) -> bool:
    return max(x.shape[0], W.shape[0], W.shape[1]) <= SMALL_LINEAR_MAX and W.is_contiguous()


def small_linear_fwd(rec, y, x, W, b):
    """y = x W^T + b for the small layers on the critical chain (one latency-lean launch; include/fxhip.h)."""
    for t, n in ((y, "y"), (x, "x"), (W, "W")):
        _chk2d(t, "small_linear_fwd." + n)
    R, K = x.shape
    O = W.shape[0]
    if W.shape != (O, K) or y.shape != (R, O) or not W.is_contiguous():
        raise FxError("small_linear_fwd: shape mismatch")
    rec.emit("fx_small_linear_fwd", y.data_ptr(), x.data_ptr(), W.data_ptr(), _ptr(b), R, O, K, _ld(x), _ld(y))


def small_linear_bwd(rec, dx, gW, gb, dy, x, W, dx_accumulate=False):
    """dx (+)= dy W (dx may be None), gW = dy^T x, gb = colsum(dy) (gb may be None) in ONE launch."""
    for t, n in ((dy, "dy"), (x, "x"), (W, "W"), (gW, "gW")):
        _chk2d(t, "small_linear_bwd." + n)
    R, K = x.shape
    O = W.shape[0]
    if W.shape != (O, K) or dy.shape != (R, O) or gW.shape != W.shape or not (W.is_contiguous() and gW.is_contiguous()):
        raise FxError("small_linear_bwd: shape mismatch")
    if dx is not None and dx.shape != (R, K):
        raise FxError("small_linear_bwd: dx shape mismatch")
    rec.emit("fx_small_linear_bwd", _ptr(dx), gW.data_ptr(), _ptr(gb), dy.data_ptr(), x.data_ptr(), W.data_ptr(), R, O, K,
             _ld(x), _ld(dy), _ld(dx) if dx is not None else K, int(bool(dx_accumulate)))


def small_linear_bwd_group(rec, jobs):
    """Several small layers' backward in ONE launch.  jobs = [dict(dx=, gW=, gb=, dy=, dy_mul=, x=, W=)]: as small_linear_bwd, the
    upstream gradient being dy * dy_mul elementwise when dy_mul is given."""
    n = len(jobs)
    arr = (_lib.SmallLinearJob * n)()
    for a, j in zip(arr, jobs):
        dy, x, W, gW, dx, mulv = j["dy"], j["x"], j["W"], j["gW"], j.get("dx"), j.get("dy_mul")
        for t, nm in ((dy, "dy"), (x, "x"), (W, "W"), (gW, "gW")):
            _chk2d(t, "small_linear_bwd_group." + nm)
        R, K = x.shape
        O = W.shape[0]
        if W.shape != (O, K) or dy.shape != (R, O) or gW.shape != W.shape or not (W.is_contiguous() and gW.is_contiguous()):
            raise FxError("small_linear_bwd_group: shape mismatch")
        if (dx is not None and dx.shape != (R, K)) or (mulv is not None and mulv.shape != dy.shape):
            raise FxError("small_linear_bwd_group: dx / dy_mul shape mismatch")
        a.dx, a.gW, a.gb, a.dy, a.dy_mul, a.x, a.W = _ptr(dx), gW.data_ptr(), _ptr(j.get("gb")), dy.data_ptr(), _ptr(mulv), x.data_ptr(), W.data_ptr()
        a.R, a.O, a.K, a.dx_accumulate = R, O, K, 0
        a.ldx, a.lddy, a.ldmul, a.lddx = _ld(x), _ld(dy), _ld(mulv) if mulv is not None else O, _ld(dx) if dx is not None else K
    if hasattr(rec, "keep"):
        rec.keep(arr)
    rec.emit("fx_small_linear_bwd_group", C.addressof(arr), n)


def linear_dw_adam(rec, W, m, v, dy, x, ctrl):
    for t, n in ((W, "W"), (m, "m"), (v, "v"), (dy, "dY"), (x, "X")):
        _chk2d(t, "linear_dw_adam." + n)
    B, n_out = dy.shape
    k_in = x.shape[1]
    if W.shape != (n_out, k_in) or x.shape[0] != B or m.shape != W.shape or v.shape != W.shape:
        raise FxError("linear_dw_adam: shape mismatch")
    if not (_ld(m) == _ld(W) == _ld(v)):
        raise FxError("linear_dw_adam: W/m/v must share a leading dimension")
    rec.emit("fx_linear_dw_adam_f32", W.data_ptr(), m.data_ptr(), v.data_ptr(), dy.data_ptr(), x.data_ptr(),
             B, n_out, k_in, _ld(dy), _ld(x), _ld(W), ctrl.data_ptr())


def pad32(n: int) -> int:
    return (int(n) + 31) // 32 * 32


def new_split(rows: int, cols: int, device):
    """(hi, lo) bf16 buffers [rows, pad32(cols)] (row-major, contraction dim contiguous) for fx_split_bf16_t."""
    return (torch.zeros(rows, pad32(cols), dtype=torch.bfloat16, device=device),
            torch.zeros(rows, pad32(cols), dtype=torch.bfloat16, device=device))


def pad128(n: int) -> int:
    return (n + 127) // 128 * 128


def new_split_kb(rows: int, cols: int, device):
    """(hi, lo) K-BLOCKED bf16 buffers [pad32(cols)/32, pad128(rows), 32] (include/fxhip.h): the X operand of the
    wide forward.  Padding rows/columns are zero (allocated zero, never written)."""
    shape = (pad32(cols) // 32, pad128(rows), 32)
    return (torch.zeros(shape, dtype=torch.bfloat16, device=device), torch.zeros(shape, dtype=torch.bfloat16, device=device))


def unblock(t: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """K-blocked [Kb, Rp, 32] -> row-major [rows, cols] view-copy (tests / debugging)."""
    return t.permute(1, 0, 2).reshape(t.shape[1], -1)[:rows, :cols]


def _chk_kb(hi, lo, R, K, what):
    want = (pad32(K) // 32, pad128(R), 32)
    if tuple(hi.shape) != want or tuple(lo.shape) != want or hi.dtype != torch.bfloat16 or not hi.is_contiguous():
        raise FxError(f"{what}: hi/lo must be contiguous K-blocked bf16 {want} (ops.new_split_kb), got {tuple(hi.shape)}")


def _lo(rec, t):
    """The `lo` operand of a wide kernel: its address, or NULL on a plain-bf16 tape (TapeRecorder.products == 1)."""
    return None if getattr(rec, "products", 3) == 1 else t.data_ptr()


def split_bf16(rec, hi, lo, x):
    """hi/lo K-blocked (new_split_kb(R, C)) <- x [R, C]  (x ~= hi + lo)."""
    _chk2d(x, "split_bf16.x")
    R, Cc = x.shape
    _chk_kb(hi, lo, R, Cc, "split_bf16")
    rec.emit("fx_split_bf16", hi.data_ptr(), lo.data_ptr(), x.data_ptr(), R, Cc, _ld(x), hi.shape[1])


def split_bf16_t(rec, hiT, loT, x):
    """hiT/loT [C, pad32(R)] <- x[R, C] transposed (contraction dimension becomes contiguous)."""
    _chk2d(x, "split_bf16_t.x")
    R, Cc = x.shape
    if hiT.shape != (Cc, pad32(R)) or loT.shape != hiT.shape or hiT.dtype != torch.bfloat16:
        raise FxError(f"split_bf16_t: hiT/loT must be bf16 [{Cc},{pad32(R)}]")
    rec.emit("fx_split_bf16_t", hiT.data_ptr(), loT.data_ptr(), x.data_ptr(), R, Cc, _ld(x), _ld(hiT))


def linear_fwd_bf16x3(rec, y, xhi, xlo, W, b, ws):
    """y[M,N] = x W^T + b with x pre-split (xhi/xlo [M, pad32(K)]) and W [N,K] fp32 split in-kernel."""
    _chk2d(y, "linear_fwd_bf16x3.y")
    _chk2d(W, "linear_fwd_bf16x3.W")
    M, N = y.shape
    K = W.shape[1]
    if W.shape[0] != N:
        raise FxError("linear_fwd_bf16x3: shape mismatch")
    _chk_kb(xhi, xlo, M, K, "linear_fwd_bf16x3")
    need = int(lib.fx_linear_fwd_bf16x3_workspace_bytes(M, N, K))
    t = TUNE
    if t["fwd_splitk"] or t["fwd_wn"] or t["fwd_no_mt"] or t["fwd_nt"]:
        need = max(need, max(t["fwd_splitk"], 1) * M * N * 4)
        ws.reserve(need)
        rec.emit("fx_linear_fwd_bf16x3_ex", y.data_ptr(), xhi.data_ptr(), _lo(rec, xlo), W.data_ptr(), _ptr(b), M, N, K,
                 xhi.shape[1], _ld(W), _ld(y), ws.buf.data_ptr(), ws.nbytes, t["fwd_splitk"], t["fwd_wn"], t["fwd_no_mt"], t["fwd_nt"])
        return
    ws.reserve(need)
    rec.emit("fx_linear_fwd_bf16x3", y.data_ptr(), xhi.data_ptr(), _lo(rec, xlo), W.data_ptr(), _ptr(b), M, N, K,
             xhi.shape[1], _ld(W), _ld(y), ws.buf.data_ptr(), ws.nbytes)


def linear_bwd_x_bf16x3(rec, dx, dyhi, dylo, W, ws: Workspace):
    """dx[M,N] = dy[M,K] W[K,N] for a wide W (dy K-blocked split, new_split_kb(M, K))."""
    _chk2d(dx, "linear_bwd_x_bf16x3.dx")
    _chk2d(W, "linear_bwd_x_bf16x3.W")
    M, N = dx.shape
    K = W.shape[0]
    if W.shape[1] != N:
        raise FxError("linear_bwd_x_bf16x3: shape mismatch")
    _chk_kb(dyhi, dylo, M, K, "linear_bwd_x_bf16x3")
    ws.reserve(int(lib.fx_linear_fwd_bf16x3_workspace_bytes(M, N, K)))
    rec.emit("fx_linear_bwd_x_bf16x3", dx.data_ptr(), dyhi.data_ptr(), _lo(rec, dylo), W.data_ptr(), M, N, K, dyhi.shape[1],
             _ld(W), _ld(dx), ws.buf.data_ptr(), ws.nbytes)


def linear_dw_adam_bf16x3(rec, W, m, v, dyT_hi, dyT_lo, xT_hi, xT_lo, ctrl, tile_order=None):
    """W[N,K] <- Adam(clip * dY^T X) with dY^T [N, Bp] and X^T [K, Bp] pre-split.  tile_order: None = the process-wide
    choice (TUNE), 0 auto, 1 linear, 2 XCD-partitioned (bit-identical results)."""
    for t, n in ((W, "W"), (m, "m"), (v, "v")):
        _chk2d(t, "linear_dw_adam_bf16x3." + n)
    N, K = W.shape
    Bp = dyT_hi.shape[1]
    if dyT_hi.shape != (N, Bp) or xT_hi.shape != (K, Bp) or dyT_lo.shape != dyT_hi.shape or xT_lo.shape != xT_hi.shape:
        raise FxError("linear_dw_adam_bf16x3: shape mismatch")
    if not (_ld(m) == _ld(W) == _ld(v)):
        raise FxError("linear_dw_adam_bf16x3: W/m/v must share a leading dimension")
    order = TUNE["adam_order"] if tile_order is None else int(tile_order)
    if order or TUNE["adam_wn"] or TUNE["adam_plain"]:
        rec.emit("fx_linear_dw_adam_bf16x3_ex", W.data_ptr(), m.data_ptr(), v.data_ptr(), dyT_hi.data_ptr(), _lo(rec, dyT_lo),
                 xT_hi.data_ptr(), _lo(rec, xT_lo), Bp, N, K, _ld(dyT_hi), _ld(xT_hi), _ld(W), ctrl.data_ptr(), order,
                 TUNE["adam_wn"], TUNE["adam_plain"])
        return
    rec.emit("fx_linear_dw_adam_bf16x3", W.data_ptr(), m.data_ptr(), v.data_ptr(), dyT_hi.data_ptr(), _lo(rec, dyT_lo),
             xT_hi.data_ptr(), _lo(rec, xT_lo), Bp, N, K, _ld(dyT_hi), _ld(xT_hi), _ld(W), ctrl.data_ptr())


def _fused_flags(nt=True, mapping=0) -> int:
    return int(bool(nt)) | ((int(mapping) & 3) << 1) | ((TUNE["fused_runs"] & 0xFF) << 8) | ((TUNE["fused_prio"] & 3) << 17)


def dw_adam_fwd_slabs(n_out: int, k_in: int, batch_padded: int = 128, mapping: Optional[int] = None) -> int:
    """Partial-sum slabs linear_dw_adam_fwd_bf16x3 writes for a weight [n_out, k_in] (mapping None = the configured default)."""
    m = TUNE["fused_map"] if mapping is None else mapping
    return int(lib.fx_linear_dw_adam_fwd_bf16x3_slabs_ex(int(n_out), int(k_in), int(batch_padded), _fused_flags(True, m)))


def linear_dw_adam_fwd_bf16x3(rec, W, m, v, dyT_hi, dyT_lo, xT_hi, xT_lo, ctrl, xn_hi, xn_lo, next_rows, y_slabs, nt=True,
                              mapping=0):
    """W[N,K] <- Adam(clip * dY^T X) as linear_dw_adam_bf16x3, and y_slabs[s] = partial sums of x_next W_new^T
    (x_next: K-blocked split of the NEXT batch, new_split_kb(next_rows, K)); reduce with reduce_slabs."""
    for t, n in ((W, "W"), (m, "m"), (v, "v")):
        _chk2d(t, "linear_dw_adam_fwd_bf16x3." + n)
    N, K = W.shape
    Bp = dyT_hi.shape[1]
    if dyT_hi.shape != (N, Bp) or xT_hi.shape != (K, Bp) or dyT_lo.shape != dyT_hi.shape or xT_lo.shape != xT_hi.shape:
        raise FxError("linear_dw_adam_fwd_bf16x3: shape mismatch")
    if not (_ld(m) == _ld(W) == _ld(v)):
        raise FxError("linear_dw_adam_fwd_bf16x3: W/m/v must share a leading dimension")
    _chk_kb(xn_hi, xn_lo, next_rows, K, "linear_dw_adam_fwd_bf16x3")
    S = dw_adam_fwd_slabs(N, K, Bp, mapping)
    if y_slabs.dtype != torch.float32 or not y_slabs.is_contiguous() or y_slabs.numel() < S * next_rows * N:
        raise FxError(f"linear_dw_adam_fwd_bf16x3: y_slabs must hold {S} x {next_rows} x {N} fp32")
    rec.emit("fx_linear_dw_adam_fwd_bf16x3", W.data_ptr(), m.data_ptr(), v.data_ptr(), dyT_hi.data_ptr(), _lo(rec, dyT_lo),
             xT_hi.data_ptr(), _lo(rec, xT_lo), Bp, N, K, _ld(dyT_hi), _ld(xT_hi), _ld(W), ctrl.data_ptr(), xn_hi.data_ptr(),
             _lo(rec, xn_lo), xn_hi.shape[1], int(next_rows), y_slabs.data_ptr(), y_slabs.numel() * 4,
             _fused_flags(nt, mapping))


def reduce_slabs(rec, y, slabs, bias, n_slabs):
    """y[M,N] = sum of the first n_slabs slabs [n_slabs][M][N] (+ bias), fixed order."""
    _chk2d(y, "reduce_slabs.y")
    M, N = y.shape
    if slabs.numel() < n_slabs * M * N:
        raise FxError("reduce_slabs: slab buffer too small")
    rec.emit("fx_reduce_slabs", y.data_ptr(), slabs.data_ptr(), _ptr(bias), M, N, _ld(y), int(n_slabs), M * N)


def placement_probe_us(W, m=None, v=None, launches: int = 3) -> float:
    """Microseconds per pass of the fused dW + Adam kernel's W / m / v traffic pattern over these three arrays (contents unchanged):
    a rating of WHERE they landed in physical memory (include/fxhip.h: fx_placement_probe).  Synchronises the current stream."""
    for t in (W, m, v):
        if t is not None:
            _chk2d(t, "placement_probe")
    n_out, k_in = W.shape
    args = (W.data_ptr(), _ptr(m), _ptr(v), n_out, k_in, _ld(W))
    for t in (m, v):          # (None: W on its own)
        if t is not None and (_ld(t) != _ld(W) or t.shape != W.shape):
            raise FxError("placement_probe: W / m / v must share shape and leading dimension")
    IMMEDIATE.emit("fx_placement_probe", *args)
    best = float("inf")
    for _ in range(max(1, launches)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        IMMEDIATE.emit("fx_placement_probe", *args)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def placement_probe_oop_us(dst, src, launches: int = 3) -> float:
    """Microseconds per pass of the OUT-OF-PLACE twin of the fused kernel's traffic pattern: ``src`` = (W, m, v) are read, the same
    values are written to ``dst`` = (W', m', v') (m, v and their destinations may be None together).  Rates a (source partitions,
    destination partitions) layout for fx_linear_dw_adam_fwd_bf16x3_oop (include/fxhip.h: fx_placement_probe_oop)."""
    W, m, v = src
    Wd, md, vd = dst
    for t in (W, m, v, Wd, md, vd):
        if t is not None:
            _chk2d(t, "placement_probe_oop")
            if t.shape != W.shape or _ld(t) != _ld(W):
                raise FxError("placement_probe_oop: all arrays must share shape and leading dimension")
    n_out, k_in = W.shape
    args = (Wd.data_ptr(), _ptr(md), _ptr(vd), W.data_ptr(), _ptr(m), _ptr(v), n_out, k_in, _ld(W))
    IMMEDIATE.emit("fx_placement_probe_oop", *args)
    best = float("inf")
    for _ in range(max(1, launches)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        IMMEDIATE.emit("fx_placement_probe_oop", *args)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def reduce_slabs_par(rec, y, slabs, bias, n_slabs):
    """reduce_slabs for many slabs of a small output: partial sums over contiguous ranges of slabs, combined in range order."""
    _chk2d(y, "reduce_slabs_par.y")
    M, N = y.shape
    if slabs.numel() < n_slabs * M * N:
        raise FxError("reduce_slabs_par: slab buffer too small")
    rec.emit("fx_reduce_slabs_par", y.data_ptr(), slabs.data_ptr(), _ptr(bias), M, N, _ld(y), int(n_slabs), M * N)


def gemm_slabs(rec, layout, slabs, A, Bm, M, N):
    """Contraction whose split-K partial sums stay in ``slabs`` [splitk, M, N]; returns splitk."""
    _chk2d(A, "gemm_slabs.A")
    _chk2d(Bm, "gemm_slabs.B")
    K = A.shape[0] if layout == GEMM_TN else A.shape[1]
    s = int(lib.fx_gemm_splitk(M, N, K))
    if slabs.numel() < s * M * N:
        raise FxError("gemm_slabs: slab buffer too small")
    rec.emit("fx_gemm_f32_slabs", layout, slabs.data_ptr(), A.data_ptr(), Bm.data_ptr(), M, N, K, _ld(A), _ld(Bm))
    return s


def fwd_slabs_splitk(M: int, N: int, K: int, wave_cols: int = 0) -> int:
    """Slabs linear_fwd_bf16x3_slabs writes for this shape and output tile (wave_cols 8 = the 128 x 256 tile)."""
    return int(lib.fx_linear_fwd_bf16x3_splitk_ex(int(M), int(N), int(K), int(wave_cols)))


def linear_fwd_bf16x3_slabs(rec, slabs, xhi, xlo, W, M, wave_cols: int = 0):
    """Wide forward contraction, partial sums left in ``slabs`` [splitk, M, N]; returns splitk."""
    _chk2d(W, "linear_fwd_bf16x3_slabs.W")
    N, K = W.shape
    M = int(M)
    s = fwd_slabs_splitk(M, N, K, wave_cols)
    _chk_kb(xhi, xlo, M, K, "linear_fwd_bf16x3_slabs")
    if slabs.numel() < s * M * N:
        raise FxError("linear_fwd_bf16x3_slabs: bad buffer shapes")
    rec.emit("fx_linear_fwd_bf16x3_slabs_ex", slabs.data_ptr(), slabs.numel() * 4, xhi.data_ptr(), _lo(rec, xlo),
             W.data_ptr(), M, N, K, xhi.shape[1], _ld(W), int(wave_cols))
    return s


def bn_act_fwd_slabs(rec, out, x_out, slabs, nslabs, slab_stride, lin_bias, gamma, beta, rmean, rvar, save_mean,
                     save_invstd, pre_act, post_act, train, drop_p=0.0, mask=None, seed=0, offset=0, ctrl=None):
    """BatchNorm block fed by split-K slabs [nslabs, B, C]; also materialises x = sum(slabs)+bias in x_out."""
    _chk2d(out, "bn_slabs.out")
    B, Cc = out.shape
    rec.emit("fx_bn_act_fwd_slabs", out.data_ptr(), _ptr(x_out), slabs.data_ptr(), int(nslabs), int(slab_stride),
             _ptr(lin_bias),
             gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), _ptr(save_mean), _ptr(save_invstd),
             _ptr(mask), B, Cc, _ld(x_out) if x_out is not None else Cc, _ld(out), pre_act, post_act, int(train),
             float(drop_p), int(seed), int(offset), _ptr(ctrl))


def gram_hadamard_blocks(n) -> int:
    return int(lib.fx_gram_hadamard_blocks(n))


def gram_hadamard(rec, slots, slabs_x, nx, slabs_d, nd, n):
    rec.emit("fx_gram_hadamard", slots.data_ptr(), slabs_x.data_ptr(), int(nx), slabs_d.data_ptr(), int(nd), int(n))


def gather_split(rec, x, hi, lo, hiT, loT, src, idx, ctrl_cursor=None, cursor_stride=0, n_rows=None):
    """x[r,:] = src[idx[r],:] plus its bf16 splits: K-blocked hi/lo (new_split_kb(R, F), the forward operand) and
    transposed hiT/loT [F, pad32(R)] (the weight-gradient operand)."""
    R = int(n_rows) if n_rows is not None else x.shape[0]
    Fc = src.shape[1]
    _chk_kb(hi, lo, R, Fc, "gather_split")
    if hiT.shape != (Fc, pad32(R)) or idx.dtype != torch.int64:
        raise FxError("gather_split: bad buffer shapes")
    rec.emit("fx_gather_split", _ptr(x), hi.data_ptr(), lo.data_ptr(), hiT.data_ptr(), loT.data_ptr(), src.data_ptr(),
             idx.data_ptr(), R, Fc, _ld(src), _ld(x) if x is not None else Fc, hi.shape[1], _ld(hiT), _ptr(ctrl_cursor),
             int(cursor_stride))


def block_bwd_blocks(C: int) -> int:
    return int(lib.fx_block_bwd_blocks(int(C)))


def block_bwd(rec, ups, x, out, gamma, save_mean, save_invstd, dgamma, dbeta, dbias, pre_act, post_act, drop_p,
              dy=None, dyT=None, gram_x=None, slots=None, accumulate=False, dy_kb=None):
    """Backward of "wide Linear -> BatchNorm block -> small Linears" in one launch (include/fxhip.h: fx_block_bwd).
    ``ups`` = [(dE [B, L], W [L, C], gW [L, C], gb [L] | None), ...] (1 or 2 entries).  ``dy_kb`` = (hi, lo, row0): dY also in the
    K-blocked split layout (new_split_kb(total rows, C)), this pass's rows starting at row0 (fx_block_bwd_ex)."""
    _chk2d(x, "block_bwd.x")
    _chk2d(out, "block_bwd.out")
    B, Cc = x.shape
    n = len(ups)
    PA, LA, IA = C.c_void_p * n, C.c_long * n, C.c_int * n
    for (dE, W, gW, gb) in ups:
        _chk2d(dE, "block_bwd.dE")
        _chk2d(W, "block_bwd.W")
        if dE.shape[0] != B or W.shape != (dE.shape[1], Cc) or gW.shape != W.shape or not W.is_contiguous() or not gW.is_contiguous():
            raise FxError("block_bwd: upstream shapes must be dE [B, L], W / gW [L, C] contiguous")
    arrs = (PA(*[u[0].data_ptr() for u in ups]), LA(*[_ld(u[0]) for u in ups]), PA(*[u[1].data_ptr() for u in ups]),
            PA(*[u[2].data_ptr() for u in ups]), PA(*[_ptr(u[3]) for u in ups]), IA(*[u[0].shape[1] for u in ups]))
    if hasattr(rec, "keep"):
        rec.keep(arrs)
    common = [*[C.addressof(a_) for a_ in arrs], n, x.data_ptr(), out.data_ptr(), gamma.data_ptr(),
              save_mean.data_ptr(), save_invstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _ptr(dbias), _ptr(dy),
              _ptr(dyT[0]) if dyT else None, _ptr(dyT[1]) if dyT else None, _ld(dyT[0]) if dyT else 0, _ptr(gram_x),
              _ptr(slots), B, Cc, _ld(x), _ld(out), int(pre_act), int(post_act), float(drop_p), int(bool(accumulate))]
    if dy_kb is None:
        rec.emit("fx_block_bwd", *common)
    else:
        khi, klo, row0 = dy_kb
        if (khi.dim() != 3 or khi.shape[0] != pad32(Cc) // 32 or khi.shape[2] != 32 or klo.shape != khi.shape or khi.dtype != torch.bfloat16
                or not khi.is_contiguous() or not klo.is_contiguous()):
            raise FxError(f"block_bwd: dy_kb must be K-blocked bf16 [{pad32(Cc) // 32}, rows, 32] (ops.new_split_kb), got {tuple(khi.shape)}")
        rec.emit("fx_block_bwd_ex", *common, khi.data_ptr(), klo.data_ptr(), int(khi.shape[1]), int(row0))
    return arrs


def block_bwd_desc(ups, x, out, gamma, save_mean, save_invstd, dgamma, dbeta, dbias, dy=None, dyT=None, gram_x=None, slots=None,
                   accumulate=False):
    """One fx_block_bwd_desc (arguments as block_bwd)."""
    _chk2d(x, "block_bwd.x")
    _chk2d(out, "block_bwd.out")
    B, Cc = x.shape
    d = _lib.BlockBwdDesc()
    if not 1 <= len(ups) <= 2:
        raise FxError("block_bwd: 1 or 2 upstream Linears")
    for k, (dE, W, gW, gb) in enumerate(ups):
        _chk2d(dE, "block_bwd.dE")
        _chk2d(W, "block_bwd.W")
        if dE.shape[0] != B or W.shape != (dE.shape[1], Cc) or gW.shape != W.shape or not W.is_contiguous() or not gW.is_contiguous():
            raise FxError("block_bwd: upstream shapes must be dE [B, L], W / gW [L, C] contiguous")
        d.dE[k], d.ldE[k], d.W[k], d.gW[k], d.gb[k], d.L[k] = dE.data_ptr(), _ld(dE), W.data_ptr(), gW.data_ptr(), _ptr(gb), dE.shape[1]
    d.n_up = len(ups)
    d.x, d.out, d.gamma, d.save_mean, d.save_invstd = x.data_ptr(), out.data_ptr(), gamma.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr()
    d.dgamma, d.dbeta, d.dbias, d.dy = dgamma.data_ptr(), dbeta.data_ptr(), _ptr(dbias), _ptr(dy)
    d.dyT_hi, d.dyT_lo, d.ldt = (_ptr(dyT[0]), _ptr(dyT[1]), _ld(dyT[0])) if dyT else (None, None, 0)
    d.gram_x, d.slots, d.C, d.ldx, d.ldo, d.accumulate = _ptr(gram_x), _ptr(slots), Cc, _ld(x), _ld(out), int(bool(accumulate))
    return d


def block_bwd_group(rec, descs, B, pre_act, post_act, drop_p):
    """fx_block_bwd of several independent encoder tails (one per modality) in one launch."""
    arr = (_lib.BlockBwdDesc * len(descs))(*descs)
    if hasattr(rec, "keep"):
        rec.keep(arr)
    rec.emit("fx_block_bwd_group", C.addressof(arr), len(descs), int(B), int(pre_act), int(post_act), float(drop_p))
    return arr


HEADS_MAX = dict(B=128, L=128, hidden=32, n_out=32, heads=8)     # limits of fx_heads_fwd / fx_heads_bwd


def head_desc(**kw):
    """One fx_head_desc; tensors are converted to device pointers (None -> NULL)."""
    d = _lib.HeadDesc()
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if torch.is_tensor(v) else v)
    return d


def _head_array(rec, descs):
    arr = (_lib.HeadDesc * len(descs))(*descs)
    if hasattr(rec, "keep"):
        rec.keep(arr)                   # the tape re-reads the host array at every launch
    return arr


def heads_fwd(rec, descs, x, B, L, train, drop_p, ctrl=None):
    """All supervisor heads forward in one launch (one workgroup per head)."""
    _chk2d(x, "heads_fwd.x")
    arr = _head_array(rec, descs)
    rec.emit("fx_heads_fwd", C.addressof(arr), len(descs), x.data_ptr(), _ld(x), int(B), int(L), int(train), float(drop_p),
             _ptr(ctrl))
    return arr


def gather_split_group(rec, items, idx, ctrl_cursor, cursor_stride, n_rows):
    """gather_split of several cohort layers in one launch.  items = [(x, hi, lo, hiT, loT, src)]."""
    R = int(n_rows)
    arr = (_lib.GatherSplitDesc * len(items))()
    for d, (x, hi, lo, hiT, loT, src) in zip(arr, items):
        Fc = src.shape[1]
        _chk_kb(hi, lo, R, Fc, "gather_split_group")
        if hiT.shape != (Fc, pad32(R)) or idx.dtype != torch.int64:
            raise FxError("gather_split_group: bad buffer shapes")
        d.x, d.hi, d.lo, d.hiT, d.loT, d.src = _ptr(x), hi.data_ptr(), lo.data_ptr(), hiT.data_ptr(), loT.data_ptr(), src.data_ptr()
        d.n_cols, d.ld_src, d.ldx, d.ldo, d.ldt = Fc, _ld(src), _ld(x) if x is not None else Fc, hi.shape[1], _ld(hiT)
    if hasattr(rec, "keep"):
        rec.keep(arr)
    rec.emit("fx_gather_split_group", C.addressof(arr), len(items), idx.data_ptr(), R, _ptr(ctrl_cursor), int(cursor_stride))


def gram_kb_slices(k_in: int) -> int:
    return int(lib.fx_gram_kb_slices(int(k_in)))


def gram_kb_group(rec, splits, slabs, k_ins, R):
    """X X^T partial sums (one slab per K slice) of several modalities in one launch; splits = [(hi, lo)] K-blocked."""
    n = len(splits)
    hi = (C.c_void_p * n)(*[s_[0].data_ptr() for s_ in splits])
    lo = (C.c_void_p * n)(*[s_[1].data_ptr() for s_ in splits])
    sl = (C.c_void_p * n)(*[t.data_ptr() for t in slabs])
    ks = (C.c_int * n)(*[int(k) for k in k_ins])
    for t, k in zip(slabs, k_ins):
        if t.numel() < gram_kb_slices(k) * R * R:
            raise FxError("gram_kb_group: slab buffer too small")
    if hasattr(rec, "keep"):
        rec.keep(hi, lo, sl, ks)
    rec.emit("fx_gram_kb_group", C.addressof(hi), C.addressof(lo), C.addressof(sl), C.addressof(ks), n, int(R))


def reduce_group(rec, jobs):
    """Ordered slab sums of several jobs in one launch.  jobs = [(y, slabs, n_slabs, bias | None)]; y.numel() % 4 == 0."""
    n = len(jobs)
    ys = (C.c_void_p * n)(*[j[0].data_ptr() for j in jobs])
    sl = (C.c_void_p * n)(*[j[1].data_ptr() for j in jobs])
    bs = (C.c_void_p * n)(*[_ptr(j[3]) for j in jobs])
    ln = (C.c_long * n)(*[int(j[0].numel()) for j in jobs])
    ns = (C.c_int * n)(*[int(j[2]) for j in jobs])
    bn = (C.c_int * n)(*[int(j[3].numel()) if j[3] is not None else 4 for j in jobs])
    for (y, slabs, k, b) in jobs:
        if not y.is_contiguous() or slabs.numel() < k * y.numel():
            raise FxError("reduce_group: y must be contiguous and the slab buffer hold n_slabs copies of it")
    if hasattr(rec, "keep"):
        rec.keep(ys, sl, bs, ln, ns, bn)
    rec.emit("fx_reduce_group", C.addressof(ys), C.addressof(sl), C.addressof(bs), C.addressof(ln), C.addressof(ns), C.addressof(bn), n)


def enc_tail_blocks(H: int) -> int:
    return int(lib.fx_enc_tail_blocks(int(H)))


def enc_tail_desc(*, slabs, n_slabs, slab_stride, lin_bias, x, out, gamma, beta, running_mean, running_var, save_mean, save_invstd,
                  mask, ups, seed, offset):
    """One fx_enc_tail_desc.  ups = [(W_k [L_k, H], part_k [blocks, B, L_k])] (one or two following Linears)."""
    d = _lib.EncTailDesc()
    H = x.shape[1]
    for t, n in ((x, "x"), (out, "out")):
        if not (t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda):
            raise FxError(f"enc_tail_desc.{n}: expected a contiguous fp32 GPU tensor")
    d.slabs, d.slab_stride, d.lin_bias = _ptr(slabs), int(slab_stride), _ptr(lin_bias)
    d.x, d.out, d.gamma, d.beta = x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    d.running_mean, d.running_var = running_mean.data_ptr(), running_var.data_ptr()
    d.save_mean, d.save_invstd, d.mask = _ptr(save_mean), _ptr(save_invstd), _ptr(mask)
    if not 1 <= len(ups) <= 2:
        raise FxError("enc_tail_desc: one or two following Linears")
    for k, (W, part) in enumerate(ups):
        if W.shape[1] != H or not W.is_contiguous() or part.numel() < enc_tail_blocks(H) * x.shape[0] * W.shape[0]:
            raise FxError("enc_tail_desc: following Linear / partial-product buffer mismatch")
        d.W[k], d.part[k], d.L[k] = W.data_ptr(), part.data_ptr(), int(W.shape[0])
    d.seed, d.offset, d.n_slabs, d.H, d.n_up = int(seed), int(offset), int(n_slabs), int(H), len(ups)
    return d


def enc_tail_fwd(rec, descs, B, pre_act, post_act, train, drop_p, ctrl=None):
    """Encoder tails of all modalities forward in one launch (slab sum + bias, BatchNorm block, partial products of the
    following small Linears)."""
    arr = (_lib.EncTailDesc * len(descs))(*descs)
    if hasattr(rec, "keep"):
        rec.keep(arr)
    rec.emit("fx_enc_tail_fwd", C.addressof(arr), len(descs), int(B), int(pre_act), int(post_act), int(bool(train)), float(drop_p),
             _ptr(ctrl))
    return arr


def _fusion_widths(parts, width):
    """Logical width of every part [blocks, B, pitch]: ``width`` (one for all; the pitch is then that rounded up to 4) or the pitch."""
    wd = [int(width) if width is not None else int(p.shape[-1]) for p, _ in parts]
    for (p, _), w_ in zip(parts, wd):
        if int(p.shape[-1]) != (w_ + 3) // 4 * 4:
            raise FxError(f"fusion_fwd: a part of width {w_} must have a row pitch of {(w_ + 3) // 4 * 4} (got {int(p.shape[-1])})")
    return wd


def fusion_fwd(rec, emb, ecat, parts, biases, W=None, b=None, width=None):
    """ecat = concatenated ordered sums of ``parts`` [(part [blocks, B, pitch], blocks)] (+ biases), emb = ecat W^T + b.  ``width``: the
    number of columns of a part that are real (the latent size; pitch = that rounded up to 4), default the pitch."""
    n = len(parts)
    B = ecat.shape[0] if ecat is not None else emb.shape[0]
    pp = (C.c_void_p * n)(*[p.data_ptr() for p, _ in parts])
    nb = (C.c_int * n)(*[int(k) for _, k in parts])
    bb = (C.c_void_p * n)(*[_ptr(x) for x in biases])
    wds = _fusion_widths(parts, width)
    wd = (C.c_int * n)(*wds)
    if hasattr(rec, "keep"):
        rec.keep(pp, nb, bb, wd)
    if W is not None and (not W.is_contiguous() or W.shape[1] != sum(wds)):
        raise FxError("fusion_fwd: fusion weight must be contiguous [L, sum of widths]")
    rec.emit("fx_fusion_fwd", _ptr(emb), _ld(emb) if emb is not None else 0, _ptr(ecat), _ld(ecat) if ecat is not None else 0,
             C.addressof(pp), C.addressof(nb), C.addressof(bb), C.addressof(wd), n, _ptr(W), _ptr(b), int(B),
             int(W.shape[0]) if W is not None else 0)


def fusion_fwd_pair(rec, embs, ecats, parts2, biases2, Ws, bs, width=None):
    """Two fusion_fwd calls over the same rows in one launch: embs / ecats / Ws / bs are pairs, parts2 / biases2 pairs of the per-layer
    lists (same widths in both)."""
    n = len(parts2[0])
    B = ecats[0].shape[0]
    wd0 = _fusion_widths(parts2[0], width)
    if len(parts2[1]) != n or _fusion_widths(parts2[1], width) != wd0:
        raise FxError("fusion_fwd_pair: both layers take the same number and widths of parts")
    for W in Ws:
        if not W.is_contiguous() or W.shape[1] != sum(wd0) or W.shape[0] != Ws[0].shape[0]:
            raise FxError("fusion_fwd_pair: fusion weights must be contiguous [L, sum of widths]")
    em = (C.c_void_p * 2)(*[e.data_ptr() for e in embs])
    le = (C.c_long * 2)(*[_ld(e) for e in embs])
    ec = (C.c_void_p * 2)(*[e.data_ptr() for e in ecats])
    lc = (C.c_long * 2)(*[_ld(e) for e in ecats])
    pp = (C.c_void_p * (2 * n))(*[p.data_ptr() for ps in parts2 for p, _ in ps])
    nb = (C.c_int * (2 * n))(*[int(k) for ps in parts2 for _, k in ps])
    bb = (C.c_void_p * (2 * n))(*[_ptr(x) for bsl in biases2 for x in bsl])
    wd = (C.c_int * n)(*wd0)
    ww = (C.c_void_p * 2)(*[W.data_ptr() for W in Ws])
    wb = (C.c_void_p * 2)(*[_ptr(b) for b in bs])
    if hasattr(rec, "keep"):
        rec.keep(em, le, ec, lc, pp, nb, bb, wd, ww, wb)
    rec.emit("fx_fusion_fwd_pair", C.addressof(em), C.addressof(le), C.addressof(ec), C.addressof(lc), C.addressof(pp), C.addressof(nb),
             C.addressof(bb), C.addressof(wd), n, C.addressof(ww), C.addressof(wb), int(B), int(Ws[0].shape[0]))


def heads_bwd_scratch(n_heads: int, B: int, L: int, device) -> torch.Tensor:
    """Zero-filled scratch for fx_heads_bwd's per-head dx workgroups (shares + arrival counter) and fx_heads_step (the same + one
    shadow block per head for the weight-gradient role's copies of the saved tensors)."""
    return torch.zeros(int(lib.fx_heads_step_scratch_floats(int(n_heads), int(B), int(L))), dtype=torch.float32, device=device)


def heads_bwd(rec, descs, x, dx, B, L, drop_p, dx_accumulate=False, scratch=None):
    """All supervisor heads backward + the summed embedding gradient in one launch."""
    _chk2d(x, "heads_bwd.x")
    arr = _head_array(rec, descs)
    if scratch is not None and scratch.numel() < len(descs) * int(B) * int(L) + 1:
        raise FxError("heads_bwd: scratch too small")
    rec.emit("fx_heads_bwd", C.addressof(arr), len(descs), x.data_ptr(), _ld(x), _ptr(dx), _ld(dx) if dx is not None else 0,
             int(bool(dx_accumulate)), int(B), int(L), float(drop_p), _ptr(scratch))
    return arr


LOSS_MSE, LOSS_CE, LOSS_COX = 0, 1, 2


def heads_step(rec, descs, kinds, labels, durations, logvars, losses, x, dx, B, L, drop_p, ctrl, scratch, term_losses, term_logvars,
               term_dlogvars, weighted, total_out, epoch_acc, dx_accumulate=False):
    """Forward + loss + backward of all supervisor heads, summed embedding gradient and the total loss in one launch."""
    _chk2d(x, "heads_step.x")
    arr = _head_array(rec, descs)
    n = len(descs)
    if scratch is None or scratch.numel() < int(lib.fx_heads_step_scratch_floats(n, int(B), int(L))):
        raise FxError("heads_step: needs heads_bwd_scratch(n_heads, B, L)")

    def parr(ts):
        return (C.c_void_p * max(len(ts), 1))(*[_ptr(t) for t in ts])
    k = (C.c_int * n)(*[int(v) for v in kinds])
    lab, dur, lv, ls = parr(labels), parr(durations), parr(logvars), parr(losses)
    tl, tv, td = parr(term_losses), parr(term_logvars if weighted else []), parr(term_dlogvars if weighted else [])
    if hasattr(rec, "keep"):
        rec.keep(k, lab, dur, lv, ls, tl, tv, td)
    rec.emit("fx_heads_step", C.addressof(arr), n, C.addressof(k), C.addressof(lab), C.addressof(dur), C.addressof(lv), C.addressof(ls),
             x.data_ptr(), _ld(x), _ptr(dx), _ld(dx) if dx is not None else 0, int(bool(dx_accumulate)), int(B), int(L), float(drop_p),
             _ptr(ctrl), _ptr(scratch), len(term_losses), int(bool(weighted)), C.addressof(tl),
             C.addressof(tv) if weighted else None, C.addressof(td) if weighted else None, total_out.data_ptr(), _ptr(epoch_acc))
    return arr


def colsum(rec, out, x):
    _chk2d(x, "colsum.x")
    rec.emit("fx_colsum", out.data_ptr(), x.data_ptr(), x.shape[0], x.shape[1], _ld(x))


def bn_act_fwd(rec, out, x, gamma, beta, rmean, rvar, save_mean, save_invstd, pre_act, post_act, train,
               drop_p=0.0, mask=None, mask_out=None, seed=0, offset=0, ctrl=None):
    _chk2d(out, "bn.out")
    _chk2d(x, "bn.x")
    B, Cc = x.shape
    if mask is not None and (mask.shape != x.shape or not mask.is_contiguous()):
        raise FxError("bn_act_fwd: mask must be a contiguous [B,C] tensor")
    rec.emit("fx_bn_act_fwd", out.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(),
             rvar.data_ptr(), _ptr(save_mean), _ptr(save_invstd), _ptr(mask), _ptr(mask_out), B, Cc,
             _ld(x), _ld(out), pre_act, post_act, int(train), float(drop_p), int(seed), int(offset),
             _ptr(ctrl))


def bn_act_bwd(rec, dx, dgamma, dbeta, dbias, dout, x, out, gamma, save_mean, save_invstd, pre_act, post_act,
               drop_p=0.0, accumulate=False):
    for t, n in ((dx, "dx"), (dout, "dout"), (x, "x")):
        _chk2d(t, "bn_bwd." + n)
    B, Cc = x.shape
    rec.emit("fx_bn_act_bwd", dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _ptr(dbias), dout.data_ptr(),
             x.data_ptr(), _ptr(out), gamma.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(), B, Cc,
             _ld(x), _ld(out) if out is not None else 0, _ld(dout), _ld(dx), pre_act, post_act,
             float(drop_p), int(accumulate))


def bn_eval_bwd(rec, dx, dout, x, out, gamma, running_var, pre_act, post_act):
    """Input gradient of a BatchNorm block in eval mode (running statistics, no dropout); dx may alias dout."""
    for t, n in ((dx, "dx"), (dout, "dout")):
        _chk2d(t, "bn_eval_bwd." + n)
    B, Cc = dout.shape
    rec.emit("fx_bn_eval_bwd", dx.data_ptr(), dout.data_ptr(), _ptr(x), _ptr(out), gamma.data_ptr(), running_var.data_ptr(), B, Cc,
             _ld(x) if x is not None else 0, _ld(out) if out is not None else 0, _ld(dout), _ld(dx), pre_act, post_act)


def gather_rows(rec, dst, src, idx, ctrl_cursor=None, cursor_stride=0):
    """dst[r,:] = src[idx[r],:]; 1-D src (labels) is treated as [N,1]."""
    if src.dim() == 1:
        src2, dst2 = src.unsqueeze(1), dst.unsqueeze(1) if dst.dim() == 1 else dst
    else:
        src2, dst2 = src, dst
    if idx.dtype != torch.int64 or not idx.is_cuda:
        raise FxError("gather_rows: idx must be an int64 GPU tensor")
    n_rows = dst2.shape[0]
    rec.emit("fx_gather_rows", dst2.data_ptr(), src2.data_ptr(), idx.data_ptr(), n_rows, src2.shape[1],
             _ld(src2), _ld(dst2), _ptr(ctrl_cursor), int(cursor_stride))


def mse_masked(rec, loss_out, dyhat, yhat, y, logvar=None, extra_scale=1.0):
    rec.emit("fx_mse_masked", loss_out.data_ptr(), dyhat.data_ptr(), yhat.data_ptr(), y.data_ptr(), yhat.shape[0],
             _ld(yhat), _ld(dyhat), _ptr(logvar), float(extra_scale))


def ce_masked(rec, loss_out, dlogits, logits, y, logvar=None, extra_scale=1.0):
    rec.emit("fx_ce_masked", loss_out.data_ptr(), dlogits.data_ptr(), logits.data_ptr(), y.data_ptr(),
             logits.shape[0], logits.shape[1], _ld(logits), _ld(dlogits), _ptr(logvar), float(extra_scale))


def cox_ph(rec, loss_out, dout, out, durations, events, logvar=None, extra_scale=1.0):
    rec.emit("fx_cox_ph", loss_out.data_ptr(), dout.data_ptr(), out.data_ptr(), durations.data_ptr(),
             events.data_ptr(), out.shape[0], _ld(out), _ld(dout), _ptr(logvar), float(extra_scale))


def triplet(rec, loss_out, da, dp, dn, a, p, n, margin=1.0, logvar=None, extra_scale=1.0):
    for t in (da, dp, dn, a, p, n):
        if _ld(t) != _ld(a):
            raise FxError("triplet: all operands must share a leading dimension")
    rec.emit("fx_triplet", loss_out.data_ptr(), da.data_ptr(), dp.data_ptr(), dn.data_ptr(), a.data_ptr(),
             p.data_ptr(), n.data_ptr(), a.shape[0], a.shape[1], _ld(a), float(margin), _ptr(logvar),
             float(extra_scale))


def mmd_rows(rec, row_sums, dz, prior, z, logvar=None, extra_scale=1.0, tiled=None, overwrite=False):
    """MMD(prior, z) row sums and (dz given) the gradient with respect to z, added to dz -- or stored, with ``overwrite``.
    tiled: fx_mmd_rows_ex (16-byte row loads, bit-identical results, the default); TUNE["mmd_tiled"] / FX_MMD_TILED=0: the first entry."""
    if not prior.is_contiguous():
        raise FxError("mmd_rows: prior must be contiguous")
    if tiled is None:
        tiled = TUNE["mmd_tiled"]
    if tiled:
        rec.emit("fx_mmd_rows_ex", row_sums.data_ptr(), _ptr(dz), prior.data_ptr(), z.data_ptr(), prior.shape[0], z.shape[0],
                 z.shape[1], _ld(z), _ptr(logvar), float(extra_scale), int(bool(overwrite)))
        return
    if overwrite and dz is not None:
        fill(rec, dz, 0.0)
    rec.emit("fx_mmd_rows", row_sums.data_ptr(), _ptr(dz), prior.data_ptr(), z.data_ptr(), prior.shape[0], z.shape[0],
             z.shape[1], _ld(z), _ptr(logvar), float(extra_scale))


def recon_sigmoid(rec, partial, dlogits, xhat_out, logits, x, logvar=None, extra_scale=1.0):
    if not (logits.is_contiguous() and x.is_contiguous()):
        raise FxError("recon_sigmoid: logits and x must be contiguous")
    rec.emit("fx_recon_sigmoid", partial.data_ptr(), _ptr(dlogits), _ptr(xhat_out), logits.data_ptr(), x.data_ptr(),
             logits.numel(), _ptr(logvar), float(extra_scale))


def recon_sigmoid_slabs_blocks(B, F) -> int:
    return int(lib.fx_recon_sigmoid_slabs_blocks(int(B), int(F)))


def recon_sigmoid_slabs(rec, partial, dlogits, split, slabs, nslabs, bias, x, logvar=None, extra_scale=1.0):
    """The reconstruction term as the epilogue of FC_output's forward: slabs [nslabs, B * F] -> partial sums of (x_hat - x)^2,
    dlogits [B, F] and their K-blocked bf16 split (``split`` = new_split_kb(B, F) or None)."""
    _chk2d(x, "recon_sigmoid_slabs.x")
    B, F = x.shape
    if not x.is_contiguous() or (dlogits is not None and (dlogits.shape != x.shape or not dlogits.is_contiguous())):
        raise FxError("recon_sigmoid_slabs: x and dlogits must be contiguous [B, F]")
    if slabs.numel() < int(nslabs) * B * F or partial.numel() < recon_sigmoid_slabs_blocks(B, F):
        raise FxError("recon_sigmoid_slabs: slab / partial buffers too small")
    if split is not None:
        _chk_kb(split[0], split[1], B, F, "recon_sigmoid_slabs")
    rec.emit("fx_recon_sigmoid_slabs", partial.data_ptr(), _ptr(dlogits), _ptr(split[0]) if split is not None else None,
             _ptr(split[1]) if split is not None else None, slabs.data_ptr(), int(nslabs), B * F, _ptr(bias), x.data_ptr(), B, F,
             split[0].shape[1] if split is not None else 0, _ptr(logvar), float(extra_scale))


def mmd_finalize(rec, loss_acc, row_sums, P, B, recon_partial, n_partial, n_recon, extra_scale, accumulate):
    rec.emit("fx_mmd_finalize", loss_acc.data_ptr(), row_sums.data_ptr(), P, B, _ptr(recon_partial), n_partial,
             float(n_recon), float(extra_scale), int(accumulate))


def total_loss(rec, total_out, losses, logvars, dlogvars, weighted, epoch_acc=None):
    n = len(losses)
    arr = C.c_void_p * n
    la = arr(*[t.data_ptr() for t in losses])
    lv = arr(*[t.data_ptr() for t in logvars]) if weighted else None
    dl = arr(*[(t.data_ptr() if t is not None else None) for t in dlogvars]) if weighted else None
    if hasattr(rec, "keep"):
        rec.keep(la, lv, dl)
    rec.emit("fx_total_loss", total_out.data_ptr(), n, int(weighted), C.cast(la, C.c_void_p),
             C.cast(lv, C.c_void_p) if lv is not None else None, C.cast(dl, C.c_void_p) if dl is not None else None,
             _ptr(epoch_acc))


def fill(rec, y, value=0.0):
    rec.emit("fx_fill", y.data_ptr(), y.numel(), float(value))


def scale_by(rec, x, scale):
    """x *= scale[0] (device scalar; no-op when it is 1)."""
    rec.emit("fx_scale_by", x.data_ptr(), x.numel(), scale.data_ptr())


def stream_copy(rec, dst, src):
    rec.emit("fx_stream_copy", dst.data_ptr(), src.data_ptr(), src.numel())


def step_begin(rec, ctrl, lr, n_batches=0):
    rec.emit("fx_step_begin", ctrl.data_ptr(), float(lr), int(n_batches))


def sumsq_blocks(n) -> int:
    return int(lib.fx_sumsq_blocks(n))


def sumsq(rec, slots, x):
    rec.emit("fx_sumsq", slots.data_ptr(), x.data_ptr(), x.numel())


def hadamard_sum(rec, slot, g1, g2):
    rec.emit("fx_hadamard_sum", slot.data_ptr(), g1.data_ptr(), g2.data_ptr(), g1.numel())


def clip_finalize(rec, ctrl, slots, n_slots, max_norm):
    rec.emit("fx_clip_finalize", ctrl.data_ptr(), slots.data_ptr(), int(n_slots), float(max_norm))


def adam_flat(rec, p, g, m, v, ctrl, trainable=None):
    """``trainable``: optional 0/1 fp32 mask over the arena; elements with 0 are left untouched (frozen groups)."""
    rec.emit("fx_adam_flat", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), ctrl.data_ptr(),
             _ptr(trainable))


def adam_flat_clip(rec, p, g, m, v, ctrl, slots, n_slots, max_norm, trainable=None):
    """clip_finalize + adam_flat in one launch (see include/fxhip.h)."""
    rec.emit("fx_adam_flat_clip", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), ctrl.data_ptr(),
             _ptr(trainable), slots.data_ptr(), int(n_slots), float(max_norm))


def sigmoid(rec, y, x):
    rec.emit("fx_sigmoid", y.data_ptr(), x.data_ptr(), x.numel())


def sigmoid_bwd(rec, dx, dy, y):
    rec.emit("fx_sigmoid_bwd", dx.data_ptr(), dy.data_ptr(), y.data_ptr(), y.numel())


def softmax_rows(rec, y, x):
    _chk2d(x, "softmax_rows.x")
    _chk2d(y, "softmax_rows.y")
    rec.emit("fx_softmax_rows", y.data_ptr(), x.data_ptr(), x.shape[0], x.shape[1], _ld(x), _ld(y))


def reparam(rec, z, mean, log_var, eps=None, eps_out=None, seed=0, offset=0, ctrl=None):
    rec.emit("fx_reparam", z.data_ptr(), _ptr(eps_out), mean.data_ptr(), log_var.data_ptr(), _ptr(eps), z.numel(),
             int(seed), int(offset), _ptr(ctrl))


def mul(rec, y, a, b):
    rec.emit("fx_mul", y.data_ptr(), a.data_ptr(), b.data_ptr(), y.numel())


def fill_normal(rec, y, seed, offset, ctrl=None):
    rec.emit("fx_fill_normal", y.data_ptr(), y.numel(), int(seed), int(offset), _ptr(ctrl))


# ---- device-side ingest (csrc/fx_ingest.hip; reference data.py:360-452,519-545) -----------------------------------
IN_F32, IN_F64 = 0, 1


def _in_dtype(x: torch.Tensor, name: str) -> int:
    if not x.is_cuda or x.dim() != 2 or x.stride(1) != 1 or x.dtype not in (torch.float32, torch.float64):
        raise FxError(f"{name}: expected a row-major fp32/fp64 [n_samples, n_features] matrix on the GPU, got "
                      f"{x.dtype} {tuple(x.shape)} on {x.device}")
    return IN_F32 if x.dtype == torch.float32 else IN_F64


def _i32(t: Optional[torch.Tensor], name: str):
    if t is not None and not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise FxError(f"{name}: index lists are contiguous int32 tensors on the GPU")
    return _ptr(t)


def _f64(t: Optional[torch.Tensor], name: str):
    if t is not None and not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise FxError(f"{name}: expected a contiguous fp64 tensor on the GPU")
    return _ptr(t)


def col_moments(rec, x, rows=None, med=None, log1p=False):
    """Per-column (count int32, mean fp64, m2 fp64) over ``rows`` (None = all), NaN skipped, after imputation with
    ``med`` and the optional log1p."""
    dt = _in_dtype(x, "col_moments")
    n_rows = int(rows.numel()) if rows is not None else x.shape[0]
    F = x.shape[1]
    dev = x.device
    count = torch.empty(F, dtype=torch.int32, device=dev)
    mean = torch.empty(F, dtype=torch.float64, device=dev)
    m2 = torch.empty(F, dtype=torch.float64, device=dev)
    ws = torch.empty(int(lib.fx_col_moments_workspace_bytes(n_rows, F)) // 8 + 1, dtype=torch.float64, device=dev)
    rec.emit("fx_col_moments", x.data_ptr(), dt, x.stride(0), n_rows, F, _i32(rows, "col_moments"),
             _f64(med, "col_moments"), int(bool(log1p)), count.data_ptr(), mean.data_ptr(), m2.data_ptr(), ws.data_ptr())
    return count, mean, m2


def col_median(rec, x, cols, med_out):
    """med_out[c] = median of the non-NaN entries of column c for every c in ``cols`` (int32)."""
    dt = _in_dtype(x, "col_median")
    if med_out.numel() != x.shape[1]:
        raise FxError("col_median: med_out is an n_features-long fp64 vector")
    rec.emit("fx_col_median", x.data_ptr(), dt, x.stride(0), x.shape[0], _i32(cols, "col_median"), int(cols.numel()),
             _f64(med_out, "col_median"))


def row_moments(rec, x, cols, med=None):
    """Per-sample ddof=1 variance (fp64) over the listed columns after imputation."""
    dt = _in_dtype(x, "row_moments")
    var = torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
    rec.emit("fx_row_moments", x.data_ptr(), dt, x.stride(0), x.shape[0], _i32(cols, "row_moments"), int(cols.numel()),
             _f64(med, "row_moments"), var.data_ptr())
    return var


def ingest_transform(rec, x, out, rows=None, cols=None, med=None, log1p=False, mean=None, scale=None):
    """out[i, j] = fp32((value(x[rows[i], cols[j]]) - mean[j]) / scale[j]) (see include/fxhip.h)."""
    dt = _in_dtype(x, "ingest_transform")
    _chk2d(out, "ingest_transform.out")
    n_rows = int(rows.numel()) if rows is not None else x.shape[0]
    n_cols = int(cols.numel()) if cols is not None else x.shape[1]
    if tuple(out.shape) != (n_rows, n_cols):
        raise FxError(f"ingest_transform: out is {tuple(out.shape)}, expected {(n_rows, n_cols)}")
    for v in (mean, scale):
        if v is not None and v.numel() != n_cols:
            raise FxError("ingest_transform: mean / scale have one entry per output column")
    rec.emit("fx_ingest_transform", x.data_ptr(), dt, x.stride(0), _i32(rows, "ingest_transform"), n_rows,
             _i32(cols, "ingest_transform"), n_cols, _f64(med, "ingest_transform"), int(bool(log1p)),
             _f64(mean, "ingest_transform"), _f64(scale, "ingest_transform"), out.data_ptr(), _ld(out))


# ---- graph-convolution encoder (csrc/fx_gnn.hip; reference modules.py:153-262) -------------------------------------
GACT = {"relu": 0, "sigmoid": 1, "leakyrelu": 2, "tanh": 3, "gelu": 4}      # flexGCN act_options (modules.py:210-216)


def _chk_act3(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 3):
        raise FxError(f"{name}: expected a contiguous fp32 [B, nodes, C] tensor on the GPU, got {t.dtype} {tuple(t.shape)}")


def spmm_rows(rec, out, x, rowptr, idx, w):
    """out[b, i, :] = sum_e w[e] * x[b, idx[e], :] over the CSR row i (message passing over one shared graph)."""
    _chk_act3(x, "spmm_rows.x")
    _chk_act3(out, "spmm_rows.out")
    B, nodes, Cc = x.shape
    if out.shape != x.shape or rowptr.numel() != nodes + 1:
        raise FxError("spmm_rows: shape mismatch")
    rec.emit("fx_spmm_rows", out.data_ptr(), x.data_ptr(), _i32(rowptr, "spmm_rows"), _i32(idx, "spmm_rows"),
             w.data_ptr(), B, nodes, Cc, int(idx.numel()))


def rowlin2(rec, out, a, Wa, b=None, Wb=None, bias=None, trans=False, accumulate=False):
    """out[r, :] (+)= a[r, :] Wa^T (+ b[r, :] Wb^T) (+ bias) over the rows of the flattened [R, C] views."""
    R = a.numel() // a.shape[-1]
    Ca, Cout = a.shape[-1], out.shape[-1]
    Cb = b.shape[-1] if b is not None else 0
    want = (Ca, Cout) if trans else (Cout, Ca)
    if tuple(Wa.shape) != want or out.numel() != R * Cout:
        raise FxError(f"rowlin2: weight {tuple(Wa.shape)} / out {tuple(out.shape)} do not fit a {tuple(a.shape)} input")
    rec.emit("fx_rowlin2", out.data_ptr(), a.data_ptr(), Wa.data_ptr(), Ca, _ptr(b), _ptr(Wb), Cb, _ptr(bias), R, Cout,
             int(bool(trans)), int(bool(accumulate)))


def gnn_scratch(R: int, C: int, device) -> torch.Tensor:
    """Scratch big enough for fx_rowlin_wgrad (Cin, Cout <= C) and fx_bn_rows_* over R rows of C channels; give each
    call chain that may overlap another its own."""
    n = max(int(lib.fx_rowlin_wgrad_workspace_bytes(R, 32, 32)), int(lib.fx_bn_rows_workspace_bytes(R, 32)))
    return torch.empty(n // 8 + 2, dtype=torch.float64, device=device)


def rowlin_wgrad(rec, dW, db, dy, x, ws: torch.Tensor, accumulate=False):
    """dW [Cout, Cin] (+)= dy^T x and db [Cout] (+)= colsum(dy) over all rows, in a fixed summation order."""
    Cin, Cout = x.shape[-1], dy.shape[-1]
    R = x.numel() // Cin
    if ws.numel() * ws.element_size() < int(lib.fx_rowlin_wgrad_workspace_bytes(R, Cin, Cout)):
        raise FxError("rowlin_wgrad: scratch too small")
    rec.emit("fx_rowlin_wgrad", _ptr(dW), _ptr(db), dy.data_ptr(), x.data_ptr(), R, Cin, Cout, int(bool(accumulate)),
             ws.data_ptr())


def bn_rows_fwd(rec, out, x, gamma, beta, rmean, rvar, save_mean, save_invstd, act, train, drop_p, ws: torch.Tensor,
                mask=None, seed=0, offset=0, ctrl=None):
    C_ = x.shape[-1]
    R = x.numel() // C_
    if ws.numel() * ws.element_size() < int(lib.fx_bn_rows_workspace_bytes(R, C_)):
        raise FxError("bn_rows_fwd: scratch too small")
    rec.emit("fx_bn_rows_fwd", out.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(),
             rvar.data_ptr(), _ptr(save_mean), _ptr(save_invstd), _ptr(mask), R, C_, int(act), int(bool(train)),
             float(drop_p), int(seed), int(offset), _ptr(ctrl), ws.data_ptr())


def bn_rows_bwd(rec, da, dgamma, dbeta, x, gamma, beta, save_mean, save_invstd, act, drop_p, ws: torch.Tensor, mask=None,
                seed=0, offset=0, ctrl=None):
    C_ = x.shape[-1]
    R = x.numel() // C_
    if ws.numel() * ws.element_size() < int(lib.fx_bn_rows_workspace_bytes(R, C_)):
        raise FxError("bn_rows_bwd: scratch too small")
    rec.emit("fx_bn_rows_bwd", da.data_ptr(), _ptr(dgamma), _ptr(dbeta), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
             save_mean.data_ptr(), save_invstd.data_ptr(), _ptr(mask), R, C_, int(act), float(drop_p), int(seed),
             int(offset), _ptr(ctrl), ws.data_ptr())


# ---- index sampling (csrc/fx_sampling.hip) -------------------------------------------------------------------------------------
class DeviceRng:
    """A Philox stream position on the host: ``seed`` plus a running offset.  Every sampling launch consumes a disjoint range of
    counters, so a fit seeded the same way draws the same permutations and triplets whatever else runs beside it."""

    def __init__(self, seed: int):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0

    def take(self, n_counters: int) -> int:
        o = self.offset
        self.offset += int(n_counters) + 1
        return o


_RP_SCRATCH: dict = {}


def randperm(n: int, rng: DeviceRng, device, src: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A uniformly random permutation of range(n) (``src`` given: of src's first n entries), int64 on ``device``, launched on the
    current stream: DataLoader(shuffle=True)'s per-epoch torch.randperm (reference main.py:289-298) as one HIP launch."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise FxError("randperm: flexynesis_amd samples on the GPU only (no CPU fallback)")
    if out is None:
        out = torch.empty(int(n), dtype=torch.int64, device=dev)
    if src is not None and not (src.is_cuda and src.dtype == torch.int64 and src.is_contiguous() and src.numel() >= n):
        raise FxError("randperm: src must be a contiguous int64 tensor on the GPU with at least n entries")
    need = int(lib.fx_randperm_scratch_bytes(int(n)))
    scratch = None
    if need:
        import threading
        key = (dev.index, threading.get_ident())
        scratch = _RP_SCRATCH.get(key)
        if scratch is None or scratch.numel() < need:
            scratch = _RP_SCRATCH[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.call("fx_randperm", out.data_ptr(), _ptr(src), int(n), rng.seed, rng.take((int(n) + 3) // 4), _ptr(scratch),
              need, _stream())
    return out


def triplet_sample(anchors: torch.Tensor, gid, order, starts, counts, rank_in_group, n_groups: int, rng: DeviceRng, err_flag: torch.Tensor):
    """Positive / negative sample indices for ``anchors`` (TripletMultiOmicDataset.__getitem__, reference data.py:1106-1131)."""
    n = anchors.numel()
    pos, neg = torch.empty_like(anchors), torch.empty_like(anchors)
    for t in (anchors, gid, order, starts, counts, rank_in_group):
        if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
            raise FxError("triplet_sample: contiguous int64 GPU tensors expected")
    _lib.call("fx_triplet_sample", pos.data_ptr(), neg.data_ptr(), anchors.data_ptr(), n, gid.data_ptr(), order.data_ptr(),
              starts.data_ptr(), counts.data_ptr(), rank_in_group.data_ptr(), int(n_groups), rng.seed, rng.take(n),
              err_flag.data_ptr(), _stream())
    return pos, neg
