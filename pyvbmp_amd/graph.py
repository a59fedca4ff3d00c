"""hipGraph replay of one VB iteration.

The conjugate-update path at small problem sizes (BASELINE configs[0]: a K=4, D=2 mixture on 400 points) is a few
dozen KB-sized kernels per iteration, i.e. bound by launch latency, not by the GPU.  `GraphedStep` captures ONE
iteration -- the libvbmp_hip.so launches and the surrounding torch glue alike, they are all enqueued on torch's
current stream -- into a HIP graph and replays it.

The reference's classes rebind their attributes on every update (`self.mu = ...`), so an iteration reads the
state tensors it finds and leaves NEW tensors behind.  A graph replays fixed addresses; therefore, inside the
capture, every state tensor that the iteration replaced is copied back into the buffer the iteration read it
from, and the attributes are pointed at those static buffers again.  Replays then carry the state forward
exactly like the eager loop does (the floating-point atomics of the reductions make the sums order-dependent:
compare with the eager loop at 1e-10, not bitwise).

Restrictions (checked where possible, and capture fails loudly otherwise): the iteration must not synchronise
(no `.item()`, no printing of device values), must not change tensor shapes from one iteration to the next, and
the data tensors passed to it must stay alive and in place.
"""
import gc
import weakref

import torch

from ._common import state_epoch

# Destroying a HIP graph while a stream is capturing aborts the process.  Graphs therefore never die during a capture:
# (1) a GraphedStep holds its model only weakly and gets the model passed into its step function, so there is no
# model -> cache -> GraphedStep -> closure -> model cycle and a graph's lifetime ends deterministically with its model
# (or with close()), not whenever the cyclic collector runs; (2) a GraphedStep that is finalised while some capture is
# in progress (the step under capture dropped the last reference to another model) parks its graph here until that
# capture has ended; (3) the collector is paused for the duration of a capture as a last line of defence.
_capture_depth = 0
_graveyard = []


def _walk(obj, path, out, seen):
    """collect every tensor reachable through instance attributes, lists, tuples and dicts: path -> (owner, key)"""
    if id(obj) in seen:
        return
    if isinstance(obj, torch.Tensor):
        return
    seen.add(id(obj))
    if isinstance(obj, dict):
        items = list(obj.items())
    elif isinstance(obj, (list, tuple)):
        items = list(enumerate(obj))
    elif hasattr(obj, "__dict__"):
        items = list(vars(obj).items())
    else:
        return
    for k, v in items:
        if isinstance(k, str) and k.startswith("_vbmp_"):
            continue  # the graph cache itself
        if isinstance(v, torch.Tensor):
            if v.is_cuda and not isinstance(obj, tuple):
                out[path + (k,)] = (obj, k)
        elif isinstance(v, (dict, list, tuple)) or (hasattr(v, "__dict__") and not isinstance(v, type)
                                                     and not callable(v)):
            _walk(v, path + (k,), out, seen)


def _get(owner, key):
    return owner[key] if isinstance(owner, (dict, list)) else getattr(owner, key)


def _set(owner, key, value):
    if isinstance(owner, (dict, list)):
        owner[key] = value
    else:
        setattr(owner, key, value)


def state_tensors(model):
    out = {}
    _walk(model, (), out, set())
    return out


class GraphedStep():
    """graph = GraphedStep(model, lambda m: m.update(X, iters=1)); graph.run(n) == n eager iterations.
    `step` (and `post`) receive the model as their argument: they must not close over it (see the note on lifetimes)."""

    def __init__(self, model, step, warmup=2, post=None):
        """post: optional eager epilogue run after every iteration (bookkeeping whose shapes grow, e.g. an ELBO trace)"""
        global _capture_depth
        self._model = weakref.ref(model)
        self._step, self._post = step, post
        self.graph = None
        # bound methods as LOCALS only (a stored one would tie this object into a cycle with itself)
        step = self._run_step
        post = self._run_post if post is not None else None
        dev = next((_get(o, k).device for (o, k) in state_tensors(model).values()), None)
        assert dev is not None and dev.type == "cuda", "GraphedStep needs a model whose state lives on the GPU"
        self.device = dev
        self.warmup_iters = warmup
        # lazily created attributes (responsibilities, cached statistics) must exist with their final shapes before
        # the state is snapshotted; torch asks for warm-up work to run on a side stream before capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step()
                if post is not None:
                    post()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        before = state_tensors(model)
        static = {path: _get(o, k) for path, (o, k) in before.items()}
        self.graph = torch.cuda.CUDAGraph()
        # No garbage collection while the stream is capturing: models hold their graphs in reference cycles
        # (model -> cache -> GraphedStep -> step closure -> model), so an older model's graph is destroyed whenever the
        # cyclic collector happens to run -- and destroying a HIP graph during a capture aborts the process.
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        _capture_depth += 1
        state_epoch[0] += 1
        try:
            self._capture(model, step, before, static)
        finally:
            _capture_depth -= 1
            if gc_was_on:
                gc.enable()
            if _capture_depth == 0:
                del _graveyard[:]  # graphs whose owners went away during the capture are released now
        self.static = static
        self.iterations = warmup + 1  # the capture pass does not execute; accounted for by the first replay below
        self.graph.replay()
        state_epoch[0] += 1
        if post is not None:
            post()

    def _run_step(self):
        self._step(self.model)

    def _run_post(self):
        if self._post is not None:
            self._post(self.model)

    @property
    def model(self):
        m = self._model()
        if m is None:
            raise RuntimeError("the model of this GraphedStep no longer exists")
        return m

    def close(self):
        """release the HIP graph and its memory pool now (deferred to the end of a capture that is in progress)"""
        g, self.graph = self.graph, None
        self.static = {}
        if g is not None and _capture_depth > 0:
            _graveyard.append(g)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def _capture(self, model, step, before, static):
        with torch.cuda.graph(self.graph):
            step()
            after = state_tensors(model)
            for path, (o, k) in after.items():
                new = _get(o, k)
                old = static.get(path)
                if old is None:
                    static[path] = new  # created inside the capture: lives in the graph's pool, address is stable
                    continue
                if new is old:
                    continue
                if new.shape != old.shape or new.dtype != old.dtype:
                    raise RuntimeError(f"state tensor {'.'.join(map(str, path))} changed from {tuple(old.shape)} "
                                       f"{old.dtype} to {tuple(new.shape)} {new.dtype} within one iteration; "
                                       "a replayed graph needs fixed shapes")
                # stride-0 expanded state (the reference's expanded priors): write the one stored copy
                idx = tuple(slice(0, 1) if (st == 0 and sz > 1) else slice(None)
                            for st, sz in zip(old.stride(), old.shape))
                old[idx].copy_(new[idx])
                _set(o, k, old)

    def sync_in(self):
        """state that was rebound outside the graph (an eager update in between) goes back into the static buffers"""
        for path, (o, k) in state_tensors(self.model).items():
            old = self.static.get(path)
            cur = _get(o, k)
            if old is None or cur is old:
                continue
            if cur.shape != old.shape or cur.dtype != old.dtype:
                # bookkeeping outside the graph (grown by the `post` epilogue) is simply re-pointed
                self.static[path] = cur
                continue
            idx = tuple(slice(0, 1) if (st == 0 and sz > 1) else slice(None) for st, sz in zip(old.stride(), old.shape))
            old[idx].copy_(cur[idx])
            _set(o, k, old)

    def run(self, iters=1):
        for _ in range(iters):
            self.graph.replay()
            state_epoch[0] += 1  # the replay rewrote state in place: quantities cached from it are stale
            self._run_post()
        self.iterations += iters
        return self


def close(model):
    """drop the cached graph of a model explicitly (its HIP graph and private memory pool are released)"""
    cache = model.__dict__.get("_vbmp_graphs")
    if cache:
        for g in list(cache.values()):
            g.close()
        cache.clear()


def run_iterations(model, step, iters, key, warmup=2, post=None):
    """`iters` VB iterations of `step(model)` (one iteration per call) through a cached GraphedStep; iterations that the
    construction of the graph already performed (warm-up + first replay) count towards `iters`.  The graph is
    cached on the model under `key` (data pointer / shape / hyper-parameters of the call): a different key builds
    a new graph."""
    cache = model.__dict__.setdefault("_vbmp_graphs", {})
    g = cache.get(key)
    done = 0
    if g is None:
        if iters < warmup + 1:
            for _ in range(iters):
                step(model)
                if post is not None:
                    post(model)
            return
        close(model)  # one graph (and one private memory pool) per model at a time
        g = cache[key] = GraphedStep(model, step, warmup=warmup, post=post)
        done = warmup + 1
    else:
        g.sync_in()
    g.run(iters - done)
