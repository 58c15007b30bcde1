"""TrainStep: one optimisation step of train.py:33-41, optionally replayed from HIP graphs.

Why graphs: the step is ~1000 kernel launches (DenseNet-121 has 120 conv + 121 BatchNorm layers, each with
fused forward, data-gradient and weight-gradient kernels); in eager mode the Python launch path needs longer per
step on the host than the GPU work itself.  Capturing removes the host from the critical path
(MI355X-first: "HIP streams and graphs instead of a tracing compiler").

Structure (identical for 1 GPU and for data parallel, so RCCL collectives are never inside a capture):

  (one process without collectives: ONE graph = A + the InfoNCE kernels + B; the split below is the data-parallel form)
    graph A   forward up to the (B, P) embeddings            [model.embed]
    eager     symmetric InfoNCE fwd+bwd (+ 2 all-gathers)     [model.loss_and_grads]  -> loss, dE
    graph B   zero_grad + backward from the embeddings        [torch.autograd.backward captured]
    eager     gradient all-reduce (DP), FusedAdam step        [a handful of launches]

Graph B is captured right after graph A in the same memory pool, while A's autograd graph is alive; replays
re-launch the same kernels on the same static buffers.  The first ``warmup`` calls run eagerly (MIOpen solver
search, FusedAdam flat-bucket construction) and a batch whose shapes differ from the captured ones (ragged
last batch, train.py:49 has no drop_last) falls back to the eager path.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

Tensor = torch.Tensor


class TrainStep:
    def __init__(self, model, optimizer, reducer=None, graphs: bool = True, warmup: int = 3,
                 single_graph: Optional[bool] = None, equal_shards: bool = False):
        """``equal_shards``: the caller guarantees that every rank sees the same per-rank batch size on every step
        (DistributedSampler with drop_last, synthetic data): the per-step host size exchange -- a blocking gloo
        all-gather that keeps the ranks' host threads in lock-step -- is skipped."""
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.equal_shards = bool(equal_shards)
        self.graphs = graphs and torch.cuda.is_available()
        self.warmup = max(2, warmup)
        self.calls = 0
        self.static_in: Optional[Dict[str, Tensor]] = None
        self.ga = self.gb = None
        self.es = self.ei = self.d_es = self.d_ei = self.loss = None
        can_single = reducer is None and getattr(model, "process_group", None) is None
        if single_graph is None:
            single_graph = os.environ.get("MCL_SINGLE_GRAPH", "1") != "0"
        self.single_graph = bool(single_graph) and can_single
        self._sizes_ex = None
        self._all_regular = True
        self.opt_in_graph = False
        self._one = None

    # ------------------------------------------------------------------ eager (reference order, train.py:36-39)
    def _eager(self, batch) -> Tensor:
        loss = self.model(batch)
        self.opt.zero_grad()
        # (an explicit, cached root gradient: a bare loss.backward() makes autograd launch an ATen fill for ones_like(loss))
        if self._one is None or self._one.device != loss.device or self._one.dtype != loss.dtype:
            self._one = torch.ones_like(loss)
        loss.backward(self._one)
        self._reduce_and_step()
        return loss.detach()

    def _sequence(self, inp):
        """The launch sequence of the single step graph up to the optimizer, shared by the capture and by
        ``run_sequence_eager``: forward, closed-form InfoNCE forward + backward, zero_grad, backward from the embeddings."""
        m = self.model
        es, ei = m.embed(inp)
        loss, d_es, d_ei = m.loss_and_grads(es, ei)
        self.opt.zero_grad()
        torch.autograd.backward((es, ei), (d_es, d_ei))
        return es, ei, loss

    def run_sequence_eager(self, batch) -> Tensor:
        """One optimisation step issued EAGERLY with exactly the calls the single step graph records (``_sequence`` +
        ``optimizer.step()``, the position tables updated from inside the backward): what a kernel-level audit of the
        captured step profiles (tests/test_own_kernels_gpu.py) -- kernels of a replayed graph are not individually visible
        to torch.profiler on this stack."""
        if self.reducer is not None or getattr(self.model, "process_group", None) is not None:
            raise RuntimeError("run_sequence_eager mirrors the single-process step graph")
        m = self.model
        batch = {k: batch[k] for k in ("image", "expression", "position")}
        sink = getattr(m, "sparse_grads", None)
        early = (self.opt_will_be_in_graph() and sink is not None and hasattr(self.opt, "_early_tables")
                 and getattr(m, "embedding_grad", "dense") == "rowsparse" and "hook" not in sink
                 and os.environ.get("MCL_EARLY_TABLES", "1") != "0")
        if early:
            sink["hook"] = self.opt._early_tables
        try:
            _, _, loss = self._sequence(batch)
            self.opt.step()
        finally:
            if early:
                sink.pop("hook", None)
        return loss

    def _reduce_and_step(self) -> None:
        """Data parallel: the 63 MB flat-gradient all-reduce is enqueued asynchronously and overlaps the HBM-bound
        position-table update (which needs only the small row exchange issued just before it)."""
        if self.reducer is None:
            self.opt.step()
            return
        if hasattr(self.opt, "prefetch_table_rows"):
            handles = self.reducer.reduce(self.opt, async_flat=True)
            self.opt.step(wait=handles)
        else:
            self.reducer.reduce(self.opt)
            self.opt.step()

    # ------------------------------------------------------------------ capture
    def _capture(self, batch) -> None:
        m = self.model
        self.static_in = {k: v.clone(memory_format=torch.preserve_format) for k, v in batch.items()}
        if hasattr(self.opt, "sync_hyper"):
            self.opt.sync_hyper()                # the captured step reads lr / betas / eps / wd from device memory
        torch.cuda.synchronize()
        # capture_error_mode "thread_local": under data parallelism the RCCL watchdog thread polls events while we
        # capture; in the default "global" mode any such call from another thread invalidates the capture
        self.ga = torch.cuda.CUDAGraph()
        if self.single_graph:
            # one process, no collectives: forward, InfoNCE (closed-form forward + backward) and backward replay as ONE
            # graph -- no host round trip between the three phases (the eager InfoNCE launches left the GPU idle
            # while the second graph was being submitted)
            # the optimizer is part of this graph: the position tables may be updated from inside the backward (side stream,
            # under the latency-bound blocks of the backbone) -- armed for the capture only, see FusedAdam.attach_model
            sink = getattr(m, "sparse_grads", None)
            early = (self.opt_will_be_in_graph() and sink is not None and hasattr(self.opt, "_early_tables")
                     and getattr(m, "embedding_grad", "dense") == "rowsparse" and "hook" not in sink
                     and os.environ.get("MCL_EARLY_TABLES", "1") != "0")
            if early:
                sink["hook"] = self.opt._early_tables
            with torch.cuda.graph(self.ga, capture_error_mode="thread_local"):
                if getattr(m, "embedding_grad", "dense") == "rowsparse":
                    m.sparse_grads["static"] = True          # the captured backward's dout / ix / iy are static buffers
                self.es, self.ei, self.loss = self._sequence(self.static_in)
                # the optimizer too: FusedAdam keeps its step counter and constants on the device (optim._begin_step), so
                # its launches replay unchanged -- no eager launches between two replays
                self.opt_in_graph = hasattr(self.opt, "_begin_step") and os.environ.get("MCL_OPT_IN_GRAPH", "1") != "0"
                if self.opt_in_graph:
                    self.opt.step()
                    self.opt._step_count -= 1    # capture records launches, it does not run them: the replay counts
            if early:
                sink.pop("hook", None)           # eager backward calls (ragged batches) keep the plain semantics
            torch.cuda.synchronize()
            return
        # two graphs with eager work between them: the early position-table update (a hook inside backward that also
        # advances the device step counter) would be replayed by graph B AND repeated by the eager step() -> off here
        if getattr(m, "sparse_grads", None) is not None:
            m.sparse_grads.pop("hook", None)
        with torch.cuda.graph(self.ga, capture_error_mode="thread_local"):
            self.es, self.ei = m.embed(self.static_in)
        self.d_es = torch.zeros_like(self.es)
        self.d_ei = torch.zeros_like(self.ei)
        self.gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gb, pool=self.ga.pool(), capture_error_mode="thread_local"):
            self.opt.zero_grad()
            torch.autograd.backward((self.es, self.ei), (self.d_es, self.d_ei))
        if getattr(m, "embedding_grad", "dense") == "rowsparse":
            m.sparse_grads["static"] = True
        torch.cuda.synchronize()

    def opt_will_be_in_graph(self) -> bool:
        return hasattr(self.opt, "_begin_step") and os.environ.get("MCL_OPT_IN_GRAPH", "1") != "0"

    def _eager_ragged(self, batch) -> Tensor:
        """A batch whose shapes differ from the captured ones (ragged last batch: train.py:49 has no drop_last) runs
        eagerly.  The captured backward refills the STATIC sink entries (dout / ix / iy of the position tables) on
        every replay; the eager backward overwrites them with its own tensors and FusedAdam.step then clears the
        sink -- so the static entries are set aside here and put back afterwards, otherwise every later replay
        would silently stop updating x_embed / y_embed."""
        sink = getattr(self.model, "sparse_grads", None)
        rowsparse = sink is not None and getattr(self.model, "embedding_grad", "dense") == "rowsparse"
        saved = {}
        if rowsparse:
            saved = {k: sink[k] for k in ("dout", "ix", "iy", "hook") if k in sink}
            sink.clear()
            if "hook" in saved:
                sink["hook"] = saved["hook"]
        try:
            return self._eager(batch)
        finally:
            if rowsparse:
                sink.clear()
                sink.update(saved)
                sink["static"] = True

    def _same_shapes(self, batch) -> bool:
        return all(k in batch and batch[k].shape == v.shape and batch[k].dtype == v.dtype
                   for k, v in self.static_in.items())

    # ------------------------------------------------------------------ call
    def _agree_sizes(self, batch) -> bool:
        """Data parallel: the ranks agree (on the host, dist.SizeExchange) on this step's per-rank batch sizes, so
        that every rank takes the SAME path -- graph replay only when every shard has the captured size -- and the
        collectives of a ragged step (last batch of an epoch, train.py:49) know the true shard sizes.  Returns
        True when all shards are equal."""
        if self.reducer is None:
            return True
        if self.equal_shards:
            self._all_regular = (self.static_in is not None
                                 and batch["expression"].shape[0] == self.static_in["expression"].shape[0])
            return True
        from . import dist as mdist
        if self._sizes_ex is None:
            self._sizes_ex = mdist.SizeExchange(self.reducer.pg)
        sizes = self._sizes_ex(batch["expression"].shape[0])
        if min(sizes) == 0:
            raise RuntimeError(f"data-parallel step with an empty shard (sizes {sizes}): drop or re-balance the "
                               "last batch")
        mdist.set_step_sizes(sizes)
        self._all_regular = self.static_in is not None and all(
            v == self.static_in["expression"].shape[0] for v in sizes)
        return len(set(sizes)) == 1

    def __call__(self, batch: Dict[str, Tensor]) -> Tensor:
        batch = {k: batch[k] for k in ("image", "expression", "position")}
        self.calls += 1
        equal = self._agree_sizes(batch)
        if not self.graphs or self.calls <= self.warmup:
            return self._eager(batch)
        if self.ga is None:
            if not equal:
                return self._eager(batch)            # never capture on a ragged step
            self._capture(batch)
            self._all_regular = True
        if not self._same_shapes(batch) or (self.reducer is not None and not self._all_regular):
            return self._eager_ragged(batch)
        for k, v in self.static_in.items():
            v.copy_(batch[k], non_blocking=True)
        if self.opt_in_graph:
            self.opt.sync_hyper()                # an LR schedule / param_groups edit reaches the replayed optimizer
        self.ga.replay()
        if self.single_graph:
            if self.opt_in_graph:
                self.opt._step_count += 1        # host mirror of the device step counter
            else:
                self._reduce_and_step()
            return self.loss.clone()
        loss, d_es, d_ei = self.model.loss_and_grads(self.es, self.ei)
        self.d_es.copy_(d_es)
        self.d_ei.copy_(d_ei)
        self.gb.replay()
        self._reduce_and_step()
        return loss
