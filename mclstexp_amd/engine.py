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
                 single_graph: Optional[bool] = None, equal_shards: bool = False, strict_shards: Optional[bool] = None,
                 allow_short_last_batch: bool = False):
        """``equal_shards``: the caller guarantees that every rank sees the same per-rank batch size on every step
        (DistributedSampler with drop_last, synthetic data): the per-step host size exchange -- a blocking gloo
        all-gather that keeps the ranks' host threads in lock-step -- is skipped.  A step whose per-rank size differs from
        the first one then RAISES (a different size on ONE rank would send the ranks down different paths with mismatched
        collective shapes, and without the exchange that cannot be detected across ranks); ``allow_short_last_batch=True``
        is the explicit promise that such a step is a uniformly smaller last batch on EVERY rank (no drop_last with a
        dataset size that leaves the same remainder everywhere): it then takes the eager path on all of them."""
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.equal_shards = bool(equal_shards)
        self.strict_shards = (not allow_short_last_batch) if strict_shards is None else bool(strict_shards)
        self.graphs = graphs and torch.cuda.is_available()
        self.warmup = max(2, warmup)
        self.calls = 0
        self.static_in: Optional[Dict[str, Tensor]] = None
        self.ga = self.gb = None
        self.es = self.ei = self.d_es = self.d_ei = self.loss = None
        can_single = reducer is None and getattr(model, "process_group", None) is None
        if single_graph is None:
            single_graph = True
        self.single_graph = bool(single_graph) and can_single
        self._sizes_ex = None
        self._all_regular = True
        self.opt_in_graph = False
        self._one = None
        self._first_size = None
        self.seg_graphs, self.seg_ranges = [], []
        self._tables_stream = None
        self._img_bf16 = False
        self._in_sig = {}
        self.error_poll_every = int(os.environ.get("MCL_ERROR_POLL_EVERY", "16"))
        self._err_host = None
        self._err_event = None

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
        from . import densenet_fused as dn
        m = self.model
        dn.stamp("step start (main)")
        # zero_grad rides on the spot lane's stream, in front of the spot encoder (the lane is forked during the image forward
        # and joined before the loss: no new edge in the graph) -- on the main chain between loss and backward it cost 10 us
        zeroed = []

        def _zero_on_spot_lane():
            self.opt.zero_grad()
            zeroed.append(True)
        m._spot_lane_prologue = _zero_on_spot_lane
        try:
            es, ei = m.embed(inp)
        finally:
            m._spot_lane_prologue = None
        dn.stamp("forward done (main)")
        loss, d_es, d_ei = m.loss_and_grads(es, ei)
        if not zeroed:
            self.opt.zero_grad()
        dn.stamp("backward start (main)")
        torch.autograd.backward((es, ei), (d_es, d_ei))
        dn.stamp("backward done (main)")
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
                 and getattr(m, "embedding_grad", "dense") == "rowsparse" and "hook" not in sink)
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
    def _image_staged_bf16(self, batch) -> bool:
        """The static image input of the step graph can hold the bf16 channels-last activation directly when the image
        encoder's first kernel consumes exactly that (DenseNet / ResNet on the fused bf16 path): the per-step copy of the fp32
        image into the graph's input buffer and the cast kernel inside the graph become ONE launch outside it."""
        from . import backbones
        m = self.model
        enc = getattr(m, "image_encoder", None)
        img = batch.get("image")
        return (getattr(m, "fused_backbone", False)
                and getattr(m, "backbone_dtype", None) == torch.bfloat16
                and isinstance(enc, (backbones.ImageEncoder, backbones.ImageEncoder_Resnet, backbones.ImageEncdoer_res18,
                                     backbones.ImageEncdoer_res101))
                and img is not None and img.is_cuda and img.dtype == torch.float32 and img.dim() == 4
                and not img.requires_grad)

    def _stage_inputs(self, batch) -> None:
        for k, v in self.static_in.items():
            if k == "image" and self._img_bf16:
                from . import densenet_fused as dn
                dn.image_to_act(batch[k], torch.bfloat16, out=v)
            else:
                v.copy_(batch[k], non_blocking=True)

    def _capture(self, batch) -> None:
        m = self.model
        self.static_in = {k: v.clone(memory_format=torch.preserve_format) for k, v in batch.items()}
        self._img_bf16 = self._image_staged_bf16(batch)
        if self._img_bf16:
            from . import densenet_fused as dn
            self.static_in["image"] = dn.image_to_act(batch["image"], torch.bfloat16)
        self._in_sig = {k: (tuple(v.shape), v.dtype) for k, v in batch.items()}
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
                     and getattr(m, "embedding_grad", "dense") == "rowsparse" and "hook" not in sink)
            if early:
                sink["hook"] = self.opt._early_tables
            with torch.cuda.graph(self.ga, capture_error_mode="thread_local"):
                if getattr(m, "embedding_grad", "dense") == "rowsparse":
                    m.sparse_grads["static"] = True          # the captured backward's dout / ix / iy are static buffers
                self.es, self.ei, self.loss = self._sequence(self.static_in)
                # the optimizer too: FusedAdam keeps its step counter and constants on the device (optim._begin_step), so
                # its launches replay unchanged -- no eager launches between two replays
                self.opt_in_graph = hasattr(self.opt, "_begin_step")
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
        # Segmented backward (data parallel, DenseNet on the fused kernels): the backbone's backward is cut at dense-block
        # inputs and captured as one graph PER SEGMENT; after each segment's replay the all-reduce of exactly the gradient
        # range that segment finished is enqueued on the communicator's stream, where it runs beside the next segment's
        # kernels (what DDP's bucketed overlap does for /root/reference/baselines/Bleep/BLEEP_main.py:76-78,147 -- here with
        # ranges that follow the flat bucket's layout, no packing copies), and the position tables are updated beside the
        # remaining segments as soon as segment 0 has produced their gradient rows.  MCL_DP_SEGMENTS = number of segments
        # (default 1 = one backward graph and one all-reduce after it: the only form MEASURED so far -- on one GPU every cut
        # costs time and there is nothing to overlap; 2 = [heads, spot branch, norm5, last dense block] = 70 % of the gradient
        # bytes | the rest; up to one per dense block.  Until a multi-GPU run decides, the default follows the measurement).  Every cut adds a join of the
        # weight-gradient side stream at its block's end and a graph launch: measured on ONE GPU (size-1 RCCL group, nothing
        # to overlap) 2 segments cost +0.27 ms/step, 4 segments +0.35 (profiles/r04_dp_segments_size1.txt).
        # (a group of ONE rank -- MCL_FORCE_DIST=1 on a single GPU -- has nothing to overlap: one backward graph by default)
        try:
            world = torch.distributed.get_world_size(self.reducer.pg) if self.reducer is not None else 1
        except Exception:
            world = 1
        n_seg = int(os.environ.get("MCL_DP_SEGMENTS", "1"))
        n_blocks = 0
        enc = getattr(m, "image_encoder", None)
        feats = enc.model[0] if enc is not None and hasattr(enc, "model") and hasattr(enc, "forward_fused") else None
        while feats is not None and hasattr(feats, f"denseblock{n_blocks + 1}"):
            n_blocks += 1
        n_seg = max(1, min(n_seg, n_blocks)) if n_blocks else 1
        want_seg = (self.reducer is not None and n_seg > 1 and hasattr(m, "segment_backward")
                    and hasattr(self.opt, "flat_location"))
        m.segment_backward = tuple(range(n_blocks - n_seg + 2, n_blocks + 1)) if want_seg else ()
        # (zero_grad rides on the spot lane inside graph A, as in the single-graph step: nothing touches .grad between the
        #  previous step's optimizer and graph B)
        zeroed = []

        def _zero_on_spot_lane():
            self.opt.zero_grad()
            zeroed.append(True)
        m._spot_lane_prologue = _zero_on_spot_lane
        try:
            with torch.cuda.graph(self.ga, capture_error_mode="thread_local"):
                self.es, self.ei = m.embed(self.static_in)
        finally:
            m.segment_backward = ()
            m._spot_lane_prologue = None
        cuts = list(getattr(m, "backward_cuts", None) or []) if want_seg else []
        m.backward_cuts = None
        self.d_es = torch.zeros_like(self.es)
        self.d_ei = torch.zeros_like(self.ei)
        self.gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gb, pool=self.ga.pool(), capture_error_mode="thread_local"):
            if not zeroed:
                self.opt.zero_grad()
            torch.autograd.backward((self.es, self.ei), (self.d_es, self.d_ei))
        self.seg_graphs, self.seg_ranges = [], []
        if cuts:
            for up, slot in reversed(cuts):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.ga.pool(), capture_error_mode="thread_local"):
                    torch.autograd.backward((up,), (slot["grad"],))
                self.seg_graphs.append(g)
            self.seg_ranges = self._segment_ranges(n_blocks - len(cuts) + 1, n_blocks)
            if self.seg_ranges is None:                  # parameters not laid out as expected: one all-reduce at the end
                self.seg_ranges = []
        if getattr(m, "embedding_grad", "dense") == "rowsparse":
            m.sparse_grads["static"] = True
        torch.cuda.synchronize()

    def _segment_ranges(self, first_cut_block: int, last_block: int):
        """[(flat bucket, lo, hi)] per backward segment, in backward order: segment 0 (heads, spot branch, norm5, last
        dense block) owns the tail of the image encoder's flat range and everything behind it, segment k the range from
        the first parameter of the k-th cut block (counted from the end) up to the next cut block's.  None if the layout
        does not allow it."""
        enc = getattr(self.model, "image_encoder", None)
        feats = enc.model[0] if enc is not None and hasattr(enc, "model") else None
        if feats is None:
            return None
        starts, gi = [], None
        for i in range(first_cut_block, last_block + 1):
            blk = getattr(feats, f"denseblock{i}", None)
            if blk is None:
                return None
            loc = self.opt.flat_location(next(blk.parameters()))
            if loc is None or (gi is not None and loc[0] != gi):
                return None
            gi = loc[0]
            starts.append(loc[1])
        bucket = self.opt.flat_bucket(gi)
        if bucket is None or starts != sorted(starts) or len(set(starts)) != len(starts):
            return None
        bounds = [0] + starts + [bucket.numel()]
        # EVERY parameter a segment back-propagates must lie inside that segment's range (an optimizer group built in another
        # order than the module tree would otherwise have a range reduced before the segment that fills it has replayed):
        # children of the feature extractor in front of the first cut block -> the last segment (range 0), a cut block and what
        # follows it up to the next cut -> its own range, everything outside the backbone (heads, spot branch) -> the tail range
        seg_of = {}
        k = 0
        cut_names = [f"denseblock{i}" for i in range(first_cut_block, last_block + 1)]
        for name, child in feats.named_children():
            if name in cut_names:
                k = cut_names.index(name) + 1
            for p in child.parameters():
                seg_of[id(p)] = k
        tail = len(bounds) - 2
        for group in self.opt.param_groups:
            for p in group["params"]:
                loc = self.opt.flat_location(p)
                if loc is None:
                    continue
                if loc[0] != gi:
                    return None
                kk = seg_of.get(id(p), tail)
                if not (bounds[kk] <= loc[1] and loc[1] + p.numel() <= bounds[kk + 1]):
                    return None
        return [(bucket, bounds[k], bounds[k + 1]) for k in range(len(bounds) - 2, -1, -1)]

    def _replay_backward_dp(self) -> None:
        """Graph B (and its segments) + the gradient exchange + the optimizer, data parallel."""
        if not self.seg_graphs or not self.seg_ranges:
            self.gb.replay()
            self._reduce_and_step()
            return
        red, handles = self.reducer, []
        main = torch.cuda.current_stream()
        self.gb.replay()                                   # heads, spot branch, norm5, last dense block
        # the position tables: their gradient rows exist now (the spot branch's backward is in segment 0).  Row exchange
        # (a small all-gather, first on the communicator) + the HBM-streaming table Adam run on a side stream beside the
        # remaining, latency-bound segments -- what the single-process step graph does from inside its backward
        tables_side = None
        if hasattr(self.opt, "_early_tables") and getattr(self.model, "embedding_grad", "dense") == "rowsparse":
            if self._tables_stream is None:
                self._tables_stream = torch.cuda.Stream(device=main.device)
            tables_side = self._tables_stream
            tables_side.wait_stream(main)
            with torch.cuda.stream(tables_side):
                self.opt._early_tables()
        elif hasattr(self.opt, "prefetch_table_rows"):
            self.opt.prefetch_table_rows()
        bucket, lo, hi = self.seg_ranges[0]
        handles += red.reduce_range(bucket, lo, hi, async_flat=True)
        for g, (bucket, lo, hi) in zip(self.seg_graphs, self.seg_ranges[1:]):
            g.replay()
            handles += red.reduce_range(bucket, lo, hi, async_flat=True)
        # whatever lives outside the segmented bucket (other param groups, parameters that are not flat-managed)
        handles += red.reduce(self.opt, async_flat=True, skip_flat={self.seg_ranges[0][0].data_ptr()})
        if tables_side is not None:
            main.wait_stream(tables_side)
        if hasattr(self.opt, "prefetch_table_rows"):
            self.opt.step(wait=handles)
        else:
            for h in handles:
                h.wait()
            self.opt.step()

    def opt_will_be_in_graph(self) -> bool:
        return hasattr(self.opt, "_begin_step")

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
        return all(k in batch and tuple(batch[k].shape) == sh and batch[k].dtype == dt for k, (sh, dt) in self._in_sig.items())

    # ------------------------------------------------------------------ call
    def _agree_sizes(self, batch) -> bool:
        """Data parallel: the ranks agree (on the host, dist.SizeExchange) on this step's per-rank batch sizes, so
        that every rank takes the SAME path -- graph replay only when every shard has the captured size -- and the
        collectives of a ragged step (last batch of an epoch, train.py:49) know the true shard sizes.  Returns
        True when all shards are equal."""
        if self.reducer is None:
            return True
        if self.equal_shards:
            # the caller promised that every rank sees the SAME per-rank batch size on every step (a different size on ONE rank
            # would send the ranks down different paths with mismatched collective shapes; it cannot be detected across ranks
            # without the exchange this flag switches off).  A size that differs from the captured one -- a uniformly smaller last
            # batch without drop_last -- is an error unless the caller passed allow_short_last_batch=True (then: eager on every rank).
            n = int(batch["expression"].shape[0])
            if self._first_size is None:
                self._first_size = n
            elif n != self._first_size and self.strict_shards:
                raise RuntimeError(f"TrainStep(equal_shards=True): this step's per-rank batch has {n} pairs, the first one had "
                                   f"{self._first_size}; pass allow_short_last_batch=True if EVERY rank sees the same smaller "
                                   "last batch, or equal_shards=False for the per-step size exchange")
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

    # ------------------------------------------------------------------ device error words (ops.device_error_words)
    def _device(self):
        return next(self.model.parameters()).device

    def check_errors(self) -> None:
        """Host sync: raises IndexError (position outside the tables) / ops.SeamTimeoutError (a persistent dense-block launch
        timed out at a BatchNorm seam: that step is invalid) if any step since the last check flagged one."""
        from . import ops
        self._err_event = None
        ops.check_device_errors(self._device())

    def _poll_errors(self) -> None:
        """The same check WITHOUT a host sync: every ``error_poll_every`` calls the device's error words are copied to pinned
        host memory behind the step; a later call finds the copy finished and raises -- at most ``error_poll_every`` + 1 steps
        after the faulty one (the step that timed out and its successors are invalid: the caller restores a checkpoint or
        rebuilds the model).  ``error_poll_every = 0`` switches the polling off (``check_errors`` remains)."""
        if not self.error_poll_every or not torch.cuda.is_available():
            return
        from . import ops
        dev = self._device()
        if dev.type != "cuda":
            return
        if self._err_event is not None and self._err_event.query():
            self._err_event = None
            words = self._err_host.tolist()
            ops.raise_for_error_words(words, ops.device_error_words(dev).zero_)
        if self._err_event is None and self.calls % self.error_poll_every == 0:
            if self._err_host is None:
                self._err_host = torch.zeros(ops._ERR_WORDS, dtype=torch.int32).pin_memory()
            self._err_host.copy_(ops.device_error_words(dev), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._err_event = ev

    def __call__(self, batch: Dict[str, Tensor]) -> Tensor:
        out = self._step(batch)
        self._poll_errors()
        return out

    def _step(self, batch: Dict[str, Tensor]) -> Tensor:
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
        self._stage_inputs(batch)
        if self.opt_in_graph:
            # an LR schedule / param_groups edit reaches the replayed optimizer; the lazy position tables are materialised
            # before the constants' history ring wraps
            (self.opt.pre_replay if hasattr(self.opt, "pre_replay") else self.opt.sync_hyper)()
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
        if self.reducer is not None:
            self._replay_backward_dp()
        else:
            self.gb.replay()
            self._reduce_and_step()
        return loss
