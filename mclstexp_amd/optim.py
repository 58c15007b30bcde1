"""FusedAdam: torch.optim.Adam(lr, betas, eps, weight_decay) semantics (train.py:118-120) on the HIP
kernels of csrc/adam.hip.

MI355X-first layout: all dense parameters of a group are re-homed into ONE flat fp32 buffer (their
``.data`` / ``.grad`` become views), with flat exp_avg / exp_avg_sq beside it, so the whole update is a
single HBM-streaming launch (28 B/element) and, under data parallelism, the flat gradient buffer is
the all-reduce bucket (no packing copies).  The two (65536, G) position tables are updated by a
second kernel that reads no dense gradient: untouched rows see g = wd*p exactly as in the reference
(whose dense autograd gradient is zero there), touched rows add the row-sparse data gradient.

Lazy-exact tables (default, ``lazy_tables=True``; SURVEY section 7 hard part 1(b)): an untouched row follows a
recurrence in its own (p, m, v) and the step constants only, so it is not advanced until somebody looks at it.
Every row carries a "valid through step" stamp; the rows a batch gathers are brought up to date just before the
forward reads them (``catch_up``), the rows that receive a data gradient are replayed and updated in ``step()``,
and everything else is replayed -- bit-identically to the dense pass, same fp32 operation sequence with the
recorded per-step constants -- when the table is *materialised*: ``state_dict()`` of the model or the optimizer,
``load_state_dict``, a direct call of ``model.x_embed`` / ``model.y_embed``, ``materialize_tables()``, or when
the constants' history ring (HIST_LEN steps) is about to wrap.  Reading ``model.x_embed.weight`` directly
between steps without one of these sees rows as of their last touch: call ``materialize_tables()`` first.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Callable, Dict, List, Optional

import torch

from . import _lib, ops
from ._lib import check

Tensor = torch.Tensor


# model -> weakref(the FusedAdam that owns its lazy position tables).  Not an attribute of the model: ``copy.deepcopy(model)`` must
# not inherit an owner (the copy is attached to its own optimizer).
_TABLE_OWNERS = weakref.WeakKeyDictionary()

HIST_LEN = 8192      # per-step Adam constants kept for the lazy tables' replay (power of two; 32 B per step)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-3, process_group=None, lazy_tables: bool = True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.process_group = process_group
        self.lazy_tables = bool(lazy_tables)
        self._lazy_flushed_at = 0                 # step count at which every table row was last current
        self._lazy_dirty = False                  # some row may be behind the step counter
        self.materialize_count = 0
        self._step_count = 0
        self._flat: Dict[int, dict] = {}          # group index -> flat buffers
        self._tables: Dict[int, dict] = {}        # id(param) -> table state
        self._where: Dict[int, tuple] = {}        # id(param) -> (group index, element offset in the flat buffer)
        self._sink: Optional[dict] = None
        self._pver: Dict[int, int] = {}           # id(param) -> autograd version at the last shadow refresh
        # device-resident step counter + derived constants per param group (csrc/adam.hip: mcl_adam_consts_update): nothing
        # step-dependent is baked into a launch, so step() can be captured in a HIP graph and replayed
        self._dev: Dict[int, dict] = {}
        self._began = False                       # this step's constants are already on the device (early table update)
        self._group_of: Dict[int, int] = {}       # id(param) -> group index
        self._hook_handles: list = []             # what attach_model() registered on the model (detach_model removes them)
        self._model_ref = None

    def __deepcopy__(self, memo):
        """``copy.deepcopy(model)`` reaches the optimizer through the hooks attach_model() registered (bound methods): the copied
        model's hooks keep pointing at THIS optimizer (they materialise this optimizer's tables -- harmless for the copy, which
        gets an optimizer of its own), never at a field-less clone (torch's Optimizer.__getstate__ drops every attribute added
        here)."""
        return self

    # ------------------------------------------------------------------ wiring
    def attach_model(self, model) -> "FusedAdam":
        """Registers the model's position tables for the row-sparse update (model.embedding_grad ==
        'rowsparse') and inherits its process group."""
        if getattr(model, "embedding_grad", "dense") == "rowsparse":
            self._sink = model.sparse_grads
            for key, emb in (("ix", model.x_embed), ("iy", model.y_embed)):
                self._tables[id(emb.weight)] = {"key": key, "param": emb.weight}
        if self.process_group is None:
            self.process_group = getattr(model, "process_group", None)
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                self._group_of[id(p)] = gi
        ref = _TABLE_OWNERS.get(model)
        prev = ref() if ref is not None else None
        if prev is not None and prev is not self:
            # a second optimizer for the same model: the first one's pending replays must land in the tables before this
            # one declares every row current, and its hooks must not pile up on the model
            prev.materialize_tables()
            prev.detach_model()
        self._model_ref = model
        if self._tables and self.lazy_tables:
            # the forward brings the rows it gathers up to date (model._spot_features); whoever reads a whole table
            # through the module API gets it materialised first.  Bound methods (picklable with the optimizer), handles
            # kept for detach_model().
            model._table_catchup = self.catch_up
            _TABLE_OWNERS[model] = weakref.ref(self)
            hs = self._hook_handles
            hs.append(model.register_state_dict_pre_hook(self._hook_state_dict_pre))
            if hasattr(model, "register_load_state_dict_pre_hook"):
                hs.append(model.register_load_state_dict_pre_hook(self._hook_any))
            for emb in (model.x_embed, model.y_embed):
                hs.append(emb.register_forward_pre_hook(self._hook_any))
                hs.append(emb.register_state_dict_pre_hook(self._hook_state_dict_pre))   # model.x_embed.state_dict()
        # The position tables can be updated as soon as their gradient rows exist -- from inside PosEmbedAddFn.backward, on
        # the side stream, under the latency-bound part of the image backbone's backward -- instead of in step().  That
        # changes what a bare ``loss.backward()`` does, so it is engine.TrainStep that switches it on, only while it captures
        # its single step graph (a backward that is always followed by step()).  Data parallel always updates in step()
        # (gathered rows of every rank).
        # load_state_dict() into a live model rewrites the flat fp32 buffer behind the bf16 shadows' back (captured
        # HIP graphs read the shadows directly, so the lazy per-parameter check in shadow() cannot catch that case)
        if hasattr(model, "register_load_state_dict_post_hook"):
            self._hook_handles.append(model.register_load_state_dict_post_hook(self._hook_loaded))
        # bf16 shadow weights for the fused backbone: one flat cast per step instead of ~120 per-weight casts
        from . import densenet_fused
        densenet_fused.set_weight_provider(self.shadow)
        return self

    def _hook_state_dict_pre(self, module, prefix, keep_vars):
        self.materialize_tables()

    def _hook_any(self, *args, **kwargs):
        self.materialize_tables()

    def _hook_loaded(self, module, incompatible):
        self._refresh_shadows()

    def detach_model(self) -> None:
        """Removes every hook attach_model() registered (the tables are materialised first)."""
        if self._lazy_active():
            self.materialize_tables()
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        m = self._model_ref
        if m is not None:
            ref = _TABLE_OWNERS.get(m)
            if ref is not None and ref() is self:
                del _TABLE_OWNERS[m]
            if getattr(m, "_table_catchup", None) == self.catch_up:
                m._table_catchup = None
        self._model_ref = None

    # ------------------------------------------------------------------ flat buffers
    def _build_flat(self, gi: int, group) -> None:
        ps = [p for p in group["params"] if p.grad is not None and id(p) not in self._tables
              and p.dtype == torch.float32 and p.is_cuda and not p.grad.is_sparse]
        if not ps:
            self._flat[gi] = {"params": [], "n": 0}
            return
        dev = ps[0].device
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4          # keep every segment 16-byte aligned
        flat_p = torch.zeros(n, device=dev, dtype=torch.float32)
        flat_g = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, o in zip(ps, offs):
            strides = p.stride()
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            if not dense:
                p.data = p.data.contiguous()
                p.grad = p.grad.contiguous()
                strides = p.stride()
            vp = flat_p.as_strided(p.shape, strides, o)
            vg = flat_g.as_strided(p.shape, strides, o)
            vp.copy_(p.data)
            vg.copy_(p.grad)        # copy_ is layout-aware: logical element (i,j,..) -> same logical slot
            p.data = vp
            p.grad = vg
            self._where[id(p)] = (gi, o)
        flat_m, flat_v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
        for p, o in zip(ps, offs):
            # moments that load_state_dict() parked in self.state before the flat buffers existed (a resumed run)
            stt = self.state.get(p)
            if stt and "exp_avg" in stt:
                flat_m.as_strided(p.shape, p.stride(), o).copy_(stt["exp_avg"])
                flat_v.as_strided(p.shape, p.stride(), o).copy_(stt["exp_avg_sq"])
                del self.state[p]
        self._flat[gi] = {"params": ps, "n": n, "p": flat_p, "g": flat_g, "m": flat_m, "v": flat_v}

    def ensure_flat(self) -> None:
        for gi, group in enumerate(self.param_groups):
            if gi not in self._flat:
                self._build_flat(gi, group)

    def flat_location(self, p: Tensor):
        """(group index, element offset) of a flat-managed parameter inside its group's flat buffers, or None."""
        return self._where.get(id(p))

    def flat_bucket(self, gi: int) -> Optional[Tensor]:
        """The flat gradient bucket of param group ``gi`` (None when the group has no flat-managed parameter)."""
        f = self._flat.get(gi)
        return f["g"] if f and f.get("n", 0) > 0 else None

    def flat_param_ids(self) -> set:
        return {id(p) for f in self._flat.values() for p in f["params"]}

    def shadow(self, p: Tensor, dtype: torch.dtype) -> Optional[Tensor]:
        """Low-precision view of parameter ``p`` inside a flat shadow copy of the flat fp32 buffer (same
        offsets/strides), refreshed by ONE cast kernel after every step.  None if ``p`` is not flat-managed."""
        loc = self._where.get(id(p))
        if loc is None:
            return None
        gi, off = loc
        f = self._flat[gi]
        key = "shadow_" + str(dtype)
        if key not in f:
            f[key] = f["p"].to(dtype)
            f.setdefault("shadow_keys", []).append((key, dtype))
            self._pver.update({id(q): q._version for q in f["params"]})
        v = f[key].as_strided(p.shape, p.stride(), off)
        if self._pver.get(id(p)) != p._version:
            # the parameter was written outside step() (p.copy_, load_state_dict without the model hook, an in-place
            # edit): its autograd version counter moved, the kernels' own writes do not move it -> re-cast this segment
            with torch.no_grad():
                for k2, _ in f.get("shadow_keys", []):
                    f[k2].as_strided(p.shape, p.stride(), off).copy_(p)
            self._pver[id(p)] = p._version
        return v

    @torch.no_grad()
    def _refresh_shadows(self, fused_bf16=()) -> None:
        """``fused_bf16``: group indices whose bf16 shadow was just written by the Adam kernel itself
        (mcl_adam_step_dev_shadow) -- no cast pass for those."""
        for gi, f in self._flat.items():
            if not f.get("shadow_keys"):
                continue
            for key, dt in f["shadow_keys"]:
                if gi in fused_bf16 and dt == torch.bfloat16:
                    continue
                f[key].copy_(f["p"])       # (outside the steady-state step: first use, load_state_dict, non-bf16 shadows)
            self._pver.update({id(q): q._version for q in f["params"]})

    refresh_shadows = _refresh_shadows      # public: call after editing parameters in place outside step()

    def flat_grads(self) -> List[Tensor]:
        """Flat gradient buffers (the data-parallel all-reduce buckets)."""
        return [f["g"] for f in self._flat.values() if f.get("n", 0) > 0]

    # ------------------------------------------------------------------ API
    def zero_grad(self, set_to_none: bool = True) -> None:
        flat_ids = set()
        for f in self._flat.values():
            if f.get("n", 0) > 0:
                if f["g"].is_cuda:      # own fill kernel (the captured step graph holds kernel nodes only)
                    check(_lib.lib().mcl_fill_zero(f["g"].data_ptr(), 4 * f["g"].numel(), ops._stream()), "mcl_fill_zero")
                else:
                    f["g"].zero_()
                flat_ids.update(id(p) for p in f["params"])
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) in flat_ids or p.grad is None:
                    continue
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def prefetch_table_rows(self) -> None:
        """Data parallel: run the (small) all-gather of the table gradient rows now, ahead of a large asynchronous
        all-reduce, so that the table update can overlap it."""
        if self._sink is not None and "dout" in self._sink:
            self._gathered()

    @torch.no_grad()
    def step(self, closure=None, wait=None):
        """``wait``: work handles of an asynchronous flat-gradient all-reduce (dist.GradReducer.reduce(async_flat=
        True)); the position tables -- which do not depend on it -- are updated first, then the handles are waited for
        and the flat bucket is updated."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        st = ops._stream()
        lazy = self._lazy_active()
        tables_pending = self._sink is not None and (self._sink.get("tables_done", False) or "dout" in self._sink)
        skipped_tables = lazy and not tables_pending and any(self.state.get(t["param"]) for t in self._tables.values())
        if skipped_tables:
            # a step in which the tables received no gradient at all: torch.optim.Adam skips a parameter whose .grad is
            # None, so no row may move -- every row is made current first and stamped with the new step count below
            self.materialize_tables()
        self._begin_step()
        if skipped_tables:
            for t in self._tables.values():
                if self.state.get(t["param"]):
                    self._row_step(t["param"]).fill_(self._step_count + 1)
            self._lazy_flushed_at = self._step_count + 1
        tables_done = self._sink is not None and self._sink.pop("tables_done", False)
        if lazy and tables_pending and not tables_done:
            self._tables_step_lazy(L, st)
            tables_done = True
        deferred = []
        for gi, group in enumerate(self.param_groups):
            if gi not in self._flat:
                self._build_flat(gi, group)
            f = self._flat[gi]
            flat_ids = {id(p) for p in f["params"]}
            if f["n"] > 0:
                deferred.append((gi, f))                       # after the tables / the pending all-reduce
            for p in group["params"]:
                if id(p) in flat_ids:
                    continue
                tab = self._tables.get(id(p))
                if tab is not None and self._sink is not None and (tables_done or "dout" in self._sink):
                    if not tables_done:
                        self._table_step(L, st, p, tab, gi)
                    continue
                if p.grad is None:
                    continue                        # torch.optim.Adam skips parameters without a gradient
                state = self.state[p]
                if not state:
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g = p.grad
                if g.stride() != p.stride():
                    g = torch.empty_like(p).copy_(g)   # same memory order as p (the update is elementwise)
                check(L.mcl_adam_step_dev(p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(),
                                          state["exp_avg_sq"].data_ptr(), p.numel(),
                                          self._dev_state(gi, p.device)["consts"].data_ptr(), st), "mcl_adam_step_dev")
        for h in (wait or []):
            h.wait()
        fused_shadow = []
        for gi, f in deferred:
            sh = f.get("shadow_" + str(torch.bfloat16))
            if sh is not None:
                # the update also writes the bf16 shadow the backbone kernels read: no cast pass after the step
                check(L.mcl_adam_step_dev_shadow(f["p"].data_ptr(), f["g"].data_ptr(), f["m"].data_ptr(), f["v"].data_ptr(),
                                                 f["n"], self._dev_state(gi, f["p"].device)["consts"].data_ptr(),
                                                 sh.data_ptr(), st), "mcl_adam_step_dev_shadow")
                fused_shadow.append(gi)
                continue
            check(L.mcl_adam_step_dev(f["p"].data_ptr(), f["g"].data_ptr(), f["m"].data_ptr(), f["v"].data_ptr(),
                                      f["n"], self._dev_state(gi, f["p"].device)["consts"].data_ptr(), st),
                  "mcl_adam_step_dev")
        if self._sink is not None:
            if self._sink.get("static"):
                # graph-captured backward: dout/ix/iy are static buffers refreshed by every replay
                for k in ("g_dout", "g_ix", "g_iy"):
                    self._sink.pop(k, None)
            else:
                hook = self._sink.get("hook")
                self._sink.clear()
                if hook is not None:
                    self._sink["hook"] = hook
        self._began = False
        self._step_count += 1
        self._refresh_shadows(fused_bf16=fused_shadow)
        return loss

    # ------------------------------------------------------------------ device-resident step state
    def _dev_state(self, gi: int, device) -> dict:
        d = self._dev.get(gi)
        if d is None:
            d = {"step": torch.full((1,), self._step_count, dtype=torch.int64, device=device),
                 "consts": torch.zeros(8, dtype=torch.float32, device=device),
                 # {lr, beta1, beta2, eps, weight_decay} as the kernels read them (device) and as last uploaded (host)
                 "hyper": torch.zeros(5, dtype=torch.float64, device=device), "hyper_host": None,
                 # the constants of the last HIST_LEN steps (slot = step mod HIST_LEN): what the lazy tables replay with
                 "hist": torch.zeros(8 * HIST_LEN, dtype=torch.float32, device=device)}
            self._dev[gi] = d
        return d

    def sync_hyper(self) -> None:
        """Upload the param groups' hyper-parameters to their device buffers when they changed (LR scheduler, manual
        ``param_groups[i]['lr'] = ...``).  mcl_adam_consts_update reads them from device memory, so a step captured
        in a HIP graph honours the change on its next replay; engine.TrainStep calls this before every replay.
        Never called while a stream is capturing (a host-to-device copy cannot be captured from pageable memory):
        ``_begin_step`` checks and raises instead."""
        for gi, group in enumerate(self.param_groups):
            dev = next((p.device for p in group["params"] if p.is_cuda), None)
            if dev is None:
                continue
            d = self._dev_state(gi, dev)
            b1, b2 = group["betas"]
            cur = (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]))
            if d["hyper_host"] != cur:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("FusedAdam: hyper-parameters changed inside a graph capture; call "
                                       "optimizer.sync_hyper() before capturing")
                # pinned staging ring + asynchronous copy: a per-iteration LR schedule must not put a host-synchronising
                # pageable copy in front of every replay (stream order ahead of the replay guarantees visibility); a slot is
                # reused only after the copy that read it has completed
                ring = d.setdefault("hyper_ring", [])
                if not ring:
                    for _ in range(4):
                        ring.append([torch.empty(5, dtype=torch.float64, pin_memory=True), None])
                    d["hyper_slot"] = 0
                slot = ring[d["hyper_slot"]]
                d["hyper_slot"] = (d["hyper_slot"] + 1) % len(ring)
                if slot[1] is not None:
                    slot[1].synchronize()
                for i, v in enumerate(cur):
                    slot[0][i] = v
                d["hyper"].copy_(slot[0], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(d["hyper"].device))
                slot[1] = ev
                d["hyper_host"] = cur

    def _begin_step(self) -> None:
        """Advance the device step counters and refresh the constants -- once per optimisation step, by whoever comes
        first: the early table update (inside backward) or step()."""
        if self._began:
            return
        self._began = True
        self.sync_hyper()
        self._flush_if_due()
        self._lazy_dirty = True
        L = _lib.lib()
        st = ops._stream()
        for gi, group in enumerate(self.param_groups):
            dev = next((p.device for p in group["params"] if p.is_cuda), None)
            if dev is None:
                continue
            d = self._dev_state(gi, dev)
            check(L.mcl_adam_consts_update_hist(d["step"].data_ptr(), d["consts"].data_ptr(), d["hyper"].data_ptr(),
                                                d["hist"].data_ptr(), HIST_LEN, st), "mcl_adam_consts_update_hist")

    def pre_replay(self) -> None:
        """Host-side work that must precede the replay of a captured step (engine.TrainStep): hyper-parameter upload and,
        every HIST_LEN - 2 steps, the materialisation that keeps every table row inside the constants' history ring."""
        self._lazy_dirty = True                  # the replay advances the step counter: every untouched row falls behind
        self.sync_hyper()
        self._flush_if_due()

    def _early_tables(self) -> None:
        """Called by ops.PosEmbedAddFn.backward once sink['dout'|'ix'|'iy'] exist (single process)."""
        if self._sink is None or "dout" not in self._sink:
            return
        L = _lib.lib()
        st = ops._stream()
        self._begin_step()
        if self._lazy_active():
            self._tables_step_lazy(L, st)
        else:
            for tab in self._tables.values():
                p = tab["param"]
                self._table_step(L, st, p, tab, self._group_of[id(p)])
        self._sink["tables_done"] = True

    # ------------------------------------------------------------------ lazy-exact tables
    def _lazy_active(self) -> bool:
        return self.lazy_tables and bool(self._tables) and self._sink is not None

    def _lazy_state(self, p) -> dict:
        state = self.state[p]
        if not state:
            state["exp_avg"] = torch.zeros_like(p)
            state["exp_avg_sq"] = torch.zeros_like(p)
        self._row_step(p)
        return state

    def _row_step(self, p) -> Tensor:
        """The int32 "valid through step" stamp per table row.  Kept OUTSIDE ``self.state``: torch's
        ``Optimizer.load_state_dict`` casts every state tensor of a floating-point parameter to the parameter's dtype, and
        the kernel would read a float's bits as a step number.  After ``materialize_tables()`` -- which
        ``state_dict()`` runs -- every row is current, so the stamps carry no information a checkpoint needs."""
        tab = self._tables[id(p)]
        rs = tab.get("row_step")
        if rs is None or rs.device != p.device:
            # everything is current as of now
            rs = tab["row_step"] = torch.full((p.shape[0],), self._step_count, device=p.device, dtype=torch.int32)
        return rs

    def _lazy_groups(self):
        """[(group index, [table params])] -- the tables of one param group share its step counter and history."""
        by = {}
        for tab in self._tables.values():
            p = tab["param"]
            by.setdefault(self._group_of.get(id(p), 0), []).append((tab["key"], p))
        return [(gi, [p for _, p in sorted(v, key=lambda kv: kv[0])], [k for k, _ in sorted(v, key=lambda kv: kv[0])])
                for gi, v in by.items()]

    def _lazy_launch(self, L, st, gi, ps, pos=None, owners=None, grads=None, ld_rg=0, n_owner=None) -> None:
        d = self._dev_state(gi, ps[0].device)
        sts = [self._lazy_state(p) for p in ps]
        for i in range(0, len(ps), 2):
            two = i + 1 < len(ps)
            p0, s0 = ps[i], sts[i]
            p1, s1 = (ps[i + 1], sts[i + 1]) if two else (None, None)
            own = owners[i:i + 2] if owners is not None else [None, None]
            gr = grads[i:i + 2] if grads is not None else [None, None]

            def ptr(t):
                return t.data_ptr() if t is not None else None
            check(L.mcl_adam_table_lazy(p0.data_ptr(), s0["exp_avg"].data_ptr(), s0["exp_avg_sq"].data_ptr(),
                                        self._row_step(p0).data_ptr(), ptr(p1), ptr(s1["exp_avg"]) if two else None,
                                        ptr(s1["exp_avg_sq"]) if two else None, ptr(self._row_step(p1)) if two else None,
                                        p0.shape[0], p0.shape[1], ptr(pos), ptr(own[0]), ptr(own[1]) if two else None,
                                        n_owner if n_owner is not None else p0.shape[0], ptr(gr[0]),
                                        ptr(gr[1]) if two else None, ld_rg, d["step"].data_ptr(), d["hist"].data_ptr(),
                                        HIST_LEN, st), "mcl_adam_table_lazy")

    def catch_up(self, pos: Tensor) -> None:
        """Bring the table rows a batch is about to gather up to the current step (pos: (B, 2) fp32 on the device, column 0
        indexes x_embed, column 1 y_embed).  Called by the model's forward in every mode; one small launch."""
        if not self._lazy_active() or not (self._lazy_dirty or torch.cuda.is_current_stream_capturing()):
            return                  # (a capture always records the launch: the flag describes the host's present, not a replay's)
        L, st = _lib.lib(), ops._stream()
        for gi, ps, keys in self._lazy_groups():
            if keys != ["ix", "iy"] or ps[0].shape != ps[1].shape:
                raise RuntimeError("lazy position tables expect x_embed and y_embed in one param group")
            self._lazy_launch(L, st, gi, ps, pos=pos, n_owner=pos.shape[0])

    def _tables_step_lazy(self, L, st) -> None:
        """This step's update of the rows that received a data gradient (both tables, one launch): each is replayed
        through the previous step and then updated with g = wd*p + its gradient row."""
        dout, ix, iy = self._gathered()
        for gi, ps, keys in self._lazy_groups():
            rss = [ops.embed_rowgrad(dout, ix if k == "ix" else iy) for k in keys]
            self._lazy_launch(L, st, gi, ps, owners=[r.owner_idx for r in rss], grads=[r.rows for r in rss],
                              ld_rg=rss[0].rows.stride(0), n_owner=rss[0].rows.shape[0])
        self._lazy_dirty = True

    @torch.no_grad()
    def materialize_tables(self) -> None:
        """Replay every row of the position tables up to the current step (bit-identical to the dense per-step update)."""
        if not self._lazy_active() or not self._lazy_dirty:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FusedAdam.materialize_tables() inside a graph capture")
        L, st = _lib.lib(), ops._stream()
        for gi, ps, keys in self._lazy_groups():
            ps = [p for p in ps if self.state.get(p)]
            if ps:
                self._lazy_launch(L, st, gi, ps)
        self._lazy_flushed_at = self._step_count
        self._lazy_dirty = False
        self.materialize_count += 1

    def _flush_if_due(self) -> None:
        if (self._lazy_active() and self._lazy_dirty and self._step_count + 1 - self._lazy_flushed_at >= HIST_LEN - 1
                and not torch.cuda.is_current_stream_capturing()):
            self.materialize_tables()

    def state_dict(self):
        """torch.optim.Optimizer.state_dict() layout (``exp_avg`` / ``exp_avg_sq`` per parameter) + ``mcl_step_count``.  As in
        torch the tensors are REFERENCES to the live state (``copy.deepcopy`` / ``torch.save`` it for a snapshot).  The
        flat-managed parameters' moments live in the flat buffers, not in ``self.state``: exported here as views of them."""
        self.materialize_tables()
        added = []
        for f in self._flat.values():
            for p in f.get("params", []):
                _, o = self._where[id(p)]
                self.state[p] = {"exp_avg": f["m"].as_strided(p.shape, p.stride(), o),
                                 "exp_avg_sq": f["v"].as_strided(p.shape, p.stride(), o)}
                added.append(p)
        try:
            sd = super().state_dict()
        finally:
            for p in added:
                del self.state[p]
        sd["mcl_step_count"] = self._step_count
        return sd

    def load_state_dict(self, state_dict):
        """Resume: the checkpoint's tables are materialised as of its step count (``state_dict()`` does that), so every row is
        stamped current at that count and the replay only ever needs constants recorded after the resume.  Tensors that a
        captured step graph may already point to (table moments, row stamps, flat moments) are overwritten IN PLACE."""
        state_dict = dict(state_dict)
        n = state_dict.pop("mcl_step_count", None)
        if self._lazy_active():
            self.materialize_tables()            # pending replays belong to the state that is about to be overwritten
        live = {}
        for tab in self._tables.values():
            stt = self.state.get(tab["param"])
            if stt:
                live[id(tab["param"])] = dict(stt)
        super().load_state_dict(state_dict)
        for tab in self._tables.values():
            p = tab["param"]
            stt = self.state.get(p)
            if not stt:
                continue
            for k in ("row_step", "row_slot"):      # checkpoints written while these still lived in the state
                stt.pop(k, None)
            for k, old in live.get(id(p), {}).items():
                if k in stt and torch.is_tensor(old) and old.shape == stt[k].shape:
                    old.copy_(stt[k])
                    stt[k] = old
        for f in self._flat.values():               # flat buffers already built: the loaded moments go into them
            for q in f.get("params", []):
                stt = self.state.get(q)
                if stt and "exp_avg" in stt:
                    _, o = self._where[id(q)]
                    f["m"].as_strided(q.shape, q.stride(), o).copy_(stt["exp_avg"])
                    f["v"].as_strided(q.shape, q.stride(), o).copy_(stt["exp_avg_sq"])
                    del self.state[q]
        # (not built yet -- a fresh optimizer: the moments stay parked in self.state until _build_flat() adopts them)
        if n is not None:
            self._step_count = int(n)
        for d in self._dev.values():
            d["step"].fill_(self._step_count)
            d["hyper_host"] = None                  # the loaded param_groups' hyper-parameters are uploaded again
        for tab in self._tables.values():
            if tab.get("row_step") is not None:
                tab["row_step"].fill_(self._step_count)
        self._lazy_flushed_at = self._step_count
        self._lazy_dirty = False
        self._began = False

    # ------------------------------------------------------------------ tables
    def _gathered(self):
        """(dout, ix, iy) over the global batch (identical on every rank) -- cached per step."""
        s = self._sink
        if "g_dout" in s:
            return s["g_dout"], s["g_ix"], s["g_iy"]
        dout, ix, iy = s["dout"], s["ix"], s["iy"]
        pg = self.process_group
        if pg is not None and torch.distributed.get_world_size(pg) > 1:
            from . import dist as mdist
            dout, ix, iy = mdist.gather_rows(dout, ix, iy, pg)
        s["g_dout"], s["g_ix"], s["g_iy"] = dout, ix, iy
        return dout, ix, iy

    def _table_step(self, L, st, p, tab, gi: int) -> None:
        state = self.state[p]
        if not state:
            state["exp_avg"] = torch.zeros_like(p)
            state["exp_avg_sq"] = torch.zeros_like(p)
        if tab.get("row_slot") is None or tab["row_slot"].device != p.device:
            # (outside self.state: load_state_dict would cast the int32 map to the parameter's dtype)
            tab["row_slot"] = torch.full((p.shape[0],), -1, device=p.device, dtype=torch.int32)
        dout, ix, iy = self._gathered()
        rs = ops.embed_rowgrad(dout, ix if tab["key"] == "ix" else iy)
        B = rs.rows.shape[0]
        slot = tab["row_slot"]
        check(L.mcl_row_slot_update(slot.data_ptr(), rs.owner_idx.data_ptr(), B, 1, st), "mcl_row_slot_update")
        check(L.mcl_adam_table_step_dev(p.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                        p.shape[0], p.shape[1], slot.data_ptr(), rs.rows.data_ptr(), rs.rows.stride(0),
                                        self._dev_state(gi, p.device)["consts"].data_ptr(), st),
              "mcl_adam_table_step_dev")
        check(L.mcl_row_slot_update(slot.data_ptr(), rs.owner_idx.data_ptr(), B, 0, st), "mcl_row_slot_update")
