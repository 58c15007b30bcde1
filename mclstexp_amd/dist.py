"""Data parallelism for the contrastive step: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm), gloo for CPU logic tests.

New capability relative to the reference (single ``cuda:0``, train.py:107).  Rank r owns
pairs [r*B_loc, (r+1)*B_loc).  Everything up to the (B_loc, P) embeddings is rank-local (so the spot
encoder attends over the LOCAL spots and BatchNorm uses LOCAL statistics -- SURVEY R2); the loss is
the reference's symmetric InfoNCE over the GLOBAL batch:

  1. ONE all-gather of [E_spot | E_img] (B_loc x 2P per rank; 256 KiB at B_loc=128 fp32);
  2. each rank forms its row strip  S_r = E_spot_loc E_img_all^T / T  (complete row LSEs) and its
     column strip  S_c = E_spot_all E_img_loc^T / T  (complete column LSEs): no partial-LSE merge;
  3. ONE all-gather of [row_lse | col_lse | diag] (3*B_loc floats) -> global loss on every rank and the
     LSE vectors the closed-form gradient needs;
  4. dE_spot_loc = dS_r E_img_all and dE_img_loc = dS_c^T E_spot_all are purely local: no reduce-scatter.

xGMI is point-to-point and both collectives are latency-bound (KBs), so the design minimises their
COUNT (two) rather than their bytes.  Parameter gradients are then summed (not averaged: dE already
carries the 1/(2*B_glob) of the global mean) with one all-reduce over FusedAdam's flat gradient
buffer; the two position tables exchange only the upstream gradient rows (B_glob x G) instead of a
dense 2 x 65536 x G all-reduce.

The collective plumbing is device-agnostic; the arithmetic goes through a small primitive set that
defaults to the HIP kernels (``HipPrims``).  CPU tests inject oracle-backed primitives.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as td

Tensor = torch.Tensor


# --------------------------------------------------------------------------- process group
def init_from_env() -> Tuple[Optional[td.ProcessGroup], int, int]:
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (group, rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    # MCL_FORCE_DIST=1: build the (size-1) RCCL group anyway, so that the complete data-parallel code path --
    # collectives, reducer, gathered table rows -- can be exercised and profiled on a single-GPU box
    if world <= 1 and os.environ.get("MCL_FORCE_DIST", "0") != "1":
        return None, 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost", "::1"):
        # single node over the loopback rendezvous: the container's hostname may not resolve, so do not let gloo (size
        # exchange of ragged shards) or the RCCL bootstrap pick an interface by hostname
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    if not td.is_initialized():
        backend = os.environ.get("MCL_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        td.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return td.group.WORLD, rank, world


def shutdown() -> None:
    if td.is_available() and td.is_initialized():
        td.destroy_process_group()


def _host_staged(x: Tensor, pg) -> bool:
    """gloo has no device collectives for every op: GPU tensors are staged through the host.  Only the
    single-GPU multi-process correctness test uses gloo with GPU tensors; production is RCCL."""
    return x.is_cuda and td.get_backend(pg) == "gloo"


def _all_gather_cat(x: Tensor, pg) -> Tensor:
    """Concatenation over ranks along dim 0 (equal shapes on every rank)."""
    x = x.contiguous()
    world = td.get_world_size(pg)
    if _host_staged(x, pg):
        xc = x.cpu()
        outc = torch.empty((world * xc.shape[0],) + tuple(xc.shape[1:]), dtype=xc.dtype)
        td.all_gather_into_tensor(outc, xc, group=pg)
        return outc.to(x.device)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    td.all_gather_into_tensor(out, x, group=pg)
    return out


# --------------------------------------------------------------------------- ragged shards
# The reference's DataLoader has no drop_last (train.py:49): the last batch of an epoch is short, and under data
# parallelism the ranks may then hold DIFFERENT numbers of pairs.  Shard sizes are agreed on the HOST every step
# (``SizeExchange``: a tiny gloo all-gather that never touches the GPU stream, so the launch-ahead of the step is
# kept) and handed to the collectives below, which pad to the largest shard, gather, and drop the padding rows.
_step_sizes: Optional[List[int]] = None


def set_step_sizes(sizes: Optional[Sequence[int]]) -> None:
    """Shard sizes (pairs per rank, rank order) of the CURRENT step; None = equal shards (the fast path)."""
    global _step_sizes
    _step_sizes = None if sizes is None or len(set(sizes)) == 1 else [int(v) for v in sizes]


def step_sizes() -> Optional[List[int]]:
    return _step_sizes


class SizeExchange:
    """Host-side agreement on the per-rank batch sizes of a step (one int per rank over a gloo group)."""

    def __init__(self, pg):
        self.pg = pg
        self.world = td.get_world_size(pg)
        self._cpu_pg = None

    def _group(self):
        if self._cpu_pg is None:
            # a gloo group over exactly the ranks of ``pg`` (which may be a subgroup of the world; like every
            # new_group call this is collective over the WORLD: all processes must construct their SizeExchange)
            self._cpu_pg = (self.pg if td.get_backend(self.pg) == "gloo"
                            else td.new_group(ranks=td.get_process_group_ranks(self.pg), backend="gloo"))
        return self._cpu_pg

    def __call__(self, b_loc: int) -> List[int]:
        if self.world == 1:
            return [int(b_loc)]
        mine = torch.tensor([int(b_loc)], dtype=torch.int64)
        out = torch.empty((self.world,), dtype=torch.int64)
        td.all_gather_into_tensor(out, mine, group=self._group())
        return [int(v) for v in out.tolist()]


def _all_gather_rows(x: Tensor, pg, sizes: Optional[Sequence[int]] = None) -> Tensor:
    """Rank-ordered concatenation along dim 0; ``sizes`` = rows per rank when the shards are ragged."""
    if sizes is None:
        return _all_gather_cat(x, pg)
    world, rank = td.get_world_size(pg), td.get_rank(pg)
    if len(sizes) != world or x.shape[0] != sizes[rank]:
        raise RuntimeError(f"ragged all-gather: this rank holds {x.shape[0]} rows, sizes say {list(sizes)}")
    cap = max(sizes)
    if cap == 0:
        return x.new_empty((0,) + tuple(x.shape[1:]))
    pad = x.new_zeros((cap,) + tuple(x.shape[1:]))
    pad[: x.shape[0]] = x
    full = _all_gather_cat(pad, pg).view((world, cap) + tuple(x.shape[1:]))
    return torch.cat([full[r, : sizes[r]] for r in range(world)], dim=0)


def _all_reduce_sum(x: Tensor, pg) -> None:
    if _host_staged(x, pg):
        xc = x.cpu()
        td.all_reduce(xc, op=td.ReduceOp.SUM, group=pg)
        x.copy_(xc)
        return
    td.all_reduce(x, op=td.ReduceOp.SUM, group=pg)


# --------------------------------------------------------------------------- arithmetic primitives
class HipPrims:
    """The gfx950 kernels (ops.py) behind the five primitives the DP InfoNCE needs."""

    @staticmethod
    def logits(a: Tensor, b: Tensor, inv_t: float) -> Tensor:
        from . import ops
        R, P = a.shape
        Cn = b.shape[0]
        S = torch.empty((R, Cn), device=a.device, dtype=torch.float32)
        ops.gemm_raw(R, Cn, P, 1, a, a.stride(0), 1, 0, b, 1, b.stride(0), 0, S, Cn, 0, alpha=inv_t)
        return S

    @staticmethod
    def row_lse(S: Tensor) -> Tensor:
        from . import _lib, ops
        out = torch.empty((S.shape[0],), device=S.device, dtype=torch.float32)
        _lib.check(_lib.lib().mcl_infonce_lse(S.data_ptr(), S.stride(0), S.shape[0], S.shape[1], out.data_ptr(), None,
                                              ops._stream()), "mcl_infonce_lse")
        return out

    @staticmethod
    def col_lse(S: Tensor) -> Tensor:
        from . import _lib, ops
        out = torch.empty((S.shape[1],), device=S.device, dtype=torch.float32)
        _lib.check(_lib.lib().mcl_infonce_lse(S.data_ptr(), S.stride(0), S.shape[0], S.shape[1], None, out.data_ptr(),
                                              ops._stream()), "mcl_infonce_lse")
        return out

    @staticmethod
    def dlogits(S: Tensor, row_lse: Tensor, col_lse: Tensor, row0: int, col0: int, coef: float) -> Tensor:
        from . import _lib, ops
        dS = torch.empty_like(S)
        _lib.check(_lib.lib().mcl_infonce_dlogits(S.data_ptr(), S.stride(0), row_lse.data_ptr(), col_lse.data_ptr(),
                                                  S.shape[0], S.shape[1], row0, col0, coef, dS.data_ptr(),
                                                  dS.stride(0), ops._stream()), "mcl_infonce_dlogits")
        return dS

    @staticmethod
    def mm_nn(dS: Tensor, e: Tensor) -> Tensor:
        from . import ops
        R, Cn = dS.shape
        P = e.shape[1]
        out = torch.empty((R, P), device=dS.device, dtype=torch.float32)
        ops.gemm_raw(R, P, Cn, 1, dS, dS.stride(0), 1, 0, e, e.stride(0), 1, 0, out, P, 0)
        return out

    @staticmethod
    def mm_tn(dS: Tensor, e: Tensor) -> Tensor:
        from . import ops
        R, Cn = dS.shape
        P = e.shape[1]
        out = torch.empty((Cn, P), device=dS.device, dtype=torch.float32)
        ops.gemm_raw(Cn, P, R, 1, dS, 1, dS.stride(0), 0, e, e.stride(0), 1, 0, out, P, 0)
        return out


def dist_infonce_fwd_bwd(e_spot: Tensor, e_img: Tensor, temperature: float, pg, prims=HipPrims,
                         sizes: Optional[Sequence[int]] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Global symmetric InfoNCE from per-rank embeddings.  Returns (loss [identical on every rank],
    dE_spot_loc, dE_img_loc, S_row_strip).  ``sizes`` (default: the step's agreed sizes, ``set_step_sizes``): pairs
    per rank when the shards are ragged."""
    world, rank = td.get_world_size(pg), td.get_rank(pg)
    sizes = _step_sizes if sizes is None else (None if len(set(sizes)) == 1 else list(sizes))
    b_loc, P = e_spot.shape
    b_glob = world * b_loc if sizes is None else sum(sizes)
    row0 = rank * b_loc if sizes is None else sum(sizes[:rank])
    all_e = _all_gather_rows(torch.cat([e_spot, e_img], dim=1), pg, sizes)   # collective 1: (B_glob, 2P)
    es_all, ei_all = all_e[:, :P], all_e[:, P:]
    inv_t = 1.0 / temperature
    s_rows = prims.logits(e_spot.contiguous(), ei_all, inv_t)                # (B_loc, B_glob)
    s_cols = prims.logits(es_all, e_img.contiguous(), inv_t)                 # (B_glob, B_loc)
    rl = prims.row_lse(s_rows)
    cl = prims.col_lse(s_cols)
    idx = torch.arange(b_loc, device=e_spot.device)
    diag = s_rows[idx, row0 + idx]
    packed = _all_gather_rows(torch.stack([rl, cl, diag], dim=1), pg, sizes)  # collective 2: (B_glob, 3)
    rl_all = packed[:, 0].contiguous()
    cl_all = packed[:, 1].contiguous()
    diag_all = packed[:, 2]
    loss = ((rl_all - diag_all).sum() + (cl_all - diag_all).sum()) / (2.0 * b_glob)
    coef = 1.0 / (2.0 * b_glob * temperature)
    ds_rows = prims.dlogits(s_rows, rl, cl_all, row0, 0, coef)
    ds_cols = prims.dlogits(s_cols, rl_all, cl, 0, row0, coef)
    d_es = prims.mm_nn(ds_rows, ei_all)
    d_ei = prims.mm_tn(ds_cols, es_all)
    return loss, d_es, d_ei, s_rows


# --------------------------------------------------------------------------- fused (bf16, logits never in HBM)
class HipFusedPrims:
    """csrc/infonce_fused.hip behind the three primitives of the fused data-parallel InfoNCE."""

    @staticmethod
    def cast(x: Tensor) -> Tensor:
        from . import ops
        return ops.cast_bf16(x)

    @staticmethod
    def lse(a16: Tensor, b16: Tensor, inv_t: float, diag_off: int) -> Tuple[Tensor, Tensor]:
        from . import ops
        return ops.infonce_fused_lse(a16, b16, inv_t, diag_off)

    @staticmethod
    def grad(a16: Tensor, b16: Tensor, inv_t: float, lse_a: Tensor, lse_b: Tensor, coef: float, diag_off: int) -> Tensor:
        from . import ops
        return ops.infonce_fused_grad(a16, b16, inv_t, lse_a, lse_b, coef, diag_off)


class HipFp8Prims:
    """fp8 similarity contraction under data parallelism: the all-gather ships the packed e4m3 rows (256 bytes + the
    scale byte, 272 B per embedding instead of 1024 B fp32 / 512 B bf16); LSEs on the fp8 MFMA kernel, gradients on the
    bf16 strip kernel over the (exact) dequantised copies."""
    width = 272

    @staticmethod
    def cast(x: Tensor) -> Tensor:
        from . import ops
        return ops.quant_e4m3(x, want_deq=False)[0]

    @staticmethod
    def lse(a8: Tensor, b8: Tensor, inv_t: float, diag_off: int) -> Tuple[Tensor, Tensor]:
        from . import ops
        a8c, b8c = a8.contiguous(), b8.contiguous()
        lse = ops.infonce_fp8_lse(a8c, b8c, inv_t)
        diag = ops.infonce_rowdot(ops.dequant_e4m3(a8c), ops.dequant_e4m3(b8c), inv_t, diag_off)
        return lse, diag

    @staticmethod
    def grad(a8: Tensor, b8: Tensor, inv_t: float, lse_a: Tensor, lse_b: Tensor, coef: float, diag_off: int) -> Tensor:
        from . import ops
        return ops.infonce_fused_grad(ops.dequant_e4m3(a8.contiguous()), ops.dequant_e4m3(b8.contiguous()), inv_t,
                                      lse_a, lse_b, coef, diag_off)


def dist_infonce_fused_fwd_bwd(e_spot: Tensor, e_img: Tensor, temperature: float, pg, prims=HipFusedPrims,
                               sizes: Optional[Sequence[int]] = None) -> Tuple[Tensor, Tensor, Tensor, None]:
    """Same global symmetric InfoNCE as ``dist_infonce_fwd_bwd`` on the fused kernels: the embeddings are rounded
    to bf16 (or e4m3, ``prims``) BEFORE the all-gather (half / a quarter of the bytes on xGMI), each rank runs the strip
    kernel in both orientations (own spots vs all images: row LSEs + dE_spot; own images vs all spots: column LSEs +
    dE_img) and no logits strip is ever written.  Still two collectives: [E_spot | E_img], then [row_lse | col_lse |
    diag].  ``sizes``: pairs per rank for ragged shards (default: ``set_step_sizes``)."""
    world, rank = td.get_world_size(pg), td.get_rank(pg)
    sizes = _step_sizes if sizes is None else (None if len(set(sizes)) == 1 else list(sizes))
    b_loc, P = e_spot.shape
    b_glob = world * b_loc if sizes is None else sum(sizes)
    doff = rank * b_loc if sizes is None else sum(sizes[:rank])
    inv_t = 1.0 / temperature
    loc16 = torch.cat([prims.cast(e_spot), prims.cast(e_img)], dim=1)       # (B_loc, 2W) low precision
    all16 = _all_gather_rows(loc16, pg, sizes)                               # collective 1: (B_glob, 2W)
    Wc = getattr(prims, "width", P)                                          # columns of one embedding in cast form
    es_loc, ei_loc = loc16[:, :Wc], loc16[:, Wc:]                            # row stride 2W: read in place
    es_all, ei_all = all16[:, :Wc], all16[:, Wc:]
    rl, diag = prims.lse(es_loc, ei_all, inv_t, doff)                        # rows of S owned by this rank
    cl, _ = prims.lse(ei_loc, es_all, inv_t, doff)                           # columns of S owned by this rank
    packed = _all_gather_rows(torch.stack([rl, cl, diag], dim=1), pg, sizes)  # collective 2: (B_glob, 3)
    rl_all = packed[:, 0].contiguous()
    cl_all = packed[:, 1].contiguous()
    diag_all = packed[:, 2]
    loss = ((rl_all - diag_all).sum() + (cl_all - diag_all).sum()) / (2.0 * b_glob)
    coef = inv_t / (2.0 * b_glob)
    d_es = prims.grad(es_loc, ei_all, inv_t, rl, cl_all, coef, doff)
    d_ei = prims.grad(ei_loc, es_all, inv_t, cl, rl_all, coef, doff)
    return loss, d_es, d_ei, None


def dist_infonce_auto(e_spot: Tensor, e_img: Tensor, temperature: float, pg, fused):
    """Routes the global InfoNCE like the single-process ``ops.InfoNCEFn``: ``fused`` False -> exact fp32 kernels,
    "fp8" -> e4m3 exchange + fp8 similarity, True -> the flash-style bf16 kernels from a GLOBAL batch of
    ``ops.FUSED_MIN_BATCH`` pairs, the exact kernels below it (same rule as one process at that batch)."""
    if fused == "fp8":
        return dist_infonce_fused_fwd_bwd(e_spot, e_img, temperature, pg, prims=HipFp8Prims)
    if fused:
        from . import ops
        glob = sum(_step_sizes) if _step_sizes is not None else e_spot.shape[0] * td.get_world_size(pg)
        fused = glob >= ops.FUSED_MIN_BATCH
    fn = dist_infonce_fused_fwd_bwd if fused else dist_infonce_fwd_bwd
    return fn(e_spot, e_img, temperature, pg)


class DistInfoNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e_spot, e_img, temperature, pg, stash, fused=False):
        loss, d_es, d_ei, s_rows = dist_infonce_auto(e_spot, e_img, temperature, pg, fused)
        if stash is not None:
            stash["logits"] = s_rows
        ctx.save_for_backward(d_es, d_ei)
        return loss

    @staticmethod
    def backward(ctx, gl):
        d_es, d_ei = ctx.saved_tensors
        return d_es * gl, d_ei * gl, None, None, None, None


# --------------------------------------------------------------------------- gradient exchange
def gather_rows(dout: Tensor, ix: Tensor, iy: Tensor, pg, sizes: Optional[Sequence[int]] = None
                ) -> Tuple[Tensor, Tensor, Tensor]:
    """Sparse exchange for the two position tables: both share one upstream gradient (B_loc, G), so a
    single all-gather of it (+ the two index vectors) replaces a dense 2 x (65536, G) all-reduce
    (4 MB vs 524 MB at B_glob=1024, G=1000).  Every rank then reduces the identical global rows."""
    sizes = _step_sizes if sizes is None else (None if len(set(sizes)) == 1 else list(sizes))
    g_dout = _all_gather_rows(dout, pg, sizes)
    g_idx = _all_gather_rows(torch.stack([ix, iy], dim=1), pg, sizes)
    return g_dout, g_idx[:, 0].contiguous(), g_idx[:, 1].contiguous()


class _WireHandle:
    """Work handle of a gradient all-reduce that travelled in a narrower wire format: ``wait()`` waits for the collective
    and writes the summed values back into the fp32 gradient slice."""

    def __init__(self, work, dst: Tensor, wire: Tensor):
        self.work, self.dst, self.wire = work, dst, wire

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.dst.copy_(self.wire)


class GradReducer:
    """Sum parameter gradients over ranks: the flat FusedAdam bucket (the gradients already live contiguously there) in
    ``buckets`` contiguous chunks, each its own collective, plus stragglers that are not in a flat bucket.

    ``buckets`` (MCL_GRAD_BUCKETS, default 1): several collectives in flight let RCCL pipeline the ring over xGMI; the
    result is independent of the chunking (every element is still summed once over the same ranks).
    ``wire`` (MCL_GRAD_WIRE, default "fp32"): "bf16" sends the gradient as bf16 -- half the bytes of the one large
    message of the step (63 MB fp32 with DenseNet-121: ring time over 7 x 153 GB/s xGMI links 0.7 -> 0.35 ms) at bf16
    rounding of the summands and of the sum (the usual DDP compression hook trade)."""

    def __init__(self, pg, buckets: Optional[int] = None, wire: Optional[str] = None):
        self.pg = pg
        self.buckets = max(1, int(os.environ.get("MCL_GRAD_BUCKETS", "1")) if buckets is None else int(buckets))
        self.wire = (os.environ.get("MCL_GRAD_WIRE", "fp32") if wire is None else wire).lower()
        if self.wire not in ("fp32", "bf16"):
            raise ValueError("GradReducer wire format must be 'fp32' or 'bf16'")

    def _chunks(self, g: Tensor) -> List[Tensor]:
        n = g.numel()
        if self.buckets == 1 or n < 4 * self.buckets:
            return [g]
        step = ((n + self.buckets - 1) // self.buckets + 3) // 4 * 4          # 16-byte aligned chunk starts
        return [g[o:min(n, o + step)] for o in range(0, n, step)]

    def _reduce_chunk(self, c: Tensor, async_flat: bool, handles: list) -> None:
        if self.wire == "bf16":
            w = c.to(torch.bfloat16)
            if async_flat and not _host_staged(w, self.pg):
                handles.append(_WireHandle(td.all_reduce(w, op=td.ReduceOp.SUM, group=self.pg, async_op=True), c, w))
            else:
                _all_reduce_sum(w, self.pg)
                c.copy_(w)
        elif async_flat and not _host_staged(c, self.pg):
            handles.append(td.all_reduce(c, op=td.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            _all_reduce_sum(c, self.pg)

    def reduce_range(self, g: Tensor, lo: int, hi: int, async_flat: bool = True) -> list:
        """All-reduce (SUM) of the element range [lo, hi) of a flat gradient bucket as ONE collective -- issued by
        engine.TrainStep as soon as the backward segment that produces the range has been enqueued, so that it runs on the
        communicator's stream beside the next segment's kernels.  Returns the work handles (empty when it ran synchronously).
        Every element is summed once over the same ranks, whatever the partition into ranges."""
        handles: list = []
        if hi > lo:
            self._reduce_chunk(g[lo:hi], async_flat, handles)
        return handles

    def reduce(self, optimizer, async_flat: bool = False, skip_flat=()):
        """Sums gradients over ranks.  ``async_flat``: the (large) flat-bucket all-reduce is only ENQUEUED and its
        work handles are returned -- the caller overlaps it with independent work (the position-table update, which
        needs no flat gradient) and waits before the flat Adam step: ``optimizer.step(wait=handles)``.  The small
        table-row all-gather is issued first so that it does not queue behind the big all-reduce on the
        communicator's stream.  ``skip_flat``: data pointers of flat buckets that were already reduced range by range
        (``reduce_range``)."""
        if hasattr(optimizer, "ensure_flat"):
            optimizer.ensure_flat()
        if async_flat and hasattr(optimizer, "prefetch_table_rows"):
            optimizer.prefetch_table_rows()
        flat_ids = set()
        handles = []
        if hasattr(optimizer, "flat_grads"):
            for g in optimizer.flat_grads():
                if g.data_ptr() in skip_flat:
                    continue
                for c in self._chunks(g):
                    self._reduce_chunk(c, async_flat, handles)
            flat_ids = optimizer.flat_param_ids()
        for group in optimizer.param_groups:
            for p in group["params"]:
                if id(p) in flat_ids or p.grad is None:
                    continue
                _all_reduce_sum(p.grad, self.pg)
        return handles
