"""Drop-in model classes for mclSTExp's contrastive hot path (mirrors /root/reference/model.py).

Same class names, constructor signatures, attribute names and ``state_dict`` keys as the reference,
so ``train.py``-style loops and the ``evel_*.py`` sub-module calls (evel_her2st.py:48-69) work
unchanged; underneath, every operation outside the image backbone is a hand-written gfx950 kernel
reached through the C ABI (``ops.py``).  Inputs must live on the GPU -- there is no CPU path.

Differences from the reference, all deliberate:
  * an unknown ``encoder_name`` raises (the reference silently builds a model without
    ``image_encoder``, model.py:206-215); ``"identity"`` is an extension meaning ``batch["image"]``
    already holds (B, image_dim) features;
  * ``torch.eye(B).cuda()`` (model.py:243) is never built: the closed-form InfoNCE kernel uses the
    diagonal directly;
  * keyword-only extras: ``compute`` ("f32" exact / "bf16" MFMA), ``backbone_dtype`` (autocast dtype
    for the delegated backbone), ``embedding_grad`` ("dense" = stock-optimizer compatible,
    "rowsparse" = hand the 2 x (65536, G) tables' gradients to FusedAdam as touched rows only),
    ``process_group`` (data-parallel global InfoNCE), ``infonce`` ("exact" = fp32 logits through the
    GEMM + LSE kernels, the reference's numerics; "fused" = flash-style bf16 MFMA kernel that never writes the
    B x B logits, csrc/infonce_fused.hip).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import backbones, ops
from .backbones import (ENCODERS, ImageEncdoer_res18, ImageEncdoer_res101, ImageEncoder,  # noqa: F401
                        ImageEncoder_Resnet, ImageEncoder_VIT)

Tensor = torch.Tensor


class _LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) [gelu] through mcl_gemm, with hand-written backward."""

    @staticmethod
    def forward(ctx, x, W, b, gelu):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        y, pre = ops.linear_fwd(x2, W, b, gelu=gelu, save_pre=gelu)
        ctx.save_for_backward(x2, W, pre if gelu else None)
        ctx.has_bias, ctx.shp = b is not None, shp
        return y.view(*shp[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, W, pre = ctx.saved_tensors
        dy2 = dy.reshape(-1, W.shape[0])
        if pre is not None:
            dy2 = ops.gelu_bwd(dy2, pre)
        dW = ops.linear_bwd_weight(dy2, x2)
        db = ops.colsum(dy2) if ctx.has_bias else None
        dx = ops.linear_bwd_data(dy2, W).view(ctx.shp) if ctx.needs_input_grad[0] else None
        return dx, dW, db, None


def _dropout(mod: nn.Dropout, x: Tensor) -> Tensor:
    """nn.Dropout as an own kernel when it is active (training, p > 0; never in the reference: model.py:217), else identity."""
    if mod.training and mod.p > 0:
        return ops.DropoutFn.apply(x, mod.p)
    return x


class _AttnCoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads, dim_head):
        out, aux = ops.attention_core_fwd(qkv, heads, dim_head)
        ctx.save_for_backward(qkv, out, aux)
        ctx.hd = (heads, dim_head)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, aux = ctx.saved_tensors
        return ops.attention_core_bwd(dout, qkv, out, aux, *ctx.hd), None, None


def _linear(m: nn.Linear, x: Tensor, gelu: bool = False) -> Tensor:
    return _LinearFn.apply(x, m.weight, m.bias, gelu)


def _layer_norm(m: nn.LayerNorm, x: Tensor) -> Tensor:
    return ops.LayerNormFn.apply(x, m.weight, m.bias, m.eps)


class PreNorm(nn.Module):
    """model.py:10-17."""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(_layer_norm(self.norm, x), **kwargs)


class FeedForward(nn.Module):
    """model.py:20-32."""

    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))

    def forward(self, x):
        h = _dropout(self.net[2], _linear(self.net[0], x, gelu=True))
        return _dropout(self.net[4], _linear(self.net[3], h))


class Attention(nn.Module):
    """model.py:35-57 (batch-as-sequence: x is (1, B, G) or (B, G))."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout)) if project_out else nn.Identity()

    def forward(self, x):
        shp = x.shape
        if x.dim() == 3 and shp[0] != 1:
            return torch.cat([self.forward(xi.unsqueeze(0)) for xi in x], 0)
        x2 = x.reshape(-1, shp[-1])
        qkv = _linear(self.to_qkv, x2)
        out = _AttnCoreFn.apply(qkv, self.heads, self.dim_head)
        if isinstance(self.to_out, nn.Identity):
            return out.view(*shp[:-1], -1)
        out = _dropout(self.to_out[1], _linear(self.to_out[0], out))
        return out.view(*shp[:-1], -1)


class attn_block(nn.Module):
    """model.py:60-69.  ``forward`` runs the whole layer as one hand-scheduled forward/backward
    (ops.AttnBlockFn) when its dropouts are inactive -- which is always the case in the reference
    (model.py:217 hard-codes dropout=0.)."""

    def __init__(self, dim, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.attn = PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout))
        self.ff = PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout))
        self._p = dropout

    def forward(self, x):
        a, f = self.attn.fn, self.ff.fn
        fused_ok = (not (self.training and self._p > 0)) and not isinstance(a.to_out, nn.Identity)
        if not fused_ok:                                 # dropout > 0: the unfused sequence, every op an own kernel
            x = ops.AddFn.apply(self.attn(x), x)
            return ops.AddFn.apply(self.ff(x), x)
        shp = x.shape
        if x.dim() == 3 and shp[0] != 1:
            return torch.cat([self.forward(xi.unsqueeze(0)) for xi in x], 0)
        y = ops.AttnBlockFn.apply(x.reshape(-1, shp[-1]), self.attn.norm.weight, self.attn.norm.bias,
                                  a.to_qkv.weight, a.to_out[0].weight, a.to_out[0].bias,
                                  self.ff.norm.weight, self.ff.norm.bias,
                                  f.net[0].weight, f.net[0].bias, f.net[3].weight, f.net[3].bias,
                                  a.heads, a.dim_head)
        return y.view(shp)


class ProjectionHead(nn.Module):
    """model.py:151-168."""

    def __init__(self, embedding_dim, projection_dim, dropout=0.):
        super().__init__()
        self.projection = nn.Linear(embedding_dim, projection_dim)
        self.gelu = nn.GELU()
        self.fc = nn.Linear(projection_dim, projection_dim)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(projection_dim)

    def forward(self, x):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if self.training and self.dropout.p > 0:
            projected = _linear(self.projection, x2)
            y = ops.AddFn.apply(_dropout(self.dropout, _linear(self.fc, ops.GeluFn.apply(projected))), projected)
            y = _layer_norm(self.layer_norm, y)
        else:
            y = ops.ProjectionHeadFn.apply(x2, self.projection.weight, self.projection.bias, self.fc.weight,
                                           self.fc.bias, self.layer_norm.weight, self.layer_norm.bias)
        return y.view(*shp[:-1], -1)


class _ContrastiveBase(nn.Module):
    """Shared forward tail of mclSTExp_Attention / mclSTExp_MLP (model.py:187-198, 225-247)."""

    def _init_common(self, temperature, compute, backbone_dtype, embedding_grad, process_group, infonce="exact"):
        self.temperature = temperature
        if infonce not in ("exact", "fused", "fp8"):
            raise ValueError("infonce must be 'exact' (fp32 logits, reference numerics), 'fused' (bf16 MFMA, logits "
                             "never written to HBM) or 'fp8' (e4m3 similarity contraction, BASELINE configs[4])")
        self.infonce = infonce
        if embedding_grad not in ("dense", "rowsparse"):
            raise ValueError("embedding_grad must be 'dense' or 'rowsparse'")
        self.compute = compute
        self.backbone_dtype = backbone_dtype
        self.embedding_grad = embedding_grad
        self.process_group = process_group
        self.fused_backbone = True      # DenseNet-121: concat-free BN+ReLU kernels (densenet_fused.py)
        self.segment_backward = ()      # dense blocks whose input cuts the backbone's backward (data-parallel overlap, engine.py)
        self.backward_cuts = None
        self.capture = False
        self.last: Dict[str, Tensor] = {}
        self.sparse_grads: Dict[str, ops.RowSparseGrad] = {}

    # The spot branch (embedding add, spot Transformer, projection head: ~85 small latency-bound launches forward +
    # backward) shares nothing with the image branch until the loss: run it on a side stream so it overlaps the
    # DenseNet kernels.  Autograd replays each branch's backward on the stream its forward ran on, and under HIP-graph
    # capture the fork/join becomes a parallel branch of both graphs.
    overlap_branches = True           # (class attribute: bench.py switches it off for its per-kernel timing pass)
    _branch_streams: Dict[int, "torch.cuda.Stream"] = {}

    def _branch_stream(self, device) -> Optional["torch.cuda.Stream"]:
        if not (self.overlap_branches and device.type == "cuda"):
            return None
        if (self.fused_backbone
                and isinstance(getattr(self, "image_encoder", None), (backbones.ImageEncoder, backbones.ImageEncoder_VIT))
                and self.backbone_dtype == torch.bfloat16):
            # ONE side stream for everything off the critical chain.  A replayed graph runs on two hardware queues here:
            # during the backward they are the main chain and the backbone's weight-gradient stream, and a third branch
            # is multiplexed onto the MAIN one (a pure delay in the spot backward showed up 1:1 in the step time).  On
            # the weight-gradient stream -- ahead of that work, see embed() -- the spot backward rides in its slack.
            from . import densenet_fused
            return densenet_fused._side_stream(device)
        s = _ContrastiveBase._branch_streams.get(device.index)
        if s is None:
            s = torch.cuda.Stream(device=device)
            _ContrastiveBase._branch_streams[device.index] = s
        return s

    def _encode_image(self, encoder: nn.Module, image: Tensor) -> Tensor:
        if isinstance(encoder, nn.Identity):
            return image
        if self.fused_backbone and isinstance(encoder, backbones.ImageEncoder) and encoder.training and image.is_cuda:
            # segment_backward (set by engine.TrainStep while it captures the data-parallel step): the backbone's backward is
            # cut at the dense-block inputs; the (upstream tensor, slot) pairs are left in self.backward_cuts
            seg = self.segment_backward if self.backbone_dtype == torch.bfloat16 else ()
            self.backward_cuts = [] if seg else None
            return encoder.forward_fused(image, self.backbone_dtype or torch.float32, cuts=self.backward_cuts,
                                         cut_blocks=tuple(seg) if seg else ())
        if self.fused_backbone and isinstance(encoder, backbones.ImageEncoder) and not encoder.training and image.is_cuda:
            # inference (evel_her2st.py:48-50): running statistics, bf16 fused kernels or the fp32 generic own-kernel path
            if torch.is_grad_enabled() and (image.requires_grad or any(p.requires_grad for p in encoder.parameters())):
                raise RuntimeError("the fused DenseNet path has no backward in eval mode: call model.train() to train, or wrap "
                                   "inference in torch.no_grad()")
            return encoder.forward_eval_fused(image, self.backbone_dtype or torch.float32)
        if (self.fused_backbone and image.is_cuda and image.dim() == 4 and isinstance(
                encoder, (backbones.ImageEncoder_Resnet, backbones.ImageEncdoer_res18, backbones.ImageEncdoer_res101))):
            # ResNet selector values (model.py:88-148) on the generic own-kernel path, bf16 or fp32 activations
            if not encoder.training and torch.is_grad_enabled() and (image.requires_grad or any(
                    p.requires_grad for p in encoder.parameters())):
                # eval-mode BatchNorm is a raw affine kernel on this path (no autograd node): a gradient would silently stop
                # at the first BatchNorm.  The reference never differentiates in eval mode (train.py:31, evel_*.py: no_grad).
                raise RuntimeError("the fused ResNet path has no backward in eval mode: call model.train() to train, or wrap "
                                   "inference in torch.no_grad()")
            return encoder.forward_fused(image, self.backbone_dtype or torch.float32)
        if self.fused_backbone and isinstance(encoder, backbones.ImageEncoder_VIT) and image.is_cuda:
            if self.backbone_dtype is None or self.backbone_dtype == torch.float32:
                # "reference numerics": the whole encoder in fp32 on mcl_gemm (exact-fp32 MFMA), the fp32 LayerNorm / GELU
                # kernels and the fp32 attention core (csrc/attention.hip, one sequence per image)
                from .vit_fused import vit_features_fp32
                return vit_features_fp32(encoder.model, image)
            if self.backbone_dtype != torch.bfloat16:
                raise RuntimeError(f"the ViT image encoder runs in bf16 or fp32 (backbone_dtype={self.backbone_dtype})")
            from .vit_fused import vit_features_fused           # ViT on the hand-written bf16 kernels (csrc/gemm_bf16.hip)
            return vit_features_fused(encoder.model, image)
        if self.fused_backbone and image.is_cuda and isinstance(encoder, tuple(backbones.ENCODERS.values())):
            raise RuntimeError(f"no own-kernel path for {type(encoder).__name__} in this mode (training={encoder.training}, "
                               f"backbone_dtype={self.backbone_dtype}); set model.fused_backbone = False to run the plain "
                               "torch modules explicitly")
        # explicit plain-module path: CPU tensors (BASELINE configs[0] plumbing), fused_backbone = False (A/B), custom encoders
        if self.backbone_dtype is not None and self.backbone_dtype != torch.float32:
            if image.dim() == 4:
                image = image.contiguous(memory_format=torch.channels_last)
            with torch.autocast("cuda", dtype=self.backbone_dtype):
                feats = encoder(image)
            return feats.float()
        return encoder(image)

    def encode_image(self, image: Tensor) -> Tensor:
        """``self.image_encoder(image)`` through the execution path this model is configured for (fused bf16 kernels
        when ``backbone_dtype`` is bf16, else the plain module) -- what retrieval.get_embeddings calls."""
        enc = getattr(self, "image_encoder", None)
        if enc is None:
            enc = self.image_ecode
        return self._encode_image(enc, image)

    # FusedAdam(lazy_tables=True).attach_model installs its catch-up here: the rows this batch gathers are brought up to the
    # current optimisation step before they are read (optim.py; the untouched rows of the tables are advanced lazily)
    _table_catchup = None

    def _spot_features(self, batch) -> Tensor:
        sink = self.sparse_grads if (self.embedding_grad == "rowsparse" and torch.is_grad_enabled()) else None
        pos = batch["position"]
        if self._table_catchup is not None and pos.is_cuda:
            if pos.dtype != torch.float32:
                pos = pos.to(torch.float32)
            pos = pos.contiguous()
            self._table_catchup(pos)
        return ops.PosEmbedAddFn.apply(batch["expression"], pos, self.x_embed.weight, self.y_embed.weight, sink)

    # "infonce": the reference's identity-target symmetric cross entropy (model.py:242-247).  "bleep" / "bleep_vit": the
    # soft-target CLIP loss of the reference's main baseline (baselines/Bleep/models.py:34-43 / 66-76) on the same
    # kernels; single process only.
    loss_kind = "infonce"

    def _soft_clip(self) -> Optional[bool]:
        if self.loss_kind == "infonce":
            return None
        if self.loss_kind not in ("bleep", "bleep_vit"):
            raise ValueError("loss_kind must be 'infonce', 'bleep' or 'bleep_vit'")
        if self.process_group is not None:
            raise RuntimeError("the BLEEP soft-target loss is implemented for a single process only")
        return self.loss_kind == "bleep_vit"

    def _loss(self, spot_embeddings: Tensor, image_embeddings: Tensor) -> Tensor:
        soft = self._soft_clip()
        if soft is not None:
            return ops.SoftClipLossFn.apply(spot_embeddings, image_embeddings, float(self.temperature), soft)
        stash = self.last if self.capture else None
        if self.capture:
            self.last["spot_embeddings"] = spot_embeddings.detach()
            self.last["image_embeddings"] = image_embeddings.detach()
        if self.process_group is not None:
            from . import dist as mdist
            return mdist.DistInfoNCEFn.apply(spot_embeddings, image_embeddings, float(self.temperature),
                                             self.process_group, stash, self._fused_mode())
        return ops.InfoNCEFn.apply(spot_embeddings, image_embeddings, float(self.temperature), stash,
                                   self._fused_mode())

    def _fused_mode(self):
        return {"exact": False, "fused": True, "fp8": "fp8"}[self.infonce]

    def infonce_effective(self, global_batch: int) -> str:
        """The InfoNCE kernels a step at this GLOBAL batch actually runs: ``infonce="fused"`` takes the exact fp32 kernels
        below ``ops.FUSED_MIN_BATCH`` pairs (they are faster AND exact there; same rule under data parallelism)."""
        if self.infonce == "fused" and global_batch < ops.FUSED_MIN_BATCH:
            return f"exact (fused from {ops.FUSED_MIN_BATCH} pairs)"
        return self.infonce


    # ---- split form of forward(), used by engine.TrainStep (HIP-graph capture around the collectives)
    def embed(self, batch):
        """(spot_embeddings, image_embeddings): everything of forward() up to the logits."""
        raise NotImplementedError

    def loss_and_grads(self, spot_embeddings: Tensor, image_embeddings: Tensor):
        """Eager (no autograd): (loss, dE_spot, dE_img) of the symmetric InfoNCE -- global over the process
        group if one is set.  Forward and backward of model.py:242-247 in closed form."""
        es, ei = spot_embeddings.detach(), image_embeddings.detach()
        if self.capture:
            self.last["spot_embeddings"], self.last["image_embeddings"] = es, ei
        soft = self._soft_clip()
        if soft is not None:
            return ops.soft_clip_fwd_bwd(es, ei, float(self.temperature), soft)
        fused = self.infonce != "exact"
        if self.process_group is not None:
            from . import dist as mdist
            loss, d_es, d_ei, s = mdist.dist_infonce_auto(es, ei, float(self.temperature), self.process_group,
                                                          self._fused_mode())
        elif self.infonce == "fp8":
            loss, d_es, d_ei, s = ops.infonce_fp8_fwd_bwd(es, ei, float(self.temperature))
        elif fused:
            loss, d_es, d_ei, s = ops.infonce_fused_fwd_bwd(es, ei, float(self.temperature))
        else:
            loss, d_es, d_ei, s = ops.infonce_fwd_bwd(es, ei, float(self.temperature), want_logits=self.capture)
        if self.capture:
            self.last["logits"] = s
        return loss, d_es, d_ei


class mclSTExp_MLP(_ContrastiveBase):
    """model.py:171-198 (ablation without the spot encoder; attribute name ``image_ecode`` kept)."""

    def __init__(self, temperature, image_embedding, spot_embedding, projection_dim, dropout=0., *,
                 encoder_name="densenet121", compute="f32", backbone_dtype=None, embedding_grad="dense",
                 process_group=None, infonce="exact"):
        super().__init__()
        self.x_embed = nn.Embedding(65536, spot_embedding)
        self.y_embed = nn.Embedding(65536, spot_embedding)
        self.image_ecode = _make_encoder(encoder_name)
        self.image_projection = ProjectionHead(embedding_dim=image_embedding, projection_dim=projection_dim)
        self.spot_projection = ProjectionHead(embedding_dim=spot_embedding, projection_dim=projection_dim)
        self._init_common(temperature, compute, backbone_dtype, embedding_grad, process_group, infonce)

    def embed(self, batch):
        ops.set_compute(self.compute)
        image_features = self._encode_image(self.image_ecode, batch["image"])
        image_embeddings = self.image_projection(image_features)
        spot_embeddings = self.spot_projection(self._spot_features(batch))
        return spot_embeddings, image_embeddings

    def forward(self, batch):
        return self._loss(*self.embed(batch))


class mclSTExp_Attention(_ContrastiveBase):
    """model.py:201-247."""

    def __init__(self, encoder_name, temperature, image_dim, spot_dim, projection_dim, heads_num, heads_dim,
                 head_layers, dropout=0., *, compute="f32", backbone_dtype=None, embedding_grad="dense",
                 process_group=None, infonce="exact"):
        super().__init__()
        self.x_embed = nn.Embedding(65536, spot_dim)
        self.y_embed = nn.Embedding(65536, spot_dim)
        self.image_encoder = _make_encoder(encoder_name)
        # model.py:216-218: mlp_dim = spot_dim, dropout hard-coded to 0.
        self.spot_encoder = nn.Sequential(
            *[attn_block(spot_dim, heads=heads_num, dim_head=heads_dim, mlp_dim=spot_dim, dropout=0.)
              for _ in range(head_layers)])
        self.image_projection = ProjectionHead(embedding_dim=image_dim, projection_dim=projection_dim)
        self.spot_projection = ProjectionHead(embedding_dim=spot_dim, projection_dim=projection_dim)
        self._init_common(temperature, compute, backbone_dtype, embedding_grad, process_group, infonce)

    def _embed_spots(self, batch):
        spot_features = self._spot_features(batch).unsqueeze(dim=0)        # (1, B, G): batch is the sequence
        spot_embeddings = self.spot_encoder(spot_features)
        return self.spot_projection(spot_embeddings).squeeze(dim=0)

    def embed(self, batch):
        ops.set_compute(self.compute)
        side = self._branch_stream(batch["expression"].device)
        late = 2
        # only the train-mode fused forward (densenet_features_fused) fires the block hook: eval / no_grad calls take
        # forward_eval_fused or the module and must use the plain fork below, or the side stream would wait on an event
        # that is never recorded (= not wait at all)
        if (side is not None and late > 0 and self.fused_backbone and isinstance(self.image_encoder, backbones.ImageEncoder)
                and self.backbone_dtype == torch.bfloat16 and self.image_encoder.training and torch.is_grad_enabled()
                and batch["image"].is_cuda):
            # The spot branch starts when the DenseNet has finished dense block `late`: the first two blocks (56 x 56 and
            # 28 x 28 maps) are throughput-bound, anything running beside them costs its full duration; the
            # later blocks are latency-bound chains with idle CUs.
            from . import densenet_fused
            main = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            fired = []

            def _hook():
                ev.record(main)
                fired.append(True)
            densenet_fused.FORWARD_BLOCK_HOOKS[late] = _hook
            try:
                image_features = self._encode_image(self.image_encoder, batch["image"])
            finally:
                densenet_fused.FORWARD_BLOCK_HOOKS.pop(late, None)
            if fired:
                side.wait_event(ev)
            else:                                   # MCL_SPOT_AFTER_BLOCK beyond the last block: order after everything
                side.wait_stream(main)
            with torch.cuda.stream(side):
                pro = getattr(self, "_spot_lane_prologue", None)
                if pro is not None:                 # (engine.TrainStep: the step's zero_grad, off the main chain)
                    pro()
                spot_embeddings = self._embed_spots(batch)
            image_embeddings = self.image_projection(image_features)
            main.wait_stream(side)
            return spot_embeddings, image_embeddings
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                spot_embeddings = self._embed_spots(batch)
        image_features = self._encode_image(self.image_encoder, batch["image"])
        image_embeddings = self.image_projection(image_features)
        if side is not None:
            main.wait_stream(side)
        else:
            spot_embeddings = self._embed_spots(batch)
        return spot_embeddings, image_embeddings

    def forward(self, batch):
        return self._loss(*self.embed(batch))


def _make_encoder(encoder_name: str) -> nn.Module:
    if encoder_name == "identity":
        return nn.Identity()
    if encoder_name not in ENCODERS:
        raise ValueError(f"unknown encoder_name '{encoder_name}' (reference accepts {sorted(ENCODERS)}; "
                         "'identity' = precomputed features)")
    return ENCODERS[encoder_name]()


def load_reference_state_dict(model: nn.Module, state: Dict[str, Tensor], strict: bool = True):
    """Checkpoint loader with the reference's key rewrites (evel_her2st.py:33-37): strips a DataParallel
    'module.' prefix and renames 'well' -> 'spot'."""
    new = {}
    for k, v in state.items():
        k = k.replace("module.", "").replace("well", "spot")
        new[k] = v
    return model.load_state_dict(new, strict=strict)
