"""CPU restatement (fp32 torch, optional fp64) of mclSTExp's contrastive training step.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  The product path
(``mclstexp_amd``) never imports this file and has no CPU fallback.

Parity status: PINNED.  Every function here is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were produced by importing the reference's
own ``model.py`` in the build container (``tests/golden/gen_goldens.py``; harness-side
stubs for the absent ``timm``/``torchvision`` and an identity ``Tensor.cuda``).  The
image backbones (torchvision DenseNet-121 / timm ViT) are third-party, un-vendored and
version-unpinned in the reference (README lists only torch>=2.1); their restatement in
``densenet121_features`` follows the published torchvision architecture and is
"parity unpinned" (no reference artefact exists to pin it).

Everything is written as explicit arithmetic on a flat ``dict`` of tensors keyed by the
reference's ``state_dict`` names, so a reference checkpoint is
directly usable as ``params``.

Reference citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]

LN_EPS = 1e-5  # nn.LayerNorm default, model.py:13,158


# --------------------------------------------------------------------------- blocks
def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = LN_EPS) -> Tensor:
    """nn.LayerNorm over the last dim (biased variance).  model.py:13,17,158,166."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * weight + bias


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form.  model.py:25,156."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def gelu_erf_grad(x: Tensor) -> Tensor:
    """d/dx gelu_erf(x) = Phi(x) + x*phi(x)."""
    cdf = 0.5 * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))
    pdf = torch.exp(-0.5 * x * x) * (1.0 / math.sqrt(2.0 * math.pi))
    return cdf + x * pdf


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """nn.Linear: y = x @ W^T + b, W is (out, in)."""
    y = x @ weight.t()
    return y if bias is None else y + bias


def attention(x: Tensor, w_qkv: Tensor, w_out: Tensor, b_out: Tensor, heads: int, dim_head: int) -> Tensor:
    """Attention.forward, model.py:49-57, on a (B, G) batch-as-sequence (the leading
    singleton batch dim of the reference's (1, B, G) is dropped).

    qkv = x W_qkv^T (no bias, model.py:43); chunk order q,k,v (model.py:51); head split
    '(h d)' with h outer (model.py:52); softmax(q k^T * d^-0.5) v; merge; to_out Linear.
    """
    n = x.shape[0]
    inner = heads * dim_head
    qkv = x @ w_qkv.t()                                   # (B, 3*inner)
    q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
    q = q.reshape(n, heads, dim_head).permute(1, 0, 2)    # (h, B, d)
    k = k.reshape(n, heads, dim_head).permute(1, 0, 2)
    v = v.reshape(n, heads, dim_head).permute(1, 0, 2)
    dots = (q @ k.transpose(1, 2)) * (dim_head ** -0.5)   # (h, B, B)
    attn = torch.softmax(dots, dim=-1)
    out = attn @ v                                        # (h, B, d)
    out = out.permute(1, 0, 2).reshape(n, inner)          # 'b h n d -> b n (h d)'
    return out @ w_out.t() + b_out


def feed_forward(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """FeedForward.forward, model.py:20-32 (both dropouts are p=0, model.py:217)."""
    return linear(gelu_erf(linear(x, w1, b1)), w2, b2)


def attn_block(x: Tensor, p: Params, prefix: str, heads: int, dim_head: int) -> Tensor:
    """attn_block.forward, model.py:66-69: x = attn(LN(x)) + x ; x = ff(LN(x)) + x."""
    u = layer_norm(x, p[prefix + "attn.norm.weight"], p[prefix + "attn.norm.bias"])
    x = attention(u, p[prefix + "attn.fn.to_qkv.weight"], p[prefix + "attn.fn.to_out.0.weight"],
                  p[prefix + "attn.fn.to_out.0.bias"], heads, dim_head) + x
    u = layer_norm(x, p[prefix + "ff.norm.weight"], p[prefix + "ff.norm.bias"])
    x = feed_forward(u, p[prefix + "ff.fn.net.0.weight"], p[prefix + "ff.fn.net.0.bias"],
                     p[prefix + "ff.fn.net.3.weight"], p[prefix + "ff.fn.net.3.bias"]) + x
    return x


def projection_head(x: Tensor, p: Params, prefix: str) -> Tensor:
    """ProjectionHead.forward, model.py:160-168 (dropout p=0)."""
    projected = linear(x, p[prefix + "projection.weight"], p[prefix + "projection.bias"])
    y = linear(gelu_erf(projected), p[prefix + "fc.weight"], p[prefix + "fc.bias"]) + projected
    return layer_norm(y, p[prefix + "layer_norm.weight"], p[prefix + "layer_norm.bias"])


def pos_embed_add(expression: Tensor, position: Tensor, x_table: Tensor, y_table: Tensor) -> Tensor:
    """model.py:230-235: expr + x_embed(pos[:,0].long()) + y_embed(pos[:,1].long()).
    .long() truncates toward zero."""
    ix = position[:, 0].long()
    iy = position[:, 1].long()
    return expression + x_table[ix] + y_table[iy]


def spot_encoder(x: Tensor, p: Params, layers: int, heads: int, dim_head: int) -> List[Tensor]:
    """nn.Sequential of attn_block, model.py:216-218.  Returns the output of every layer."""
    outs = []
    for l in range(layers):
        x = attn_block(x, p, f"spot_encoder.{l}.", heads, dim_head)
        outs.append(x)
    return outs


# --------------------------------------------------------------------------- loss
def logits(e_spot: Tensor, e_img: Tensor, temperature: float) -> Tensor:
    """model.py:242: cos_smi = spot_embeddings @ image_embeddings.T / temperature
    (raw dot product of LayerNorm-ed embeddings, no L2 normalisation)."""
    return (e_spot @ e_img.t()) / temperature


def symmetric_infonce(s: Tensor) -> Tensor:
    """model.py:243-247 with the float identity as soft label:
    loss = 0.5*[ mean_i(LSE_j S_ij - S_ii) + mean_j(LSE_i S_ij - S_jj) ]."""
    diag = torch.diagonal(s)
    row = torch.logsumexp(s, dim=1) - diag
    col = torch.logsumexp(s, dim=0) - diag
    return 0.5 * (row.mean() + col.mean())


def symmetric_infonce_grad(s: Tensor) -> Tensor:
    """dloss/dS = (softmax_rows(S) + softmax_cols(S) - 2I) / (2B)."""
    b = s.shape[0]
    eye = torch.eye(b, dtype=s.dtype)
    return (torch.softmax(s, dim=1) + torch.softmax(s, dim=0) - 2.0 * eye) / (2.0 * b)


def infonce_strip(e_spot_loc: Tensor, e_img_loc: Tensor, e_spot_all: Tensor, e_img_all: Tensor,
                  row_offset: int, temperature: float) -> Tuple[Tensor, Tensor, Tensor]:
    """Data-parallel form (new capability): this rank owns global
    rows/cols [row_offset, row_offset+B_loc).  Returns (sum_i (LSE_row_i - S_ii),
    sum_j (LSE_col_j - S_jj), S_row_strip).  Summing both partials over ranks and
    dividing by 2*B_glob gives ``symmetric_infonce`` of the global logits."""
    b_loc = e_spot_loc.shape[0]
    s_rows = (e_spot_loc @ e_img_all.t()) / temperature          # (B_loc, B_glob)
    s_cols = (e_spot_all @ e_img_loc.t()) / temperature          # (B_glob, B_loc)
    idx = torch.arange(b_loc)
    diag = s_rows[idx, row_offset + idx]
    row_part = (torch.logsumexp(s_rows, dim=1) - diag).sum()
    col_part = (torch.logsumexp(s_cols, dim=0) - diag).sum()
    return row_part, col_part, s_rows


# --------------------------------------------------------------------------- model
def forward_from_features(p: Params, image_features: Tensor, expression: Tensor, position: Tensor,
                          temperature: float, layers: int, heads: int, dim_head: int) -> Dict[str, Tensor]:
    """mclSTExp_Attention.forward, model.py:225-247, downstream of the image backbone."""
    image_embeddings = projection_head(image_features, p, "image_projection.")
    x0 = pos_embed_add(expression, position, p["x_embed.weight"], p["y_embed.weight"])
    layer_outs = spot_encoder(x0, p, layers, heads, dim_head)
    spot_embeddings = projection_head(layer_outs[-1] if layer_outs else x0, p, "spot_projection.")
    s = logits(spot_embeddings, image_embeddings, temperature)
    return {
        "image_embeddings": image_embeddings,
        "spot_features0": x0,
        "layer_outs": layer_outs,
        "spot_embeddings": spot_embeddings,
        "cos_smi": s,
        "loss": symmetric_infonce(s),
    }


def forward_mlp_from_features(p: Params, image_features: Tensor, expression: Tensor, position: Tensor,
                              temperature: float) -> Dict[str, Tensor]:
    """mclSTExp_MLP.forward, model.py:187-198 (no spot encoder)."""
    image_embeddings = projection_head(image_features, p, "image_projection.")
    x0 = pos_embed_add(expression, position, p["x_embed.weight"], p["y_embed.weight"])
    spot_embeddings = projection_head(x0, p, "spot_projection.")
    s = logits(spot_embeddings, image_embeddings, temperature)
    return {"image_embeddings": image_embeddings, "spot_embeddings": spot_embeddings, "cos_smi": s,
            "loss": symmetric_infonce(s)}


# --------------------------------------------------------------------------- optimiser
def adam_l2_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int,
                 lr: float = 1e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                 weight_decay: float = 1e-3) -> None:
    """torch.optim.Adam (L2-coupled weight decay) single-tensor update, in place.
    train.py:118-120; ordering follows torch/optim/adam.py::_single_tensor_adam:
      g = g + wd*p ; m = lerp(m, g, 1-b1) ; v = b2*v + (1-b2)*g*g ;
      denom = sqrt(v)/sqrt(1-b2^t) + eps ; p -= (lr/(1-b1^t)) * m / denom
    """
    g = grad + weight_decay * param
    exp_avg.mul_(beta1).add_(g, alpha=1.0 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-(lr / bc1))


# --------------------------------------------------------------------------- backbone
def _bn_train(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mean = x.mean(dim=(0, 2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def densenet121_features(p: Params, x: Tensor, prefix: str = "image_encoder.model.0.",
                         block_config=(6, 12, 24, 16)) -> Tensor:
    """torchvision DenseNet-121 ``features`` in train mode (batch-stat BN) followed by the
    reference's pooling, model.py:81-85: adaptive_avg_pool2d directly on norm5 (no ReLU).
    PARITY UNPINNED (torchvision absent from the container; architecture from the
    published torchvision definition)."""
    F = torch.nn.functional
    g = lambda k: p[prefix + k]
    x = F.conv2d(x, g("conv0.weight"), stride=2, padding=3)
    x = torch.relu(_bn_train(x, g("norm0.weight"), g("norm0.bias")))
    x = F.max_pool2d(x, 3, 2, 1)
    for bi, nl in enumerate(block_config, start=1):
        for li in range(1, nl + 1):
            q = f"denseblock{bi}.denselayer{li}."
            y = torch.relu(_bn_train(x, g(q + "norm1.weight"), g(q + "norm1.bias")))
            y = F.conv2d(y, g(q + "conv1.weight"))
            y = torch.relu(_bn_train(y, g(q + "norm2.weight"), g(q + "norm2.bias")))
            y = F.conv2d(y, g(q + "conv2.weight"), padding=1)
            x = torch.cat([x, y], dim=1)
        if bi != len(block_config):
            q = f"transition{bi}."
            x = torch.relu(_bn_train(x, g(q + "norm.weight"), g(q + "norm.bias")))
            x = F.conv2d(x, g(q + "conv.weight"))
            x = F.avg_pool2d(x, 2, 2)
    x = _bn_train(x, g("norm5.weight"), g("norm5.bias"))
    return x.mean(dim=(2, 3))


# --------------------------------------------------------------------------- whole step (CPU baseline / smoke)
def train_step(params: Params, state: Dict[str, Tuple[Tensor, Tensor]], batch: Dict[str, Tensor], step: int,
               temperature: float = 1.0, layers: int = 2, heads: int = 8, dim_head: int = 64,
               backbone: str = "densenet121") -> float:
    """One iteration of train.py:33-41 on CPU: forward (model.py:225-247), autograd backward (dense
    (65536, G) table gradients, as in the reference), Adam(lr=1e-4, weight_decay=1e-3) over ALL
    parameters.  ``params`` are leaf tensors with requires_grad=True; ``state`` maps name -> (m, v).
    ``backbone`` = "densenet121" (image (B,3,H,W)) or "identity" (image = (B, D) features)."""
    for p in params.values():
        p.grad = None
    feats = batch["image"] if backbone == "identity" else densenet121_features(params, batch["image"])
    out = forward_from_features(params, feats, batch["expression"], batch["position"], temperature, layers,
                                heads, dim_head)
    out["loss"].backward()
    with torch.no_grad():
        for n, p in params.items():
            if p.grad is None:
                continue
            if n not in state:
                state[n] = (torch.zeros_like(p), torch.zeros_like(p))
            adam_l2_step(p, p.grad, state[n][0], state[n][1], step)
    return float(out["loss"])


# --------------------------------------------------------------------------- BLEEP soft-target CLIP loss
def bleep_soft_clip_loss(e_spot: Tensor, e_img: Tensor, temperature: float, targets_times_temperature: bool = False
                         ) -> Tensor:
    """baselines/Bleep/models.py:34-43 (CLIPModel) and :66-76 (CLIPModel_ViT, ``targets_times_temperature``):
    logits = E_s E_i^T / T; targets = softmax((E_i E_i^T + E_s E_s^T)/2 {/T | *T}); loss = mean of
    (CE(logits, targets) + CE(logits^T, targets^T)) / 2 with soft targets (cross_entropy, models.py:228-234).
    The targets are NOT detached: gradients flow through them as in the reference."""
    logits_ = (e_spot @ e_img.T) / temperature
    sim = (e_img @ e_img.T + e_spot @ e_spot.T) / 2
    targets = torch.softmax(sim * temperature if targets_times_temperature else sim / temperature, dim=-1)
    spots_loss = (-targets * torch.log_softmax(logits_, dim=-1)).sum(1)
    images_loss = (-targets.T * torch.log_softmax(logits_.T, dim=-1)).sum(1)
    return ((images_loss + spots_loss) / 2.0).mean()
