"""Teacher-forced per-layer parity of the generic own-kernel convolution path (conv_generic.py: im2col + this library's GEMM,
col2im, split-K weight gradient) inside ResNet-50 AT A TRAINING SHAPE (batch 128, 224 x 224: the batch of BASELINE configs[1]) --
VERDICT r04 "next" #3(a).  The end-to-end comparison of tests/test_generic_conv_gpu.py cannot bound a kernel (a random-init
50-layer BatchNorm net amplifies rounding by > 100 % on both the own and the stock path), so ONE forward + backward runs with
conv_generic.CAPTURE_CONVS recording the operands every convolution's kernels READ and the tensors they WROTE, and each sampled
convolution -- the 7 x 7 / 2 stem, stride-1 and stride-2 3 x 3, 1 x 1, the strided 1 x 1 downsample -- is re-evaluated in fp64
from exactly those operands (torch unfold + fp64 matmul on the GPU; reference ResNet = /root/reference/model.py:88-101 via
torchvision's Bottleneck):

    y  = conv(x, w)            output        bf16: 1e-2 of max|y|      fp32: 1e-5
    dx = conv^T(dy, w)         data gradient bf16: 1e-2 of max|dx|     fp32: 1e-5
    dW = dy^T * im2col(x)      weight grad.  bf16: 5e-3 of max|dW|     fp32: 2e-4      (an fp32 sum over up to 4e5 pixels whose
                                                                                 terms cancel: 7.8e-5 measured on the stem)
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
CL = torch.channels_last


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def _conv64(x, w, stride, pad):
    """fp64 conv2d by unfold + matmul (no library convolution): x (B, Ci, H, W), w (Co, Ci, k, k)."""
    B, Ci, H, W = x.shape
    Co, _, k, _ = w.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Co, OH, OW), device=x.device, dtype=torch.float64)
    wm = w.double().reshape(Co, -1)
    for b0 in range(0, B, 16):                                            # (chunks of images: the unfolded operand is k*k times x)
        cols = F.unfold(x[b0:b0 + 16].double(), k, padding=pad, stride=stride)       # (b, Ci*k*k, OH*OW)
        out[b0:b0 + 16] = (wm @ cols).view(-1, Co, OH, OW)
    return out


def _conv64_bwd(x, w, dy, stride, pad):
    B, Ci, H, W = x.shape
    Co, _, k, _ = w.shape
    wm = w.double().reshape(Co, -1)
    dx = torch.empty((B, Ci, H, W), device=x.device, dtype=torch.float64)
    dw = torch.zeros((Co, Ci * k * k), device=x.device, dtype=torch.float64)
    for b0 in range(0, B, 16):
        d = dy[b0:b0 + 16].double().reshape(-1, Co, dy.shape[2] * dy.shape[3])        # (b, Co, S)
        cols = F.unfold(x[b0:b0 + 16].double(), k, padding=pad, stride=stride)        # (b, K, S)
        dw += torch.einsum("bcs,bks->ck", d, cols)
        dcols = wm.t() @ d                                                            # (b, K, S)
        dx[b0:b0 + 16] = F.fold(dcols, (H, W), k, padding=pad, stride=stride)
    return dx, dw.view(Co, Ci, k, k)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_resnet50_convolutions_teacher_forced(dtype):
    from mclstexp_amd import conv_generic as cg, densenet_fused as dn
    from mclstexp_amd.backbones import ImageEncoder_Resnet
    torch.manual_seed(0)
    B = 128 if dtype == torch.bfloat16 else 32                     # (fp32 activations of ResNet-50 at 128 x 224^2: 3x the memory)
    enc = ImageEncoder_Resnet().to(DEV).to(memory_format=CL).train()
    x = torch.rand(B, 3, 224, 224, device=DEV).contiguous(memory_format=CL)
    g = torch.Generator(device=DEV).manual_seed(1)
    cg.CAPTURE_CONVS = []
    try:
        f = enc.forward_fused(x, dtype)
        f.backward(torch.randn(f.shape, device=DEV, generator=g) / f.shape[1] ** 0.5)
        torch.cuda.synchronize()
        caps = cg.CAPTURE_CONVS
    finally:
        cg.CAPTURE_CONVS = None
        dn.set_weight_provider(None)
    # (the bf16 stem runs on the DenseNet conv0 kernel, checked in test_layerwise_gpu.py; fp32: the generic path, captured here)
    kinds = {}
    for c in caps:
        k = c["wk"].shape[2] if c["wk"].dim() == 4 else 1
        kinds.setdefault((int(c["param"].shape[2]), c["stride"]), []).append(c)
    assert (3, 1) in kinds and (3, 2) in kinds and (1, 1) in kinds and (1, 2) in kinds, sorted(kinds)
    tol_y, tol_dx, tol_dw = (1e-2, 1e-2, 5e-3) if dtype == torch.bfloat16 else (1e-5, 1e-5, 2e-4)
    worst = {}
    n = 0
    for (k, stride), lst in sorted(kinds.items()):
        # first, middle and last convolution of every (kernel, stride) kind: every map size of the network appears
        for c in {id(v): v for v in (lst[0], lst[len(lst) // 2], lst[-1])}.values():
            xk, wk = c["x"], c["param"].detach().to(c["x"].dtype)            # the operands as the kernels read them
            y64 = _conv64(xk, wk, stride, c["pad"])
            dx64, dw64 = _conv64_bwd(xk, wk, c["dy"], stride, c["pad"])
            e = (_rel(c["y"], y64), _rel(c["dx"], dx64) if c["dx"] is not None else 0.0, _rel(c["dw"], dw64))
            key = f"{k}x{k}/{stride}"
            worst[key] = tuple(max(a, b) for a, b in zip(worst.get(key, (0.0, 0.0, 0.0)), e))
            n += 1
            shape = f"{tuple(xk.shape)} -> {tuple(c['y'].shape)}"
            assert e[0] <= tol_y and e[1] <= tol_dx and e[2] <= tol_dw, (key, shape, e)
    print(f"ResNet-50 {dtype}, batch {B}, 224^2: {n} convolutions re-evaluated in fp64 from the operands their kernels read; "
          f"worst (y, dx, dW) of the tensor maximum per kind: " + "; ".join(f"{k}: {v[0]:.1e} {v[1]:.1e} {v[2]:.1e}" for k, v in worst.items()))
