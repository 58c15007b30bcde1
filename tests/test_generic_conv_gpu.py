"""Generic own-kernel convolution / pooling (conv_generic.py: im2col + mcl_gemm / mcl_gemm_bf16, csrc/pool_generic.hip) and
what is built on it: the fp32 ("reference numerics") mode of the DenseNet backbone and the ResNet encoders
(/root/reference/model.py:72-148) -- against fp64 torch on the same data, on the MI355X."""
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close, assert_close_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda"
CL = torch.channels_last


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H,W,k,stride,pad", [(2, 16, 9, 7, 3, 1, 1), (3, 8, 8, 8, 3, 2, 1), (2, 3, 16, 24, 7, 2, 3),
                                                  (2, 24, 5, 6, 1, 2, 0), (1, 64, 4, 4, 1, 1, 0), (2, 12, 7, 7, 3, 2, 1)])
def test_im2col_col2im_vs_unfold(dtype, B, C, H, W, k, stride, pad):
    """im2col == F.unfold (column order (ky, kx, c)), also on a channel slice of a wider buffer; col2im == F.fold."""
    from mclstexp_amd import conv_generic as cg
    wide = (_rand(B, H, W, C + 8, seed=1) - 0.5).to(dtype).to(DEV)
    x = wide.permute(0, 3, 1, 2)[:, 8 if C % 8 == 0 else 0: (8 if C % 8 == 0 else 0) + C]
    cols = cg.im2col(x, k, stride, pad)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    ref = F.unfold(x.float(), k, padding=pad, stride=stride)                       # (B, C*k*k, L) order (c, ky, kx)
    ref = ref.view(B, C, k * k, OH * OW).permute(0, 3, 2, 1).reshape(B * OH * OW, k * k * C)
    assert torch.equal(cols.float(), ref)
    d = (_rand(B * OH * OW, k * k * C, seed=2) - 0.5).to(dtype).to(DEV)
    dx = cg.col2im(d, (B, C, H, W), k, stride, pad)
    dref = d.float().view(B, OH * OW, k * k, C).permute(0, 3, 2, 1).reshape(B, C * k * k, OH * OW)
    dref = F.fold(dref.double(), (H, W), k, padding=pad, stride=stride)
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert_close_scaled(dx.float().cpu(), dref.cpu(), tol, what="col2im")
    dx2 = cg.col2im(d, (B, C, H, W), k, stride, pad, out=dx.clone(memory_format=CL), accumulate=True)
    assert_close_scaled(dx2.float().cpu(), 2 * dref.cpu(), 2 * tol, what="col2im accumulate")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Ci,Co,H,W,k,stride,pad", [(4, 64, 128, 14, 14, 1, 1, 0), (3, 64, 64, 15, 13, 3, 1, 1),
                                                      (2, 128, 256, 16, 16, 3, 2, 1), (2, 256, 512, 8, 8, 1, 2, 0),
                                                      (2, 3, 64, 32, 32, 7, 2, 3), (5, 32, 32, 7, 7, 3, 1, 1),
                                                      # data gradient as a same-convolution of dy (C_out <= C_in / 2)
                                                      (3, 128, 32, 14, 14, 3, 1, 1), (2, 64, 16, 9, 11, 5, 1, 2),
                                                      (2, 96, 48, 7, 7, 3, 1, 1)])
def test_conv_fn_vs_fp64(dtype, B, Ci, Co, H, W, k, stride, pad):
    """ConvFn forward / backward-data / weight gradient (direct accumulation into a channels-last fp32 .grad AND the autograd
    hand-over) against fp64 conv2d on the same (rounded) operands."""
    from mclstexp_amd import conv_generic as cg
    if dtype == torch.bfloat16 and Ci % 8:
        pytest.skip("bf16 GEMM operands need K % 8 == 0 (the 3-channel stem runs the dedicated conv0 kernel in bf16)")
    x = (_rand(B, Ci, H, W, seed=3) - 0.5).to(dtype).to(DEV).contiguous(memory_format=CL)
    w = ((_rand(Co, Ci, k, k, seed=4) - 0.5) / (k * Ci ** 0.5)).to(DEV).contiguous(memory_format=CL)
    wq = w.to(dtype).float()
    for direct in (False, True):
        if direct:
            wp = torch.nn.Parameter(w.clone(memory_format=CL))
            wp.grad = torch.full_like(wp, 0.25, memory_format=CL)
        else:       # a plain tensor (not an nn.Parameter) never gets a .grad created for it: the autograd hand-over
            wp = w.clone(memory_format=CL).requires_grad_(True)
        xin = x.clone(memory_format=CL).requires_grad_(Ci != 3)
        y = cg.conv2d(xin, wp, stride, pad)
        dy = (_rand(*y.shape, seed=5) - 0.5).to(dtype).to(DEV).contiguous(memory_format=CL)
        y.backward(dy)
        xr = x.double().requires_grad_(True)
        wr = wq.double().requires_grad_(True)
        yr = F.conv2d(xr, wr, stride=stride, padding=pad)
        yr.backward(dy.double())
        tol = 2e-6 if dtype == torch.float32 else 1e-2
        assert_close_scaled(y.detach().float().cpu(), yr.detach().cpu(), tol, what="conv forward")
        if Ci != 3:
            assert_close_scaled(xin.grad.float().cpu(), xr.grad.cpu(), tol, what="conv backward-data")
        gw = wp.grad.float() - (0.25 if direct else 0.0)
        assert_close_scaled(gw.cpu(), wr.grad.cpu(), 2e-5 if dtype == torch.float32 else 4e-3, what=f"conv weight gradient (direct={direct})")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_generic_pools_and_residual(dtype):
    from mclstexp_amd import conv_generic as cg
    x = torch.relu(_rand(3, 16, 9, 11, seed=6) - 0.4).to(dtype).to(DEV).contiguous(memory_format=CL)
    # max pool: forward bit-exact, backward with ATen's first-maximum tie rule
    xa = x.detach().clone().float().requires_grad_(True)
    ya = F.max_pool2d(xa, 3, 2, 1)
    xb = x.clone(memory_format=CL).requires_grad_(True)
    yb = cg.max_pool_3s2(xb)
    assert torch.equal(yb.float(), ya.detach())
    dy = (_rand(*ya.shape, seed=7) - 0.5).to(dtype).to(DEV).contiguous(memory_format=CL)
    ya.backward(dy.float()); yb.backward(dy)
    assert_close(xb.grad.float().cpu(), xa.grad.to(dtype).float().cpu(), 1e-2 if dtype == torch.bfloat16 else 1e-6, what="max pool bwd")
    # avg pool with floor on odd maps
    xa = x.detach().clone().float().requires_grad_(True)
    ya = F.avg_pool2d(xa, 2, 2)
    xb = x.clone(memory_format=CL).requires_grad_(True)
    yb = cg.avg_pool_2(xb)
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert_close(yb.detach().float().cpu(), ya.detach().cpu(), tol, tol, what="avg pool fwd")
    d2 = (_rand(*ya.shape, seed=8) - 0.5).to(dtype).to(DEV).contiguous(memory_format=CL)
    ya.backward(d2.float()); yb.backward(d2)
    assert_close(xb.grad.float().cpu(), xa.grad.cpu(), tol, tol, what="avg pool bwd")
    # global average pool
    xa = x.double().requires_grad_(True)
    ga = F.adaptive_avg_pool2d(xa, (1, 1)).flatten(1)
    xb = x.clone(memory_format=CL).requires_grad_(True)
    gb = cg.global_avg_pool(xb)
    g = (_rand(3, 16, seed=9) - 0.5).to(DEV)
    ga.backward(g.double()); gb.backward(g)
    assert_close(gb.detach().cpu(), ga.detach().cpu(), 1e-6, 1e-6, what="gap fwd")
    assert_close(xb.grad.float().cpu(), xa.grad.cpu(), tol, tol, what="gap bwd")
    # residual add + relu
    a = (_rand(3, 16, 9, 11, seed=10) - 0.5).to(dtype).to(DEV).contiguous(memory_format=CL).requires_grad_(True)
    b = (_rand(3, 16, 9, 11, seed=11) - 0.5).to(dtype).to(DEV).contiguous(memory_format=CL).requires_grad_(True)
    y = cg.add_relu(a, b)
    ref = torch.relu(a.detach().float() + b.detach().float())
    assert_close(y.detach().float().cpu(), ref.cpu(), tol, tol, what="add_relu")
    dyy = (_rand(3, 16, 9, 11, seed=12) - 0.5).to(dtype).to(DEV).contiguous(memory_format=CL)
    y.backward(dyy)
    assert torch.equal(a.grad, b.grad) and torch.equal(a.grad.float(), dyy.float() * (y.detach().float() > 0))


import copy


def _copy_fp64(m):
    return copy.deepcopy(m).double()


LIB_MARKERS = ("Cijk_", "miopen", "MIOpen", "batch_norm", "max_pool", "avg_pool", "igemm", "naive_conv", "gridwise")


@pytest.mark.parametrize("name,dtype", [("res18", torch.float32), ("res18", torch.bfloat16), ("resnet50", torch.bfloat16),
                                        ("res101", torch.float32)])
def test_resnet_fused_vs_fp64_module(name, dtype):
    """ResNet encoders (model.py:88-148) on the generic own-kernel path against an fp64 run of the same module: pooled features
    and parameter gradients; running statistics and num_batches_tracked as nn.BatchNorm2d updates them; no MIOpen / ATen
    convolution, BatchNorm or pooling kernel in the profiler's kernel records."""
    from mclstexp_amd import backbones, kernel_audit
    torch.manual_seed(0)
    enc = backbones.ENCODERS[name]().to(DEV).to(memory_format=CL).train()
    ref = _copy_fp64(enc)
    B, HW = 8, 96
    x = _rand(B, 3, HW, HW, seed=13).to(DEV)
    dy = (_rand(B, enc.out_dim, seed=14) - 0.5).to(DEV)
    out = {}

    def run():
        y = enc.forward_fused(x, dtype)
        y.backward(dy)
        out["y"] = y.detach()
    ks = kernel_audit.step_kernels(run)
    bad = [n for n in ks if any(m in n for m in LIB_MARKERS)]
    assert not bad, bad
    y = out["y"]
    xin = x.double() if dtype == torch.float32 else x.to(torch.bfloat16).double()
    yr = ref(xin)
    yr.backward(dy.double())
    # the same module on the stock PyTorch-ROCm ops in the same activation type: a random-init 18..101-layer BatchNorm net at a
    # small batch amplifies rounding chaotically (as the DenseNet does), so the bound is "as accurate as the stock
    # path", measured against fp64 on both sides
    stock = copy.deepcopy(ref).float().to(memory_format=CL)
    for p in stock.parameters():
        p.grad = None
    if dtype == torch.float32:
        ys = stock(x)
    else:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ys = stock(x.contiguous(memory_format=CL))
    ys.float().backward(dy)

    def dev(model, yy):
        e = float((yy.detach().double() - yr.detach()).abs().max() / yr.detach().abs().max())
        g = sorted((float((p.grad.double() - q.grad).abs().max() / (q.grad.abs().max() + 1e-30)), n)
                   for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()))
        return e, g[len(g) // 2][0], g[-1]
    for n, p in enc.named_parameters():
        assert p.grad is not None, n
    e_f, med, worst = dev(enc, y)
    e_s, med_s, worst_s = dev(stock, ys)
    print(f"{name} {dtype}: features {e_f:.2e} of max (stock ops {e_s:.2e}); parameter-gradient deviation median {med:.2e} "
          f"(stock {med_s:.2e}), worst {worst[0]:.2e} ({worst[1]}; stock {worst_s[0]:.2e})")
    if dtype == torch.float32:
        # (ResNet-18: 3e-6; the 50 / 101-layer nets amplify even fp32 rounding at this batch: bounded relative to the stock
        # fp32 modules, and kernel by kernel in tests/test_resnet_layerwise_gpu.py)
        assert e_f <= 2.0 * e_s + 1e-5 and med <= 2.0 * med_s + 1e-5, (e_f, e_s, med, med_s)
    else:
        # bf16 end to end: a random-init BatchNorm net at a small batch amplifies rounding chaotically on EVERY bf16 path
        # (own and stock alike deviate by tens of percent): no kernel can be bounded here.  The kernels are bounded one by one,
        # from the operands they read, in tests/test_resnet_layerwise_gpu.py; this test keeps the plumbing (all parameters
        # receive finite gradients, running statistics, no library kernel) and only a loose sanity relation to the stock path.
        assert all(torch.isfinite(p.grad).all() for p in enc.parameters())
        assert e_f <= 2.0 * e_s + 2e-2 and med <= 2.0 * med_s + 2e-2, (e_f, e_s, med, med_s)
    bn, bnr = enc.model[1], ref.model[1]
    assert_close(bn.running_mean.cpu(), bnr.running_mean.float().cpu(), 2e-3 if dtype == torch.bfloat16 else 1e-5, what="running_mean")
    assert int(bn.num_batches_tracked) == int(bnr.num_batches_tracked) == 1
    # eval mode: running statistics
    enc.eval(); ref.eval()
    with torch.no_grad():
        ye, yre = enc.forward_fused(x, dtype), ref(xin)
    e_e = float((ye.double() - yre).abs().max() / yre.abs().max())
    print(f"{name} {dtype}: eval-mode features {e_e:.2e} of max")
    assert e_e <= (1e-4 if dtype == torch.float32 else 0.1), e_e


def test_model_with_resnet_encoder_trains_on_own_kernels():
    """mclSTExp_Attention(encoder_name='res18') (model.py:206-215 selector value): a FusedAdam training step through the
    model class; the image branch's kernels are this library's."""
    from mclstexp_amd import kernel_audit, synth
    from mclstexp_amd.model import mclSTExp_Attention
    from mclstexp_amd.optim import FusedAdam
    torch.manual_seed(0)
    G = 171
    m = mclSTExp_Attention("res18", 1.0, 512, G, 256, 8, 64, 2, backbone_dtype=torch.bfloat16, embedding_grad="rowsparse",
                           infonce="exact").to(DEV)
    m.to(memory_format=CL).train()
    opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(m)
    b = {k: v.to(DEV) for k, v in synth.make_batch(8, G, image_hw=64, seed=0).items()}
    losses = []

    def step():
        loss = m(b)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    for _ in range(3):
        step()
    ks = kernel_audit.step_kernels(step)
    bad = [n for n in ks if any(mk in n for mk in LIB_MARKERS)]
    assert not bad, bad
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]


def test_densenet_fp32_mode_runs_on_own_kernels():
    """``backbone_dtype=None`` (fp32 activations, the reference's numerics): no MIOpen / ATen convolution, BatchNorm or pooling
    kernel in a forward + backward of the DenseNet backbone (VERDICT r03 missing #5), and the result matches an fp64 run."""
    from mclstexp_amd import backbones, kernel_audit
    torch.manual_seed(0)
    enc = backbones.ImageEncoder().to(DEV).train()
    ref = _copy_fp64(enc)
    x = _rand(4, 3, 64, 64, seed=15).to(DEV)
    ks = kernel_audit.step_kernels(lambda: enc.forward_fused(x, torch.float32).sum().backward())
    bad = [n for n in ks if any(m in n for m in LIB_MARKERS)]
    assert not bad, bad
    for p in enc.parameters():
        p.grad = None
    enc.load_state_dict(ref.float().state_dict())       # undo the first run's running-statistics update
    ref = _copy_fp64(enc)
    y = enc.forward_fused(x, torch.float32)
    dy = (_rand(*y.shape, seed=16) - 0.5).to(DEV)
    y.backward(dy)
    yr = ref(x.double())
    yr.backward(dy.double())
    assert_close_scaled(y.detach().cpu(), yr.detach().cpu(), 2e-4, what="fp32 features vs fp64")
    errs = sorted(float((p.grad.double() - q.grad).abs().max() / (q.grad.abs().max() + 1e-30))
                  for p, q in zip(enc.parameters(), ref.parameters()))
    print(f"densenet fp32 mode: parameter-gradient deviation from fp64: median {errs[len(errs) // 2]:.2e}, max {errs[-1]:.2e}")
    assert errs[len(errs) // 2] <= 2e-4
