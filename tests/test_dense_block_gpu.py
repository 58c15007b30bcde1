"""csrc/dense_block.hip: a whole DenseNet dense block (7 x 7 maps) as ONE persistent launch -- torchvision's denseblock4 behind
/root/reference/model.py:75-76.  Teacher-forced fp64 re-evaluation of every layer from the tensors the kernel itself produced
(its own concat buffer, z, statistics), determinism, and agreement with the per-layer launch sequence it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
CL = torch.channels_last


def _block(seed=0):
    from mclstexp_amd.backbones import densenet121_features_module
    torch.manual_seed(seed)
    feats = densenet121_features_module()
    blk = feats.denseblock4
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in blk.named_parameters():            # non-trivial affine parameters (default init: gamma = 1, beta = 0)
            if "norm" in n and n.endswith("weight"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif "norm" in n:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
    return blk.to(DEV).train()


def _run(blk, x, persistent):
    from mclstexp_amd import densenet_fused as dn
    old = dn.USE_BLOCK_PERSISTENT
    dn.USE_BLOCK_PERSISTENT = persistent
    dn.CAPTURE_BLOCKS = []
    try:
        rec = dn._RunningStats()
        with torch.enable_grad():
            buf, stats = dn.dense_block(blk, x.clone().requires_grad_(True), rec)
        cap = dn.CAPTURE_BLOCKS[0]
        torch.cuda.synchronize()
        assert not dn.block_persistent_error(x.device)
        return buf.detach(), stats, cap
    finally:
        dn.USE_BLOCK_PERSISTENT = old
        dn.CAPTURE_BLOCKS = None


def _input(B, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 512, 7, 7, generator=g).mul_(1.5).add_(0.2)
    return x.to(DEV).to(torch.bfloat16).contiguous(memory_format=CL)


@pytest.mark.parametrize("B", [128, 33, 5, 1])
def test_persistent_block_teacher_forced_vs_fp64(B):
    blk = _block()
    x = _input(B)
    buf, stats, cap = _run(blk, x, True)
    layers = list(blk.values())
    bufd = buf.double().cpu()
    worst = {"stat": 0.0, "z": 0.0, "zstat": 0.0, "y": 0.0}
    for l, ly in enumerate(layers):
        cin = 512 + 32 * l
        xin = bufd[:, :cin]
        mean = xin.mean(dim=(0, 2, 3))
        var = xin.var(dim=(0, 2, 3), unbiased=False)
        scale = xin.abs().amax(dim=(0, 2, 3)).clamp_min(1e-6)
        worst["stat"] = max(worst["stat"], float(((stats.mean[:cin].double().cpu() - mean).abs() / scale).max()),
                            float(((stats.var[:cin].double().cpu() - var).abs() / scale ** 2).max()))
        rstd = (var + ly.norm1.eps).rsqrt()
        assert torch.allclose(stats.rstd[:cin].double().cpu(), rstd, rtol=2e-5, atol=0)
        a = F.relu((xin - mean[None, :, None, None]) * (rstd * ly.norm1.weight.double().cpu())[None, :, None, None]
                   + ly.norm1.bias.double().cpu()[None, :, None, None])
        a = a.to(torch.bfloat16).double()                  # the kernel's MFMA operand is the bf16-rounded activation
        w1 = cap["wcast"][2 * l].double().cpu()
        z_ref = F.conv2d(a, w1)
        z = cap["z"][l].double().cpu()
        worst["z"] = max(worst["z"], float((z.detach() - z_ref.detach()).abs().max() / z_ref.detach().abs().max()))
        m2, v2, r2 = (t.double().cpu() for t in cap["bn2"][l])
        zm, zv = z.mean(dim=(0, 2, 3)), z.var(dim=(0, 2, 3), unbiased=False)
        zs = z.abs().amax(dim=(0, 2, 3)).clamp_min(1e-6)
        worst["zstat"] = max(worst["zstat"], float(((m2 - zm).abs() / zs).max()), float(((v2 - zv).abs() / zs ** 2).max()))
        assert torch.allclose(r2, (zv + ly.norm2.eps).rsqrt(), rtol=2e-5, atol=0)
        a2 = F.relu((z - zm[None, :, None, None]) * ((zv + ly.norm2.eps).rsqrt() * ly.norm2.weight.double().cpu())[None, :, None, None]
                    + ly.norm2.bias.double().cpu()[None, :, None, None]).to(torch.bfloat16).double()
        w2 = cap["wcast"][2 * l + 1].double().cpu()
        y_ref = F.conv2d(a2, w2, padding=1)
        y = bufd[:, cin:cin + 32]
        worst["y"] = max(worst["y"], float((y - y_ref).abs().max() / y_ref.abs().max()))
    print(f"B={B} persistent block, worst over 16 layers (of the tensor / channel maximum):", worst)
    assert worst["stat"] < 2e-6 and worst["zstat"] < 2e-6, worst
    assert worst["z"] < 6e-3 and worst["y"] < 6e-3, worst      # bf16 output rounding: 2^-9 of the value, operands exact


def test_persistent_block_deterministic_and_close_to_per_layer_path():
    blk = _block(seed=1)
    x = _input(128, seed=5)
    b1, s1, c1 = _run(blk, x, True)
    b2, s2, c2 = _run(blk, x, True)
    assert torch.equal(b1, b2) and torch.equal(s1.mean, s2.mean) and torch.equal(s1.rstd, s2.rstd)
    for za, zb in zip(c1["z"], c2["z"]):
        assert torch.equal(za, zb)
    b0, s0, c0 = _run(blk, x, False)
    # layer 0 sees identical inputs in both forms: only the accumulation order differs (one bf16 ulp on a few elements)
    z1, z0 = c1["z"][0].float(), c0["z"][0].float()
    assert float((z1 - z0).abs().max()) <= 2.0 ** -7 * float(z0.abs().max())
    assert torch.allclose(c1["bn2"][0][0], c0["bn2"][0][0], rtol=0, atol=2e-4 * float(z0.abs().max()))
    # the whole block: rounding differences travel through 16 BatchNorm layers
    d = (b1.float() - b0.float()).abs()
    assert float(d.max()) < 0.15 * float(b0.float().abs().max()) and float(d.mean()) < 4e-3 * float(b0.float().abs().max())
    assert torch.allclose(s1.mean, s0.mean, rtol=0, atol=3e-3) and torch.allclose(s1.rstd, s0.rstd, rtol=2e-2, atol=0)


def test_persistent_block_backward_runs_on_its_outputs():
    """The per-layer backward kernels consume exactly what the persistent forward left (buf, z, statistics)."""
    from mclstexp_amd import densenet_fused as dn
    blk = _block(seed=2)
    x = _input(16, seed=7)
    outs = []
    for persistent in (True, False):
        old = dn.USE_BLOCK_PERSISTENT
        dn.USE_BLOCK_PERSISTENT = persistent
        try:
            for p in blk.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            buf, _ = dn.dense_block(blk, xi, dn._RunningStats(), force_join=True)
            g = torch.Generator().manual_seed(11)
            gout = torch.randn(buf.shape, generator=g).to(DEV).to(torch.bfloat16).contiguous(memory_format=CL)
            buf.backward(gout)
            torch.cuda.synchronize()
            outs.append((xi.grad.float().clone(), {n: p.grad.float().clone() for n, p in blk.named_parameters()}))
        finally:
            dn.USE_BLOCK_PERSISTENT = old
    (gx1, gp1), (gx0, gp0) = outs
    assert torch.isfinite(gx1).all()
    assert float((gx1 - gx0).abs().mean()) < 8e-2 * float(gx0.abs().mean()) + 1e-6
    for n in gp0:
        assert torch.isfinite(gp1[n]).all(), n
        assert float((gp1[n] - gp0[n]).abs().mean()) < 0.12 * float(gp0[n].abs().mean()) + 1e-6, n


def _fwd_bwd(blk, x, gout, persistent_bwd):
    from mclstexp_amd import densenet_fused as dn
    old = dn.USE_BLOCK_PERSISTENT_BWD
    dn.USE_BLOCK_PERSISTENT_BWD = persistent_bwd
    dn.CAPTURE_BLOCKS = []
    try:
        for p in blk.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        buf, _ = dn.dense_block(blk, xi, dn._RunningStats(), force_join=True)
        buf.backward(gout.clone())
        torch.cuda.synchronize()
        assert not dn.block_persistent_error(x.device)
        cap = dn.CAPTURE_BLOCKS[0]
        dyc = cap.get("dyc") or [None] * 16
        dys = [(dyc[l] if dyc[l] is not None else cap["gbuf"][:, 512 + 32 * l: 544 + 32 * l]).float().clone() for l in range(16)]
        return (xi.grad.float().clone(), {n: p.grad.float().clone() for n, p in blk.named_parameters()},
                [t.float().clone() for t in cap["dz"]], dys)
    finally:
        dn.USE_BLOCK_PERSISTENT_BWD = old
        dn.CAPTURE_BLOCKS = None


@pytest.mark.parametrize("B", [128, 33, 5])
def test_persistent_block_backward_vs_per_layer_kernels(B):
    """csrc/dense_block.hip dense_block_bwd_kernel against the per-layer launch sequence it replaces (same forward tensors, same
    arithmetic; only the order of the batch sums differs): every layer's dz and consumed dy, the block-input gradient and all
    parameter gradients; and run-to-run determinism."""
    blk = _block(seed=4)
    x = _input(B, seed=9)
    g = torch.Generator().manual_seed(13)
    gout = torch.randn(B, 1024, 7, 7, generator=g).to(DEV).to(torch.bfloat16).contiguous(memory_format=CL)
    gx1, gp1, dz1, dy1 = _fwd_bwd(blk, x, gout, True)
    gx2, gp2, dz2, dy2 = _fwd_bwd(blk, x, gout, True)
    assert torch.equal(gx1, gx2)
    for n in gp1:
        assert torch.equal(gp1[n], gp2[n]), n
    gx0, gp0, dz0, dy0 = _fwd_bwd(blk, x, gout, False)
    worst = {"dz": 0.0, "dy": 0.0}
    for l in range(16):
        worst["dz"] = max(worst["dz"], float((dz1[l] - dz0[l]).abs().max() / dz0[l].abs().max()))
        worst["dy"] = max(worst["dy"], float((dy1[l] - dy0[l]).abs().max() / dy0[l].abs().max()))
    e_gx = float((gx1 - gx0).abs().max() / gx0.abs().max())
    e_gp = max(float((gp1[n] - gp0[n]).abs().max() / (gp0[n].abs().max() + 1e-20)) for n in gp0)
    print(f"B={B}: persistent backward vs per-layer kernels (of max): dz {worst['dz']:.2e}  dy consumed {worst['dy']:.2e}  "
          f"block-input gradient {e_gx:.2e}  parameter gradients {e_gp:.2e}")
    # identical arithmetic up to the summation order of the batch means: differences are single bf16 roundings that flipped
    assert worst["dz"] < 2e-2 and worst["dy"] < 2e-2 and e_gx < 2e-2 and e_gp < 2e-2, (worst, e_gx, e_gp)


def test_seam_timeout_is_raised_not_silent():
    """A persistent launch whose workgroups cannot see each other's BatchNorm records gives up and sets the device error word
    (csrc/dense_block.hip seam_wait).  Forced here with a one-poll bound (the `max_spins` argument of the C ABI): the word must
    surface as ops.SeamTimeoutError from ops.check_device_errors (what train.train calls at its host sync) and from
    TrainStep's sync-free polling -- never a silent step on garbage."""
    from mclstexp_amd import densenet_fused as dn, ops
    blk = _block()
    x = _input(128)
    ops.check_device_errors(x.device)
    dn.SEAM_MAX_SPINS = 1
    try:
        rec = dn._RunningStats()
        with torch.enable_grad():
            buf, _ = dn.dense_block(blk, x.clone().requires_grad_(True), rec, force_join=True)
            torch.cuda.synchronize()
            with pytest.raises(ops.SeamTimeoutError):
                ops.check_device_errors(x.device)
            ops.check_device_errors(x.device)                  # cleared by the raise
            buf.backward(torch.ones_like(buf))                 # the persistent backward has seams of its own
            torch.cuda.synchronize()
        with pytest.raises(ops.SeamTimeoutError):
            ops.check_device_errors(x.device)
    finally:
        dn.SEAM_MAX_SPINS = 0
        dn.set_weight_provider(None)
    # with the default bound the same launches are clean
    _run(blk, x, True)
    ops.check_device_errors(x.device)


def test_trainstep_polls_the_error_words_without_a_sync():
    from mclstexp_amd import densenet_fused as dn, ops, synth
    from mclstexp_amd.engine import TrainStep
    from mclstexp_amd.model import mclSTExp_Attention
    from mclstexp_amd.optim import FusedAdam
    G = 171
    torch.manual_seed(0)
    m = mclSTExp_Attention("densenet121", 1.0, 1024, G, 256, 8, 64, 2, embedding_grad="rowsparse",
                           backbone_dtype=torch.bfloat16).to(DEV).train()
    opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(m)
    tr = TrainStep(m, opt, None, graphs=True, warmup=2)
    tr.error_poll_every = 1
    ops.check_device_errors(torch.device(DEV, torch.cuda.current_device()))
    dn.SEAM_MAX_SPINS = 1                                       # captured into the graph's kernel arguments too
    try:
        with pytest.raises(ops.SeamTimeoutError):
            for s in range(8):
                batch = {k: v.to(DEV) for k, v in synth.make_batch(32, G, seed=s, image_hw=224).items()}
                tr(batch)
                torch.cuda.synchronize()                         # (only so that the NEXT call is sure to find the copy done)
    finally:
        dn.SEAM_MAX_SPINS = 0
        dn.set_weight_provider(None)
        torch.cuda.synchronize()
        ops.device_error_words(torch.device(DEV, torch.cuda.current_device())).zero_()
