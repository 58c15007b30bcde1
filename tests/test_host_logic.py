"""CPU: host-side surface that mirrors the reference (flags, module tree, checkpoint keys, meters,
synthetic data contract).  No kernels are launched."""
import numpy as np
import pytest
import torch

from mclstexp_amd import synth


def test_generate_args_keeps_reference_flags_and_defaults():
    """train.py:11-27: the 13 flags, names/defaults unchanged."""
    from mclstexp_amd.train import generate_args
    a = generate_args([])
    ref = dict(batch_size=128, max_epochs=90, temperature=1.0, fold=0, dim=785, image_embedding_dim=1024,
               projection_dim=256, heads_num=8, heads_dim=64, heads_layers=2, dropout=0.0, dataset="her2st",
               encoder_name="densenet121")
    for k, v in ref.items():
        assert getattr(a, k) == v, k
    b = generate_args(["--batch_size", "8", "--dim", "1000", "--encoder_name", "vit", "--image_embedding_dim", "768"])
    assert (b.batch_size, b.dim, b.encoder_name, b.image_embedding_dim) == (8, 1000, "vit", 768)


def test_state_dict_keys_match_reference_layout():
    """SURVEY Appendix A.3 keys (captured from the imported reference)."""
    from mclstexp_amd.model import mclSTExp_Attention
    m = mclSTExp_Attention("densenet121", 1.0, 1024, 171, 256, 8, 64, 2)
    keys = set(m.state_dict())
    want = set(synth.param_spec(171, 1024))
    assert want <= keys
    assert "image_encoder.model.0.denseblock1.denselayer1.conv1.weight" in keys
    assert "image_encoder.model.0.norm5.running_mean" in keys
    assert "image_encoder.model.0.transition3.conv.weight" in keys
    non_backbone = {k for k in keys if not k.startswith("image_encoder.")}
    assert non_backbone == want
    for attr in ("image_encoder", "image_projection", "x_embed", "y_embed", "spot_encoder", "spot_projection",
                 "temperature"):
        assert hasattr(m, attr)                 # evel_her2st.py:48-69 uses these directly
    assert m.x_embed.weight.shape == (65536, 171)
    n = sum(p.numel() for p in m.image_encoder.parameters())
    assert n == 6953856                          # torchvision densenet121.features parameter count


def test_unknown_encoder_raises_and_mlp_attr_name():
    from mclstexp_amd.model import mclSTExp_Attention, mclSTExp_MLP
    with pytest.raises(ValueError):
        mclSTExp_Attention("nope", 1.0, 1024, 171, 256, 8, 64, 2)
    m = mclSTExp_MLP(1.0, 1024, 171, 256, encoder_name="identity")
    assert hasattr(m, "image_ecode")             # model.py:176 (sic)


def test_checkpoint_key_rewrites():
    """evel_her2st.py:33-37: strip 'module.', rename 'well' -> 'spot'."""
    from mclstexp_amd.model import load_reference_state_dict, mclSTExp_Attention
    m = mclSTExp_Attention("identity", 1.0, 1024, 171, 256, 8, 64, 1)
    sd = {("module." + k).replace("spot", "well"): v.clone() + 1 for k, v in m.state_dict().items()}
    load_reference_state_dict(m, sd)
    assert torch.equal(m.state_dict()["spot_projection.fc.bias"], sd["module.well_projection.fc.bias"])


def test_avgmeter_and_get_lr():
    from mclstexp_amd.utils import AvgMeter, get_lr
    a = AvgMeter()
    a.update(2.0, 128); a.update(4.0, 64)
    assert abs(a.avg - (2 * 128 + 4 * 64) / 192) < 1e-12 and "Metric" in repr(a)
    p = torch.nn.Parameter(torch.zeros(1))
    assert get_lr(torch.optim.Adam([p], lr=1e-4)) == 1e-4


def test_synth_is_deterministic_and_row_addressable():
    full = synth.uniform_tensor("x_embed.weight", (65536, 7), -1, 1, seed=0)
    rows = synth.uniform_tensor("x_embed.weight", (65536, 7), -1, 1, seed=0, rows=np.array([0, 5, 60000]))
    assert torch.equal(rows, full[[0, 5, 60000]])
    b = synth.make_batch(8, 785, image_dim=1024, seed=0)
    assert b["expression"].shape == (8, 785) and b["position"].shape == (8, 2) and b["image"].shape == (8, 1024)
    assert float((b["expression"] == 0).float().mean()) > 0.6
    assert torch.equal(b["position"], torch.floor(b["position"])) and b["position"].max() < 64
    assert torch.equal(synth.make_batch(8, 785, image_dim=1024, seed=0)["expression"], b["expression"])


def test_vit_and_resnet_shapes_cpu():
    from mclstexp_amd.backbones import ImageEncdoer_res18, ImageEncoder_VIT
    x = torch.rand(2, 3, 224, 224)
    with torch.no_grad():
        assert ImageEncoder_VIT().eval()(x).shape == (2, 768)       # model.py:104-116, patch32
        assert ImageEncdoer_res18().eval()(x[:, :, :64, :64]).shape == (2, 512)


def test_kernel_audit_classifies_library_kernels():
    """mclstexp_amd.kernel_audit.foreign(): library kernels (hipBLASLt / Tensile, ATen, MIOpen, rocPRIM) are flagged, this
    library's kernels and the runtime's copy / fill helpers are not."""
    from mclstexp_amd import kernel_audit
    names = ["Cijk_Ailk_Bljk_BBS_BH_Bias_HA_S_SAV_UserArgs_MT256x224x32",
             "void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>, std::array<char*, 1ul> >(int)",
             "void at::native::(anonymous namespace)::multi_tensor_apply_kernel<...>", "miopenSp3AsmConv_v30_3_1_gfx9_fp32",
             "void rocprim::detail::sort_kernel<...>",
             "(anonymous namespace)::conv1x1_fwd_kernel<2, 32, 4>(unsigned short const*, long long)",
             "(anonymous namespace)::fill_zero_kernel(unsigned int*, long long)", "__amd_rocclr_copyBuffer", "Memcpy DtoD (Device -> Device)",
             "void (anonymous namespace)::gemm_bf16_kernel<false, true, 2>((anonymous namespace)::GemmB)"]
    bad = kernel_audit.foreign(names)
    assert len(bad) == 5 and all(("conv1x1" not in n and "fill_zero" not in n and "rocclr" not in n and "gemm_bf16" not in n)
                                 for n in bad)


def test_train_flags_include_fp8_infonce():
    from mclstexp_amd.train import generate_args
    a = generate_args(["--infonce", "fp8"])
    assert a.infonce == "fp8" and a.encoder_name == "densenet121" and a.dim == 785     # reference defaults untouched


def test_retrieval_filter_plan_keeps_the_expected_list_inside_its_capacity():
    """mclstexp_amd.retrieval._filter_plan (host arithmetic only): for every key count the expected candidate-list length
    stays well inside the capacity and above k -- the r05 plan overflowed for N > ~262k."""
    from mclstexp_amd import retrieval as rt
    for n in (8192, 20000, 65536, 262144, 300000, 1_000_000, 10_000_000):
        for k in (1, 50, 200, 600, 1024):
            if 16 * k > n:
                continue
            ns, step, r, cap = rt._filter_plan(n, k)
            assert ns <= n and step >= 1 and ns * step <= n and step <= rt.FUSED_MAX_STRIDE
            expect = r * n / ns
            assert expect >= 1.4 * k, (n, k)
            assert expect * (1 + 4 / r ** 0.5) < cap, (n, k, expect, cap)      # four standard deviations of the count
