"""Input pipeline on the MI355X against PIL's outputs (tests/golden/input_patches.npz) and the numpy
oracle (oracle/ref_input.py).  pytest -m gpu."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR, INPUT_CASE, assert_close, synthetic_slide

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ip():
    from mclstexp_amd import _lib, input_pipeline
    _lib.lib()
    return input_pipeline


def test_eval_patches_bit_exact_vs_pil(ip):
    z = np.load(os.path.join(GOLDEN_DIR, "input_patches.npz"))
    img = ip.to_device_image(synthetic_slide())
    out = ip.her2st_eval_patches(img, INPUT_CASE["centers_xy"], r=INPUT_CASE["r"])
    assert out.dtype == torch.float32 and tuple(out.shape) == z["eval"].shape
    assert np.array_equal(out.cpu().numpy(), z["eval"])              # uint8 / 255 in fp32: bit-exact
    # bf16 NHWC form for the backbone kernels = the fp32 result rounded to bf16
    o16 = ip.her2st_eval_patches(img, INPUT_CASE["centers_xy"], r=INPUT_CASE["r"], layout="nhwc_bf16")
    assert o16.dtype == torch.bfloat16 and o16.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(o16.float().cpu(), torch.from_numpy(z["eval"]).to(torch.bfloat16).float())


def test_tenx_patches_bit_exact_vs_pil(ip):
    z = np.load(os.path.join(GOLDEN_DIR, "input_patches.npz"))
    img = ip.to_device_image(synthetic_slide())
    r = INPUT_CASE["r"]
    out = ip.extract_patches(img, INPUT_CASE["tenx_centers"], r, INPUT_CASE["hflip"], INPUT_CASE["vflip"],
                             [(a % 360) // 90 for a in INPUT_CASE["angle"]], divisor=1.0)
    assert np.array_equal(out.cpu().numpy(), z["tenx"])
    with pytest.raises(ValueError):
        ip.tenx_patches(img, [(50, 50)], [0], [0], [45])


def test_patches_at_scale_vs_oracle(ip):
    """A 2000 x 3000 slide, 300 patches of 224 x 224 with random flips / quarter turns, border-crossing centres."""
    from oracle import ref_input
    rng = np.random.default_rng(0)
    slide = rng.integers(0, 256, size=(2000, 3000, 3), dtype=np.uint8)
    n, r = 300, 112
    centers = np.stack([rng.integers(-50, 2050, n), rng.integers(-50, 3050, n)], 1)
    hf, vf, k = rng.integers(0, 2, n), rng.integers(0, 2, n), rng.integers(0, 4, n)
    out = ip.extract_patches(ip.to_device_image(slide), centers, r, hf, vf, k, divisor=255.0).cpu().numpy()
    for i in range(0, n, 7):
        ref = ref_input.to_tensor(ref_input.tenx_transform(ref_input.crop(slide, centers[i, 0], centers[i, 1], r),
                                                           bool(hf[i]), bool(vf[i]), int(k[i]) * 90))
        assert np.array_equal(out[i], ref), i


def test_log_library_size_normalize(ip):
    from oracle import ref_input
    rng = np.random.default_rng(1)
    counts = rng.poisson(2.0, size=(333, 785)).astype(np.float32) * (rng.random((333, 785)) < 0.3)
    counts[5] = 0.0
    y = ip.log_library_size_normalize(counts)
    assert_close(y.cpu().numpy(), ref_input.log_library_size_normalize(counts), 2e-6, 2e-6, what="log-normalised expression")
    assert (y[5] == 0).all()


def test_her2st_train_patches_bit_exact_vs_pil(ip):
    """The HER2ST / cSCC TRAIN transform (dataset.py:63-68) on the GPU against PIL's outputs for the same explicit draws
    (tests/golden/input_augment.npz): every order of the colour adjustments, factors below / at / above 1, flips,
    quarter-turn and arbitrary rotation angles, border-crossing crops -- bit-exact."""
    from helpers import AUG_CASE
    z = np.load(os.path.join(GOLDEN_DIR, "input_augment.npz"))
    img = ip.to_device_image(synthetic_slide())
    draws = {k: np.asarray(AUG_CASE[k]) for k in ("order", "brightness", "contrast", "saturation", "hflip", "angle")}
    out = ip.her2st_train_patches(img, AUG_CASE["centers_xy"], r=AUG_CASE["r"], draws=draws)
    assert out.dtype == torch.float32 and tuple(out.shape) == z["train"].shape
    assert np.array_equal(out.cpu().numpy(), z["train"])
    o16 = ip.her2st_train_patches(img, AUG_CASE["centers_xy"], r=AUG_CASE["r"], draws=draws, layout="nhwc_bf16")
    assert torch.equal(o16.float().cpu(), torch.from_numpy(z["train"]).to(torch.bfloat16).float())


def test_her2st_train_patches_at_scale_vs_oracle(ip):
    """128 patches of 224 x 224 (the reference's size) from a 2000 x 3000 slide with sampled draws: every 5th patch against
    the PIL-pinned numpy oracle, bit-exact; values stay in [0, 1]."""
    from oracle import ref_input
    rng = np.random.default_rng(3)
    slide = rng.integers(0, 256, size=(2000, 3000, 3), dtype=np.uint8)
    n, r = 128, 112
    cxy = np.stack([rng.integers(-40, 3040, n), rng.integers(-40, 2040, n)], 1)
    g = torch.Generator().manual_seed(0)
    draws = ip.sample_her2st_draws(n, g)
    draws["angle"][:4] = [0.0, 90.0, 180.0, -90.0]
    out = ip.her2st_train_patches(ip.to_device_image(slide), cxy, r, draws=draws).cpu().numpy()
    assert out.min() >= 0.0 and out.max() <= 1.0
    for i in range(0, n, 5):
        ref = ref_input.her2st_train_transform(ref_input.crop(slide, cxy[i, 1], cxy[i, 0], r), draws["order"][i],
                                               float(draws["brightness"][i]), float(draws["contrast"][i]),
                                               float(draws["saturation"][i]), bool(draws["hflip"][i]), float(draws["angle"][i]))
        assert np.array_equal(out[i], ref), i
    # sampled draws have the reference's ranges
    assert (draws["brightness"] >= 0.5).all() and (draws["brightness"] <= 1.5).all()
    assert (np.abs(draws["angle"]) <= 180.0).all() and sorted(set(map(tuple, draws["order"].tolist()))) <= sorted(
        [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)])
