"""Algorithmic HBM bytes and FLOPs of the hot C-ABI launch units of the training step.

``TABLE[abi_name] = {"kernels": ..., "bytes": f(args), "strict": f(args), "flops": f(args), "bound": ...}`` where ``args``
is the argument tuple of the ctypes call (positions as in ``_lib.PROTOTYPES`` / include/mclstexp_hip.h).

* ``bytes``  -- "algorithmic" as the launch unit is BUILT: every operand element read once and every result element
  written once PER PASS (a unit made of two kernels that hand a tensor over through HBM counts it written and re-read).
* ``strict`` -- single-pass bytes: every distinct operand / result element once, whatever the unit's internal passes.
* ``flops``  -- 2 x MACs of the contraction the unit performs (0 for the streaming units).
* ``bound``  -- which roofline the unit is priced against first: "mfma/lds" (the 3x3 family: MFMA- and LDS-paced), "hbm",
  or None = decided per launch (bench.py labels a unit "latency" when its working set is < 256 MB -- it lives in the
  Infinity Cache -- and its average launch is < 30 us).

Re-reads, partial sums and workspaces are NOT counted -- they are what ``roofline.traffic`` (PMC counters) exposes.
Used by bench.py (`roofline`), tools/pmc_summary.py and the DESIGN.md tables; nothing here touches the GPU.
"""
from __future__ import annotations


def _conv1x1_wrw_det(a):        # (dz, ldz, a, lda, gamma, beta, mean, rstd, ws, dW, acc, S, M, N, stream)
    S, M, N = a[11], a[12], a[13]
    return 2 * S * (M + N) + 4 * M * N


def _conv1x1_wrw_det_flops(a):
    return 2 * a[11] * a[12] * a[13]


def _conv3x3_wrw_det(a):        # (dy, lddy, z, S, H, W, gamma, beta, mean, rstd, ws, dW, acc, stream)
    S = a[3]
    return 2 * S * (32 + 128) + 4 * 32 * 1152


def _conv3x3_flops_S3(a):
    return 2 * a[3] * 32 * 1152


def _bn1_bwd(a):                # (dz, w16, C, x, ldx, S, ...): reduce pass reads dz + x; dx pass reads dz + x + g, writes g
    C, S = a[2], a[5]
    return 2 * S * (128 + C) + 2 * S * (128 + 3 * C)


def _bn1_bwd_strict(a):         # dz, x, g read once, g written once
    C, S = a[2], a[5]
    return 2 * S * (128 + 3 * C)


def _bn1_flops(a):              # g = dz W1 per pass
    return 2 * a[5] * 128 * a[2]


def _conv3x3_bwd(a):            # (dy, lddy, S, H, W, ...): dy + z read, da2 written; then da2 + z read, dz written
    S = a[2]
    return 2 * S * (32 + 128 + 128) + 2 * S * (128 + 128 + 128)


def _conv3x3_bwd_strict(a):     # dy + z read, dz written
    return 2 * a[2] * (32 + 128 + 128)


def _conv3x3_flops_S2(a):
    return 2 * a[2] * 32 * 1152


def _conv3x3_bwd_fix(a):        # + the layer's 32 concat-buffer channels read and the corrected dy written (folded bn1_fix)
    return _conv3x3_bwd(a) + 2 * a[2] * (32 + 32)


def _conv3x3_bwd_fix_strict(a):
    return _conv3x3_bwd_strict(a) + 2 * a[2] * (32 + 32)


def _conv1x1_fwd(a):            # (x, ldx, S, K, ...)
    S, K = a[2], a[3]
    return 2 * S * (K + 128)


def _conv1x1_fwd_flops(a):
    return 2 * a[2] * a[3] * 128


def _conv3x3_fwd(a):            # (z, S, H, W, ...)
    S = a[1]
    return 2 * S * (128 + 32)


def _conv3x3_flops_S1(a):
    return 2 * a[1] * 32 * 1152


def _bn1_wrw(a):                # (dz, W1, C, x, ldx, S, ...): dz + x read once, dW1 written
    C, S = a[2], a[5]
    return 2 * S * (128 + C) + 4 * 128 * C


def _bn1_wrw_flops(a):          # two Gram matrices
    return 2 * 2 * a[5] * 128 * a[2]


def _bn1_dx(a):                 # (dz, W1, C, x, ldx, S, ...): dz + x + g read, g written
    C, S = a[2], a[5]
    return 2 * S * (128 + 3 * C)


def _bn1_dx_window(a):          # (dz, W1, C, c0, nc, x, ldx, S, ...): dz + x + g read, g written on nc of the layer's C channels
    nc, S = a[4], a[7]
    return 2 * S * (128 + 3 * nc)


def _bn1_dx_window_flops(a):
    return 2 * a[7] * 128 * a[4]


def _bn1_dx_pair(a):            # (dzA, W1A, ldwA, gA, bA, cA, dzB, W1B, gB, bB, cB, C, x, ldx, S, ...): both dz + x + g read, g written
    C, S = a[11], a[14]
    return 2 * S * (2 * 128 + 3 * C)


def _bn1_dx_pair_flops(a):      # two dz W1 products
    return 2 * 2 * a[14] * 128 * a[11]


def _bn1_dx_sums(a):            # (dz, W1, C, x, ldx, S, ...): dz + x + g read, g written -- the whole BatchNorm-1 backward of a layer
    C, S = a[2], a[5]
    return 2 * S * (128 + 3 * C)


def _bn1_fix(a):                # (x, ldx, gbuf, ldg, S, c0, nc, ...): x + g read, g written on nc channels
    return 2 * a[4] * 3 * a[6]


def _adam_table(a):             # (p, m, v, rows, cols, ...): read p, m, v; write p, m, v
    return 24 * a[3] * a[4]


def _adam_table_lazy(a):        # (p0, m0, v0, rs0, p1, ..., n_rows, cols, pos, own0, own1, n_owner, rg0, rg1, ...): <= n_owner rows per table
    tables = 2 if a[4] else 1
    return tables * a[13] * a[9] * (24 + (4 if a[14] else 0))


def _block_fwd(a):              # (buf, B, H, W, Ct, C0, L, ...): block input read once, z + the new channels written once, weights
    B, HW, Ct, C0, L = a[1], a[2] * a[3], a[4], a[5], a[6]
    w = sum(128 * (C0 + 32 * l) * 2 for l in range(L)) + L * 32 * 1152 * 2
    return 2 * B * HW * C0 + L * 2 * B * HW * 128 + 2 * B * HW * 32 * L + w


def _block_fwd_flops(a):
    B, HW, C0, L = a[1], a[2] * a[3], a[5], a[6]
    return sum(2 * B * HW * (128 * (C0 + 32 * l) + 1152 * 32) for l in range(L))


def _block_bwd(a):              # (buf, gbuf, B, H, W, Ct, C0, L, ...): gbuf + buf read once, z read, dz + dy' written, dx written, weights
    B, HW, Ct, C0, L = a[2], a[3] * a[4], a[5], a[6], a[7]
    w = sum(128 * (C0 + 32 * l) * 2 for l in range(L)) + L * 32 * 1152 * 2
    return 2 * 2 * B * HW * Ct + 2 * B * HW * C0 + L * 2 * B * HW * (128 + 128 + 32) + w


def _block_bwd_flops(a):
    B, HW, C0, L = a[2], a[3] * a[4], a[6], a[7]
    return sum(2 * B * HW * (128 * (C0 + 32 * l) + 1152 * 32) for l in range(L))


def _adam(a):                   # (p, g, m, v, n, ...): read p, g, m, v; write p, m, v
    return 28 * a[4]


def _adam_shadow(a):            # + the bf16 shadow written
    return 30 * a[4]


def _zero(a):
    return 0


def _e(kernels, b, strict=None, flops=_zero, bound=None, unit=None):
    """``unit``: the launch unit this entry point is reported under (two entry points that run the same kernels on different
    layers -- with / without the folded bn1_fix -- are one row of the roofline table)."""
    return {"kernels": kernels, "bytes": b, "strict": strict or b, "flops": flops, "bound": bound, "unit": unit}


TABLE = {
    "mcl_conv1x1_wrw_det": _e("wrw_partial_kernel + wrw_merge_kernel", _conv1x1_wrw_det, flops=_conv1x1_wrw_det_flops),
    "mcl_dense_conv3x3_wrw_det": _e("conv3x3_wrw_rows_kernel (56x56, 28x28 maps) / conv3x3_wrw_ky_kernel + wrw_merge_kernel",
                                    _conv3x3_wrw_det, flops=_conv3x3_flops_S3,
                                    bound="mfma/lds"),
    "mcl_dense_bn1_wrw": _e("wrw_partial_kernel<Gram> + wrw_merge_kernel", _bn1_wrw, flops=_bn1_wrw_flops),
    "mcl_dense_bn1_dx": _e("bn1_bwd_kernel<1> (one layer) / bn1_dx_pair_kernel (two consecutive layers' passes as one: "
                           "mcl_dense_bn1_dx_pair, preceded by the 32-channel mcl_dense_bn1_dx_window)", _bn1_dx, flops=_bn1_flops),
    "mcl_dense_bn1_dx_window": _e("bn1_bwd_kernel<1> on a 32-channel window", _bn1_dx_window, flops=_bn1_dx_window_flops,
                                  unit="mcl_dense_bn1_dx"),
    "mcl_dense_bn1_dx_pair": _e("bn1_dx_pair_kernel", _bn1_dx_pair, flops=_bn1_dx_pair_flops, unit="mcl_dense_bn1_dx"),
    "mcl_dense_bn1_dx_sums": _e("bn1_bwd_kernel<2> (single pass: dx data term + previous layer's mean terms + sums) + "
                                "bn1_bwd_finalize_kernel", _bn1_dx_sums, flops=_bn1_flops),
    "mcl_dense_bn1_fix": _e("bn1_fix_kernel", _bn1_fix, bound="latency"),
    "mcl_dense_bn1_bwd": _e("bn1_bwd_kernel<0> + bn1_bwd_finalize_kernel + bn1_bwd_kernel<1>", _bn1_bwd, _bn1_bwd_strict,
                            flops=lambda a: 2 * _bn1_flops(a)),
    "mcl_dense_conv3x3_bwd": _e("conv3x3_bwd_rows_kernel (56x56, 28x28 maps) / conv3x3_bwd_kernel (14x14, 7x7; with the "
                                "folded bn1_fix as mcl_dense_conv3x3_bwd_fix) + bn1_bwd_finalize_kernel + bn2_dz_kernel",
                                _conv3x3_bwd, _conv3x3_bwd_strict, flops=_conv3x3_flops_S2, bound="mfma/lds"),
    "mcl_dense_conv3x3_bwd_fix": _e("conv3x3_bwd_kernel (folded bn1_fix) + bn1_bwd_finalize_kernel + bn2_dz_kernel",
                                    _conv3x3_bwd_fix, _conv3x3_bwd_fix_strict, flops=_conv3x3_flops_S2, bound="mfma/lds",
                                    unit="mcl_dense_conv3x3_bwd"),
    "mcl_dense_conv1x1_fwd": _e("conv1x1_fwd_kernel + tile_stats_finalize_kernel", _conv1x1_fwd, flops=_conv1x1_fwd_flops),
    "mcl_dense_conv3x3_fwd": _e("conv3x3_fwd_rows_kernel + sums_finalize_kernel (56x56, 28x28 maps) / conv3x3_fwd_kernel + "
                                "tile_stats_finalize_kernel", _conv3x3_fwd, flops=_conv3x3_flops_S1, bound="mfma/lds"),
    "mcl_adam_table_step_dev": _e("adam_table_kernel", _adam_table, bound="hbm"),
    "mcl_dense_block_fwd": _e("dense_block_fwd_kernel (a whole 7x7 dense block forward: one persistent launch, in-launch batch-statistics "
                              "seams) + zero_words_kernel", _block_fwd, flops=_block_fwd_flops, bound="latency"),
    "mcl_dense_block_bwd": _e("dense_block_bwd_kernel (the block's data-gradient chain: one persistent launch) + zero_words_kernel",
                              _block_bwd, flops=_block_bwd_flops, bound="latency"),
    "mcl_adam_table_lazy": _e("adam_table_lazy_kernel (catch-up of gathered rows / update of the rows with a gradient)",
                              _adam_table_lazy, bound="latency"),
    "mcl_adam_step_dev": _e("adam_kernel", _adam, bound="hbm"),
    "mcl_adam_step_dev_shadow": _e("adam_kernel<shadow> (update + bf16 shadow of the parameters)", _adam_shadow, bound="hbm"),
}
