"""Algorithmic HBM bytes of the hot C-ABI launch units of the training step.

``TABLE[abi_name] = {"kernels": "<kernel names the call enqueues>", "bytes": f(args) -> bytes of ONE call}`` where
``args`` is the argument tuple of the ctypes call (positions as in ``_lib.PROTOTYPES`` / include/mclstexp_hip.h).
"Algorithmic" = every operand element read once and every result element written once, at the element sizes the
kernel is specified for (bf16 activations, fp32 parameters / statistics); re-reads, partial sums and workspaces
are NOT counted -- they are what ``roofline.traffic`` (PMC counters) exposes.

Used by bench.py (`roofline`), tools/pmc_summary.py and the DESIGN.md tables; nothing here touches the GPU.
"""
from __future__ import annotations


def _conv1x1_wrw(a):            # (dz, ldz, a, lda, gamma, beta, mean, rstd, dW, lddw, S, M, N, stream)
    S, M, N = a[10], a[11], a[12]
    return 2 * S * (M + N) + 4 * M * N


def _conv3x3_wrw(a):            # (dy, lddy, z, S, H, W, gamma, beta, mean, rstd, dW, stream)
    S = a[3]
    return 2 * S * (32 + 128) + 4 * 32 * 1152


def _bn1_bwd(a):                # (dz, w16, C, x, ldx, S, ...): reduce pass reads dz + x; dx pass reads dz + x + g, writes g
    C, S = a[2], a[5]
    return 2 * S * (128 + C) + 2 * S * (128 + 3 * C)


def _conv3x3_bwd(a):            # (dy, lddy, S, H, W, ...): dy + z read, da2 written; then da2 + z read, dz written
    S = a[2]
    return 2 * S * (32 + 128 + 128) + 2 * S * (128 + 128 + 128)


def _conv1x1_fwd(a):            # (x, ldx, S, K, ...)
    S, K = a[2], a[3]
    return 2 * S * (K + 128)


def _conv3x3_fwd(a):            # (z, S, H, W, ...)
    S = a[1]
    return 2 * S * (128 + 32)


def _conv1x1_wrw_det(a):        # (dz, ldz, a, lda, gamma, beta, mean, rstd, ws, dW, acc, S, M, N, stream)
    S, M, N = a[11], a[12], a[13]
    return 2 * S * (M + N) + 4 * M * N


def _conv3x3_wrw_det(a):        # (dy, lddy, z, S, H, W, gamma, beta, mean, rstd, ws, dW, acc, stream)
    S = a[3]
    return 2 * S * (32 + 128) + 4 * 32 * 1152


def _bn1_wrw(a):                # (dz, W1, C, x, ldx, S, ...): dz + x read once, dW1 written
    C, S = a[2], a[5]
    return 2 * S * (128 + C) + 4 * 128 * C


def _bn1_dx(a):                 # (dz, W1, C, x, ldx, S, ...): dz + x + g read, g written
    C, S = a[2], a[5]
    return 2 * S * (128 + 3 * C)


def _adam_table(a):             # (p, m, v, rows, cols, ...): read p, m, v; write p, m, v
    return 24 * a[3] * a[4]


def _adam(a):                   # (p, g, m, v, n, ...): read p, g, m, v; write p, m, v
    return 28 * a[4]


TABLE = {
    "mcl_conv1x1_wrw_det": {"kernels": "wrw_partial_kernel + wrw_merge_kernel", "bytes": _conv1x1_wrw_det},
    "mcl_dense_conv3x3_wrw_det": {"kernels": "conv3x3_wrw_kernel (56x56 maps) / conv3x3_wrw_ky_kernel + wrw_merge_kernel", "bytes": _conv3x3_wrw_det},
    "mcl_dense_bn1_wrw": {"kernels": "wrw_partial_kernel<Gram> + wrw_merge_kernel", "bytes": _bn1_wrw},
    "mcl_dense_bn1_dx": {"kernels": "bn1_bwd_kernel<1>", "bytes": _bn1_dx},
    "mcl_conv1x1_wrw_bf16": {"kernels": "conv1x1_wrw_kernel (atomics, A/B only)", "bytes": _conv1x1_wrw},
    "mcl_dense_conv3x3_wrw": {"kernels": "conv3x3_wrw_kernel (atomics, A/B only)", "bytes": _conv3x3_wrw},
    "mcl_dense_bn1_bwd": {"kernels": "bn1_bwd_kernel<0> + bn1_bwd_finalize_kernel + bn1_bwd_kernel<1>", "bytes": _bn1_bwd},
    "mcl_dense_conv3x3_bwd": {"kernels": "conv3x3_bwd_kernel + finalize + bn2_dz_kernel", "bytes": _conv3x3_bwd},
    "mcl_dense_conv1x1_fwd": {"kernels": "conv1x1_fwd_kernel + tile_stats_finalize_kernel", "bytes": _conv1x1_fwd},
    "mcl_dense_conv3x3_fwd": {"kernels": "conv3x3_fwd_kernel + tile_stats_finalize_kernel", "bytes": _conv3x3_fwd},
    "mcl_adam_table_step_dev": {"kernels": "adam_table_kernel", "bytes": _adam_table},
    "mcl_adam_step_dev": {"kernels": "adam_kernel", "bytes": _adam},
}
