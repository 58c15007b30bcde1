#!/usr/bin/env python3
"""The DenseNet stem's kernels alone at the benched shape (B x 3 x 224 x 224): conv0 forward (+ statistics), conv0 weight
gradient, max-pool backward -- HIP-event time per launch over a replayed graph of `reps` launches."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mclstexp_amd import _lib  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--hw", type=int, default=224)
    args = ap.parse_args()
    B, H = args.batch, args.hw
    L = _lib.lib()
    dev = "cuda"
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    x = torch.randn(B, H, H, 3, device=dev).bfloat16()
    w = (torch.randn(64, 7, 7, 3, device=dev) * 0.05).bfloat16()
    y = torch.empty(B, H // 2, H // 2, 64, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(B, H // 2, H // 2, 64, device=dev).bfloat16()
    ws = torch.empty(L.mcl_conv0_workspace_floats(B, H, H), device=dev)
    mean, var, rstd = (torch.empty(64, device=dev) for _ in range(3))
    wsw = torch.empty(L.mcl_conv0_wrw_workspace_floats(B, H, H), device=dev)
    dW = torch.zeros(64, 7, 7, 3, device=dev)
    out = {"batch": B, "hw": H}
    out["conv0_fwd_us"] = round(timed(lambda: _lib.check(L.mcl_conv0_fwd(x.data_ptr(), B, H, H, w.data_ptr(), y.data_ptr(), ws.data_ptr(), 1e-5,
                                                                         mean.data_ptr(), var.data_ptr(), rstd.data_ptr(), st()))), 2)
    out["conv0_wrw_us"] = round(timed(lambda: _lib.check(L.mcl_conv0_wrw(x.data_ptr(), B, H, H, dy.data_ptr(), wsw.data_ptr(), dW.data_ptr(), 1,
                                                                         st()))), 2)
    OH = H // 4
    yp = torch.empty(B, OH, OH, 64, device=dev, dtype=torch.bfloat16)
    idx = torch.empty(B, OH, OH, 64, device=dev, dtype=torch.uint8)
    _lib.check(L.mcl_maxpool3s2_nhwc_bf16_fwd(y.data_ptr(), yp.data_ptr(), idx.data_ptr(), B, H // 2, H // 2, 64, st()))
    ga, be_ = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    out["bn_act_maxpool_fwd_us"] = round(timed(lambda: _lib.check(L.mcl_bn_act_maxpool_fwd(y.data_ptr(), B, H // 2, H // 2, 64, ga.data_ptr(), be_.data_ptr(),
                                                                                          mean.data_ptr(), rstd.data_ptr(), yp.data_ptr(), idx.data_ptr(),
                                                                                          st()))), 2)
    gbuf = torch.randn(B, OH, OH, 256, device=dev).bfloat16()          # dy = the first 64 channels of the block's gradient buffer
    dxp = torch.empty(B, H // 2, H // 2, 64, device=dev, dtype=torch.bfloat16)
    out["maxpool_bwd_us"] = round(timed(lambda: _lib.check(L.mcl_maxpool3s2_nhwc_bf16_bwd_ld(idx.data_ptr(), gbuf.data_ptr(), 256, dxp.data_ptr(),
                                                                                             B, H // 2, H // 2, 64, st()))), 2)
    out["MCL_CONV0_WRW_DBG"] = os.environ.get("MCL_CONV0_WRW_DBG", "0")
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
