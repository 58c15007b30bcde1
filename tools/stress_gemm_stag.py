#!/usr/bin/env python3
"""Race screen of gemm_bf16_stag_kernel: many interior-tile problems (random M, N multiples of 256, K multiples of 64, all four operand
layouts, with other work in flight on a second stream), every result compared bit for bit with the lockstep kernel's."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mclstexp_amd import vit_fused as vf  # noqa: E402

BF = torch.bfloat16
random.seed(0)
torch.manual_seed(0)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
side = torch.cuda.Stream()
noise = torch.randn(64 << 20, device="cuda")
bad = 0
for it in range(n_iter):
    M = 256 * random.randint(2, 24)
    N = 256 * random.randint(1, 12)
    K = 64 * random.randint(1, 40)
    akm, bkm = random.randint(0, 1), random.randint(0, 1)
    A = (torch.rand((K, M) if akm else (M, K), device="cuda") * 2 - 1).to(BF)
    B = (torch.rand((K, N) if bkm else (N, K), device="cuda") * 2 - 1).to(BF)
    outs = []
    for stag in ("1", "0"):
        os.environ["MCL_GEMM_STAG"] = stag
        C = torch.zeros((M, N), device="cuda", dtype=BF)
        if stag == "1":
            with torch.cuda.stream(side):                      # memory traffic beside the kernel under test
                noise.mul_(1.0001)
        for _ in range(3 if stag == "1" else 1):               # the staggered kernel three times: any run may expose a race
            vf.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, flags=akm * vf.A_KM | bkm * vf.B_KM)
            outs.append(C.clone())
    torch.cuda.synchronize()
    ref = outs[-1]
    for o in outs[:-1]:
        if not torch.equal(o, ref):
            bad += 1
            print("MISMATCH", M, N, K, akm, bkm, int((o != ref).sum()), flush=True)
print(f"{n_iter} problems, {bad} mismatches")
sys.exit(1 if bad else 0)
