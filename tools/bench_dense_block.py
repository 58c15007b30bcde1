"""Forward of DenseNet-121's denseblock4 (7 x 7 maps, B = 128) on the persistent one-launch kernel (csrc/dense_block.hip) vs the
per-layer launch sequence; HIP events around the whole block, eager launches and a captured graph."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mclstexp_amd import densenet_fused as dn
from mclstexp_amd.backbones import densenet121_features_module

DEV = "cuda"
B = int(os.environ.get("B", "128"))
torch.manual_seed(0)
blk = densenet121_features_module().denseblock4.to(DEV).train()
x = torch.randn(B, 512, 7, 7, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def fwd():
    with torch.no_grad():
        return dn.dense_block(blk, x, dn._RunningStats())[0]


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = {}
for mode in (True, False, True, False):
    dn.USE_BLOCK_PERSISTENT = mode
    fwd()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fwd()
    t_graph = timeit(g.replay)
    t_eager = timeit(fwd, 20)
    out.setdefault("persistent" if mode else "per_layer", []).append({"graph_us": round(t_graph, 1), "eager_us": round(t_eager, 1)})
    del g
# in-kernel phase stamps of the persistent kernel (100 MHz wall clock): per layer, median over images
from mclstexp_amd import _lib
L = 16
st = torch.zeros(B * L * 8 + 64, device=DEV, dtype=torch.int64)
dn.BLOCK_STAMPS = st
dn.USE_BLOCK_PERSISTENT = True
fwd(); torch.cuda.synchronize()
st.zero_()
fwd(); torch.cuda.synchronize()
dn.BLOCK_STAMPS = None
cs = st[B * L * 8:].cpu().view(16, 4)
t = st[:B * L * 8].view(B, L, 8).double().cpu() * 0.01          # us
def med(x):
    return " ".join(f"{v:5.1f}" for v in x.median(dim=0).values.tolist())


print("phase medians over images, per layer (us); stamps: 0 loop top, 6 seam-2(l-1) seen, 7 slice done, 2 seam-1 published, "
      "3 seam-1 seen, 4 tab2 ready, 5 seam-2 published")
print("  seam-2 wait (of layer l-1)       ", med(t[:, 1:, 6] - t[:, 1:, 0]))
print("  seam-2 reduce + 32-channel slice ", med(t[:, 1:, 7] - t[:, 1:, 6]))
print("  z epilogue + publish             ", med(t[:, :, 2] - t[:, :, 1]))
print("  seam-1 wait                      ", med(t[:, :, 3] - t[:, :, 2]))
print("  seam-1 reduce + tab2             ", med(t[:, :, 4] - t[:, :, 3]))
print("  bn2 + 3x3 + tail + publish       ", med(t[:, :, 5] - t[:, :, 4]))
print("  next layer's 1x1 (old channels)  ", med(t[:, 1:, 0] - t[:, :-1, 5]))
print("  layer total                      ", med(t[:, 1:, 0] - t[:, :-1, 0]))
print("  chunk loop of the K = 960 layer, workgroup 0, core-clock cycles: prologue (3 chunk loads + first stage + barrier) =", int(cs[1, 3]) - int(cs[0, 3]))
prev = int(cs[1, 3])
for c in range(8):
    a_, b_, c_ = int(cs[c, 0]), int(cs[c, 1]), int(cs[c, 2])
    print(f"    interval {c}: multiply + stage next {a_ - prev:6d} | issue loads {b_ - a_:6d} | barrier {c_ - b_:6d}")
    prev = c_
print("  start skew across images, layer 0 (us): max - min =", float(t[:, 0, 0].max() - t[:, 0, 0].min()))
print("  span (first layer-0 top -> last seam-2 publish):", float(t[:, -1, 5].max() - t[:, 0, 0].min()), "us")
print(json.dumps({"B": B, "block": "denseblock4 forward, 16 layers, 7x7", **out}))
assert not dn.block_persistent_error(x.device)

# ------------------------------------------------------------------------------------------------ backward
dn.USE_BLOCK_PERSISTENT = True
for p_ in blk.parameters():
    p_.grad = torch.zeros_like(p_)
gout = torch.randn(B, 1024, 7, 7, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def fwd_bwd():
    xi = x.clone().requires_grad_(True)
    buf, _ = dn.dense_block(blk, xi, dn._RunningStats(), force_join=True)
    buf.backward(gout.clone())


res = {}
for mode in (True, False, True, False):
    dn.USE_BLOCK_PERSISTENT_BWD = mode
    res.setdefault("persistent_bwd" if mode else "per_layer_bwd", []).append(round(timeit(fwd_bwd, 10), 1))
print(json.dumps({"forward(persistent)+backward eager, us": res}))
st = torch.zeros(B * L * 8 + 64, device=DEV, dtype=torch.int64)
dn.USE_BLOCK_PERSISTENT_BWD = True
fwd_bwd(); torch.cuda.synchronize()
dn.BLOCK_STAMPS = st
# (the forward kernel writes the same buffer first; the backward's stamps overwrite them)
fwd_bwd(); torch.cuda.synchronize()
dn.BLOCK_STAMPS = None
t = st[:B * L * 8].view(B, L, 8).double().cpu() * 0.01
print("BACKWARD phase medians over images, per layer l = 15 .. 0 (us)")
order = list(range(L - 1, -1, -1))
print("  a-c: dy', conv2^T, g2 sums -> seam A published ", med((t[:, :, 1] - t[:, :, 0])[:, order]))
print("  seam A wait                                    ", med((t[:, :, 2] - t[:, :, 1])[:, order]))
print("  seam A merge + dz                              ", med((t[:, :, 3] - t[:, :, 2])[:, order]))
print("  d/e: dz W1, mask, sums, G update -> B1 published", med((t[:, :, 4] - t[:, :, 3])[:, order]))
print("  hop-1 wait                                     ", med((t[:, :, 5] - t[:, :, 4])[:, order]))
print("  hop-1 merge + hop-2 wait                       ", med((t[:, :, 6] - t[:, :, 5])[:, order]))
print("  layer total                                    ", med((t[:, :, 6] - t[:, :, 0])[:, order]))
print("  span:", float(t[:, 0, 6].max() - t[:, L - 1, 0].min()), "us")
assert not dn.block_persistent_error(x.device)
