"""Probe: how fast is the CPU oracle on this host at a given thread count (bounded)?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
nt = int(sys.argv[1]); torch.set_num_threads(nt)
from mclstexp_amd import synth
from mclstexp_amd.backbones import densenet121_features_module
from oracle import ref_cpu
G = 1000
t0 = time.time()
params = synth.make_params(G, 1024, seed=0)
print(nt, "make_params", round(time.time() - t0, 2), flush=True)
net = densenet121_features_module()
for k, v in net.state_dict().items():
    if v.dtype == torch.float32 and "running_" not in k:
        params["image_encoder.model.0." + k] = v.clone()
for p in params.values():
    p.requires_grad_(True)
b = synth.make_batch(16, G, image_hw=224, seed=0)
for it in range(2):
    t0 = time.time(); feats = ref_cpu.densenet121_features(params, b["image"]); t1 = time.time()
    out = ref_cpu.forward_from_features(params, feats, b["expression"], b["position"], 1.0, 2, 8, 64); t2 = time.time()
    out["loss"].backward(); t3 = time.time()
    state = {}
    with torch.no_grad():
        for n, p in params.items():
            state[n] = (torch.zeros_like(p), torch.zeros_like(p))
            ref_cpu.adam_l2_step(p, p.grad, state[n][0], state[n][1], 1)
    t4 = time.time()
    for p in params.values(): p.grad = None
    print(nt, "iter", it, "densenet fwd %.2f spot fwd %.2f backward %.2f adam %.2f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3), flush=True)
