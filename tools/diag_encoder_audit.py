"""Lists the foreign (non-library) kernels of one encoder forward (+ backward): python tools/diag_encoder_audit.py NAME DIM DTYPE MODE"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mclstexp_amd import kernel_audit
from mclstexp_amd.model import mclSTExp_Attention
name, dim, dtype, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
torch.manual_seed(0)
bb = torch.bfloat16 if dtype == "bf16" else None
hw = 224 if name.startswith("vit") else 64
m = mclSTExp_Attention(name, 1.0, dim, 171, 256, 8, 64, 2, backbone_dtype=bb).to("cuda")
m.to(memory_format=torch.channels_last)
x = torch.rand(4, 3, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
seed = torch.randn(4, dim, device="cuda")
if mode == "train":
    m.train()
    run = lambda: m.encode_image(x).backward(seed)
else:
    m.eval()
    def run():
        with torch.no_grad():
            m.encode_image(x)
run()
ks = kernel_audit.step_kernels(run)
for k in kernel_audit.foreign(ks):
    print(ks[k], k[:260])
