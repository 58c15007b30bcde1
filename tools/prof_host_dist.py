#!/usr/bin/env python3
"""Host-side (Python) profile of the data-parallel step on a size-1 RCCL group: where the enqueue time goes."""
import cProfile, os, pstats, sys, io
os.environ.update(MCL_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mclstexp_amd import dist as mdist, synth
from mclstexp_amd.engine import TrainStep
from mclstexp_amd.model import mclSTExp_Attention
from mclstexp_amd.optim import FusedAdam
pg, rank, world = mdist.init_from_env()
dev = torch.device("cuda", 0)
m = mclSTExp_Attention("densenet121", 1.0, 1024, 1000, 256, 8, 64, 2, backbone_dtype=torch.bfloat16,
                       embedding_grad="rowsparse", process_group=pg, infonce="fused").to(dev)
m.to(memory_format=torch.channels_last); m.train()
opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(m)
step = TrainStep(m, opt, mdist.GradReducer(pg), graphs=True, warmup=3)
b = {k: v.to(dev) for k, v in synth.make_batch(128, 1000, image_hw=224, seed=0).items()}
b["image"] = b["image"].contiguous(memory_format=torch.channels_last)
for _ in range(8):
    step(b)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    step(b)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000], flush=True)
mdist.shutdown(); os._exit(0)
