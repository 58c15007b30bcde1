"""Diagnostic: eager vs graph TrainStep with lazy / dense tables, materialise between calls; reports the first divergence."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mclstexp_amd import densenet_fused as dn, synth
from mclstexp_amd.engine import TrainStep
from mclstexp_amd.model import mclSTExp_Attention
from mclstexp_amd.optim import FusedAdam
DEV = "cuda"
G = 171
sizes = [8, 8, 8, 8, 8]


def run(graphs, lazy, mat):
    torch.manual_seed(0)
    m = mclSTExp_Attention("identity", 1.0, 1024, G, 256, 8, 64, 2, embedding_grad="rowsparse")
    sd = m.state_dict()
    sd.update(synth.make_params(G, 1024, seed=0))
    m.load_state_dict(sd)
    m.to(DEV).train()
    opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3, lazy_tables=lazy).attach_model(m)
    tr = TrainStep(m, opt, None, graphs=graphs, warmup=2)
    out = []
    for s, b in enumerate(sizes):
        batch = {k: v.to(DEV) for k, v in synth.make_batch(b, G, image_dim=1024, seed=s).items()}
        loss = tr(batch).item()
        pre = None
        if lazy:
            pre = (m.x_embed.weight.detach().clone(), opt.state[m.x_embed.weight]["row_step"].clone())
        if mat:
            opt.materialize_tables()
        snap = {n: p.detach().clone() for n, p in m.named_parameters()}
        st = opt.state[m.x_embed.weight]
        out.append((loss, snap, st["exp_avg"].clone(), st["exp_avg_sq"].clone(), batch["position"].long().cpu(), pre,
                    opt._dev_state(0, torch.device(DEV))["hist"][:64].clone().cpu()))
    dn.set_weight_provider(None)
    return out


ref = run(False, False, False)
got = run(True, True, True)
for i in range(len(sizes)):
    x0, x1 = ref[i][1]["x_embed.weight"], got[i][1]["x_embed.weight"]
    rows = (x0 != x1).any(dim=1).nonzero().flatten().cpu().tolist()
    mrows = (ref[i][2] != got[i][2]).any(dim=1).nonzero().flatten().cpu().tolist()
    vrows = (ref[i][3] != got[i][3]).any(dim=1).nonzero().flatten().cpu().tolist()
    print("call", i + 1, "x rows differing", len(rows), rows[:12], "m", len(mrows), mrows[:8], "v", len(vrows), vrows[:8])
    print("   positions x", got[i][4][:, 0].tolist())
    pre_x, pre_rs = got[i][5]
    print("   row_step before materialise at batch rows:", pre_rs[got[i][4][:, 0].to(DEV)].tolist(), "min/max", int(pre_rs.min()), int(pre_rs.max()))
    if rows:
        r = rows[0]
        print("   row", r, "ref", x0[r, :4].tolist(), "got", x1[r, :4].tolist(), "pre-mat", pre_x[r, :4].tolist())
    print("   hist[1..5] lr_over_bc1:", got[i][6].view(-1, 8)[1:6, 0].tolist())
