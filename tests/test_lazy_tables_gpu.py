"""Lazy-exact position-table Adam (optim.FusedAdam(lazy_tables=True), csrc/adam.hip: adam_table_lazy_kernel) against the dense
per-step table update (adam_table_kernel): torch.optim.Adam with L2 weight decay (/root/reference/train.py:118-120) moves every
row of the two (65536, G) tables on every step; the lazy form replays a row's missed steps on demand and must be BIT-identical
-- parameters and both moments, touched rows and never-touched rows (row 60000 = the fixtures' UNTOUCHED_ROW)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
UNTOUCHED_ROW = 60000


def _model(G, lazy, lr=1e-4):
    from mclstexp_amd import synth
    from mclstexp_amd.model import mclSTExp_Attention
    from mclstexp_amd.optim import FusedAdam
    torch.manual_seed(0)
    m = mclSTExp_Attention("identity", 1.0, 1024, G, 256, 8, 64, 2, embedding_grad="rowsparse")
    sd = m.state_dict()
    sd.update(synth.make_params(G, 1024, seed=0))
    m.load_state_dict(sd)
    m.to(DEV).train()
    opt = FusedAdam(m.parameters(), lr=lr, weight_decay=1e-3, lazy_tables=lazy).attach_model(m)
    return m, opt


def _batch(B, G, seed):
    from mclstexp_amd import synth
    b = synth.make_batch(B, G, image_dim=1024, seed=seed)
    pos = b["position"]
    pos[:3, 0] = 11.0          # duplicates inside the batch
    pos[-1, 1] = 5.0 + (seed % 3)
    return {k: v.to(DEV) for k, v in b.items()}


def _tables(m, opt):
    out = {}
    for n, emb in (("x", m.x_embed), ("y", m.y_embed)):
        st = opt.state[emb.weight]
        out[n] = (emb.weight.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone())
    return out


def _train(m, opt, steps, G, B=8, lr_at=None, first_seed=0, n_seeds=7):
    from mclstexp_amd import densenet_fused as dn
    losses = []
    for s in range(steps):
        if lr_at and s in lr_at:
            for g in opt.param_groups:
                g["lr"] = lr_at[s]
        batch = _batch(B, G, first_seed + (s % n_seeds))
        loss = m(batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    dn.set_weight_provider(None)
    return losses


@pytest.mark.parametrize("G", [172, 171])          # 16-byte rows (vector kernel) / ragged rows (scalar kernel)
def test_lazy_tables_bit_identical_to_dense_after_60_steps(G):
    steps = 60
    lr_at = {20: 3e-4, 41: 5e-5}                   # an LR schedule: the replay uses the constants each step really used
    md, od = _model(G, lazy=False)
    ld = _train(md, od, steps, G, lr_at=lr_at)
    ml, ol = _model(G, lazy=True)
    init_row = ml.x_embed.weight[UNTOUCHED_ROW].detach().clone()
    ll = _train(ml, ol, steps, G, lr_at=lr_at)
    assert ld == ll                                # the forward saw the same (caught-up) rows on every step
    dense = _tables(md, od)
    stale = _tables(ml, ol)
    # before materialisation a never-touched row has not moved at all, a touched one is current
    assert torch.equal(stale["x"][0][UNTOUCHED_ROW], init_row)
    assert not torch.equal(stale["x"][0][UNTOUCHED_ROW], dense["x"][0][UNTOUCHED_ROW])
    assert torch.equal(stale["x"][0][11], dense["x"][0][11])
    ol.materialize_tables()
    lazy = _tables(ml, ol)
    for n in ("x", "y"):
        for k, what in enumerate(("param", "exp_avg", "exp_avg_sq")):
            assert torch.equal(lazy[n][k][UNTOUCHED_ROW], dense[n][k][UNTOUCHED_ROW]), f"{n}_embed {what}: row {UNTOUCHED_ROW}"
            assert torch.equal(lazy[n][k], dense[n][k]), f"{n}_embed {what}: whole table"
    assert int(ol._row_step(ml.x_embed.weight).min()) == steps
    # and the other parameters walked the same trajectory
    for (na, pa), (nb, pb) in zip(md.named_parameters(), ml.named_parameters()):
        assert torch.equal(pa, pb), na


def test_lazy_tables_history_ring_wraps(monkeypatch):
    """The per-step constants live in a ring; the host materialises before a row could fall out of it."""
    from mclstexp_amd import optim
    monkeypatch.setattr(optim, "HIST_LEN", 16)
    G, steps = 172, 50
    md, od = _model(G, lazy=False)
    _train(md, od, steps, G)
    ml, ol = _model(G, lazy=True)
    _train(ml, ol, steps, G)
    assert ol.materialize_count >= 3               # flushed by the ring bound, not by the test
    ol.materialize_tables()
    dense, lazy = _tables(md, od), _tables(ml, ol)
    for n in ("x", "y"):
        for k in range(3):
            assert torch.equal(lazy[n][k], dense[n][k]), (n, k)


def test_lazy_tables_inside_step_graph_and_state_dict():
    """engine.TrainStep replays the catch-up and the row update from ONE captured graph; model.state_dict() materialises."""
    from mclstexp_amd import densenet_fused as dn
    from mclstexp_amd.engine import TrainStep
    G, steps = 172, 24

    def run(lazy, graphs):
        m, opt = _model(G, lazy)
        tr = TrainStep(m, opt, None, graphs=graphs, warmup=2)
        losses = [tr(_batch(8, G, s % 5)).item() for s in range(steps)]
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}      # the hook materialises the tables
        mom = {n: (opt.state[e.weight]["exp_avg"].clone(), opt.state[e.weight]["exp_avg_sq"].clone())
               for n, e in (("x", m.x_embed), ("y", m.y_embed))}
        dn.set_weight_provider(None)
        return losses, sd, mom, tr

    l0, sd0, mom0, _ = run(False, False)
    l1, sd1, mom1, tr = run(True, True)
    assert tr.ga is not None
    assert l0 == l1
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k
    for n in mom0:
        assert torch.equal(mom0[n][0], mom1[n][0]) and torch.equal(mom0[n][1], mom1[n][1]), n


def test_lazy_tables_eval_module_call_sees_current_rows():
    """evel_her2st.py:48-69 calls model.x_embed(idx) directly: the module call materialises first."""
    G = 172
    md, od = _model(G, lazy=False)
    _train(md, od, 5, G)
    ml, ol = _model(G, lazy=True)
    _train(ml, ol, 5, G)
    idx = torch.tensor([UNTOUCHED_ROW, 11, 3], device=DEV)
    with torch.no_grad():
        assert torch.equal(md.eval().x_embed(idx), ml.eval().x_embed(idx))
        assert torch.equal(md.y_embed(idx), ml.y_embed(idx))


def test_step_without_table_gradient_moves_no_row():
    """torch.optim.Adam skips a parameter whose .grad is None: a step in which the spot branch did not run leaves both tables
    exactly where they were -- in the lazy form too (no replayed weight-decay step for it)."""
    G = 172
    outs = []
    for lazy in (False, True):
        m, opt = _model(G, lazy)
        _train(m, opt, 3, G)
        # a step that uses the image projection only
        x = torch.randn(8, 1024, generator=torch.Generator().manual_seed(5)).to(DEV)
        opt.zero_grad()
        m.image_projection(x).square().mean().backward()
        opt.step()
        _train(m, opt, 2, G, first_seed=3)
        opt.materialize_tables()
        outs.append(_tables(m, opt))
    for n in ("x", "y"):
        for k in range(3):
            assert torch.equal(outs[0][n][k], outs[1][n][k]), (n, k)


@pytest.mark.parametrize("graphs", [False, True])
def test_checkpoint_resume_equals_uninterrupted_dense_run(graphs, tmp_path):
    """train N steps, save model + optimizer through torch.save, load both into FRESH objects, train M more
    -- bit-identical to an uninterrupted run of the dense per-step table update: every parameter, both tables, all moments.
    (torch's Optimizer.load_state_dict casts state tensors of float parameters to float: the int32 row stamps must not be in
    the state; the flat-managed parameters' moments must be in the checkpoint.)"""
    from mclstexp_amd import densenet_fused as dn
    from mclstexp_amd.engine import TrainStep
    G, N, M = 172, 9, 11

    def steps(m, opt, first, count, tr=None):
        out = []
        for s in range(first, first + count):
            batch = _batch(8, G, s % 7)
            if tr is not None:
                out.append(tr(batch).item())
            else:
                loss = m(batch)
                opt.zero_grad()
                loss.backward()
                opt.step()
                out.append(loss.item())
        return out

    md, od = _model(G, lazy=False)
    ref = steps(md, od, 0, N + M)
    dn.set_weight_provider(None)

    m1, o1 = _model(G, lazy=True)
    tr1 = TrainStep(m1, o1, None, graphs=graphs, warmup=2) if graphs else None
    got = steps(m1, o1, 0, N, tr1)
    path = str(tmp_path / "ckpt.pt")
    torch.save({"model": m1.state_dict(), "opt": o1.state_dict()}, path)
    dn.set_weight_provider(None)
    del m1, o1, tr1

    ck = torch.load(path)
    for v in ck["opt"]["state"].values():
        assert set(v) == {"exp_avg", "exp_avg_sq"}, "only Adam's moments belong in the checkpoint"
    assert len(ck["opt"]["state"]) == len(list(md.parameters())), "every parameter's moments are in the checkpoint"
    m2, o2 = _model(G, lazy=True)
    m2.load_state_dict(ck["model"])
    o2.load_state_dict(ck["opt"])
    tr2 = TrainStep(m2, o2, None, graphs=graphs, warmup=2) if graphs else None
    got += steps(m2, o2, N, M, tr2)
    assert got == ref
    o2.materialize_tables()
    dense, lazy = _tables(md, od), _tables(m2, o2)
    for n in ("x", "y"):
        for k, what in enumerate(("param", "exp_avg", "exp_avg_sq")):
            assert torch.equal(lazy[n][k], dense[n][k]), f"{n}_embed {what}"
    for (na, pa), (nb, pb) in zip(md.named_parameters(), m2.named_parameters()):
        assert torch.equal(pa, pb), na
    sd_d, sd_l = od.state_dict(), o2.state_dict()
    for k in sd_d["state"]:
        for w in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(sd_d["state"][k][w], sd_l["state"][k][w]), (k, w)
    dn.set_weight_provider(None)


def test_load_state_dict_into_a_live_captured_step():
    """Loading a checkpoint into the SAME objects after the step graph was captured: the replayed kernels read the moments and
    row stamps through the pointers they were captured with, so the load must overwrite those tensors in place."""
    from mclstexp_amd import densenet_fused as dn
    from mclstexp_amd.engine import TrainStep
    G = 172
    m, opt = _model(G, lazy=True)
    tr = TrainStep(m, opt, None, graphs=True, warmup=2)
    for s in range(6):
        tr(_batch(8, G, s % 7))
    sd_m = {k: v.detach().clone() for k, v in m.state_dict().items()}
    import copy
    sd_o = copy.deepcopy(opt.state_dict())          # (a snapshot: state_dict() holds references to the live state, as torch's)
    a = [tr(_batch(8, G, (6 + s) % 7)).item() for s in range(5)]
    assert tr.ga is not None
    m.load_state_dict(sd_m)
    opt.load_state_dict(sd_o)
    b = [tr(_batch(8, G, (6 + s) % 7)).item() for s in range(5)]
    assert a == b
    dn.set_weight_provider(None)


def test_second_optimizer_takes_over_cleanly():
    """a second FusedAdam for the same model materialises the first one's pending replays and removes its
    hooks (they do not pile up; the model stays picklable)."""
    import pickle
    from mclstexp_amd.optim import FusedAdam
    G = 172
    md, od = _model(G, lazy=False)
    _train(md, od, 4, G)
    ml, ol = _model(G, lazy=True)
    _train(ml, ol, 4, G)
    n_hooks = len(ml._state_dict_pre_hooks)
    o2 = FusedAdam(ml.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(ml)
    assert len(ml._state_dict_pre_hooks) == n_hooks and not ol._hook_handles
    # the first optimizer's deferred weight-decay steps landed in the tables before the hand-over
    assert torch.equal(ml.x_embed.weight[UNTOUCHED_ROW], md.x_embed.weight[UNTOUCHED_ROW])
    o2.detach_model()
    pickle.dumps(ml.x_embed)
    from mclstexp_amd import densenet_fused as dn
    dn.set_weight_provider(None)
