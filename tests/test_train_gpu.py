"""The parameters.yml-driven harness: the reference's config schema drives a real (tiny) training run on the
HIP path and the loss goes down (the reference's own 'can it overfit' smoke idea, src/train.py:59-60)."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

# the reference's parameters.yml keys that are in scope (values: reference defaults, shrunk model)
PARAMS = {
    "training": {"optimizer": {"type": "adam", "start_lr": 0.001, "scheduler": False, "end_lr": 0.00001, "weight_decay": 0.0},
                 "checkpoints_path": "./checkpoints", "checkpoints_frequency": 25, "batch_size": 16, "epochs": 250, "loss": "arc"},
    "loss": {"sphere": {"margin": 4}, "cos": {"margin": 0.2, "scale": 64}, "arc": {"margin": 0.2, "scale": 30}},
    "titanet": {"enabled": True, "model_size": "s", "n_mega_blocks": 2, "attention_hidden_size": 128, "simple_pool": False,
                "dropout": 0.1},
    "generic": {"seed": 42, "workers": 2, "log_console": False, "chart_dependencies": False, "embedding_size": 192},
    "audio": {"sample_rate": 16000, "spectrogram": {"n_fft": 512, "win_length": 25, "hop_length": 10, "n_mels": 80}},
}


def test_yaml_driven_training_reduces_loss(tmp_path):
    from titanet_amd import train
    p = tmp_path / "parameters.yml"
    p.write_text(yaml.safe_dump(PARAMS))
    params = train.Struct(**yaml.load(open(p), Loader=yaml.SafeLoader))
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(16, 80, 151, generator=g) * 0.11 - 0.1).cuda()
    y = torch.randint(0, 8, (16,), generator=g).cuda()

    def fixed():
        while True:
            yield x, None, y
    model, trainer, hist = train.run(params, steps=60, n_classes=8, data=fixed(), log_every=20)
    assert hist[-1][1] < hist[0][1] * 0.7, hist            # overfits a fixed batch
    ck = tmp_path / "ck" / "epoch_1.pth"
    train.save_checkpoint(model, trainer, 1, str(ck))
    d = torch.load(ck, weights_only=True)            # tensors, numbers and plain containers only
    assert set(d) == {"model", "optimizer", "lr_scheduler", "epoch", "dropout_stream"}      # the reference's four keys + the dropout stream position
    assert "encoder.prolog.conv_block.0.weight" in d["model"] and "loss_function.fc.weight" in d["model"]


def test_flat_all_reducer_on_rccl_world1():
    """The RCCL leg of the data-parallel step (side stream, event ordering, bucketed all-reduce) runs on a
    one-rank "nccl" group: sums are the identity, so the buffer must come back unchanged."""
    import torch.distributed as dist
    from titanet_amd.trainer import FlatAllReducer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        red = FlatAllReducer(n_buckets=4)
        red.world = 2                      # force the collective path on the 1-rank group
        flat = torch.arange(5000, device="cuda", dtype=torch.float32)
        want = flat.clone()
        red.all_reduce_(flat)
        torch.cuda.synchronize()
        assert torch.equal(flat, want)
    finally:
        dist.destroy_process_group()


def test_graph_mode_matches_eager_and_varies_dropout():
    """Trainer(use_graph=True): the whole step replays as one hipGraph.  With dropout 0 the loss trajectory equals the
    eager trainer's; with dropout > 0 two replays on the same batch draw different masks (the device-resident step word)."""
    from titanet_amd import LOSSES, TitaNet
    from titanet_amd.trainer import Trainer

    def make(p, seed=3):
        torch.manual_seed(seed)
        lf = LOSSES["ce"](192, 16, device="cuda")
        return TitaNet.get_titanet(n_mega_blocks=2, model_size="s", loss_function=lf, dropout=p, device="cuda",
                                   precision="fp32").train()
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(8, 80, 120, generator=g) * 0.11 - 0.1).cuda()
    y = torch.randint(0, 16, (8,), generator=g).cuda()
    eager, graph = Trainer(make(0.0)), Trainer(make(0.0), use_graph=True, graph_warmup=2)
    le, lg = [], []
    for _ in range(8):
        le.append(float(eager.step(x, y)[2]))
        lg.append(float(graph.step(x, y)[2]))
    assert any(v["graph"] is not None for v in graph._graphs.values())      # steps 3.. were replays
    assert max(abs(a - b) for a, b in zip(le, lg)) < 1e-4 * max(1.0, abs(le[0])), (le, lg)
    assert lg[-1] < lg[0]
    # dropout: consecutive replays on the same batch must not reuse the masks
    tr = Trainer(make(0.3), lr=0.0, use_graph=True, graph_warmup=1)
    embs = [tr.step(x, y)[0].clone() for _ in range(5)]
    assert all(torch.isfinite(e).all() for e in embs)
    d = [float((embs[i] - embs[i + 1]).abs().max()) for i in range(1, 4)]       # replays (lr = 0: weights frozen)
    assert min(d) > 1e-4, d


def _small_model(p=0.0, seed=3, loss="ce", precision="fp32", n_classes=16):
    from titanet_amd import LOSSES, TitaNet
    torch.manual_seed(seed)
    kw = {} if loss == "ce" else {"scale": 30, "margin": 0.2}
    lf = LOSSES[loss](192, n_classes, device="cuda", **kw)
    return TitaNet.get_titanet(n_mega_blocks=2, model_size="s", loss_function=lf, dropout=p, device="cuda",
                               precision=precision).train()


def test_fused_adam_matches_torch_optim_adam():
    """tn_adam_step (inside the timed step of bench.py) against torch.optim.Adam (reference src/train.py:131-135) on the
    same gradients: 5 steps, weight decay on and off, parameters within 1e-6."""
    from titanet_amd.trainer import Trainer
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(8, 80, 100, generator=g) * 0.11 - 0.1).cuda()
    y = torch.randint(0, 16, (8,), generator=g).cuda()
    for wd in (0.0, 0.01):
        m = _small_model()
        tr = Trainer(m, lr=1e-3, weight_decay=wd)
        ref_p = m.flat_parameters().clone().requires_grad_(True)
        opt = torch.optim.Adam([ref_p], lr=1e-3, weight_decay=wd)
        for _ in range(5):
            tr.forward_backward(x, y)
            ref_p.grad = m.flat_gradients().clone()
            # keep the reference copy on the SAME trajectory: it sees the gradients of the fused path's weights
            opt.step()
            tr.optimizer_step()
            torch.cuda.synchronize()
            err = float((m.flat_parameters() - ref_p.detach()).abs().max())
            assert err < 1e-6, (wd, err)


def test_graph_mode_two_shapes_matches_eager_and_follows_lr():
    """Alternating input shapes (three plans under RandomChunk) share ONE step counter, and a learning-rate schedule that
    assigns trainer.lr is honoured by replayed graphs (per-plan counters drifted, lr was frozen at capture)."""
    from titanet_amd.trainer import Trainer
    g = torch.Generator().manual_seed(0)
    xs = [(torch.randn(8, 80, T, generator=g) * 0.11 - 0.1).cuda() for T in (100, 140)]
    y = torch.randint(0, 16, (8,), generator=g).cuda()
    eager, graph = Trainer(_small_model()), Trainer(_small_model(), use_graph=True, graph_warmup=1)
    for i in range(12):
        lr = 1e-3 * (0.8 ** i)
        eager.lr = graph.lr = lr
        le = float(eager.step(xs[i % 2], y)[2])
        lg = float(graph.step(xs[i % 2], y)[2])
        assert abs(le - lg) < 2e-4 * max(1.0, abs(le)), (i, le, lg)
    assert sum(v["graph"] is not None for v in graph._graphs.values()) == 2
    torch.cuda.synchronize()
    # same trajectory: compare where the gradient carries signal (Adam turns the +-1e-7 rounding noise of a mathematically
    # zero gradient, e.g. a conv bias in front of BatchNorm, into +-lr steps: those entries differ run to run by design)
    pe, pg = eager.model.flat_parameters(), graph.model.flat_parameters()
    frac = float(((pe - pg).abs() > 2e-4).float().mean())
    assert frac < 0.02, frac
    # the learning rate is read per replay: with lr = 0 a replayed step must not move the weights at all
    graph.lr = 0.0
    before = graph.model.flat_parameters().clone()
    for i in range(2):
        graph.step(xs[i % 2], y)
    torch.cuda.synchronize()
    assert torch.equal(before, graph.model.flat_parameters())
    graph.lr = 1e-3
    graph.step(xs[0], y)
    torch.cuda.synchronize()
    assert not torch.equal(before, graph.model.flat_parameters())


def test_checkpoint_optimizer_state_is_adam_layout_and_resumes(tmp_path):
    """reference src/learn.py:180-201: the checkpoint's "optimizer" loads into torch.optim.Adam(model.parameters()), and a
    run resumed from the checkpoint continues exactly like the uninterrupted one."""
    from titanet_amd import train
    from titanet_amd.trainer import Trainer
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(8, 80, 100, generator=g) * 0.11 - 0.1).cuda()
    y = torch.randint(0, 16, (8,), generator=g).cuda()
    a = Trainer(_small_model())
    for _ in range(3):
        a.step(x, y)
    ck = str(tmp_path / "epoch_3.pth")
    train.save_checkpoint(a.model, a, 3, ck)
    d = torch.load(ck, weights_only=False)
    assert d["lr_scheduler"] == {} and d["epoch"] == 3
    opt = torch.optim.Adam(a.model.parameters(), lr=1e-3)
    opt.load_state_dict(d["optimizer"])                       # torch's own loader accepts the layout
    st = opt.state_dict()["state"]
    assert len(st) == len(list(a.model.parameters())) and float(st[0]["step"]) == 3.0
    b = Trainer(_small_model(seed=99))                          # different init: everything must come from the checkpoint
    epoch, _ = train.load_checkpoint(b.model, b, ck)
    assert epoch == 3 and b.step_count == 3
    b.model._step = a.model._step                               # same dropout stream position (dropout is 0 here anyway)
    for _ in range(2):
        la, lb = float(a.step(x, y)[2]), float(b.step(x, y)[2])
        assert abs(la - lb) < 1e-5 * max(1.0, abs(la)), (la, lb)


@pytest.mark.parametrize("loss", ["ce", "arc"])
def test_standalone_loss_head_matches_oracle(loss):
    """CELoss / ArcFaceLoss called as ``loss(inputs, targets)`` (reference src/losses.py:32-44, :77-132) outside
    TitaNet.forward: values and gradients against the float64 oracle restatement."""
    from oracle import titanet_oracle as O
    from titanet_amd import LOSSES
    torch.manual_seed(5)
    B, E, NC = 32, 192, 50
    kw = {} if loss == "ce" else {"scale": 30, "margin": 0.2}
    head = LOSSES[loss](E, NC, device="cuda", **kw)
    x = torch.randn(B, E, device="cuda", requires_grad=True)
    y = torch.randint(0, NC, (B,), device="cuda")
    w0 = head.fc.weight.detach().clone()
    norm, preds, lv = head(x, y)
    (lv * 2.0).backward()
    torch.cuda.synchronize()
    xo = x.detach().cpu().double().requires_grad_(True)
    wo = w0.cpu().double().requires_grad_(True)
    sd = {"loss_function.fc.weight": wo}
    if loss == "ce":
        bo = head.fc.bias.detach().cpu().double().requires_grad_(True)
        sd["loss_function.fc.bias"] = bo
        o_norm, o_preds, o_loss, _ = O.ce_loss(xo, y.cpu(), sd)
    else:
        o_norm, o_preds, o_loss, _, w_after = O.angular_margin_loss(xo, y.cpu(), sd, **O.margin_kwargs("arc", scale=30, margin=0.2))
        assert float((head.fc.weight.detach().cpu().double() - w_after.detach()).abs().max()) < 1e-6     # in-place row normalisation
    (o_loss * 2.0).backward()
    assert abs(float(lv) - float(o_loss)) < 1e-4 * max(1.0, abs(float(o_loss)))
    assert torch.equal(preds.cpu(), o_preds)
    assert float((norm.detach().cpu().double() - o_norm.detach()).abs().max()) < 1e-5
    ex = float((x.grad.cpu().double() - xo.grad).norm() / xo.grad.norm())
    ew = float((head.fc.weight.grad.cpu().double() - wo.grad).norm() / wo.grad.norm())
    assert ex < 1e-4 and ew < 1e-4, (ex, ew)
    if loss == "ce":
        eb = float((head.fc.bias.grad.cpu().double() - bo.grad).norm() / bo.grad.norm())
        assert eb < 1e-4, eb


def test_out_of_range_target_poisons_the_loss():
    """F.cross_entropy raises on a bad label; the fused head cannot raise from a kernel: the loss becomes NaN (so the
    trainers' finite-loss guard, reference src/learn.py:110-112, stops the run) and no access leaves the logits row."""
    import math
    m = _small_model()
    x = (torch.randn(4, 80, 64) * 0.1).cuda()
    y = torch.tensor([0, 3, 16, 2], device="cuda")          # 16 == n_classes
    _, _, lv = m(x, speakers=y)
    assert math.isnan(float(lv))


@pytest.mark.parametrize("name", ["arc", "cos"])
def test_margin_head_at_the_cos_clamp_vs_reference_golden(name):
    """SURVEY.md 8c "a row with cos -> +-1 clamp": forward values against the real reference's golden.  Gradients: where the
    reference is finite they match it; where the reference yields NaN (it differentiates arccos at +-1 for columns it
    never uses: 0 * inf, src/losses.py:100) the HIP head returns the finite analytic gradient of the same loss."""
    import numpy as np
    from oracle import titanet_oracle as O
    from tests.golden.cases import head_clamp_inputs
    from tests.util import load_golden, rel_err
    from titanet_amd import LOSSES
    g = load_golden("head_clamp")
    x, w, y = head_clamp_inputs()
    head = LOSSES[name](16, 9, device="cuda", scale=30 if name == "arc" else 64, margin=0.2)
    with torch.no_grad():
        head.fc.weight.copy_(torch.from_numpy(w))
    xi = torch.from_numpy(x).cuda().requires_grad_(True)
    norm, preds, lv = head(xi, torch.from_numpy(y).cuda())
    lv.backward()
    torch.cuda.synchronize()
    assert abs(float(lv) - float(g[name + ".loss"])) < 1e-4 * abs(float(g[name + ".loss"]))
    assert np.array_equal(preds.cpu().numpy(), g[name + ".preds"])
    assert rel_err(norm.detach().cpu().numpy(), g[name + ".normalized"]) < 1e-6
    assert rel_err(head.fc.weight.detach().cpu().numpy(), g[name + ".weight_after"]) < 1e-6
    gx, gw = xi.grad.cpu().numpy(), head.fc.weight.grad.cpu().numpy()
    assert np.isfinite(gx).all() and np.isfinite(gw).all()
    for got, want in ((gx, g[name + ".grad.inputs"]), (gw, g[name + ".grad.weight"])):
        ok = np.isfinite(want)
        assert ok.any() and rel_err(got[ok], want[ok]) < 1e-4
    # the rows the reference poisons: analytic gradient with arccos taken at the target column only
    kw = O.margin_kwargs(name, scale=30 if name == "arc" else 64, margin=0.2)
    xo = torch.from_numpy(x).double().requires_grad_(True)
    wn = torch.from_numpy(g[name + ".weight_after"]).double().requires_grad_(True)
    xn = xo / xo.norm(dim=1, keepdim=True)
    cos = (xn @ wn.t()).clamp(-1, 1)
    yt = torch.from_numpy(y)[:, None]
    num = kw["scale"] * (torch.cos(kw["m1"] * torch.arccos(cos.gather(1, yt)[:, 0]) + kw["m2"]) - kw["m3"])
    expo = torch.exp(kw["scale"] * cos)
    den = torch.exp(num) + expo.sum(1) - expo.gather(1, yt)[:, 0]
    (-(num - torch.log(den + kw["eps"])).mean()).backward()
    assert rel_err(gx, xo.grad.numpy()) < 1e-4 and rel_err(gw, wn.grad.numpy()) < 1e-4


def test_resume_from_a_checkpoint_written_by_the_reference():
    """tests/golden/ref_checkpoint_tiny.pth was written by the REFERENCE's model + torch.optim.Adam in the reference's
    layout (src/learn.py:188-195) after two of its own train steps; ref_checkpoint_next.npz holds its parameters after the
    third step on the same deterministic batch (tests/golden/make_reference_checkpoint.py).  Loading it here (weights_only)
    and running ONE fused step must land on the reference's third-step parameters: key-for-key state_dict compatibility,
    Adam moments / step count taken over, same arithmetic."""
    import os
    from tests.golden.cases import CASES
    from tests.test_forward_gpu import build
    from tests.util import case_inputs
    from titanet_amd import train
    from titanet_amd.trainer import Trainer
    gold = os.path.join(os.path.dirname(__file__), "golden")
    case = CASES["tiny_k3"]
    m = build(case, "ce").train()
    tr = Trainer(m, lr=123.0)                  # overwritten by the checkpoint's param group
    epoch, sched = train.load_checkpoint(m, tr, os.path.join(gold, "ref_checkpoint_tiny.pth"))
    assert epoch == 2 and sched == {} and tr.step_count == 2 and abs(tr.lr - 1e-3) < 1e-12
    ck = torch.load(os.path.join(gold, "ref_checkpoint_tiny.pth"), weights_only=True)
    sd = m.state_dict()
    assert list(sd.keys()) == list(ck["model"].keys())
    for k, v in ck["model"].items():
        assert torch.equal(sd[k].cpu(), v), k
    x, y = case_inputs(case, torch.float32)
    _, _, loss = tr.step(x.cuda(), y.cuda())
    torch.cuda.synchronize()
    nxt = np.load(os.path.join(gold, "ref_checkpoint_next.npz"))
    assert abs(float(loss) - float(nxt["loss"])) < 2e-4 * abs(float(nxt["loss"])) + 1e-5
    sd = m.state_dict()
    # a bias in front of a BatchNorm has a TRUE gradient of zero: both implementations hand Adam rounding noise there, and
    # Adam turns noise into steps of +-lr whatever its size — those tensors may differ by up to 2 lr; everything else
    # must agree to float32 rounding amplified by 1/sqrt(v)
    zero_grad = (".conv_block.0.bias", ".conv.0.bias", ".conv.1.bias", "skip_connection.0.bias", "decoder.linear.0.bias",
                 "decoder.pool.0.out_linear.bias")      # (the last: a per-channel constant cancels in the softmax over time)
    worst, worst_k = 0.0, None
    for k in sd:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(nxt[k]) == 3, k
            continue
        a, b = sd[k].float().cpu().numpy(), nxt[k]
        if k.endswith(zero_grad):
            assert float(np.abs(a - b).max()) <= 2.1e-3, k
            continue
        e = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-6))
        if e > worst:
            worst, worst_k = e, k
    print("worst relative parameter / buffer difference after the resumed step", worst, worst_k)
    assert worst < 2e-3, (worst, worst_k)


def test_variable_length_training_fast_kernels_track_the_generic_path():
    """30 optimizer steps on a learnable synthetic task with zero-padded batches + lengths: the masked fast kernels and the
    generic masked templates drive the loss down the same way (tools/train_compare_masked.py is the long version)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("tcm", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "train_compare_masked.py"))
    tcm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tcm)
    fast = tcm.run("s", 3, False, 30)
    gen = tcm.run("s", 3, True, 30)
    assert fast[2] and gen[2]
    assert fast[0][-1] < 0.2 * fast[0][0] and gen[0][-1] < 0.2 * gen[0][0], (fast[0][::5], gen[0][::5])
    # same data, same dropout stream, bf16 noise only: the curves stay together
    for k in (5, 10, 20, 29):
        assert abs(fast[0][k] - gen[0][k]) < 0.15 * max(gen[0][k], 0.05), (k, fast[0][k], gen[0][k])
