"""The parameters.yml-driven harness: the reference's config schema drives a real (tiny) training run on the
HIP path and the loss goes down (the reference's own 'can it overfit' smoke idea, src/train.py:59-60)."""
import os

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
    d = torch.load(ck)
    assert set(d) == {"model", "optimizer", "lr_scheduler", "epoch"}
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
