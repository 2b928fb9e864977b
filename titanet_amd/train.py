#!/usr/bin/env python3
"""Training harness consuming the reference's ``parameters.yml`` schema.

    python -m titanet_amd.train -p parameters.yml [--steps N]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m titanet_amd.train -p parameters.yml

Data parallel (BASELINE configs[2]): under ``torchrun`` (WORLD_SIZE > 1) one process per GPU, RCCL
(``torch.distributed`` backend "nccl"), ``training.batch_size`` is the GLOBAL batch and rank r trains on its contiguous
shard (synthetic data: generator seed ``generic.seed + rank``), gradients all-reduced overlapped with backward
(:class:`titanet_amd.trainer.Trainer`), checkpoints and logs from rank 0 only.

Reproduces the step protocol of the reference (``train.train``, src/train.py:11-183; ``learn.train_one_epoch``
/ ``training_loop`` / ``save_checkpoint``, src/learn.py:64-310) for the part that is in scope: it reads
``training.{batch_size,epochs,loss,optimizer.*,checkpoints_*}``, ``loss.*``, ``titanet.*``,
``generic.{seed,embedding_size}``, ``audio.spectrogram.*`` from the same YAML, builds the loss through
``LOSSES[name](embedding_size, n_classes, **loss.<name>)`` and the model through ``TitaNet.get_titanet``,
always uses Adam (the reference's ``== "sgd"`` test can never be true, src/train.py:130), an optional
per-epoch cosine LR (src/learn.py:257-258), and writes checkpoints as
``{"model", "optimizer", "lr_scheduler", "epoch"}`` (src/learn.py:188-195).  Datasets / W&B / plots are out of
scope: batches are synthetic ``[B, n_mels, T]`` tensors with the collate_fn layout
(src/datasets.py:48-73) unless a data iterator is passed to :func:`run`.
"""
import argparse
import math
import os

import torch
import torch.distributed as dist
import yaml

from . import LOSSES, TitaNet
from .trainer import Trainer


class Struct:
    """reference src/utils.py:31-63: nested dict -> attribute access, keeps ``.entries``."""

    def __init__(self, **entries):
        self.entries = entries
        for k, v in entries.items():
            setattr(self, k, Struct(**v) if isinstance(v, dict) else v)


def synthetic_batches(batch_size, n_mels, n_classes, device, frames=(151, 201, 301), seed=42):
    """collate_fn layout: spectrograms [B, n_mels, T] fp32 (T from the RandomChunk lengths 1.5/2/3 s),
    lengths, speaker ids (reference src/datasets.py:48-73, src/transforms.py:206-233)."""
    g = torch.Generator().manual_seed(seed)
    while True:
        T = frames[int(torch.randint(0, len(frames), (1,), generator=g))]
        x = torch.randn(batch_size, n_mels, T, generator=g) * 0.11 - 0.10
        y = torch.randint(0, n_classes, (batch_size,), generator=g)
        yield x.to(device), torch.full((batch_size,), T), y.to(device)


def init_distributed(backend="nccl", same_device=False):
    """(rank, world, device) from the torchrun environment; initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if same_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
    return rank, world, device


def run(params, steps=None, n_classes=251, data=None, device="cuda", precision="fp32", log_every=10, resume=None,
        use_graph=False, rank=0, world=1, grad_groups=2):
    torch.manual_seed(params.generic.seed)            # same initial weights on every rank (rank 0's are broadcast anyway)
    loss_name = params.training.loss
    loss_kw = dict(getattr(params.loss, loss_name).entries) if hasattr(params.loss, loss_name) else {}
    loss_function = LOSSES[loss_name](params.generic.embedding_size, n_classes, device=device, **loss_kw)
    model = TitaNet.get_titanet(embedding_size=params.generic.embedding_size, n_mels=params.audio.spectrogram.n_mels,
                                n_mega_blocks=params.titanet.n_mega_blocks, model_size=params.titanet.model_size,
                                attention_hidden_size=params.titanet.attention_hidden_size, simple_pool=params.titanet.simple_pool,
                                loss_function=loss_function, dropout=params.titanet.dropout, device=device, precision=precision)
    model.train()
    opt = params.training.optimizer
    trainer = Trainer(model, lr=opt.start_lr, weight_decay=opt.weight_decay, use_graph=use_graph, n_buckets=grad_groups)
    first = 1
    if resume:
        first = load_checkpoint(model, trainer, resume)[0] + 1
    if params.training.batch_size % world:
        raise ValueError(f"training.batch_size {params.training.batch_size} is the global batch: not divisible by {world} ranks")
    if data is None:
        # rank r's shard of the global batch: its own generator stream (seed + rank)
        data = synthetic_batches(params.training.batch_size // world, params.audio.spectrogram.n_mels, n_classes, device,
                                 seed=params.generic.seed + rank)
        for _ in range(first - 1):          # resumed runs continue the data stream where the checkpointed run stopped
            next(data)
    epochs = params.training.epochs
    total = steps if steps is not None else epochs
    history = []
    for step in range(first, total + 1):
        if opt.scheduler:
            trainer.lr = cosine_lr(opt, step - 1, epochs)
        spectrograms, _, speakers = next(data)
        emb, preds, loss = trainer.step(spectrograms, speakers)
        if step % log_every == 0 or step == total:
            lv = float(loss.item())
            if not math.isfinite(lv):                                   # src/learn.py:110-112
                raise SystemExit(f"Loss is {lv}, stopping training")
            acc = float((preds == speakers).float().mean().item())
            history.append((step, lv, acc))
            if rank == 0:
                print(f"step {step:5d}  loss {lv:.4f}  acc {acc:.3f}" + (f"  (rank 0 of {world})" if world > 1 else ""), flush=True)
    return model, trainer, history


def cosine_lr(opt, epoch, epochs):
    """CosineAnnealingLR(T_max=epochs, eta_min=end_lr) stepped once per epoch (reference src/train.py:137-144, src/learn.py:257-258)"""
    return opt.end_lr + 0.5 * (opt.start_lr - opt.end_lr) * (1 + math.cos(math.pi * min(epoch, epochs) / epochs))


def save_checkpoint(model, trainer, epoch, path, scheduler=None):
    """reference src/learn.py:180-201: ``{"model", "optimizer", "lr_scheduler", "epoch"}``.  ``"optimizer"`` is in
    ``torch.optim.Adam.state_dict()`` layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), ``"lr_scheduler"`` a
    ``CosineAnnealingLR``-shaped dict when a schedule is on, else ``dict()`` as the reference writes."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    # (the dropout stream position rides along so that a resumed eager run does not replay the masks of steps 1..k)
    torch.save({"model": model.state_dict(), "optimizer": trainer.optimizer_state_dict(),
                "lr_scheduler": dict(scheduler) if scheduler else dict(), "epoch": epoch,
                "dropout_stream": {"seed_base": int(model._seed_base), "step": int(model._step)}}, path)


def load_checkpoint(model, trainer, path, strict=True):
    """Resume from a checkpoint written by :func:`save_checkpoint` or by the reference (same dict layout): model weights
    and BatchNorm buffers, Adam moments and step, learning rate.  Returns ``(epoch, lr_scheduler dict)``."""
    # the layout holds tensors, numbers and plain containers only: no arbitrary pickle code is executed
    ck = torch.load(path, map_location=model.flat_parameters().device, weights_only=True)
    model.load_state_dict(ck["model"], strict=strict)
    if trainer is not None and ck.get("optimizer"):
        trainer.load_optimizer_state_dict(ck["optimizer"])
    ds = ck.get("dropout_stream")
    if ds:
        model._seed_base, model._step = int(ds["seed_base"]), int(ds["step"])
    elif trainer is not None:
        model._step = trainer.step_count          # reference checkpoints: continue the stream at the optimizer's step
    return ck.get("epoch", 0), ck.get("lr_scheduler") or {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-p", "--params", default="parameters.yml")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp8"])
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--resume", default=None, help="checkpoint to resume from (reference layout, src/learn.py:188-195)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend under torchrun (nccl == RCCL)")
    ap.add_argument("--grad-groups", type=int, default=2, help="gradient buckets of mega blocks (overlapped all-reduce)")
    args = ap.parse_args()
    with open(args.params) as fh:
        params = Struct(**yaml.load(fh, Loader=yaml.SafeLoader))
    rank, world, device = init_distributed(args.backend)
    model, trainer, _ = run(params, steps=args.steps, precision=args.precision, resume=args.resume, device=device, rank=rank,
                            world=world, grad_groups=args.grad_groups)
    if args.checkpoint and rank == 0:          # replicas are identical: rank 0 writes (its BatchNorm running statistics, as DDP)
        opt = params.training.optimizer
        epoch = args.steps or params.training.epochs
        sched = {"T_max": params.training.epochs, "eta_min": opt.end_lr, "base_lrs": [opt.start_lr], "last_epoch": epoch,
                 "_last_lr": [trainer.lr]} if opt.scheduler else None
        save_checkpoint(model, trainer, epoch, args.checkpoint, scheduler=sched)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
