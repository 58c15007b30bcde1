"""Training entry point with the reference's config surface (/root/reference/train.py).

``generate_args`` keeps the 13 flags and defaults of train.py:11-27 verbatim and adds new ones
without renaming any.  ``train`` is the step loop of train.py:30-42 (H2D, forward, zero_grad,
backward, Adam step, running-mean meter).  The reference's dataset plumbing (dataset.py, hard-coded
Windows paths, data absent) is out of scope: ``--data synthetic`` (default) feeds batches in the
reference's batch contract from ``synth.make_batch``; a user-supplied DataLoader yielding
{"image","expression","position"} dicts works unchanged.

    python -m mclstexp_amd.train --dataset her2st --batch_size 8 --dim 785 --max_epochs 1 --steps_per_epoch 4
    torchrun --nproc-per-node 8 -m mclstexp_amd.train --batch_size 128 --dim 1000      # data parallel
"""
from __future__ import annotations

import argparse
import os
from typing import Iterable, Optional

import torch

from . import ops, synth
from .model import mclSTExp_Attention
from .optim import FusedAdam
from .utils import AvgMeter, get_lr


def generate_args(argv=None):
    parser = argparse.ArgumentParser()
    # --- the reference's 13 flags, train.py:13-25 (names, types and defaults unchanged)
    parser.add_argument('--batch_size', type=int, default=128, help='')
    parser.add_argument('--max_epochs', type=int, default=90, help='')
    parser.add_argument('--temperature', type=float, default=1., help='temperature')
    parser.add_argument('--fold', type=int, default=0, help='fold')
    parser.add_argument('--dim', type=int, default=785, help='spot_embedding dimension (# HVGs)')  # 171, 785, 685
    parser.add_argument('--image_embedding_dim', type=int, default=1024, help='image_embedding dimension')
    parser.add_argument('--projection_dim', type=int, default=256, help='projection_dim ')
    parser.add_argument('--heads_num', type=int, default=8, help='attention heads num')
    parser.add_argument('--heads_dim', type=int, default=64, help='attention heads dim')
    parser.add_argument('--heads_layers', type=int, default=2, help='attention heads layer num')
    parser.add_argument('--dropout', type=float, default=0., help='dropout')
    parser.add_argument('--dataset', type=str, default='her2st', help='dataset')  # her2st cscc 10x
    parser.add_argument('--encoder_name', type=str, default='densenet121', help='image encoder')
    # --- additions (none of the above renamed)
    parser.add_argument('--data', type=str, default='synthetic', help='synthetic (no dataset files needed)')
    parser.add_argument('--image_size', type=int, default=224, help='patch side in pixels (dataset.py:224: 224)')
    parser.add_argument('--steps_per_epoch', type=int, default=50, help='synthetic batches per epoch')
    parser.add_argument('--folds', type=int, default=1, help='outer fold loop (train.py:100 hard-codes 32)')
    parser.add_argument('--compute', type=str, default='f32', choices=['f32', 'bf16'],
                        help='MFMA operand type of the hand kernels (f32 = reference numerics)')
    parser.add_argument('--backbone_dtype', type=str, default='bf16', choices=['f32', 'bf16'],
                        help='activation type of the image backbone kernels (f32 = reference numerics: exact-fp32 MFMA)')
    parser.add_argument('--infonce', type=str, default='exact', choices=['exact', 'fused', 'fp8'],
                        help='exact = fp32 logits (reference numerics); fused = flash-style bf16 MFMA kernel; fp8 = the '
                             'similarity contraction on e4m3 operands with hardware block scales (BASELINE configs[4])')
    parser.add_argument('--hip_graphs', action='store_true',
                        help='replay forward / backward from HIP graphs (engine.TrainStep) instead of eager launches')
    parser.add_argument('--save_dir', type=str, default='', help='if set: torch.save(state_dict) per fold (train.py:87-95)')
    parser.add_argument('--log_every', type=int, default=10, help='loss.item() sync period (reference: every step)')
    return parser.parse_args(argv)


class SyntheticLoader:
    """Yields CPU batches in the reference's batch contract (dataset.py:188-195,226-231)."""

    def __init__(self, args, steps: int, rank: int = 0):
        self.args, self.steps, self.rank = args, steps, rank

    def __len__(self):
        return self.steps

    def __iter__(self):
        a = self.args
        for s in range(self.steps):
            yield synth.make_batch(a.batch_size, a.dim, image_hw=a.image_size, seed=s, rank=self.rank)


def train(model, train_dataLoader: Iterable, optimizer, epoch: int, log_every: int = 1, reducer=None, stepper=None):
    """train.py:30-42.  ``log_every`` > 1 relaxes the reference's per-step ``loss.item()`` device sync
    (the meter then samples every log_every-th step); 1 reproduces the reference exactly.  ``stepper``: an
    engine.TrainStep that performs the same forward / zero_grad / backward / step from captured HIP graphs."""
    loss_meter = AvgMeter()
    step = 0
    sizes_ex = None
    if reducer is not None and stepper is None:
        from . import dist as mdist
        sizes_ex = mdist.SizeExchange(reducer.pg)
    for batch in train_dataLoader:
        batch = {k: v.cuda(non_blocking=True) for k, v in batch.items() if
                 k == "image" or k == "expression" or k == "position"}
        if sizes_ex is not None:
            # ragged last batch (no drop_last, train.py:49): the ranks agree on their shard sizes on the host
            mdist.set_step_sizes(sizes_ex(batch["expression"].shape[0]))
        if stepper is not None:
            if batch["image"].dtype == torch.float32 and getattr(model, "backbone_dtype", None) is not None:
                batch["image"] = batch["image"].contiguous(memory_format=torch.channels_last)
            loss = stepper(batch)
        else:
            loss = model(batch)
            optimizer.zero_grad()
            loss.backward()
            if reducer is not None:
                reducer.reduce(optimizer)
            optimizer.step()
        step += 1
        if step % log_every == 0:
            count = batch["image"].size(0)
            loss_meter.update(loss.item(), count)
            # device error words, at the sync point the meter forces anyway: nn.Embedding's IndexError for a position outside
            # the tables, SeamTimeoutError if a persistent dense-block launch gave up at a BatchNorm seam (step invalid)
            ops.check_device_errors()
    return loss_meter


def save_model(args, model, rank: int = 0):
    """train.py:87-95: one state_dict per fold, keys = module tree names."""
    if not args.save_dir or rank != 0:
        return
    os.makedirs(os.path.join(args.save_dir, args.dataset), exist_ok=True)
    torch.save(model.state_dict(), os.path.join(args.save_dir, args.dataset, f"best_{args.fold}.pt"))


def main(argv=None):
    args = generate_args(argv)
    from . import dist as mdist
    pg, rank, world = mdist.init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError("mclstexp_amd.train needs an MI355X: the hot path has no CPU fallback")
    device = torch.device("cuda", torch.cuda.current_device())
    bb = torch.bfloat16 if args.backbone_dtype == "bf16" else None
    for i in range(args.folds):
        args.fold = i
        loader = SyntheticLoader(args, args.steps_per_epoch, rank)
        torch.manual_seed(0)
        model = mclSTExp_Attention(encoder_name=args.encoder_name, spot_dim=args.dim, temperature=args.temperature,
                                   image_dim=args.image_embedding_dim, projection_dim=args.projection_dim,
                                   heads_num=args.heads_num, heads_dim=args.heads_dim, head_layers=args.heads_layers,
                                   dropout=args.dropout, compute=args.compute, backbone_dtype=bb,
                                   embedding_grad="rowsparse", process_group=pg if world > 1 else None,
                                   infonce=args.infonce)
        model.to(device)
        if bb is not None:
            model.to(memory_format=torch.channels_last)
        optimizer = FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-3).attach_model(model)   # train.py:118-120
        reducer = mdist.GradReducer(pg) if world > 1 else None
        stepper = None
        if args.hip_graphs:
            from .engine import TrainStep
            stepper = TrainStep(model, optimizer, reducer, graphs=True)
        for epoch in range(args.max_epochs):
            model.train()
            meter = train(model, loader, optimizer, epoch, args.log_every, reducer, stepper)
            if rank == 0:
                print(f"fold {i} epoch {epoch} train_loss {meter.avg:.4f} lr {get_lr(optimizer)}")
        save_model(args, model, rank)
    mdist.shutdown()


if __name__ == '__main__':
    main()
