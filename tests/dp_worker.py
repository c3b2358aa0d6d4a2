"""Worker of tests/test_dp_gpu.py: one data-parallel rank of an L-ICV training run (licv.trainer.ICVTrainer) on a tiny Idefics.
Launched by torch.distributed.run (RANK / WORLD_SIZE / MASTER_* from the environment) with the gloo backend so that several
ranks can share the one GPU of a test box; on a multi-GPU node the same code runs over RCCL (backend nccl).  Also importable:
`run(rank, world, ...)` is what the single-process reference run of the test calls."""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "licv-vqa_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

ANS = 3           # answer tokens per question (same span in teacher and student rows: the collator contract)


def micro_batch_data(arch, index: int, B: int, dev):
    """Teacher (5 images, S=40) and student (1 image, S=16) batches of global micro-batch `index`, ending in the same answer tokens."""
    from licv.synthetic import synth_vqa_batch
    tea = synth_vqa_batch(arch, B, 40, 5, seed=7000 + index, min_len=34, dtype=torch.bfloat16)
    stu = synth_vqa_batch(arch, B, 16, 1, seed=8000 + index, min_len=12, dtype=torch.bfloat16)
    tl, sl = tea["attention_mask"].sum(1), stu["attention_mask"].sum(1)
    for b in range(B):
        stu["input_ids"][b, sl[b] - ANS: sl[b]] = tea["input_ids"][b, tl[b] - ANS: tl[b]]
    to = lambda d: {k: v.to(dev) for k, v in d.items()}
    return to(stu), to(tea), (sl - ANS).to(dev), (tl - ANS).to(dev)


def build(dev, accumulate: int, group=None):
    from icv_src.icv_module import VQAICVModule
    from licv.config import IDEFICS_TINY
    from licv.synthetic import synth_idefics_weights
    from licv.trainer import ICVTrainer
    from lmm_icl_interface import IdeficsInterface
    arch = IDEFICS_TINY
    sd = synth_idefics_weights(arch, seed=77, dtype=torch.float32)
    iface = IdeficsInterface(state_dict=sd, arch=arch, device=dev)
    mod_cfg = dict(hard_loss_weight=0.0, only_hard_loss=False, kl_eps=1e-6, init_temperature=1.0, learnable_t=False, decay_ratio=-1,
                   decay_per_step=-1, min_tmeprature=1.0, alpha_lr=1e-2, icv_lr=1e-3, weight_decay=1e-3, warm_steps=0,
                   icv_encoder=dict(use_sigmoid=True, alpha_learnable=True, alpha_init_value=0.3))
    lmm_cfg = dict(intervention_layer=-1, layer_format="model.model.layers.<LAYER_NUM>", total_layers=arch.num_layers,
                   hidden_size=arch.hidden_size)
    torch.manual_seed(426)                                             # identical initial parameters on every rank
    mod = VQAICVModule(iface, mod_cfg, lmm_cfg).to(dev)
    return arch, mod, ICVTrainer(mod, total_steps=10, accumulate_grad_batches=accumulate, grad_clip=1.0, group=group)


def run(rank: int, world: int, steps: int, accumulate_per_rank: int, dev="cuda:0"):
    """`steps` optimiser steps.  Global micro-batch g of a step goes to rank g % world (round robin, licv.trainer.shard_indices);
    with world == 1 the single process accumulates all world*accumulate micro-batches itself: the union-batch reference."""
    from licv.trainer import shard_indices
    total_micro = accumulate_per_rank * max(world, 1)
    arch, mod, tr = build(dev, accumulate_per_rank)
    logs = []
    for step in range(steps):
        mine = shard_indices(total_micro, rank, world)
        for g in mine:
            log = tr.micro_batch(*micro_batch_data(arch, step * total_micro + g, 2, dev))
        assert log is not None, "the window must close after this rank's last micro-batch"
        logs.append(log)
    enc = mod.icv_encoder
    return dict(alpha=enc.alpha.detach().cpu(), icv=enc.icv.detach().cpu(), logs=logs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--accumulate", type=int, default=2)
    ap.add_argument("--backend", default="gloo")
    args = ap.parse_args()
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist.init_process_group(args.backend)
    try:
        res = run(rank, world, args.steps, args.accumulate, dev=f"cuda:{local}")
        torch.save(res, Path(args.out) / f"rank{rank}.pt")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
