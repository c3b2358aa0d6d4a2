#!/usr/bin/env python3
"""Headline benchmark: VQA questions/sec of the ICV-injected Idefics-9B forward, 32-shot, bs=8, bf16
(BASELINE.json configs[1]; SURVEY.md §8d shape "H") on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N ...                      (starts its own N ranks: a child `torch.distributed.run`, before this
                                                       process has touched the GPU, and exits with the child's code)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        (the driver's form: RANK / LOCAL_RANK / WORLD_SIZE from the environment)

A step = one full hooked forward over one batch resident in HBM: ViT-H/14 on B*33 images, perceiver,
32 decoder + 8 gated cross-attention layers with the ICV hook on every decoder layer, LM head (logits for all
positions).  Forward-only data parallel: questions shard across ranks, no data-path collective ("weak").
Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "licv-vqa_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

WORKLOADS = {
    # name: (arch preset, B, S, n_img, min_len)
    "idefics9b_32shot_bs8": ("idefics-9b", 8, 800, 33, 720),
    "idefics9b_student_bs8": ("idefics-9b", 8, 32, 1, 24),
    # the image input pipeline (uint8 on the host -> pinned staging -> H2D on a side stream -> normalise kernel)
    # beside the headline forward: images/s of the feeder alone and with the forward running, and the headline step when every
    # step's pixel_values come from host bytes instead of a device-resident tensor
    "frontend_images_bs8": ("idefics-9b", 8, 800, 33, 720),
    "idefics_mid_debug": ("idefics-mid", 4, 96, 5, 80),
    # ref:inference.py:300-321 shape: hooked generate on the query-only prompt, 3 beams, 5 new tokens (ref:config/inference.yaml:26-30)
    "idefics9b_generate_bs8": ("idefics-9b", 8, 32, 1, 32),
    # BASELINE configs[2]: L-ICV training step = teacher (32-shot, no grad) + student (query only, with grad) + KL +
    # backward + [every 2nd micro-batch] all-reduce + clipped AdamW.  (arch, B, S_teacher, n_img_teacher, min_len)
    "idefics9b_train_bs8": ("idefics-9b", 8, 800, 33, 720),
    "idefics_mid_train_debug": ("idefics-mid", 4, 96, 5, 80),
    # the same training step with the ICV-independent work reused across steps (licv.feature_cache, ICVTrainer.enable_caches).
    #  _cached_vision : every image of the step is in the vision-feature cache (100 % hits: an image pool that has been seen, as from
    #                   the second pass over the 8000-query pool on; ref:icv_src/icv_datasets/vqa_dataset.py:90-98 re-draws the SHOTS per
    #                   step, so the teacher's (query, shots) row is new: 100 % teacher-row misses) - the ViT + perceiver leave the step;
    #  _cached_teacher: the teacher's answer rows are cached too (100 % hits: the same (query, shots) pairs come back, a fixed-shot
    #                   recipe or later epochs with a fixed sampler seed) - only the student's forward / backward remains.
    "idefics9b_train_bs8_cached_vision": ("idefics-9b", 8, 800, 33, 720),
    "idefics9b_train_bs8_cached_teacher": ("idefics-9b", 8, 800, 33, 720),
    "idefics_mid_train_debug_cached_vision": ("idefics-mid", 4, 96, 5, 80),
    "idefics_mid_train_debug_cached_teacher": ("idefics-mid", 4, 96, 5, 80),
    # BASELINE configs[3] = SURVEY.md §8d shape "I2": Idefics2-8B-base 1-shot, B=8, 2 images per question at 378x504
    # (27x36 = 972 patches), 64 <image> tokens each, S = 2*66 + 40 = 172, hook on all 32 MLP branches
    "idefics2_8b_1shot_bs8": ("idefics2-8b", 8, 172, 2, 160),
    "idefics2_mid_debug": ("idefics2-mid", 2, 40, 2, 30),
    # BASELINE configs[4] shape (SURVEY "I2-32") in bf16: 33 images per question, S ~ 33*66 + 700; the fp8-weight GEMM it names is not built
    "idefics2_8b_32shot_bs8": ("idefics2-8b", 8, 2900, 33, 2800),
    # the same with the text stack's GEMMs on e4m3 operands (W8A8, per-channel / per-row scales, fp32 accumulate)
    "idefics2_8b_32shot_fp8_bs8": ("idefics2-8b", 8, 2900, 33, 2800),
    # Idefics2 L-ICV training micro-batch: teacher 32-shot (no grad) + student query-only (grad) + KL + backward
    "idefics2_8b_train_bs8": ("idefics2-8b", 8, 2900, 33, 2800),
    "idefics2_mid_train_debug": ("idefics2-mid", 2, 60, 3, 50),
}
IDEFICS2_IMAGE = {"idefics2-8b": (378, 504), "idefics2-mid": (84, 70)}


def _host_cores() -> int:
    """CPU share of this process: cgroup quota if set, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_baseline_one(arch, S, n_img, dtype):
    """One question at full width through ONE layer of each kind in the CPU oracle at `dtype`; returns (s/question, timings)."""
    from licv.synthetic import synth_idefics_weights, synth_vqa_batch
    from oracle import icv_ref as O
    from oracle import idefics_ref as R
    small = arch.with_(v_layers=1, r_depth=1, num_layers=1, cross_layer_interval=1)
    sd = synth_idefics_weights(small, seed=1, dtype=dtype)
    batch = synth_vqa_batch(small, 1, S, n_img, seed=2, min_len=S, dtype=dtype)
    icv = torch.randn(1, 1, arch.hidden_size) * 0.01
    tm = {}

    def timed(name, fn):
        t0 = time.perf_counter()
        r = fn()
        tm[name] = time.perf_counter() - t0
        return r

    with torch.no_grad():
        pv = batch["pixel_values"].view(n_img, *batch["pixel_values"].shape[2:])
        x = timed("vit_layer", lambda: R.vision_tower(pv, sd, small))
        img = timed("perceiver_block", lambda: R.perceiver(x, sd, small)).view(1, -1, arch.v_embed)
        pos, causal, img_mask, gate = R.build_masks(batch["attention_mask"], batch["image_attention_mask"], arch.image_seq_len, dtype)
        cos, sin = R.rotary_tables(arch.head_dim, arch.max_positions, arch.rope_base, dtype)
        h = R.decoupled_embedding(batch["input_ids"], sd, arch.vocab_size)
        h = timed("xattn_layer", lambda: R.gated_xattn_layer(h, sd, 0, small, img, img_mask, gate))
        h = timed("decoder_layer+hook", lambda: O.inject_renorm(R.decoder_layer(h, sd, 0, small, causal, pos, cos, sin), icv[:, 0]))
        timed("norm+lm_head", lambda: R.lm_head(R.rms_norm(h.to(dtype), sd["model.norm.weight"], arch.rms_eps), sd))
    full = (tm["vit_layer"] * arch.v_layers + tm["perceiver_block"] * arch.r_depth + tm["xattn_layer"] * arch.num_cross_layers
            + tm["decoder_layer+hook"] * arch.num_layers + tm["norm+lm_head"])
    return full, tm


def cpu_baseline(arch, S, n_img):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores on a BOUNDED sample: one question at
    full width through ONE layer of each kind (ViT layer on all images, perceiver block, gated cross-attention layer, hooked
    decoder layer, LM head), scaled by the layer counts — in fp32 AND in bf16 (the reference's model dtype; without AMX/
    AVX512-BF16 a CPU runs bf16 slower than fp32, so `value` is the FASTER of the two: the baseline is what the host can do)."""
    cores = _host_cores()
    torch.set_num_threads(cores)
    res = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        res[name] = _cpu_baseline_one(arch, S, n_img, dt)
    best = min(res, key=lambda k: res[k][0])
    full, tm = res[best]
    return {
        "value": 1.0 / full, "unit": "questions/s", "cores": cores, "kind": "port", "dtype": best,
        "value_fp32": 1.0 / res["fp32"][0], "value_bf16": 1.0 / res["bf16"][0],
        "sample": (f"1 question (S={S}, {n_img} images) through the CPU oracle, one layer of each kind, fp32 and bf16 "
                   f"(fp32 {res['fp32'][0]:.1f} s/question, bf16 {res['bf16'][0]:.1f} s/question; value = {best}): "
                   + ", ".join(f"{k} {v:.2f}s" for k, v in tm.items())
                   + f"; scaled by layer counts ({arch.v_layers} ViT, {arch.r_depth} perceiver, {arch.num_cross_layers} x-attn, "
                   f"{arch.num_layers} decoder) = {full:.1f} s/question"),
    }


def cpu_baseline_generate(arch, S, n_img):
    """Bounded CPU sample for the hooked-generate workload (ref:inference.py:300-321 shape: 3 beams, 5 new tokens): ONE question
    through the oracle's own beam search (oracle/generate_ref.py, KV cache, hooks live at every step) on the architecture
    truncated to 4 and to 8 decoder layers (1 and 2 gated cross-attention layers, 1 ViT layer, 1 perceiver block), bf16; the
    per-4-layer difference is scaled to the full depth and the vision tower's remaining layers are added from the one-layer
    timings of `_cpu_baseline_one`."""
    from licv.synthetic import synth_idefics_weights, synth_vqa_batch
    from oracle.generate_ref import generate as oracle_generate
    cores = _host_cores()
    torch.set_num_threads(cores)
    dt = torch.bfloat16
    tm = {}
    for nl in (4, 8):
        small = arch.with_(v_layers=1, r_depth=1, num_layers=nl)
        sd = synth_idefics_weights(small, seed=1, dtype=dt)
        b = synth_vqa_batch(small, 1, S, n_img, seed=2, min_len=S, dtype=dt)
        icv = torch.randn(1, nl, arch.hidden_size) * 0.01
        with torch.no_grad():
            t0 = time.perf_counter()
            oracle_generate(sd, small, **b, icv=icv, hook_layers=list(range(nl)), num_beams=3, max_new_tokens=5, length_penalty=0.0,
                            min_new_tokens=0)
            tm[f"{nl} layers"] = time.perf_counter() - t0
        del sd
    _, one = _cpu_baseline_one(arch, S, n_img, dt)
    per4 = max(tm["8 layers"] - tm["4 layers"], 0.0)
    full = (tm["4 layers"] + per4 * (arch.num_layers - 4) / 4 + one["vit_layer"] * (arch.v_layers - 1)
            + one["perceiver_block"] * (arch.r_depth - 1))
    return {"value": 1.0 / full, "unit": "questions/s", "cores": cores, "kind": "port", "dtype": "bf16",
            "sample": (f"1 question (S={S}, {n_img} image) through the CPU oracle's hooked beam search (3 beams, 5 new tokens) at 4 and 8 "
                       f"decoder layers: " + ", ".join(f"{k} {v:.2f}s" for k, v in tm.items())
                       + f"; per-4-layer difference scaled to {arch.num_layers} layers + {arch.v_layers - 1} more ViT layers "
                       f"({one['vit_layer']:.2f}s each) + {arch.r_depth - 1} more perceiver blocks = {full:.1f} s/question")}


def frontend_bench(args, eng, arch, batch, hooks, B, n_img, dev):
    """Workload frontend_images_bs8 (one GPU): see WORKLOADS."""
    import numpy as np
    from licv.image_feeder import ImageFeeder
    side = arch.v_image
    n = B * n_img
    rng = np.random.default_rng(426)
    host = [rng.integers(0, 256, (n, side, side, 3), dtype=np.uint8) for _ in range(2)]
    feeder = ImageFeeder(dev, n, side, side)
    kw = {k: v for k, v in batch.items() if k != "pixel_values"}

    def timed(fn, iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(iters)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # (1) the feeder alone
    def alone(iters):
        for i in range(iters):
            feeder.get(feeder.submit(host[i & 1]))
    alone(4)
    t_alone = timed(alone, 40)
    # (2) headline forward on device-resident pixel_values (the bench's own configuration)
    def resident(iters):
        for _ in range(iters):
            eng.forward(**batch, **hooks)
    resident(args.warmup)
    t_res = timed(resident, args.steps)
    # (3) every step's images come from host bytes: batch i + 1 is submitted (packed, copied, normalised on the side stream) while
    # the forward of batch i runs
    def fed(iters):
        t = feeder.submit(host[0])
        for i in range(iters):
            pv, _ = feeder.get(t, B, n_img)
            nxt = feeder.submit(host[(i + 1) & 1])
            eng.forward(**kw, pixel_values=pv, **hooks)
            feeder.release(t)
            t = nxt
    fed(args.warmup)
    t_fed = timed(fed, args.steps)
    # (4) the feeder at full tilt beside the forward: `k` extra batches per step on the side stream
    k = 8
    def beside(iters):                                    # (submit only: a get() would make the main stream a consumer of every batch)
        for i in range(iters):
            eng.forward(**batch, **hooks)
            for j in range(k):
                feeder.submit(host[j & 1])
    beside(2)
    t_beside = timed(beside, args.steps)
    res = {
        "metric": "frontend images/s: uint8 host -> pinned staging -> H2D -> normalised bf16 on the device (licv.image_feeder), headline forward running concurrently",
        "value": args.steps * k * n / t_beside, "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t_beside / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 -> bf16",
        "data": "synthetic (seeded uint8 images, random-init idefics-9b weights)",
        "config": {"workload": args.workload, "images_per_batch": n, "image": f"{side}x{side}x3 uint8", "extra_batches_per_step": k,
                   "questions_per_gpu": B, "images_per_question": n_img},
        "alone_images_per_s": 40 * n / t_alone, "alone_ms_per_batch": 1e3 * t_alone / 40,
        "headline_resident": {"questions_per_s": B * args.steps / t_res, "ms_per_step": 1e3 * t_res / args.steps},
        "headline_fed_from_host_uint8": {"questions_per_s": B * args.steps / t_fed, "ms_per_step": 1e3 * t_fed / args.steps,
                                         "images_per_s": n * args.steps / t_fed},
        "forward_ms_per_step_with_feeder_at_full_tilt": 1e3 * t_beside / args.steps,
        "bytes": {"per_image_over_pcie": side * side * 3, "per_image_written_bf16": side * side * 3 * 2,
                  "note": "the reference ships float32 pixel_values (4x the bytes) produced by the HF image processor on the host"},
        "roofline": {"bound": "hbm", "kernel": "image_preprocess_k", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None,
                     "note": "451 KB of device traffic per image: 10 k images/s is 4.5 GB/s, nowhere near a device bound; the pipeline is host-side (packing into pinned memory) and PCIe"},
    }
    print(json.dumps(res), flush=True)


def gemm_src_sha16() -> str:
    """Hash of the GEMM sources of THIS tree (same function as tools/pmc_summary.py): a committed PMC summary is quoted in
    `roofline.traffic` only if it was measured on kernels built from the same sources."""
    import hashlib
    csrc = ROOT / "licv-vqa_amd" / "csrc"
    h = hashlib.sha256()
    for f in sorted([f for f in csrc.glob("gemm*") if f.is_file()] + [csrc / "common.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def gpu_unfused_baseline(arch, sd, batch, icv_eff, layers, B, steps=3):
    """"Reference on GPU" denominator: the SAME restatement of the reference path that
    serves as the oracle, run unfused on this GPU through PyTorch-ROCm's own kernels (rocBLAS/hipBLASLt GEMMs, eager attention,
    one ATen launch per elementwise op, the hook as the reference's 7 ATen ops) at the headline shape, full depth, bf16.
    A reported baseline only: nothing of the product path goes through it."""
    from oracle import idefics_ref as R
    kw = {k: v for k, v in batch.items()}
    times = []
    with torch.no_grad(), torch.device(batch["input_ids"].device):       # factory calls inside the oracle land on the GPU
        for i in range(steps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = R.forward(sd, arch, **kw, icv=icv_eff, hook_layers=layers)
            torch.cuda.synchronize()
            if i:                                                           # first pass = warm-up (library handles, autotune)
                times.append(time.perf_counter() - t0)
            del out
    t = sorted(times)[len(times) // 2]
    return {"value": B / t, "unit": "questions/s", "ms_per_step": 1e3 * t, "steps": steps, "kind": "oracle restatement on device=cuda (unfused PyTorch-ROCm)",
            "dtype": "bf16 weights, fp32 residual stream after the first hook (as the reference)"}


def cpu_baseline_idefics2(arch, S, n_img, hw):
    """Same bounded sample for Idefics2: one question through one SigLIP layer, the modality projection + one perceiver
    layer, one hooked Mistral layer and the LM head in the CPU oracle under bf16 autocast; scaled by the layer counts."""
    from licv.synthetic import synth_idefics2_weights, synth_vqa_batch_idefics2
    from oracle import idefics2_ref as R2
    cores = _host_cores()
    torch.set_num_threads(cores)
    tm = {}

    def run(small, tag):
        sd = synth_idefics2_weights(small, seed=1, dtype=torch.bfloat16)
        b = synth_vqa_batch_idefics2(small, 1, S, n_img, hw[0] // small.v_patch * small.v_patch, hw[1] // small.v_patch * small.v_patch,
                                     seed=2, min_len=S, dtype=torch.bfloat16, ragged=False)
        icv = torch.randn(1, small.num_layers, arch.hidden_size) * 0.01
        best = float("inf")
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            for _ in range(3):                                   # min of 3: the per-layer figures are differences of these
                t0 = time.perf_counter()
                R2.forward(sd, small, **b, icv=icv, hook_layers=list(range(small.num_layers)))
                best = min(best, time.perf_counter() - t0)
        tm[tag] = best

    # per-layer costs from 4-layer differences (a single layer is a few tens of ms here: below the run-to-run noise)
    run(arch.with_(v_layers=1, r_depth=1, num_layers=1), "1v+1p+1t")
    run(arch.with_(v_layers=5, r_depth=1, num_layers=1), "5v+1p+1t")
    run(arch.with_(v_layers=1, r_depth=3, num_layers=1), "1v+3p+1t")
    run(arch.with_(v_layers=1, r_depth=1, num_layers=5), "1v+1p+5t")
    base = tm["1v+1p+1t"]
    dv, dp, dt_ = max(tm["5v+1p+1t"] - base, 0.0) / 4, max(tm["1v+3p+1t"] - base, 0.0) / 2, max(tm["1v+1p+5t"] - base, 0.0) / 4
    full = base + dv * (arch.v_layers - 1) + dp * (arch.r_depth - 1) + dt_ * (arch.num_layers - 1)
    return {"value": 1.0 / full, "unit": "questions/s", "cores": cores, "kind": "port",
            "sample": (f"1 question (S={S}, {n_img} images {hw[0]}x{hw[1]}) through the CPU oracle (bf16 autocast) truncated to 1-5 layers "
                       f"of each kind (min of 3 runs each): " + ", ".join(f"{k} {v:.2f}s" for k, v in tm.items())
                       + f"; per-layer differences scaled to {arch.v_layers} SigLIP / {arch.r_depth} perceiver / {arch.num_layers} text layers "
                       f"= {full:.1f} s/question")}


def build_trainer(arch, sd, dev, B, S, n_img, min_len, rank, hw=None):
    """VQAICVModule + ICVTrainer on synthetic teacher/student batches obeying the collator contract (same answer span
    at the tail of both rows, ref:icv_src/icv_datamodule.py:73-130)."""
    from icv_src.icv_module import VQAICVModule
    from licv.synthetic import synth_vqa_batch, synth_vqa_batch_idefics2
    from licv.trainer import ICVTrainer
    from lmm_icl_interface import Idefics2Interface, IdeficsInterface
    is2 = hw is not None
    iface = (Idefics2Interface if is2 else IdeficsInterface)(state_dict=sd, arch=arch, device=dev)
    mod_cfg = dict(hard_loss_weight=0.0, only_hard_loss=False, kl_eps=1e-6, init_temperature=1.0, learnable_t=False, decay_ratio=-1,
                   decay_per_step=-1, min_tmeprature=1.0, alpha_lr=1e-2, icv_lr=1e-4, weight_decay=1e-3, warm_steps=0.1,
                   icv_encoder=dict(use_sigmoid=True, alpha_learnable=True, alpha_init_value=0.0))
    lmm_cfg = dict(intervention_layer=-1, total_layers=arch.num_layers, hidden_size=arch.hidden_size,
                   layer_format="model.model.text_model.layers.<LAYER_NUM>.mlp" if is2 else "model.model.layers.<LAYER_NUM>")
    torch.manual_seed(426)
    mod = VQAICVModule(iface, mod_cfg, lmm_cfg).to(dev)
    ans = 4                                                           # answer span = last 4 real tokens (SURVEY §8d "TR")
    if is2:
        Sq = arch.r_latents + 2 + (40 if S >= 256 else 10)              # one image (64 <image> tokens + 2) + the question
        tea = synth_vqa_batch_idefics2(arch, B, S, n_img, hw[0], hw[1], seed=426 + rank, min_len=min_len, dtype=torch.bfloat16, ragged=False)
        stu = synth_vqa_batch_idefics2(arch, B, Sq, 1, hw[0], hw[1], seed=1426 + rank, min_len=Sq - 8, dtype=torch.bfloat16, ragged=False)
    else:
        Sq = 32 if S >= 256 else 16
        tea = synth_vqa_batch(arch, B, S, n_img, seed=426 + rank, min_len=min_len, dtype=torch.bfloat16)
        stu = synth_vqa_batch(arch, B, Sq, 1, seed=1426 + rank, min_len=Sq - 8, dtype=torch.bfloat16)
    tl, sl = tea["attention_mask"].sum(1), stu["attention_mask"].sum(1)
    for b in range(B):
        stu["input_ids"][b, sl[b] - ans: sl[b]] = tea["input_ids"][b, tl[b] - ans: tl[b]]
    trainer = ICVTrainer(mod, sd, total_steps=1000, accumulate_grad_batches=2, grad_clip=1.0)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}
    return trainer, (to(stu), to(tea), (sl - ans).to(dev), (tl - ans).to(dev))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="idefics9b_32shot_bs8", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prefetch-mib", type=int, default=-1, help="decode steps: MiB of the next projection's weights a side stream reads ahead (licv_runner_option 2); -1 = the library's default")
    ap.add_argument("--no-gpu-baseline", action="store_true", help="skip the unfused PyTorch-ROCm run of the oracle on the GPU (row G0)")
    ap.add_argument("--no-hooks", action="store_true", help="teacher shape: same forward with the intervention off")
    ap.add_argument("--no-profiler", action="store_true", help="no per-launch event pairs: the language stack then runs through the native layer runner")
    ap.add_argument("--batch-streams", type=int, default=-1, help="diagnostic A/B only: slices of the batch run on HIP streams of their own (engine default: 2); 1 = off")
    ap.add_argument("--gemm-knob", type=int, nargs=2, action="append", default=[], metavar=("KNOB", "VALUE"),
                    help="diagnostic A/B only: licv_gemm_experiment(KNOB, VALUE), e.g. 12 0 = rows 129-256 stay on the 128-tile kernel")
    ap.add_argument("--gemm-select", type=int, default=0, help="diagnostic A/B only: force a GEMM kernel variant (licv_gemm_select); 0 = the library's own choice")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; default).  'gloo' only to rehearse the multi-rank code path "
                    "on a box with fewer GPUs than ranks (ranks then share devices: timings are meaningless)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start N ranks as fresh child processes (one per GPU, RCCL) BEFORE this process has
        # made any GPU call, and hand their exit code back
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
        sys.exit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU: the L-ICV hot path has no CPU fallback"
    if args.dist_backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    dev_name = f"rank{rank}:cuda:{local}:{torch.cuda.get_device_name(local)}"
    if dist is not None:
        names = [None] * world
        dist.all_gather_object(names, dev_name)
        dist_info = {"dist_backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "devices": names}
    else:
        dist_info = {"dist_backend": None, "world_size_seen": 1, "devices": [dev_name]}

    from licv import ops
    if args.prefetch_mib >= 0:
        from licv import _lib as _l
        _l.check(_l.lib().licv_runner_option(2, args.prefetch_mib))
    for knob, value in args.gemm_knob:
        from licv import _lib as _lk
        _lk.check(_lk.lib().licv_gemm_experiment(knob, value))
    if args.gemm_select:
        from licv import _lib
        _lib.lib().licv_gemm_select(args.gemm_select)
    if os.environ.get("LICV_NO_FOLD"):                        # diagnostic A/B only
        from licv import _lib
        _lib.lib().licv_runner_option(0, 0)
    from licv.config import idefics_arch
    from licv.idefics_engine import IdeficsEngine, IdeficsWeights
    from licv.roofline import PEAK_BF16_TFLOPS, PEAK_HBM_GBS, flops_per_question, inject_bytes_per_question
    mfma_peak = 2 * PEAK_BF16_TFLOPS if "fp8" in args.workload else PEAK_BF16_TFLOPS
    from licv.synthetic import synth_icv, synth_idefics_weights, synth_vqa_batch

    preset, B, S, n_img, min_len = WORKLOADS[args.workload]
    arch = idefics_arch(preset)
    is2 = preset.startswith("idefics2")
    training = "train" in args.workload
    trainer = None
    if is2:
        from licv.idefics2_engine import Idefics2Engine, Idefics2Weights
        from licv.synthetic import synth_idefics2_weights, synth_vqa_batch_idefics2
        sd = synth_idefics2_weights(arch, seed=426, dtype=torch.bfloat16, device=dev)
    else:
        sd = synth_idefics_weights(arch, seed=426, dtype=torch.bfloat16, device=dev)       # full replica per GPU
    if training:
        trainer, train_args = build_trainer(arch, sd, dev, B, S, n_img, min_len, rank, hw=IDEFICS2_IMAGE[preset] if is2 else None)
        eng = trainer.m.interface.engine
    elif is2:
        eng = Idefics2Engine(Idefics2Weights(sd, arch, dev, fp8_text="fp8" in args.workload, fp8_vision="fp8" in args.workload))
    else:
        eng = IdeficsEngine(IdeficsWeights(sd, arch, dev))
    if args.batch_streams >= 0 and hasattr(eng, "batch_streams"):
        eng.batch_streams = args.batch_streams
    if os.environ.get("LICV_NO_FOLD") and hasattr(eng, "fold_residual"):
        eng.fold_residual = False
    want_g0 = (rank == 0 and world == 1 and not is2 and not training and "generate" not in args.workload and not args.no_gpu_baseline
               and not args.no_hooks)
    if not want_g0:
        del sd
    torch.cuda.empty_cache()
    if is2:
        ih, iw = IDEFICS2_IMAGE[preset]
        batch = synth_vqa_batch_idefics2(arch, B, S, n_img, ih, iw, seed=426 + rank, min_len=min_len, dtype=torch.bfloat16, device=dev,
                                         ragged=False)
    else:
        batch = synth_vqa_batch(arch, B, S, n_img, seed=426 + rank, min_len=min_len, dtype=torch.bfloat16, device=dev)
    icv, alpha = synth_icv(arch.num_layers, arch.hidden_size, seed=426, alpha=0.1, device=dev)
    layers = list(range(arch.num_layers))
    hooks = {} if args.no_hooks else dict(icv=icv, alpha=alpha, hook_layers=layers)

    if args.workload.startswith("frontend_images"):
        assert world == 1, "the image-input workload is a one-GPU measurement"
        frontend_bench(args, eng, arch, batch, hooks, B, n_img, dev)
        return
    generating = "generate" in args.workload
    if generating:
        from licv.generation import generate as native_generate
        gen_hooks = {} if args.no_hooks else dict(icv=alpha.unsqueeze(-1) * icv, hook_layers=layers)

    cache_mode = "vision" if "cached_vision" in args.workload else ("teacher" if "cached_teacher" in args.workload else None)
    cache_step = [0]
    if cache_mode:
        # ids the dataset has on the host: one per image (stable: the pool has been seen), one per (query, shots) pair
        trainer.enable_caches(vision_images=4096, teacher_rows=4096 if cache_mode == "teacher" else 0)
        image_ids = {"query_inputs": [[f"q{rank}_{b}"] for b in range(B)],
                     "inputs": [[f"t{rank}_{b}_{k}" for k in range(n_img)] for b in range(B)]}

    def step():
        if training and cache_mode:
            cache_step[0] += 1
            keys = [f"pair{rank}_{b}" for b in range(B)] if cache_mode == "teacher" else None
            return trainer.micro_batch(*train_args, image_ids=image_ids, teacher_keys=keys)
        if training:
            return trainer.micro_batch(*train_args)
        if generating:
            return native_generate(eng, batch["input_ids"], batch["attention_mask"], batch["pixel_values"], batch["image_attention_mask"],
                                   max_new_tokens=5, num_beams=3, length_penalty=0.0, **gen_hooks)
        return eng.forward(**batch, **hooks)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if training:
        trainer.allreduce_events = []                                    # HIP event pair around the one collective of an optimiser step
    prof = []
    # per-launch event pairs (roofline.achieved) need the Python-level launches; the launch-bound workloads (hooked generate, the
    # 32-token student shape) are timed on the native layer runner instead and report no per-kernel roofline
    profiled = not (generating or "student" in args.workload or args.no_profiler)
    # sliced configuration (Idefics engine, two batch slices on two streams): the timed region below runs the product path as it is
    # (no event pairs, native layer runner); the per-launch measurements come from two further regions after it
    sliced = (profiled and not generating and int(getattr(eng, "batch_streams", 1) or 1) > 1 and B >= 4 and B * S >= 4096)
    if profiled and not sliced:
        ops.set_profiler(prof)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = step()
        marks[i + 1].record()                                          # per-step boundaries on the stream (no sync): median below
    fence()
    elapsed = time.perf_counter() - t0
    ops.set_profiler(None)
    del out
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if dist is not None:
        t = torch.tensor([elapsed, median_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, median_ms = float(t[0]), float(t[1])

    ms_per_step = 1e3 * elapsed / args.steps
    qps = B * world * args.steps / elapsed

    # With the batch cut in slices on two HIP streams (the product configuration, timed above without any instrumentation) a GEMM's
    # start-to-end time includes the share of the machine the other slice's kernels took, so it says little about the kernel.  The
    # roofline of the dominant kernel therefore comes from a further timed region of the same step with the slices off (kernels of
    # one stream never overlap): per-launch HIP event pairs over min(steps, 6) steps, its ms per step reported beside it; a short
    # region with event pairs in the sliced configuration gives the figures kept under "overlapped" (union of the launch intervals,
    # mean start-to-end time).
    overlapped = None
    if sliced:
        n_sl = int(eng.batch_streams)
        o_steps = min(args.steps, 3)                          # short region with event pairs ON in the sliced configuration
        ops.set_profiler([])                                  # (two untimed steps first: the Python-level path allocates its own buffers)
        step()
        step()
        prof = []
        ops.set_profiler(prof)
        first = torch.cuda.Event(enable_timing=True)
        last = torch.cuda.Event(enable_timing=True)
        fence()
        first.record()
        for i in range(o_steps):
            out = step()
        last.record()
        fence()
        ops.set_profiler(None)
        del out
        o_elapsed = first.elapsed_time(last) * 1e-3
        ev = sorted((first.elapsed_time(e0) * 1e-3, first.elapsed_time(e1) * 1e-3, w) for k, e0, e1, w, *_ in prof if k == "gemm")
        union, hi = 0.0, -1.0
        for a0, a1, _ in ev:
            if a1 > hi:
                union += a1 - max(a0, hi)
                hi = a1
        fl_o = sum(w for _, _, w in ev)
        overlapped = {"batch_streams": n_sl, "steps": o_steps, "ms_per_step_with_event_pairs": 1e3 * o_elapsed / o_steps,
                      "gemm_launches_per_step": len(ev) // max(o_steps, 1),
                      "gemm_busy_union_share_of_step": union / o_elapsed, "gemm_tflops_over_union": fl_o / union / 1e12 if union else None,
                      "avg_launch_us_start_to_end": 1e6 * sum(a1 - a0 for a0, a1, _ in ev) / max(len(ev), 1)}
        eng.batch_streams = 1
        serial_steps = min(args.steps, 6)
        step()
        prof = []
        ops.set_profiler(prof)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(serial_steps + 1)]
        fence()
        marks[0].record()
        for i in range(serial_steps):
            out = step()
            marks[i + 1].record()
        fence()
        ops.set_profiler(None)
        del out
        eng.batch_streams = n_sl
        ser = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(serial_steps))
        overlapped["serial_pass"] = {"steps": serial_steps, "ms_per_step_median": ser[len(ser) // 2]}
    roof_steps = overlapped["serial_pass"]["steps"] if sliced else args.steps
    roof_elapsed = 1e-3 * sum(marks[i].elapsed_time(marks[i + 1]) for i in range(roof_steps)) if sliced else elapsed

    # roofline of the dominant kernel (the bf16 MFMA GEMM) from the in-run event pairs
    def agg(kind):
        ev = [(e0.elapsed_time(e1) * 1e-3, w) for k, e0, e1, w, *_ in prof if k == kind]
        return sum(t for t, _ in ev), sum(w for _, w in ev), len(ev)
    tg, fl, ng = agg("gemm")
    # algorithmic bytes of the GEMM launches (operands once + output once; bf16 operands, output/residual in their dtype is
    # not known here -> bf16 output assumed, a lower bound) and the PMC-measured L2<->fabric traffic of the same launches
    gemm_alg = sum(2.0 * (m * k + n * k + m * n) for rec in prof if rec[0] == "gemm" and len(rec) > 4 for (m, n, k) in [rec[4]])
    # roofline.traffic comes from PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over THIS command: a profiler cannot run inside
    # the timed region); the newest committed summary is used and named, with the commit it was measured at
    pmc, pmc_why = None, None
    pmc_files = sorted((ROOT / "profiles").glob("r*_pmc_headline_gemm.json"))
    if args.workload == "idefics9b_32shot_bs8" and not args.no_hooks and pmc_files:
        pmc_file = pmc_files[-1]
        pmc = json.loads(pmc_file.read_text())
        have, want = pmc.get("gemm_src_sha16"), gemm_src_sha16()
        if have != want:                                                  # measured on other kernels than the ones that just ran
            pmc_why = (f"null: the newest committed PMC summary (profiles/{pmc_file.name}, commit {pmc.get('commit', '?')}) was measured on GEMM "
                       f"sources {have or 'without a recorded hash'}; this tree's are {want} - re-run the PMC passes (tools/r04_profile.sh <commit> partA) and commit profiles/rNN_pmc_headline_gemm.json")
            pmc = None
    ti, by, ni = agg("inject")
    fq = {"total": fl / max(roof_steps, 1) / B} if is2 else flops_per_question(arch, S, n_img)   # Idefics2: GEMM flops as launched
    res = {
        "metric": (f"VQA questions/sec (whole node), Idefics2-8B-base {'32' if n_img > 2 else '1'}-shot L-ICV training micro-batch" if is2 and training else
                   f"VQA questions/sec (whole node), Idefics2-8B-base {'32' if n_img > 2 else '1'}-shot ICV forward" if is2 else
                   "VQA questions/sec (whole node), Idefics-9B 32-shot L-ICV training micro-batch" if training else
                   "VQA questions/sec (whole node), Idefics-9B hooked generate (query-only prompt, 3 beams, 5 new tokens)" if generating else
                   "VQA questions/sec (whole node), Idefics-9B 32-shot ICV forward"),
        "value": qps, "unit": "questions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_step_median": median_ms, "value_at_median": B * world / (median_ms * 1e-3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp8 e4m3 GEMM operands in the text stack and the SigLIP tower (v_mfma_f32_16x16x128_f8f6f4, fp32 accumulate), bf16 elsewhere" if "fp8" in args.workload else "bf16", "data": f"synthetic (random-init {preset} weights, seeded image+text batches)",
        "config": {"workload": args.workload, "arch": preset, "questions_per_gpu": B, "seq_len": S, "images_per_question": n_img,
                   "hooked_layers": 0 if args.no_hooks else arch.num_layers, "parallelism": f"dp{world}",
                   **({"train": "teacher fwd + student fwd/bwd + KL; accumulate 2; 1 all-reduce of 131 105 fp32 + AdamW per optimiser step"} if training else {})},
        "roofline": {"bound": "mfma", "kernel": ("gemm_fp8_flow64_k / gemm_bf16_flow64_k" if "fp8" in args.workload else "gemm_bf16_flow64_k") + " (256x256 tile per CU, four waves of 128x128, 256 AGPR accumulators, LDS-DMA ring of ten 16 KiB units, persistent K-tile stream, register-direct epilogues; N % 128 != 0: gemm_bf16_quad64_k / gemm_bf16_flow_k; M < 512: gemm_bf16_mid_k)", "achieved": fl / tg / 1e12 if tg else None,
                     "peak": mfma_peak, "unit": "TFLOP/s", "frac": (fl / tg / 1e12) / mfma_peak if tg else None,
                     "peak_note": ("dense fp8 MFMA peak (5 PFLOP/s): the text-stack and SigLIP projections run on v_mfma_f32_16x16x128_f8f6f4; the launches "
                                   "that stay bf16 (LM head, perceiver, modality projection, patch embedding) are in the same sum and priced against the same peak"
                                   if "fp8" in args.workload else "dense bf16 MFMA peak (2.5 PFLOP/s)"),
                     "traffic": pmc["traffic_bytes_per_launch"] / 1e9 if pmc else None, "traffic_unit": "GB per launch (average over the step's GEMM launches)",
                     "traffic_source": (f"profiles/{pmc_file.name} (measured at commit {pmc.get('commit', 'of round 1')}, GEMM sources {pmc.get('gemm_src_sha16')}, kernels {pmc.get('kernel')}): "
                                        "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, summed over the GEMM launches; "
                                        "FETCH_SIZE x2 (gfx950), counts Infinity-Cache hits as well as HBM") if pmc else pmc_why,
                     "algorithmic_GB_per_launch": gemm_alg / ng / 1e9 if ng else None,
                     "achieved_GBps_algorithmic": gemm_alg / tg / 1e9 if tg else None,
                     "launches_per_step": ng // max(roof_steps, 1),
                     "avg_launch_us": 1e6 * tg / ng if ng else None, "gemm_share_of_step": tg / roof_elapsed if roof_elapsed else None,
                     **({"measured_in": f"second timed region of {roof_steps} steps with the batch slices off (one stream, kernels do not overlap): "
                                        f"{overlapped['serial_pass']['ms_per_step_median']:.1f} ms per step there; 'value' is the sliced configuration",
                         "overlapped": overlapped} if sliced else {})},
        "whole_path": {"tflop_per_question": fq["total"] / 1e12, "achieved_tflops_per_gpu": fq["total"] * B * args.steps / elapsed / 1e12,
                       "frac_of_mfma_peak": fq["total"] * B * args.steps / elapsed / 1e12 / PEAK_BF16_TFLOPS},
    }
    res.update(dist_info)
    if generating or "student" in args.workload:
        # weight-streaming shapes (what the reference really hooks): every weight matrix is read once per pass and used
        # on 24-256 rows, so the bound of the STEP is the HBM stream of the weights; achieved = those bytes / the step's time
        from licv.roofline import cross_kv_weight_bytes, weight_bytes
        wb = weight_bytes(arch)
        passes = 5 if generating else 1                                  # prefill + 4 single-token steps for 5 new tokens
        alg = wb["vision"] + wb["perceiver"] + passes * wb["language"] - (passes - 1) * cross_kv_weight_bytes(arch)
        res["roofline"] = {"bound": "hbm", "kernel": "weight-streaming GEMM family over the whole step (gemm_bf16_skinny_k for M <= 32, gemm_bf16_mid_k for M <= 256, "
                           "split-K + skinny_finalize_k; native layer runner)", "achieved": alg / (ms_per_step * 1e-3) / 1e9,
                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                           "algorithmic_GB_per_step": alg / 1e9, "weight_GB": {k: v / 1e9 for k, v in wb.items()}, "weight_passes": passes,
                           "floor_ms_at_peak": alg / PEAK_HBM_GBS / 1e6,
                           "note": "bytes = bf16 weights touched per step (vision + perceiver once, language stack once per pass, the "
                                   "cross-attention K|V projections at the prefill only); activations and KV cache are < 1 % of it"}
    if training and cache_mode:
        # what left the step and what the caches hold.  FLOPs EXECUTED per question replace the uncached figure in whole_path.
        from licv.roofline import flops_per_question as _fpq
        full = flops_per_question(arch, S, n_img)
        stu_f = _fpq(arch, train_args[0]["input_ids"].shape[1], 1)
        executed = (full["total"] - full["vision"] - full["perceiver"] if cache_mode == "vision" else 0.0) + stu_f["total"] * 3 - (stu_f["vision"] + stu_f["perceiver"]) * 3
        vc, tc = trainer.vision_cache, trainer.teacher_cache
        res["cache"] = {"mode": cache_mode, "what": ("vision-feature cache: 100 % image hits in the timed steps (teacher 33 + student 1 images per question), teacher rows recomputed"
                                                     if cache_mode == "vision" else "vision-feature cache AND teacher answer-row cache: 100 % hits in the timed steps, no teacher forward"),
                        "vision_hits": vc.hits, "vision_misses": vc.misses, "vision_pool_bytes": int(vc.pool.numel() * 2) if vc.pool is not None else 0,
                        "vision_bytes_per_image": vc.rows * vc.dim * 2, "vision_images_resident": len(vc.order),
                        "teacher_hits": tc.hits if tc else None, "teacher_misses": tc.misses if tc else None,
                        "teacher_rows_resident": tc.rows if tc else None,
                        "teacher_bytes_per_row": int(next(iter(tc.store.values())).shape[1] * 2) if tc and tc.store else None,
                        "tflop_per_question_uncached": (full["total"] + 3 * stu_f["total"]) / 1e12, "tflop_per_question_executed": executed / 1e12}
        res["whole_path"] = {"tflop_per_question": executed / 1e12, "achieved_tflops_per_gpu": executed * B * args.steps / elapsed / 1e12,
                             "frac_of_mfma_peak": executed * B * args.steps / elapsed / 1e12 / PEAK_BF16_TFLOPS,
                             "note": "FLOPs executed with the caches warm (student forward + backward counted as 3 x its forward, text stack only)"}
        res["metric"] += f" (f3 caches warm: {cache_mode})"
    if training and trainer is not None and trainer.allreduce_events:
        ar = sorted(e0.elapsed_time(e1) for e0, e1 in trainer.allreduce_events)
        res["allreduce"] = {"per_optimizer_step_ms_median": ar[len(ar) // 2], "per_optimizer_step_ms_max": ar[-1], "optimizer_steps_timed": len(ar),
                            "bytes": 4 * (trainer.flat_p.numel() + 1), "what": "one flat fp32 buffer [alpha.grad | icv.grad | kl], SUM then / world"}
    if ni:
        res["hook_kernel"] = {"bound": "hbm", "kernel": "inject_renorm_fwd_k (+fused RMSNorm)", "achieved": by / ti / 1e9,
                              "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": by / ti / 1e9 / PEAK_HBM_GBS,
                              "launches_per_step": ni // max(roof_steps, 1), "avg_launch_us": 1e6 * ti / ni,
                              "algorithmic_MB_per_question": by / max(roof_steps, 1) / B / 1e6}
    if want_g0:
        del eng
        torch.cuda.empty_cache()
        g0 = gpu_unfused_baseline(arch, sd, batch, (alpha.unsqueeze(-1) * icv), layers, B)
        g0["native_over_unfused"] = qps / g0["value"]
        res["gpu_unfused_baseline"] = g0
        res["vs_baseline_note"] = ("null: BASELINE.md holds no published number for this metric; the measured 'reference on GPU' denominator "
                                   "(row G0) is gpu_unfused_baseline, native/unfused = %.2fx" % g0["native_over_unfused"])
        del sd
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = (cpu_baseline_idefics2(arch, S, n_img, IDEFICS2_IMAGE[preset]) if is2 else
                               cpu_baseline_generate(arch, S, n_img) if generating else cpu_baseline(arch, S, n_img))
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
