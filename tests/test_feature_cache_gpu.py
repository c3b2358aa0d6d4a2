"""Reuse of ICV-independent work (ref:icv_src/icv_module.py:103-105 — the teacher never sees the ICV; shots
are re-drawn from a fixed pool, ref:icv_src/icv_datasets/vqa_dataset.py:90-98).  Cached results must BE the uncached ones:
  * perceiver features served from the per-image cache == encode_images on the same images, bit for bit, whatever subset of the
    batch was already cached (the vision tower is row-independent: no kernel on that side changes its summation order with M);
  * a training micro-batch through the caches (cold, then warm) gives the same loss and the same gradients, bit for bit, as the
    uncached trainer; the warm pass runs no vision tower and no teacher forward at all (counted)."""
import pytest
import torch

from licv.config import IDEFICS_MID
from licv.synthetic import synth_idefics_weights, synth_vqa_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_vision_feature_cache_is_bit_identical_to_encode_images():
    from licv.feature_cache import VisionFeatureCache
    from licv.idefics_engine import IdeficsEngine, IdeficsWeights
    arch = IDEFICS_MID
    eng = IdeficsEngine(IdeficsWeights(synth_idefics_weights(arch, seed=5, dtype=torch.float32), arch, DEV))
    g = torch.Generator().manual_seed(6)
    pool = torch.randn(12, 3, arch.v_image, arch.v_image, generator=g).to(torch.bfloat16).to(DEV)      # 12 distinct images, ids 100..111
    cache = VisionFeatureCache(eng, capacity_images=10)
    for step, ids in enumerate(([[100, 101, 102], [103, 101, 100]], [[104, 100, 105], [101, 106, 102]], [[107, 108, 109], [110, 111, 100]])):
        pv = torch.stack([torch.stack([pool[i - 100] for i in row]) for row in ids])
        want = eng.encode_images(pv)
        got = cache.encode(pv, ids)
        assert got.shape == want.shape and torch.equal(got, want), f"step {step}"
    assert cache.hits == 6 and cache.misses == 12 and len(cache.slot_of) == 10          # capacity 10: the two oldest ids NOT in use were evicted
    assert 100 in cache.slot_of and 101 not in cache.slot_of and 102 not in cache.slot_of


def _trainer():
    from icv_src.icv_module import VQAICVModule
    from licv.trainer import ICVTrainer
    from lmm_icl_interface import IdeficsInterface
    arch = IDEFICS_MID
    sd = synth_idefics_weights(arch, seed=15, dtype=torch.float32)
    iface = IdeficsInterface(state_dict=sd, arch=arch, device=DEV)
    mod_cfg = dict(hard_loss_weight=0.0, only_hard_loss=False, kl_eps=1e-6, init_temperature=1.0, learnable_t=False, decay_ratio=-1,
                   decay_per_step=-1, min_tmeprature=1.0, alpha_lr=1e-2, icv_lr=1e-3, weight_decay=1e-3, warm_steps=0,
                   icv_encoder=dict(use_sigmoid=True, alpha_learnable=True, alpha_init_value=0.3))
    lmm_cfg = dict(intervention_layer=-1, layer_format="model.model.layers.<LAYER_NUM>", total_layers=arch.num_layers, hidden_size=arch.hidden_size)
    torch.manual_seed(426)
    mod = VQAICVModule(iface, mod_cfg, lmm_cfg).to(DEV)
    return arch, mod, ICVTrainer(mod, total_steps=10, accumulate_grad_batches=1, grad_clip=1.0)


def test_trainer_through_caches_equals_uncached_trainer_bitwise():
    arch, mod_a, tr_a = _trainer()
    _, mod_b, tr_b = _trainer()
    tr_b.enable_caches(vision_images=64, teacher_rows=64)
    B, ans = 3, 3
    tea = synth_vqa_batch(arch, B, 60, 4, seed=21, min_len=52, dtype=torch.bfloat16)
    stu = synth_vqa_batch(arch, B, 20, 1, seed=22, min_len=16, dtype=torch.bfloat16)
    stu["pixel_values"] = tea["pixel_values"][:, -1:].clone()                            # the query image is the teacher's last image
    tl, sl = tea["attention_mask"].sum(1), stu["attention_mask"].sum(1)
    for b in range(B):
        stu["input_ids"][b, sl[b] - ans: sl[b]] = tea["input_ids"][b, tl[b] - ans: tl[b]]
    to = lambda d: {k: v.to(DEV) for k, v in d.items()}
    args = (to(stu), to(tea), (sl - ans).to(DEV), (tl - ans).to(DEV))
    image_ids = dict(inputs=[[10 * b + k for k in range(4)] for b in range(B)], query_inputs=[[10 * b + 3] for b in range(B)])
    keys = [("q", b, tuple(image_ids["inputs"][b])) for b in range(B)]
    kl_a = tr_a.loss_and_backward(*args)
    ga = {n: getattr(mod_a.icv_encoder, n).grad.clone() for n in ("icv", "alpha")}
    calls = []
    eng = mod_b.interface.engine
    enc0, fwd0 = eng.encode_images, eng.forward
    eng.encode_images = lambda *a, **k: (calls.append("vision"), enc0(*a, **k))[1]
    eng.forward = lambda *a, **k: (calls.append("teacher"), fwd0(*a, **k))[1]
    for phase in ("cold", "warm"):
        for n in ("icv", "alpha"):
            getattr(mod_b.icv_encoder, n).grad = None
        calls.clear()
        kl_b = tr_b.loss_and_backward(*args, image_ids=image_ids, teacher_keys=keys)
        assert float(kl_b) == float(kl_a), phase
        for n in ("icv", "alpha"):
            assert torch.equal(getattr(mod_b.icv_encoder, n).grad, ga[n]), f"{phase}: {n}.grad differs from the uncached trainer"
        if phase == "cold":
            # the 3 query images first (student), then only the 9 teacher images not yet seen; one teacher forward
            assert calls.count("vision") == 2 and calls.count("teacher") == 1
        else:
            assert calls == [], f"warm pass still ran {calls}"
    assert tr_b.vision_cache.misses == 12 and tr_b.teacher_cache.hits == B
