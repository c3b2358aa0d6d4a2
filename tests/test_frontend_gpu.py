"""Device-side front-end kernels (csrc/frontend.hip) against what the HF code itself produced (fixture g13:
transformers' image_attention_mask functions; patch mask / padding-image removal / NaViT position ids captured inside HF
Idefics2) — integer work, so everything is bit-exact — plus the edge cases the rules have (one-token rows, more images than
mask columns, text before the first image, end-of-document runs, a 70 x 70 patch grid, an all-zero padding image)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda"


def test_idefics_image_attention_mask_kernel_matches_hf(golden):
    from licv import frontend
    from oracle import frontend_ref as F
    z = golden("g13_frontend")
    for tag in "abcd":
        img, eod, n = (int(v) for v in z[f"m_{tag}_cfg"])
        ids = T(z[f"m_{tag}_ids"])
        got = frontend.idefics_image_attention_mask(ids.to(DEV), img, eod, n)
        assert got.dtype == torch.int32 and torch.equal(got.cpu().long(), T(z[f"m_{tag}_mask"]).long()), tag
    # the headline shape: 8 rows x 800 tokens x 33 images, against the oracle's token-by-token loop
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, 60, (8, 800), generator=g)
    got = frontend.idefics_image_attention_mask(ids.to(DEV), 33, 2, 33, dtype=torch.long)
    assert torch.equal(got.cpu(), F.image_attention_mask(ids, 33, 2, 33))


def test_idefics2_patch_front_kernel_matches_hf(golden):
    from licv import frontend
    z = golden("g13_frontend")
    for tag, n_side in (("tiny", 4), ("mid", 6)):
        pv, pam = T(z[f"v_{tag}_pixel_values"]), T(z[f"v_{tag}_pixel_attention_mask"])
        B, N = pv.shape[:2]
        pvf = pv.reshape(B * N, *pv.shape[2:]).to(torch.bfloat16).contiguous()
        real, valid, pos = frontend.idefics2_patch_front(pvf.to(DEV), pam.reshape(B * N, *pam.shape[2:]).to(DEV), 14, n_side)
        keep = real.cpu().bool()
        assert int(keep.sum()) == int(z[f"v_{tag}_n_real"]) and not bool(keep.all())            # the fixture holds one padding image
        gold_mask = T(z[f"v_{tag}_patch_mask"])
        assert torch.equal(valid.cpu()[keep].bool(), gold_mask.view(gold_mask.shape[0], -1))
        assert torch.equal(pos.cpu()[keep], T(z[f"v_{tag}_position_ids"]))
    # the full 980 x 980 canvas (70 x 70 grid): ragged images incl. a one-patch-high strip; no mask given = everything attended
    sizes = [tuple(int(v) for v in s) for s in z["v_full_sizes"]]
    pam = torch.zeros(len(sizes), 980, 980, dtype=torch.bool)
    for i, (hh, ww) in enumerate(sizes):
        pam[i, :hh, :ww] = True
    pix = torch.zeros(len(sizes), 3, 980, 980, dtype=torch.bfloat16)
    pix[1, 2, 100, 7] = -0.0                                          # a negative zero is still a zero pixel
    pix[2, 0, 3, 3] = 1e-3
    real, valid, pos = frontend.idefics2_patch_front(pix.to(DEV), pam.to(DEV), 14, 70)
    assert real.cpu().tolist() == [0, 0, 1, 0]
    assert torch.equal(pos.cpu(), T(z["v_full_position_ids"]).long())
    _, valid_all, pos_all = frontend.idefics2_patch_front(pix[:1].to(DEV), None, 14, 70)
    # (in bf16 the fractional coordinates of a 70-wide grid collide — HF's own ids are NOT arange(4900); fixture row 0 is that image)
    assert bool(valid_all.all()) and torch.equal(pos_all.cpu()[0], T(z["v_full_position_ids"]).long()[0])


def test_merge_image_rows_equals_masked_scatter():
    from licv import frontend
    g = torch.Generator().manual_seed(9)
    for M, dim, tok in ((37, 64, 5), (2900 * 8, 4096, 32001), (1, 8, 3)):
        ids = torch.randint(0, 8, (M,), generator=g)
        ids = torch.where(ids == 5, torch.full_like(ids, tok), ids)
        h = torch.randn(M, dim, generator=g).to(torch.bfloat16)
        k = int((ids == tok).sum())
        rows = torch.randn(max(k, 1), dim, generator=g).to(torch.bfloat16)[:k]
        ref = h.clone()
        if k:
            ref = h.masked_scatter((ids == tok).unsqueeze(-1), rows)                            # hf:idefics2/modeling_idefics2.py:810-814
        hd = h.to(DEV)
        cnt = frontend.merge_image_rows_(hd, ids.to(DEV), rows.to(DEV).contiguous() if k else torch.zeros(1, dim, dtype=torch.bfloat16, device=DEV), tok)
        assert int(cnt) == k and torch.equal(hd.cpu(), ref)


def test_interface_builds_the_image_mask_on_device_when_none_is_passed():
    """processor.prepare_input normally supplies image_attention_mask (ref:icv_src/icv_datamodule.py:80-124); without it the
    interface derives the same mask from input_ids on the device: identical logits."""
    from licv.config import IDEFICS_TINY
    from licv.synthetic import synth_idefics_weights, synth_vqa_batch
    from lmm_icl_interface import IdeficsInterface
    arch = IDEFICS_TINY
    iface = IdeficsInterface(state_dict=synth_idefics_weights(arch, seed=3, dtype=torch.float32), arch=arch, device=DEV)
    batch = {k: v.to(DEV) for k, v in synth_vqa_batch(arch, 3, 24, 2, seed=4, min_len=18, dtype=torch.bfloat16).items()}
    a = iface(**batch)["logits"]
    b = iface(**{k: v for k, v in batch.items() if k != "image_attention_mask"})["logits"]
    assert torch.equal(a, b)
