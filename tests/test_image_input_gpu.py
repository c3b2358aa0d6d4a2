"""Image input of the hot path (second half): uint8 HWC -> normalised bf16 CHW on the device and the pinned,
double-buffered feeder, against fixture g17 (HF's IdeficsImageProcessorPil; transformers.image_transforms rescale / normalize +
the Idefics2 padding rule).  The reference hands the model float32 that the model casts to bf16: the kernel must return exactly
bf16(reference float32) — a byte has 256 values and the kernel tabulates the reference's own IEEE operations."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = torch.from_numpy


def test_preprocess_kernel_equals_reference_rounded_to_bf16(golden):
    from licv import frontend
    z = golden("g17_image_preprocess")
    u8 = T(z["idefics_u8"]).to(DEV)
    pv, m = frontend.preprocess_images(u8, tuple(z["idefics_mean"]), tuple(z["idefics_std"]), float(z["idefics_rescale"]))
    want = T(z["idefics_f32"]).to(torch.bfloat16)
    assert m is None and pv.shape == want.shape and pv.dtype == torch.bfloat16
    assert torch.equal(pv.cpu(), want), "bf16 pixel_values differ from bf16(HF IdeficsImageProcessorPil output)"
    # every byte value occurs in every channel somewhere in the fixture: the whole table is covered
    assert all(len(np.unique(z["idefics_u8"][..., c])) == 256 for c in range(3))
    # Idefics2: ragged images inside a padded frame, one missing image
    B, N, H, W = z["idefics2_u8"].shape[:4]
    u2 = T(z["idefics2_u8"]).reshape(B * N, H, W, 3).contiguous().to(DEV)
    hw = T(z["idefics2_hw"]).reshape(B * N, 2).to(DEV)
    pv2, m2 = frontend.preprocess_images(u2, frontend.IDEFICS2_MEAN, frontend.IDEFICS2_STD, 1 / 255, valid_hw=hw, want_mask=True)
    assert torch.equal(pv2.cpu().view(B, N, 3, H, W), T(z["idefics2_f32"]).to(torch.bfloat16))
    assert torch.equal(m2.cpu().view(B, N, H, W), T(z["idefics2_mask"]).bool())
    assert not bool(m2.view(B, N, H, W)[1, 1].any()) and float(pv2.view(B, N, 3, H, W)[1, 1].abs().max()) == 0.0     # the missing image


def test_feeder_double_buffering_uniform_and_ragged(golden):
    """Five batches through two staging slots: every batch comes back as the kernel applied to ITS bytes (no slot is overwritten
    while its copy is in flight, no output is recycled before its reader is done), uniform and ragged."""
    from licv import frontend
    from licv.image_feeder import ImageFeeder
    rng = np.random.default_rng(5)
    H, W, n = 56, 56, 12
    f = ImageFeeder(DEV, n, H, W)
    batches = [rng.integers(0, 256, (n, H, W, 3)).astype(np.uint8) for _ in range(5)]
    tickets, outs = [], []
    t_prev = None
    for i, b in enumerate(batches):                       # submit batch i while batch i - 1 is "being computed on"
        t = f.submit(b)
        if t_prev is not None:
            pv, m = f.get(t_prev, 3, 4)
            outs.append(pv.clone())
            assert m is None and pv.shape == (3, 4, 3, H, W)
            f.release(t_prev)
        t_prev = t
    outs.append(f.get(t_prev)[0].clone())
    for b, o in zip(batches, outs):
        want, _ = frontend.preprocess_images(T(b).to(DEV))
        assert torch.equal(o.reshape(want.shape), want)
    # ragged, with mask (the Idefics2 form)
    z = golden("g17_image_preprocess")
    B, N, Hm, Wm = z["idefics2_u8"].shape[:4]
    f2 = ImageFeeder(DEV, B * N, Hm, Wm, frontend.IDEFICS2_MEAN, frontend.IDEFICS2_STD, with_mask=True)
    imgs = []
    for b_ in range(B):
        for n_ in range(N):
            h, w = z["idefics2_hw"][b_, n_]
            imgs.append(None if h == 0 else np.ascontiguousarray(z["idefics2_u8"][b_, n_, :h, :w]))
    for _ in range(3):                                    # slot reuse with stale bytes from the previous round in the pinned buffer
        pv, m = f2.get(f2.submit(imgs), B, N)
        assert torch.equal(pv.cpu(), T(z["idefics2_f32"]).to(torch.bfloat16))
        assert torch.equal(m.cpu(), T(z["idefics2_mask"]).bool())


def test_interface_feeder_feeds_the_forward(golden):
    """The interface's feeder produces the pixel_values the engine takes: forward on feeder output == forward on the same values
    uploaded as a float tensor."""
    from licv.config import IDEFICS_TINY
    from licv.synthetic import synth_idefics_weights, synth_vqa_batch
    from lmm_icl_interface import IdeficsInterface
    arch = IDEFICS_TINY
    iface = IdeficsInterface(state_dict=synth_idefics_weights(arch, seed=3, dtype=torch.float32), arch=arch, device=DEV)
    batch = synth_vqa_batch(arch, 2, 24, 2, seed=4, min_len=20, dtype=torch.float32)
    side = arch.v_image
    u8 = np.random.default_rng(6).integers(0, 256, (4, side, side, 3)).astype(np.uint8)
    feeder = iface.image_feeder(4)
    pv, _ = feeder.get(feeder.submit(u8), 2, 2)
    from oracle import frontend_ref as F
    from licv import frontend
    ref, _ = F.preprocess_images(u8, frontend.IDEFICS_MEAN, frontend.IDEFICS_STD)
    assert torch.equal(pv.cpu().view(4, 3, side, side), T(ref).to(torch.bfloat16))
    kw = {k: v.to(DEV) for k, v in batch.items() if k != "pixel_values"}
    a = iface(**kw, pixel_values=pv)["logits"]
    b = iface(**kw, pixel_values=T(ref).to(torch.bfloat16).view(2, 2, 3, side, side).to(DEV))["logits"]
    assert torch.equal(a, b)


def test_feeder_without_release_cannot_overwrite_a_batch_that_is_still_being_read():
    """get -> (slow reader on the main stream) -> submit x depth WITHOUT release(): the side stream's copy + preprocess kernel of the
    batch that reuses the slot must wait for the reader (the safe default; release() only moves the point earlier)."""
    import numpy as np
    from licv.image_feeder import ImageFeeder
    from licv import frontend
    rng = np.random.default_rng(11)
    H = W = 224
    n = 64
    f = ImageFeeder(DEV, n, H, W)
    batches = [rng.integers(0, 256, (n, H, W, 3)).astype(np.uint8) for _ in range(4)]
    want = [frontend.preprocess_images(T(b).to(DEV))[0] for b in batches]
    spin = torch.randn(4096, 4096, device=DEV)
    torch.cuda.synchronize()
    t = f.submit(batches[0])
    sums = []
    for i in range(len(batches)):
        pv, _ = f.get(t)
        for _ in range(40):                              # keep the main stream busy for milliseconds BEFORE the reader is issued
            spin = (spin @ spin).clamp_(-1, 1)
        sums.append((pv.float() - want[i].float()).abs().max())          # the reader: issued late, long after get()
        nxt = None
        for j in range(f.depth):                         # within `depth` submits the slot being read is reused; never released
            tj = f.submit(batches[(i + 1 + j) % len(batches)])
            nxt = tj if j == 0 else nxt
        t = nxt                                          # batch i + 1
    torch.cuda.synchronize()
    assert all(float(s) == 0.0 for s in sums), [float(s) for s in sums]
