"""IdeficsEngine.batch_streams: the batch of questions cut in slices that run on HIP streams of their own (the tail round of one
slice's GEMMs is filled by the other slice's workgroups).  Rows of a batch are independent, so the logits must be BIT-IDENTICAL to
the single-stream forward — even and odd batch sizes, ragged lengths, hooks on and off, the native runner and the Python layer loop
(with split-K off library-wide: a slice has fewer rows, and a GEMM that goes split-K for the slice but not for the whole batch adds
its partial sums in another order — the same caveat as property P2)."""
import pytest
import torch

from licv.config import IDEFICS_MID
from licv.synthetic import synth_idefics_weights, synth_vqa_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B,parts,use_runner", [(4, 2, True), (5, 2, False), (6, 3, True)])
def test_sliced_forward_is_bit_identical(B, parts, use_runner):
    from licv import ops
    from licv.idefics_engine import IdeficsEngine, IdeficsWeights
    arch = IDEFICS_MID
    S, N = 900 if B == 5 else 1100, 3
    eng = IdeficsEngine(IdeficsWeights(synth_idefics_weights(arch, seed=21, dtype=torch.float32), arch, DEV), use_runner=use_runner)
    batch = {k: v.to(DEV) for k, v in synth_vqa_batch(arch, B, S, N, seed=22, min_len=S - 300, dtype=torch.bfloat16).items()}
    icv = (torch.randn(1, arch.num_layers, arch.hidden_size, generator=torch.Generator().manual_seed(23)) * 0.05).to(DEV)
    alpha = torch.full((1, arch.num_layers), 0.2, device=DEV)
    layers = list(range(arch.num_layers))
    try:
        ops.set_splitk(False)
        for kw in ({}, dict(icv=icv, alpha=alpha, hook_layers=layers)):
            eng.batch_streams = 1
            whole = eng.forward(**batch, **kw)
            eng.batch_streams = parts
            assert B >= 2 * parts and B * S >= parts * 2048, "shape must take the sliced path"
            sliced = eng.forward(**batch, **kw)
            torch.cuda.synchronize()
            assert sliced.shape == whole.shape and torch.equal(sliced, whole), f"hooks {list(kw)}"
            assert len(eng._side_streams) >= parts
    finally:
        ops.set_splitk(True)


def test_sliced_forward_not_taken_when_something_is_captured():
    from licv.idefics_engine import IdeficsEngine, IdeficsWeights
    arch = IDEFICS_MID
    eng = IdeficsEngine(IdeficsWeights(synth_idefics_weights(arch, seed=21, dtype=torch.float32), arch, DEV))
    batch = {k: v.to(DEV) for k, v in synth_vqa_batch(arch, 4, 1100, 2, seed=24, dtype=torch.bfloat16).items()}
    cap = {}
    eng.forward(**batch, capture=cap)
    assert not eng._side_streams and "final_norm" in cap


def test_sliced_forward_with_selected_rows():
    """logits_rows (the trainer's answer rows, ascending): each slice computes the rows of its own questions; rows in one slice only,
    and rows out of order (one plain pass) are handled."""
    from licv import ops
    from licv.idefics_engine import IdeficsEngine, IdeficsWeights
    arch = IDEFICS_MID
    eng = IdeficsEngine(IdeficsWeights(synth_idefics_weights(arch, seed=21, dtype=torch.float32), arch, DEV))
    B, S = 4, 1100
    batch = {k: v.to(DEV) for k, v in synth_vqa_batch(arch, B, S, 2, seed=24, dtype=torch.bfloat16).items()}
    try:
        ops.set_splitk(False)
        for rows in (torch.tensor([5, 1099, 1100, 2300, 3299, 3300, 4399], device=DEV), torch.tensor([2200, 2201, 4000], device=DEV),
                     torch.tensor([3000, 7, 1500], device=DEV)):
            eng.batch_streams = 1
            whole = eng.forward(**batch, logits_rows=rows)
            eng.batch_streams = 2
            sliced = eng.forward(**batch, logits_rows=rows)
            torch.cuda.synchronize()
            assert sliced.shape == whole.shape and torch.equal(sliced, whole), rows.tolist()
    finally:
        ops.set_splitk(True)


def test_idefics2_sliced_forward_is_bit_identical_and_keeps_host_flags_cached():
    """Idefics2Engine.batch_streams: same property; the slices are cached view objects, so the identity-keyed host flags (real-image
    count, <image>-token count) are read back once per slice and not again on the second forward."""
    from licv import ops
    from licv.config import IDEFICS2_MID
    from licv.idefics2_engine import Idefics2Engine, Idefics2Weights
    from licv.synthetic import synth_idefics2_weights, synth_vqa_batch_idefics2
    arch = IDEFICS2_MID
    eng = Idefics2Engine(Idefics2Weights(synth_idefics2_weights(arch, seed=31, dtype=torch.float32), arch, DEV))
    B, S, N = 4, 1100, 3
    batch = synth_vqa_batch_idefics2(arch, B, S, N, 2 * arch.v_patch * 3, 2 * arch.v_patch * 4, seed=32, dtype=torch.bfloat16, device=DEV)
    icv = (torch.randn(1, arch.num_layers, arch.hidden_size, generator=torch.Generator().manual_seed(33)) * 0.05).to(DEV)
    layers = list(range(arch.num_layers))
    try:
        ops.set_splitk(False)
        eng.batch_streams = 1
        whole = eng.forward(**batch, icv=icv, hook_layers=layers)
        eng.batch_streams = 2
        sliced = eng.forward(**batch, icv=icv, hook_layers=layers)
        n_entries = len(eng._flags._entries)
        again = eng.forward(**batch, icv=icv, hook_layers=layers)
        torch.cuda.synchronize()
        assert torch.equal(sliced, whole) and torch.equal(again, whole)
        assert len(eng._flags._entries) == n_entries, "the second sliced forward must hit the host-flag cache"
    finally:
        ops.set_splitk(True)
