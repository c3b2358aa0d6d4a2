"""Host-side logic that needs no GPU: the Idefics2 host-flag cache under torch.inference_mode (ref:inference.py:246,300,324 run
every entry point inside it), the teacher-logit cache's row budget, and the `<image>` token id resolution of the interface."""
import types

import pytest
import torch


def test_host_flags_step_aside_for_inference_tensors():
    from licv.idefics2_engine import _HostFlags
    f = _HostFlags()
    a, b = torch.zeros(3), torch.ones(2)
    assert f.get(a, b) is None
    f.put((1, 2), a, b)
    assert f.get(a, b) == (1, 2)
    a.add_(1)                                          # a visible write bumps the version counter: miss
    assert f.get(a, b) is None
    with torch.inference_mode():
        c = torch.zeros(3)                             # no version counter to read
        assert c.is_inference()
        assert f.get(c, None) is None
        assert f.put((7, 7), c, None) == (7, 7)        # returned, not stored
        assert f.get(c, None) is None
        assert f.get(a, c) is None                     # one inference tensor among the keys is enough
    f.put((3, 4), a, None)
    assert f.get(a, None) == (3, 4)
    f.clear()
    assert f.get(a, None) is None


def test_slice_views_are_not_cached_for_inference_tensors():
    from licv.idefics2_engine import Idefics2Engine
    eng = Idefics2Engine.__new__(Idefics2Engine)
    eng._slice_views = {}
    t = torch.arange(8)
    v1, v2 = eng._views(t, (0, 4, 8)), eng._views(t, (0, 4, 8))
    assert v1[0] is v2[0]                              # ordinary tensors: the same view objects (identity-keyed flag cache)
    with torch.inference_mode():
        u = torch.arange(8)
        w1, w2 = eng._views(u, (0, 4, 8)), eng._views(u, (0, 4, 8))
        assert w1[0] is not w2[0] and torch.equal(w1[1], u[4:])
    assert id(u) not in eng._slice_views


def test_teacher_logit_cache_is_bounded_by_rows():
    from licv.feature_cache import TeacherLogitCache
    c = TeacherLogitCache(capacity_rows=10)
    for k in range(5):
        c.insert(k, torch.full((3, 4), float(k)))     # 3 rows each: only three questions fit in 10 rows
    assert c.rows == 9 and set(c.store) == {2, 3, 4}
    c.insert(3, torch.zeros(1, 4))                     # re-insert replaces, row count follows
    assert c.rows == 7
    c.insert("big", torch.zeros(11, 4))                # larger than the whole budget: skipped
    assert "big" not in c.store and c.rows == 7
    got, miss = c.lookup([2, 9])
    assert miss == [1] and got[0] is not None


def test_image_token_id_resolution():
    from licv.config import IDEFICS_TINY
    from lmm_icl_interface.interface import IdeficsInterface
    arch = IDEFICS_TINY
    res = IdeficsInterface._resolve_image_token_id
    fallback = arch.vocab_size + 1
    assert res(types.SimpleNamespace(), arch) == fallback                               # no convert_tokens_to_ids at all
    tok = types.SimpleNamespace(convert_tokens_to_ids=lambda t: arch.vocab_size, unk_token_id=0)
    assert res(tok, arch) == arch.vocab_size
    unk = types.SimpleNamespace(convert_tokens_to_ids=lambda t: 0, unk_token_id=0)       # tokenizer answers <unk>
    assert res(unk, arch) == fallback
    none = types.SimpleNamespace(convert_tokens_to_ids=lambda t: None, unk_token_id=0)
    assert res(none, arch) == fallback
    bad = types.SimpleNamespace(convert_tokens_to_ids=lambda t: 10 ** 6, unk_token_id=0)  # outside the embedding table
    with pytest.raises(ValueError):
        res(bad, arch)
    assert res(types.SimpleNamespace(), arch.with_(additional_vocab_size=0)) is None                  # no such token at all
