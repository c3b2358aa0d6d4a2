"""Reuse of what does not depend on the ICV.

The teacher forward (ref:icv_src/icv_module.py:103-105) and the whole vision side of BOTH passes are functions of the frozen
LMM and the inputs only — never of ``icv`` / ``alpha``.  Shots are re-drawn for every query (ref:icv_src/icv_datasets/
vqa_dataset.py:90-98) but from a fixed pool (8000 queries in the published recipe, ref:README.md:126-139), so the same images
come back step after step and epoch after epoch.  Two caches, both keyed by ids the DATASET already has on the host (no device
read-back is needed to consult them):

  * ``VisionFeatureCache``  — perceiver outputs per image id: (image_seq_len, E) bf16 = 160 KiB per image at Idefics-9B
    (64 x 1280), kept in one preallocated device pool (100 k images = 16 GB of the 288 GB).  A batch runs the vision tower
    only on the images it has not seen; HF itself accepts precomputed features (hf:idefics/modeling_idefics.py:942-943,
    :996-1007), the engine's ``forward(image_states=...)`` is that entry.  This removes ~47 % of the teacher's FLOPs when warm.
  * ``TeacherLogitCache``   — the teacher's answer-row logits per (query, shots) key: hit rows skip the teacher forward
    altogether (a full-batch hit skips the launch sequence; partial hits run the missing rows as a smaller batch).

Both return exactly what the uncached computation produced when it was first run (bf16 tensors copied, not recomputed).
"""
from __future__ import annotations

from typing import Dict, Hashable, List, Optional, Sequence

import torch


class VisionFeatureCache:
    def __init__(self, engine, capacity_images: int = 4096):
        a = engine.arch
        self.engine = engine
        self.rows, self.dim = a.image_seq_len, (a.v_embed if hasattr(a, "v_embed") else a.hidden_size)
        self.capacity = int(capacity_images)
        self.pool: Optional[torch.Tensor] = None            # (capacity, rows, dim) bf16, allocated on first use
        self.slot_of: Dict[Hashable, int] = {}
        self.order: List[Hashable] = []                     # insertion order for FIFO eviction
        self.hits = self.misses = 0

    def _alloc(self, device):
        if self.pool is None:
            self.pool = torch.empty((self.capacity, self.rows, self.dim), dtype=torch.bfloat16, device=device)

    def encode(self, pixel_values: torch.Tensor, image_ids: Sequence[Sequence[Hashable]]) -> torch.Tensor:
        """pixel_values (B, N, 3, H, W); image_ids: B lists of N hashable ids (host side).  Returns image_states
        (B, N * image_seq_len, E) bf16 = what ``IdeficsEngine.encode_images(pixel_values)`` returns, computing only unseen images."""
        B, N = pixel_values.shape[:2]
        assert len(image_ids) == B and all(len(r) == N for r in image_ids), "image_ids must be B lists of N ids"
        dev = self.engine.w.device
        self._alloc(dev)
        flat = [i for row in image_ids for i in row]
        missing, seen = [], set()
        for pos, key in enumerate(flat):
            if key not in self.slot_of and key not in seen:
                missing.append(pos)
                seen.add(key)
        self.hits += len(flat) - len(missing)
        self.misses += len(missing)
        if missing:
            idx = torch.tensor(missing, device=dev)
            pv = pixel_values.to(dev).reshape(B * N, *pixel_values.shape[2:]).index_select(0, idx).unsqueeze(0)     # (1, n_miss, 3, H, W)
            feats = self.engine.encode_images(pv).view(len(missing), self.rows, self.dim)
            slots, in_use = [], set(flat)
            assert len(in_use) <= self.capacity, "cache capacity smaller than the distinct images of one batch"
            for pos in missing:
                key = flat[pos]
                if len(self.order) >= self.capacity:                     # FIFO eviction, skipping ids this batch is about to read
                    j = next(i for i, k in enumerate(self.order) if k not in in_use)
                    slot = self.slot_of.pop(self.order.pop(j))
                else:
                    slot = len(self.order)
                self.slot_of[key] = slot
                self.order.append(key)
                slots.append(slot)
            self.pool.index_copy_(0, torch.tensor(slots, device=dev), feats)
        gather = torch.tensor([self.slot_of[k] for k in flat], device=dev)
        return self.pool.index_select(0, gather).view(B, N * self.rows, self.dim)


class TeacherLogitCache:
    """Teacher answer-row logits per question, bounded by the total number of cached ROWS (each row is V bf16 values on the
    device: 64 KB at V = 32002, so the default 65536 rows are ~4.2 GB of HBM); oldest questions are evicted first."""

    def __init__(self, capacity_rows: int = 65536):
        self.capacity = int(capacity_rows)
        self.store: Dict[Hashable, torch.Tensor] = {}
        self.rows = 0
        self.hits = self.misses = 0

    def lookup(self, keys: Sequence[Hashable]):
        """-> (list of cached (n_ans, V) tensors or None per question, indices of the questions that must be computed)."""
        got = [self.store.get(k) for k in keys]
        miss = [i for i, g in enumerate(got) if g is None]
        self.hits += len(keys) - len(miss)
        self.misses += len(miss)
        return got, miss

    def insert(self, key: Hashable, rows: torch.Tensor):
        old = self.store.pop(key, None)
        if old is not None:
            self.rows -= old.shape[0]
        n = int(rows.shape[0])
        if n > self.capacity:
            return                                           # one question larger than the whole budget: not cached
        while self.store and self.rows + n > self.capacity:
            self.rows -= self.store.pop(next(iter(self.store))).shape[0]
        self.store[key] = rows.clone()
        self.rows += n
