"""Batch assembly for L-ICV training: the producer of ``VQAICVModule.forward``'s inputs.

Mirrors the output contract of ``collator_data`` (ref:icv_src/icv_datamodule.py:73-130) — CPU-side integer work that
runs in DataLoader workers, outside the MI355X hot path.  Only the contract is restated; datasets, Lightning's
``LightningDataModule`` shell and the COCO loaders stay the reference's own (out of scope).

Per sample the dataset yields ``ice_prompt`` (the in-context demonstrations, a list of interleaved image/text items),
``query_prompt`` (query with its answer) and ``query_x`` (query without the answer).  Outputs:
  * ``query_inputs``      student batch  = tokenised query_prompt, right padded, EOS appended;
  * ``inputs``            teacher batch  = tokenised ice_prompt + query_prompt, right padded, EOS appended;
  * ``query_x_length``    #non-pad tokens of query_x                     -> first answer position in the student row;
  * ``in_context_length`` #non-pad tokens of ice + #non-pad-non-BOS of query_x -> first answer position in the teacher row.
``VQAICVModule.get_mask`` turns the two lengths into masks that select the SAME number of answer tokens in both rows.
"""
from __future__ import annotations

from functools import partial
from typing import Dict, List, Sequence

import torch


def _count(ids: torch.Tensor, *excluded: int) -> torch.Tensor:
    keep = torch.ones_like(ids, dtype=torch.bool)
    for tok in excluded:
        keep &= ids != tok
    return keep.sum(dim=1)


def collator_data(data_list: Sequence[dict], prompt_processor) -> Dict[str, object]:
    """data_list: samples with keys ``ice_prompt``, ``query_prompt``, ``query_x`` (each a prompt = list of items)."""
    ice = [d["ice_prompt"] for d in data_list]
    query = [d["query_prompt"] for d in data_list]
    query_x = [d["query_x"] for d in data_list]
    field = prompt_processor.input_ids_field
    tok = prompt_processor.tokenizer
    prep = partial(prompt_processor.prepare_input, padding=True, truncation=True)

    student = prep(query, add_eos_token=True)
    teacher = prep([i + q for i, q in zip(ice, query)], return_tensors="pt", add_eos_token=True)
    qx_ids = prep(query_x, return_tensors="pt")[field]
    ice_ids = prep(ice, return_tensors="pt")[field]
    return {
        "query_inputs": student,
        "inputs": teacher,
        "in_context_length": _count(ice_ids, tok.pad_token_id) + _count(qx_ids, tok.pad_token_id, tok.bos_token_id),
        "query_x_length": _count(qx_ids, tok.pad_token_id),
    }


class VQAICVDataModule:
    """Thin holder with the reference's constructor and ``collator_data`` attribute (ref:icv_src/icv_datamodule.py:12-27);
    building datasets needs the reference's dataset classes and data on disk, which are out of scope here."""

    def __init__(self, data_cfg, prompt_manager, prompt_processor) -> None:
        self.data_cfg, self.prompt_manager, self.prompt_processor = data_cfg, prompt_manager, prompt_processor
        self.prompt_processor.tokenizer.padding_side = "right"          # get_mask assumes right padding
        self.collator_data = partial(collator_data, prompt_processor=prompt_processor)

    def train_dataloader(self, train_ds, rank: int = 0, world: int = 1):
        """One process per GPU: each rank reads its own shard of the dataset (licv.trainer.shard_indices)."""
        from torch.utils.data import DataLoader, Subset
        from licv.trainer import shard_indices
        ds = Subset(train_ds, shard_indices(len(train_ds), rank, world)) if world > 1 else train_ds
        return DataLoader(ds, self.data_cfg.bs, num_workers=getattr(self.data_cfg, "num_workers", 0),
                          collate_fn=self.collator_data, pin_memory=True)
