"""Prompt side of the interface (CPU string work; outside the MI355X hot path, kept minimal).

Call shapes follow ref:utils.py:32-38 and ref:icv_src/icv_datamodule.py:80-103 / ref:inference.py:273-275."""
from __future__ import annotations

from typing import Dict, List, Optional


class LMMPromptManager:
    def __init__(self, prompt_template: str, column_token_map: Dict[str, str], label_field: str, sep_token: str = "",
                 query_prompt_template: Optional[str] = None):
        self.prompt_template, self.column_token_map = prompt_template, dict(column_token_map)
        self.label_field, self.sep_token = label_field, sep_token
        self.query_prompt_template = query_prompt_template or prompt_template

    def _fill(self, template: str, item: dict, with_label: bool) -> str:
        text = template
        for column, token in self.column_token_map.items():
            value = "" if (column == self.label_field and not with_label) else str(item.get(column, ""))
            text = text.replace(token, value)
        return text

    def gen_ice_text_with_label(self, item: dict, add_sep_token: bool = False) -> str:
        return self._fill(self.prompt_template, item, True) + (self.sep_token if add_sep_token else "")

    def gen_query_text_with_label(self, item: dict) -> str:
        return self._fill(self.query_prompt_template, item, True)

    def gen_query_text_without_label(self, item: dict) -> str:
        return self._fill(self.query_prompt_template, item, False).rstrip()


class LMMPromptProcessor:
    """``prepare_input(batch_prompts, padding=, truncation=, add_eos_token=, return_tensors=)`` over a transformers
    Idefics processor (needs tokenizer files; not available offline — the bench/tests use ``licv.synthetic``)."""

    def __init__(self, processor, input_ids_field: str = "input_ids"):
        if processor is None:
            raise ValueError("LMMPromptProcessor needs a transformers processor (tokenizer files are not bundled)")
        self.processor, self.tokenizer, self.input_ids_field = processor, processor.tokenizer, input_ids_field

    def prepare_input(self, batch_prompts: List[list], padding=True, truncation=None, add_eos_token=False, return_tensors="pt", **kw):
        return self.processor(batch_prompts, padding=padding, truncation=truncation, add_end_of_utterance_token=False,
                              add_eos_token=add_eos_token, return_tensors=return_tensors, **kw)
