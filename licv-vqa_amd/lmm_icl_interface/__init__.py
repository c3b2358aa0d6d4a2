"""Native stand-in for the reference's ``lmm_icl_interface`` dependency (call sites: ref:utils.py:9-14,31-80,
ref:icv_src/icv_module.py:28-30,137-147, ref:icv_src/icv_datamodule.py:22,80-124, ref:inference.py:273-320).

Only the model side is MI355X work: ``IdeficsInterface`` wraps the native engine and exposes the attributes
the reference touches (``.model .tokenizer .processor .device .input_ids_field_name``, ``requires_grad_``,
``__call__(**inputs)`` with ``["logits"]``/``["loss"]``, ``generate(**inputs, **gen_kwargs)``).  Prompt
templating and tokenisation need tokenizer files that do not exist offline; ``LMMPromptManager`` /
``LMMPromptProcessor`` are thin and delegate to a transformers processor when one is supplied.
"""
from .interface import Idefics2Interface, IdeficsInterface, LMMInterface, LMMOutput  # noqa: F401
from .prompt import LMMPromptManager, LMMPromptProcessor  # noqa: F401
