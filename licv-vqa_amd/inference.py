"""Mirror of the hot-path functions of the reference's `inference.py` (ref:inference.py:246-321): `generate_answers` — ICV
scaling, hooked generate, prompt strip, `batch_decode` — and the `icv_inference` batching loop around it, on the native
interfaces.  The Hydra `main`, dataset loading and the ICL baseline are out of scope; a maintainer who
runs the reference's own `inference.py` against this package (`PYTHONPATH=licv-vqa_amd`) gets the same two functions from the
reference file itself, driving the same `LearnableICVInterventionLMM.generate`."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch


@torch.inference_mode()
def generate_answers(inputs, model, processor, generate_kwargs, in_context_vector=None, alpha=None) -> List[str]:
    """ref:inference.py:299-321.  `model` = LearnableICVInterventionLMM; `inputs` as `processor.prepare_input` returns them
    (already on the device).  The vector handed to the hooks is alpha[..., None] * in_context_vector (1, L, H); the first
    `attention_mask.shape[1]` ids of every returned row are the (padded) prompt and are dropped before decoding."""
    icv = None
    if in_context_vector is not None:
        icv = alpha.unsqueeze(dim=-1) * in_context_vector
    out = model.generate(**inputs, **generate_kwargs, icv=icv)
    n_prompt = int(inputs["attention_mask"].shape[1])
    rows = out.tolist()
    return processor.tokenizer.batch_decode([r[n_prompt:] for r in rows], skip_special_tokens=True)


def _chunks(it: Iterable, n: int):
    buf = []
    for x in it:
        buf.append(x)
        if len(buf) == n:
            yield buf
            buf = []
    if buf:
        yield buf


@torch.inference_mode()
def icv_inference(val_ds: Sequence[dict], icv_model, prompt_manager, processor, bs: int, generate_kwargs: dict,
                  instruction: str = "", in_context_vector: Optional[torch.Tensor] = None,
                  alpha: Optional[torch.Tensor] = None) -> Dict[int, dict]:
    """ref:inference.py:246-297: zero-shot prompts [instruction?, image, "Question: ... Short answer:"] in chunks of `bs`,
    hooked generate, one result record per sample ({"prediction": text, **sample without its image}).  As in the reference a
    short LAST chunk is still `bs` prompts wide (the trailing ones carry the instruction only) and only the first
    len(batch) generations are kept."""
    results: Dict[int, dict] = {}
    index = 0
    for batch in _chunks(val_ds, bs):
        prompts = [[instruction] if instruction else [] for _ in range(bs)]
        for i, sample in enumerate(batch):
            prompts[i].extend([sample["image"], prompt_manager.gen_query_text_without_label(sample)])
        query_inputs = processor.prepare_input(prompts)
        query_inputs = {k: v.to(icv_model.lmm.device) for k, v in query_inputs.items()}
        generated = generate_answers(inputs=query_inputs, model=icv_model, processor=processor, generate_kwargs=generate_kwargs,
                                     in_context_vector=in_context_vector, alpha=alpha)
        for i in range(len(batch)):
            rec = dict(batch[i])
            rec.pop("image")
            results[index] = {"prediction": generated[i], **rec}
            index += 1
    return results
