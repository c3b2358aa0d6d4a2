from __future__ import annotations

import re
import types
from pathlib import Path
from typing import Dict, List, Optional

import torch

from licv.config import Idefics2Arch, IdeficsArch
from licv.generation import generate as native_generate
from licv.idefics_engine import IdeficsEngine, IdeficsWeights
from licv.idefics2_engine import Idefics2Engine, Idefics2Weights


class LMMOutput(dict):
    """HF-ModelOutput-like: ``out["logits"]`` and ``out.logits`` both work."""
    __getattr__ = dict.get


class _NativeModel(torch.nn.Module):
    """What ``interface.model`` exposes to the reference (ref:icv_src/icv_module.py:29-30)."""

    def __init__(self, arch, model_type: str = "idefics"):
        super().__init__()
        self.config = types.SimpleNamespace(**arch.to_dict(), model_type=model_type)

    def gradient_checkpointing_enable(self, *a, **k):      # the native engine recomputes instead
        return None


def ce_rows_and_labels(labels: torch.Tensor, keep: torch.Tensor):
    """Flat (b*S + t) row indices of the logits that predict a kept token, and those tokens: position t predicts labels[:, t+1]."""
    B, S = labels.shape
    t_idx = torch.arange(S - 1, device=labels.device).unsqueeze(0).expand(B, -1)
    rows = (torch.arange(B, device=labels.device).unsqueeze(1) * S + t_idx)[keep]
    return rows.contiguous(), labels[:, 1:][keep].contiguous()


def _ce_shifted(logits: torch.Tensor, labels: torch.Tensor, keep: torch.Tensor) -> torch.Tensor:
    """mean over kept positions of CE(logits[:, t], labels[:, t+1]) — the row reduction over the vocabulary is a HIP kernel
    (forward and, when the logits carry gradient, backward)."""
    from licv.autograd import ShiftedCEFn
    rows, tok = ce_rows_and_labels(labels, keep)
    return ShiftedCEFn.apply(logits, rows, tok)


class LMMInterface(torch.nn.Module):
    input_ids_field_name = "input_ids"
    supports_logits_rows = True          # forward(..., logits_rows=flat b*S+t indices) returns only those rows' logits (R, V)

    def __init__(self):
        super().__init__()
        self._plan = None
        self._student_pass = None

    def student_pass(self):
        """The forward-with-saved-activations + explicit backward (licv.train_engine), built on first use: it keeps transposed
        copies of the language stack's weights (+17 GB at 9B), which inference never needs."""
        if self._student_pass is None:
            from licv.train_engine import StudentPass, StudentPass2, TrainWeights, TrainWeights2
            if isinstance(self.engine, Idefics2Engine):
                self._student_pass = StudentPass2(self.engine, TrainWeights2(self.engine.w))
            else:
                self._student_pass = StudentPass(self.engine, TrainWeights(self.engine.w))
        return self._student_pass

    def _run_lmm(self, inputs: dict, logits_rows=None):
        """Hooked (or plain) forward.  When the installed ``icv`` carries gradient the pass runs behind an autograd node whose
        backward is the explicit HIP backward through the frozen LMM (ref:icv_src/icv_module.py:97-98 is then differentiable)."""
        hooks = self._hooks()
        icv = hooks.get("icv")
        if icv is not None and torch.is_grad_enabled() and icv.requires_grad:
            from licv.autograd import HookedStudentFn
            return HookedStudentFn.apply(icv, self.student_pass(), inputs, hooks["hook_layers"], logits_rows)
        return self.engine.forward(**inputs, **hooks, logits_rows=logits_rows)


class IdeficsInterface(LMMInterface):
    """IdeficsInterface(model_name_or_path, precision, device, prompt_manager, instruction, image_field, label_field)
    as called at ref:utils.py:41-50.  Extra keyword-only ``state_dict``/``arch`` build from in-memory weights
    (tests, bench); otherwise ``model_name_or_path`` must be a local HF checkpoint directory."""

    HOOK_SITE = re.compile(r"^model\.model\.layers\.(\d+)$")       # ref:config/lmm/idefics-9B.yaml:7

    def __init__(self, model_name_or_path=None, precision="bf16", device="cuda", prompt_manager=None, instruction="",
                 image_field="image", label_field="answer", *, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 arch: Optional[IdeficsArch] = None, tokenizer=None, processor=None):
        super().__init__()
        if str(precision) not in ("bf16", "bfloat16", "torch.bfloat16"):
            raise ValueError(f"the native Idefics path computes in bf16 only (got precision={precision!r})")
        if state_dict is None:
            state_dict, arch, tokenizer, processor = self._load_checkpoint(Path(model_name_or_path), tokenizer, processor)
        self.arch = arch
        self._device = torch.device(device)
        self.engine = IdeficsEngine(IdeficsWeights(state_dict, arch, self._device))
        self.model = _NativeModel(arch)
        self.prompt_manager, self.instruction = prompt_manager, instruction
        self.image_field, self.label_field = image_field, label_field
        self.tokenizer = tokenizer if tokenizer is not None else types.SimpleNamespace(
            pad_token_id=arch.pad_token_id, bos_token_id=arch.bos_token_id, eos_token_id=arch.eos_token_id, padding_side="right")
        self.processor = processor
        # `<image>` id for the device-side image_attention_mask builder: from the tokenizer when it knows the token, else the second
        # additional-vocabulary slot (where the released checkpoints put it)
        self.image_token_id = self._resolve_image_token_id(self.tokenizer, arch)

    @staticmethod
    def _resolve_image_token_id(tokenizer, arch):
        """`<image>` id: the tokenizer's, when it really knows the token (not None, not its unk id); otherwise the
        additional-vocabulary slot the released checkpoints use.  An id outside the embedding table coming from a tokenizer
        is an error (tokenizer and model do not belong together); a model without additional vocabulary and without a
        tokenizer simply has no `<image>` token: None, and building an image_attention_mask from input_ids then raises.
        A silently wrong id would give an all-zero mask, i.e. no cross-attention at all, without any error."""
        extra = getattr(arch, "additional_vocab_size", 0)
        n_embed = arch.vocab_size + extra
        fallback = arch.vocab_size + (1 if extra > 1 else 0) if extra > 0 else None
        conv = getattr(tokenizer, "convert_tokens_to_ids", None)
        if conv is None:
            return fallback
        tid = conv("<image>")
        unk = getattr(tokenizer, "unk_token_id", None)
        if tid is None or (unk is not None and tid == unk) or not isinstance(tid, int):
            return fallback
        if not 0 <= tid < max(n_embed, 1):
            raise ValueError(f"`<image>` token id {tid} is outside the embedding table of {n_embed} rows: "
                             "the tokenizer and the model configuration do not belong together")
        return tid

    # ------------------------------------------------------------------ device-side image input
    image_norm = ("IDEFICS_MEAN", "IDEFICS_STD")             # names in licv.frontend; Idefics2Interface overrides

    def image_feeder(self, max_images: int, height: Optional[int] = None, width: Optional[int] = None, with_mask: bool = False):
        """The pinned, double-buffered uint8 -> normalised-bf16 image path for this model (licv.image_feeder.ImageFeeder): what
        `processor.prepare_input` + `.to(device)` do for the image half of a batch (ref:icv_src/icv_datamodule.py:80-124,
        ref:inference.py:277-278), with the bytes crossing PCIe as uint8 and the normalisation done by a HIP kernel on a side
        stream.  Default size: the vision tower's image size."""
        from licv import frontend
        from licv.image_feeder import ImageFeeder
        side = getattr(self.arch, "v_image", 224)
        return ImageFeeder(self._device, max_images, height or side, width or side, getattr(frontend, self.image_norm[0]),
                           getattr(frontend, self.image_norm[1]), with_mask=with_mask)

    def _image_mask(self, input_ids, pixel_values, image_attention_mask):
        """image_attention_mask as processor.prepare_input builds it (hf:idefics/processing_idefics.py:89-133) when the caller
        did not pass one: from input_ids, on the device (licv.frontend, csrc/frontend.hip)."""
        if image_attention_mask is not None:
            return image_attention_mask.to(self._device)
        if self.image_token_id is None:
            raise ValueError("no image_attention_mask was passed and this model has no `<image>` token to build one from "
                             "(no additional vocabulary, no tokenizer that knows the token)")
        from licv import frontend
        return frontend.idefics_image_attention_mask(input_ids.to(self._device), self.image_token_id,
                                                     getattr(self.tokenizer, "eos_token_id", self.arch.eos_token_id), pixel_values.shape[1])

    @staticmethod
    def _load_checkpoint(path: Path, tokenizer, processor, arch_cls=IdeficsArch):
        from safetensors.torch import load_file
        from transformers import AutoConfig
        if not path.is_dir():
            raise FileNotFoundError(f"{path} is not a local checkpoint directory (no network access to fetch one)")
        arch = arch_cls.from_hf(AutoConfig.from_pretrained(path))
        sd = {}
        for f in sorted(path.glob("*.safetensors")):
            sd.update(load_file(str(f)))
        if not sd:
            raise FileNotFoundError(f"no *.safetensors shards under {path}")
        if processor is None:
            try:
                from transformers import AutoProcessor
                processor = AutoProcessor.from_pretrained(path)
                tokenizer = tokenizer or processor.tokenizer
            except Exception:                                           # tokenizer files are optional for the model side
                processor = None
        return sd, arch, tokenizer, processor

    # ---- what the reference touches
    @property
    def device(self):
        return self._device

    def requires_grad_(self, flag: bool = True):                        # LMM weights are frozen constants here
        return self

    # ---- native hook plan (licv.intervention.NativeIntervention)
    def install_intervention(self, layer_names: List[str], layer_to_icv_index: Dict[int, int], icv: torch.Tensor):
        layers = []
        for name in layer_names:
            m = self.HOOK_SITE.match(name)
            if m is None or int(m.group(1)) >= self.arch.num_layers:
                raise LookupError(f"no hook site named {name!r} in the native Idefics engine "
                                  f"(sites: model.model.layers.<0..{self.arch.num_layers - 1}>)")
            layers.append(int(m.group(1)))
        order = sorted(range(len(layers)), key=lambda i: layer_to_icv_index[layers[i]])
        self._plan = ([layers[i] for i in order], icv)

    def remove_intervention(self):
        self._plan = None

    def _hooks(self):
        if self._plan is None:
            return {}
        layers, icv = self._plan
        return dict(icv=icv, hook_layers=layers)

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, image_attention_mask=None, labels=None,
                logits_rows=None, **_):
        dev = self._device
        logits = self._run_lmm(dict(input_ids=input_ids.to(dev), attention_mask=attention_mask.to(dev), pixel_values=pixel_values.to(dev),
                                    image_attention_mask=self._image_mask(input_ids, pixel_values, image_attention_mask)), logits_rows)
        out = LMMOutput(logits=logits)
        if labels is not None:
            # shift-by-one CE with pads masked by attention_mask: transformers 4.38.2 Idefics behaviour
            keep = attention_mask[:, 1:].to(self._device) != 0
            out["loss"] = _ce_shifted(logits, labels.to(self._device), keep)
        return out

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, pixel_values=None, image_attention_mask=None,
                 max_new_tokens=20, num_beams=1, length_penalty=1.0, min_new_tokens=0, early_stopping=False,
                 eos_token_id=None, pad_token_id=None, do_sample=False, **_):
        if do_sample:
            raise NotImplementedError("sampling is not part of the reference's inference path")
        return native_generate(self.engine, input_ids.to(self._device), attention_mask.to(self._device),
                               pixel_values.to(self._device), self._image_mask(input_ids, pixel_values, image_attention_mask),
                               max_new_tokens=max_new_tokens, num_beams=num_beams, length_penalty=length_penalty,
                               min_new_tokens=min_new_tokens, early_stopping=early_stopping,
                               eos_token_id=eos_token_id if eos_token_id is not None else getattr(self.tokenizer, "eos_token_id", None),
                               pad_token_id=pad_token_id if pad_token_id is not None else getattr(self.tokenizer, "pad_token_id", None),
                               **self._hooks())


class Idefics2Interface(IdeficsInterface):
    """Idefics2Interface(model_name_or_path, precision, device, prompt_manager, instruction, image_field, label_field)
    (ref:utils.py:68-78).  Hook sites are the text layers' MLP branches (ref:config/lmm/idefics2-8B-base.yaml:8)."""
    image_norm = ("IDEFICS2_MEAN", "IDEFICS2_STD")

    HOOK_SITE = re.compile(r"^model\.model\.text_model\.layers\.(\d+)\.mlp$")

    def __init__(self, model_name_or_path=None, precision="bf16", device="cuda", prompt_manager=None, instruction="",
                 image_field="image", label_field="answer", *, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 arch: Optional[Idefics2Arch] = None, tokenizer=None, processor=None):
        LMMInterface.__init__(self)
        if str(precision) not in ("bf16", "bfloat16", "torch.bfloat16"):
            raise ValueError(f"the native Idefics2 path computes in bf16 only (got precision={precision!r})")
        if state_dict is None:
            state_dict, arch, tokenizer, processor = self._load_checkpoint(Path(model_name_or_path), tokenizer, processor, Idefics2Arch)
        self.arch = arch
        self._device = torch.device(device)
        self.engine = Idefics2Engine(Idefics2Weights(state_dict, arch, self._device))
        self.model = _NativeModel(arch, "idefics2")
        self.prompt_manager, self.instruction = prompt_manager, instruction
        self.image_field, self.label_field = image_field, label_field
        self.tokenizer = tokenizer if tokenizer is not None else types.SimpleNamespace(
            pad_token_id=arch.pad_token_id, bos_token_id=arch.bos_token_id, eos_token_id=arch.eos_token_id, padding_side="right")
        self.processor = processor

    def install_intervention(self, layer_names: List[str], layer_to_icv_index: Dict[int, int], icv: torch.Tensor):
        try:
            super().install_intervention(layer_names, layer_to_icv_index, icv)
        except LookupError as e:
            raise LookupError(str(e).replace("model.model.layers.<", "model.model.text_model.layers.<").replace(">)", ">.mlp)")
                              .replace("Idefics engine", "Idefics2 engine")) from None

    def forward(self, input_ids=None, attention_mask=None, pixel_values=None, pixel_attention_mask=None, labels=None,
                logits_rows=None, **_):
        dev = self._device
        logits = self._run_lmm(dict(input_ids=input_ids.to(dev), attention_mask=attention_mask.to(dev) if attention_mask is not None else None,
                                    pixel_values=pixel_values.to(dev) if pixel_values is not None else None,
                                    pixel_attention_mask=pixel_attention_mask.to(dev) if pixel_attention_mask is not None else None),
                               logits_rows)
        out = LMMOutput(logits=logits)
        if labels is not None:
            # hf:idefics2/modeling_idefics2.py ForConditionalGeneration loss: plain shifted CE, ignore_index=-100
            lab = labels.to(dev)
            out["loss"] = _ce_shifted(logits, lab, lab[:, 1:] != -100)
        return out

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, pixel_values=None, pixel_attention_mask=None,
                 max_new_tokens=20, num_beams=1, length_penalty=1.0, min_new_tokens=0, early_stopping=False,
                 eos_token_id=None, pad_token_id=None, do_sample=False, **_):
        if do_sample:
            raise NotImplementedError("sampling is not part of the reference's inference path")
        from licv.generation import generate_idefics2
        dev = self._device
        return generate_idefics2(self.engine, input_ids.to(dev), attention_mask.to(dev),
                                 pixel_values.to(dev) if pixel_values is not None else None,
                                 pixel_attention_mask.to(dev) if pixel_attention_mask is not None else None,
                                 max_new_tokens=max_new_tokens, num_beams=num_beams, length_penalty=length_penalty,
                                 min_new_tokens=min_new_tokens, early_stopping=early_stopping,
                                 eos_token_id=eos_token_id if eos_token_id is not None else getattr(self.tokenizer, "eos_token_id", None),
                                 pad_token_id=pad_token_id if pad_token_id is not None else getattr(self.tokenizer, "pad_token_id", None),
                                 **self._hooks())
