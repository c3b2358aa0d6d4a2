"""ctypes binding of liblicv_hip.so (the C-ABI declared in include/licv_hip.h).

This is the binding a maintainer of the reference would add.  The library is
looked up in-tree only; a missing library is a hard error — there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("LICV_HIP_LIB", _HERE / "liblicv_hip.so"))
HEADER = _HERE.parents[1] / "include" / "licv_hip.h"

LAB_PATH = Path(os.environ.get("LICV_HIP_LAB_LIB", _HERE / "liblicv_hip_lab.so"))
LAB_HEADER = _HERE.parents[1] / "include" / "licv_hip_lab.h"

LICV_BF16, LICV_F32 = 0, 1
ABI_VERSION = 6          # == LICV_ABI_VERSION of include/licv_hip.h this binding was written against (check_exports compares both)

_lib = None
_lab = None


class LicvError(RuntimeError):
    pass


class GemmEpilogue(C.Structure):
    _fields_ = [("bias_bf16", C.c_void_p), ("row_gate", C.c_void_p), ("residual", C.c_void_p),
                ("residual_dtype", C.c_int), ("ld_res", C.c_int64), ("act", C.c_int), ("swiglu", C.c_int),
                ("use_scale", C.c_int), ("scale", C.c_float), ("out_dtype", C.c_int)]


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("q_bs", C.c_int64), ("q_rs", C.c_int64),
                ("k", C.c_void_p), ("v", C.c_void_p), ("kv_bs", C.c_int64), ("kv_rs", C.c_int64),
                ("o", C.c_void_p),
                ("B", C.c_int64), ("Sq", C.c_int64), ("Sk", C.c_int64), ("n_heads", C.c_int64),
                ("n_kv_heads", C.c_int64), ("head_dim", C.c_int64),
                ("scale", C.c_float), ("mask_mode", C.c_int),
                ("key_valid", C.c_void_p), ("img_mask", C.c_void_p), ("n_img", C.c_int64), ("img_len", C.c_int64)]


class BeamStepArgs(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("logits_dtype", C.c_int), ("ld", C.c_int64), ("q_stride_rows", C.c_int64),
                ("beam_stride_rows", C.c_int64), ("B", C.c_int64), ("nb", C.c_int64), ("V", C.c_int64), ("max_len", C.c_int64),
                ("cur", C.c_int64), ("P", C.c_int64), ("eos", C.c_int64), ("suppress_eos", C.c_int), ("length_penalty", C.c_float),
                ("early_stopping", C.c_int),
                ("running_in", C.c_void_p), ("finished_in", C.c_void_p), ("run_scores_in", C.c_void_p), ("fin_scores_in", C.c_void_p),
                ("is_fin_in", C.c_void_p), ("improve_in", C.c_void_p), ("gen_len_in", C.c_void_p),
                ("running_out", C.c_void_p), ("finished_out", C.c_void_p), ("run_scores_out", C.c_void_p), ("fin_scores_out", C.c_void_p),
                ("is_fin_out", C.c_void_p), ("improve_out", C.c_void_p), ("gen_len_out", C.c_void_p),
                ("beam_src_flat", C.c_void_p), ("next_tokens", C.c_void_p), ("flags", C.c_void_p), ("sync", C.c_void_p),
                ("kv_rows_in", C.c_void_p), ("kv_rows_out", C.c_void_p), ("kv_ld", C.c_int64),
                ("scratch", C.c_void_p), ("scratch_bytes", C.c_int64)]


class DecodeAttnArgs(C.Structure):
    _fields_ = [("qkv_ws", C.c_void_p), ("splits", C.c_int), ("slice_elems", C.c_int64), ("row_stride", C.c_int64),
                ("qkv_bf16", C.c_void_p), ("ldq", C.c_int64),
                ("cos", C.c_void_p), ("sin", C.c_void_p), ("position_ids", C.c_void_p), ("n_pos", C.c_int64),
                ("cache", C.c_void_p), ("max_len", C.c_int64), ("past", C.c_int64),
                ("kv_rows", C.c_void_p), ("ld_kv_rows", C.c_int64), ("key_valid", C.c_void_p), ("out", C.c_void_p),
                ("M", C.c_int64), ("n_heads", C.c_int64), ("n_kv_heads", C.c_int64), ("head_dim", C.c_int64), ("scale", C.c_float)]


def declared_symbols(header: Path = HEADER):
    """Every function name declared in include/licv_hip.h (or the given header)."""
    text = header.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(licv_[a-z0-9_]+)\s*\(", text)))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise LicvError(
                f"{LIB_PATH} not found: the L-ICV hot path has no CPU fallback. Build it with "
                f"`python __graft_entry__.py` (hipcc --offload-arch=gfx950).")
        _lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)          # global: liblicv_hip_lab.so resolves against it
        _lib.licv_last_error.restype = C.c_char_p
        _lib.licv_version.restype = C.c_int
        _lib.licv_inject_bwd_partials.restype = C.c_int64
        _lib.licv_workspace_size.restype = C.c_int64
        _lib.licv_workspace_size.argtypes = [C.c_int64, C.c_int64, C.c_int64]
        _lib.licv_inject_bwd_partials.argtypes = [C.c_int64]
        _lib.licv_beam_step_scratch_bytes.restype = C.c_int64
        _lib.licv_beam_step_scratch_bytes.argtypes = [C.c_int64, C.c_int64]
        P, I64, F, I = C.c_void_p, C.c_int64, C.c_float, C.c_int
        sig = {
            "licv_inject_renorm_fwd": [P, I, P, P, P, I64, I64, P, P, F, P],
            "licv_inject_renorm_add_fwd": [P, I, P, P, P, I, P, I64, I64, P, P, F, I, P],
            "licv_inject_renorm_pre_fwd": [P, I, P, P, P, P, I64, I64, P, P, F, P],
            "licv_add_rmsnorm_fwd": [P, I, P, P, I, F, P, P, I64, I64, F, I, P],
            "licv_runner_option": [I, I],
            "licv_backward_option": [I, I],
            "licv_allreduce_small": [P, P, I64, I, P],
            "licv_scatter_rows": [P, P, P, I64, I64, P],
            "licv_ce_rows": [P, I, P, P, I64, I64, I64, P, F, P, P, I64, P, I, P],
            "licv_head_group_sum": [P, P, I64, I64, I64, I64, I64, I64, P],
            "licv_inject_renorm_bwd": [P, I, P, P, P, P, P, I64, I64, P],
            "licv_rmsnorm_fwd": [P, I, P, P, I64, I64, I64, I64, I64, F, I, P],
            "licv_layernorm_fwd": [P, P, P, P, I64, I64, I64, I64, I64, I64, I64, F, P],
            "licv_rotary_fwd": [P, P, P, P, I64, I64, I64, I64, I64, I, I64, P],
            "licv_rotary_kv_append": [P, P, P, P, I64, I64, I64, I64, I64, P, I64, I64, P],
            "licv_gemm_bf16": [P, I64, P, I64, P, I64, I64, I64, I64, C.POINTER(GemmEpilogue), P],
            "licv_pack_gate_up": [P, P, P, I64, I64, P],
            "licv_quantize_rows_fp8": [P, I, P, P, I64, I64, I64, I64, P],
            "licv_rmsnorm_fwd_q8": [P, I, P, P, P, P, I64, I64, F, I, P],
            "licv_add_rmsnorm_fwd_q8": [P, I, P, P, P, P, P, I64, I64, F, I, P],
            "licv_inject_renorm_add_fwd_q8": [P, I, P, P, P, I, P, I64, I64, P, P, P, P, F, I, P],
            "licv_layernorm_fwd_q8": [P, P, P, P, P, P, I64, I64, F, P],
            "licv_gemm_fp8": [P, I64, P, P, I64, P, P, I64, I64, I64, I64, P, P],
            "licv_gemm_splitk_plan": [I64, I64, I64, P, P],
            "licv_gemm_bf16_splitk": [P, I64, P, I64, P, I64, I64, I64, I64, P, I, P, I64, P],
            "licv_gemm_bf16_splitk_produce": [P, I64, P, I64, I64, I64, I64, I, P, I64, P, P, P],
            "licv_add_rmsnorm_fwd_ws": [P, I, P, I, I64, I64, P, I, F, P, P, I64, I64, F, I, P],
            "licv_inject_renorm_pre_fwd_ws": [P, I, P, I, I64, I64, P, P, P, I64, I64, P, P, F, P],
            "licv_rotary_kv_append_ws": [P, I, I64, I64, P, P, P, P, I64, I64, I64, I64, I64, P, I64, I64, P],
            "licv_idefics_image_attention_mask": [P, P, I64, I64, I64, I64, I64, P],
            "licv_idefics2_patch_front": [P, P, P, P, P, P, I64, I64, I64, I64, I64, P],
            "licv_merge_image_rows": [P, P, P, P, P, I64, I64, I64, I64, P],
            "licv_preprocess_images": [P, P, P, P, I64, I64, I64, C.c_double, P, P, P],
            "licv_gemm_flow_available": [],
            "licv_gemm_select": [I],
            "licv_lab_register": [P, P, P],
            "licv_gemm_experiment": [I, I],
            "licv_gemm_debug_timestamps": [P],
            "licv_attn_select": [I],
            "licv_attn_fwd": [C.POINTER(AttnArgs), P],
            "licv_beam_step": [C.POINTER(BeamStepArgs), P],
            "licv_decode_attn": [C.POINTER(DecodeAttnArgs), P],
            "licv_embed_gather": [P, P, P, P, I64, I64, I64, I64, P],
            "licv_im2col_patches": [P, P, I64, I64, I64, I64, I64, P],
            "licv_vit_embed_ln": [P, P, P, P, P, P, I64, I64, I64, F, P],
            "licv_tile_rows": [P, P, I64, I64, I64, P],
            "licv_swiglu": [P, P, I64, I64, P],
            "licv_rmsnorm_bwd": [P, I, P, P, I, P, I, I64, I64, I64, I64, I64, I64, F, I, I, P],
            "licv_rmsnorm_bwd_ws": [P, I, P, P, I, I64, I64, P, I, I64, I64, F, I, I, P],
            "licv_swiglu_bwd": [P, P, P, I64, I64, P],
            "licv_branch_grad": [P, P, I64, I64, F, I, P, P],
            "licv_attn_bwd_small": [C.POINTER(AttnArgs), P, P, I64, I64, P, P, I64, I64, P],
            "licv_kl_rows_bwd": [P, P, I, P, P, I64, I64, I64, I64, F, F, F, P, P, I64, P],
            "licv_kl_rows_fwd": [P, P, I, P, P, I64, I64, I64, I64, F, F, P, P],
            "licv_kl_rows_dtemp": [P, P, I, P, P, I64, I64, I64, I64, F, F, P, P],
            "licv_adamw_step": [P, P, P, P, I64, I64, F, F, F, F, F, F, I64, F, P],
        }
        for name, args in sig.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
    return _lib


def lab() -> C.CDLL:
    """liblicv_hip_lab.so: the experiment kernels and roofline probes (include/licv_hip_lab.h).  Tests and tools only - nothing under
    licv/ calls this.  Loading it registers the experiments with the product library's licv_gemm_select dispatch."""
    global _lab
    if _lab is None:
        lib()                                                   # the lab library links against the product library
        if not LAB_PATH.exists():
            raise LicvError(f"{LAB_PATH} not found: build it with `python __graft_entry__.py`")
        _lab = C.CDLL(str(LAB_PATH), mode=C.RTLD_GLOBAL)
        P, I64, I = C.c_void_p, C.c_int64, C.c_int
        for name, args in {"licv_lab_loaded": [], "licv_gemm_stagger": [I], "licv_probe_mfma_loop": [P, I, I, P],
                           "licv_probe_permlane16_swap": [P, P], "licv_probe_weight_stream": [P, I64, I64, I64, I, I, I, P, P],
                           "licv_probe_lds_dma_stream": [P, I64, I64, I64, I, I, P],
                           "licv_probe_l2_ingest": [P, I64, I, I, I, P, P]}.items():
            fn = getattr(_lab, name)
            fn.argtypes, fn.restype = args, C.c_int
    return _lab


def header_abi_version() -> int:
    m = re.search(r"#define\s+LICV_ABI_VERSION\s+(\d+)", HEADER.read_text())
    return int(m.group(1)) if m else -1


def check_exports():
    """The library exports every symbol the header declares and answers the ABI version this binding was written against
    (no compute calls)."""
    l = lib()
    missing = [s for s in declared_symbols() if not hasattr(l, s)]
    if missing:
        raise LicvError(f"liblicv_hip.so lacks symbols declared in licv_hip.h: {missing}")
    got = l.licv_version()
    if got != ABI_VERSION:
        raise LicvError(f"liblicv_hip.so answers ABI version {got}, this binding (licv/_lib.py) was written against {ABI_VERSION}: rebuild "
                        f"with `python __graft_entry__.py`")
    if HEADER.exists() and header_abi_version() != ABI_VERSION:
        raise LicvError(f"include/licv_hip.h declares LICV_ABI_VERSION {header_abi_version()}, licv/_lib.py expects {ABI_VERSION}")
    return True


def check_lab_exports():
    l = lab()
    missing = [s for s in declared_symbols(LAB_HEADER) if not hasattr(l, s)]
    if missing:
        raise LicvError(f"liblicv_hip_lab.so lacks symbols declared in licv_hip_lab.h: {missing}")
    return l.licv_lab_loaded() == 1


def check(status: int):
    if status != 0:
        msg = lib().licv_last_error().decode()
        raise LicvError(f"liblicv_hip error {status}: {msg}")
