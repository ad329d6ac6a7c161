"""Loader for the C-ABI shared library (soundswallower_amd/libssw_amd.so).

The library is the product: HIP kernels for gfx950 plus the C host code.  There is no CPU
fallback -- a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSW_AMD_LIB: load another build of the same library (the instrumented ones of
# `make -C soundswallower_amd/csrc timeline`); never set in tests or bench runs
LIB_PATH = os.environ.get("SSW_AMD_LIB") or os.path.join(_HERE, "libssw_amd.so")
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "ssw_amd.h")

_lib = None


class SswConfig(C.Structure):
    _fields_ = [("logbase", C.c_double), ("varfloor", C.c_double), ("mixwfloor", C.c_double),
                ("tmatfloor", C.c_double), ("topn", C.c_int32), ("ds", C.c_int32),
                ("aw", C.c_int32), ("device", C.c_int32)]


class SswModelInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_cb", "n_feat", "n_density", "veclen_total", "n_sen", "n_ci_sen", "n_ciphone",
                 "n_phone", "n_emit_state", "n_tmat", "n_sseq", "sil", "n_floored", "topn",
                 "has_ptm", "has_ms", "device")] + [
        ("veclen", C.c_int32 * 8), ("scan_mode", C.c_int32), ("mfma_selftest", C.c_int32),
        ("mfma_selftest_worst_u", C.c_float), ("mfma_selftest_ms", C.c_float)]


class SswAlignEntry(C.Structure):
    _fields_ = [("start", C.c_int32), ("duration", C.c_int32), ("score", C.c_int32)]


FRAME_EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                            C.POINTER(C.POINTER(C.c_float)), C.c_int32, C.c_int32)
TRANSFORM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
FREE_FN = C.CFUNCTYPE(None, C.c_void_p)


class SswMgauFuncs(C.Structure):
    _fields_ = [("name", C.c_char_p), ("frame_eval", FRAME_EVAL_FN), ("transform", TRANSFORM_FN),
                ("free", FREE_FN)]


class SswMgau(C.Structure):
    _fields_ = [("vt", C.POINTER(SswMgauFuncs)), ("frame_idx", C.c_int)]


def build(force: bool = False) -> str:
    """Compile libssw_amd.so for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
            if f.endswith((".c", ".hip", ".h", ".inc"))] + [HEADER]
    if os.environ.get("SSW_AMD_LIB"):
        return LIB_PATH
    stale = (not os.path.exists(LIB_PATH)
             or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs))
    if force or stale:
        args = ["make", "-C", CSRC] + (["-B"] if force else [])
        subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with __graft_entry__.build() or "
            f"`make -C {CSRC}`; soundswallower_amd has no CPU fallback")
    # A PyTorch-ROCm wheel bundles its own HIP / HSA / RCCL.  A process must not end up with two
    # of them (the library's RCCL gather would talk to an HSA runtime nobody initialised:
    # "no ROCm-capable device is detected"), so when torch is installed it goes first and the
    # library binds to the copies torch has loaded.  A C host without torch uses /opt/rocm's.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
    L.ssw_config_defaults.argtypes = [C.POINTER(SswConfig)]
    L.ssw_last_error.restype = C.c_char_p
    L.ssw_abi_version.restype = C.c_int
    L.ssw_model_load.restype = vp
    L.ssw_model_load.argtypes = [C.c_char_p] * 6 + [C.POINTER(SswConfig)]
    L.ssw_model_free.argtypes = [vp]
    L.ssw_model_info.argtypes = [vp, C.POINTER(SswModelInfo)]
    L.ssw_model_selftest_message.restype = C.c_char_p
    L.ssw_model_selftest_message.argtypes = [vp]
    L.ssw_model_table.restype = vp
    L.ssw_model_table.argtypes = [vp, C.c_int, C.POINTER(sz)]
    L.ssw_score_batch.argtypes = [vp, C.c_int, vp, i32, vp, i32, vp, vp]
    L.ssw_debug_score_loop.argtypes = [vp, C.c_int, vp, i32, vp, i32, vp, i32, vp]
    L.ssw_score_batch_ex.argtypes = [vp, C.c_int, vp, i32, vp, i32, vp, vp, C.c_uint32, vp, vp]
    L.ssw_score_batch_host.argtypes = [vp, C.c_int, vp, i32, vp, i32, vp]
    L.ssw_score_batch_topn.argtypes = [vp, i32, vp, vp]
    L.ssw_score_batch_stats.argtypes = [vp, vp]
    L.ssw_scan_audit_stats.argtypes = [vp, vp]
    L.ssw_align_stats.argtypes = [vp, vp]
    L.ssw_compact_plan_create.restype = vp
    L.ssw_compact_plan_create.argtypes = [vp, i32, vp, vp, vp, vp]
    L.ssw_compact_plan_free.argtypes = [vp]
    L.ssw_compact_plan_free.restype = None
    L.ssw_compact_plan_elems.argtypes = [vp]
    L.ssw_compact_plan_elems.restype = sz
    L.ssw_compact_plan_rows.argtypes = [vp, i32, vp, vp]
    L.ssw_score_batch_compact.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_uint32]
    L.ssw_align_batch_compact.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32]
    L.ssw_debug_scan_keys.argtypes = [vp, vp, i32, i32, vp]
    L.ssw_debug_mfma_f16_tiles.argtypes = [vp, vp, vp, vp, vp, i32]
    L.ssw_set_kernel_timing.argtypes = [vp, C.c_int]
    L.ssw_get_kernel_timing.argtypes = [vp, vp, C.c_int]
    L.ssw_ptm_mgau_init.restype = C.POINTER(SswMgau)
    L.ssw_ptm_mgau_init.argtypes = [vp]
    L.ssw_ms_mgau_init.restype = C.POINTER(SswMgau)
    L.ssw_ms_mgau_init.argtypes = [vp]
    L.ssw_mgau_reset_hist.argtypes = [C.POINTER(SswMgau)]
    L.ssw_mgau_prescore.argtypes = [C.POINTER(SswMgau), vp, i32]
    L.ssw_align_batch.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ssw_align_batch_active.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ssw_align_batch_active_ex.argtypes = [vp, C.c_int, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ssw_alignment_propagate.argtypes = [vp, vp, i32, vp, i32]
    L.ssw_state_align_search_init.restype = vp
    L.ssw_state_align_search_init.argtypes = [vp, C.POINTER(SswMgau), i32, vp, vp, vp, vp]
    L.ssw_state_align_search_start.argtypes = [vp]
    L.ssw_state_align_search_step.argtypes = [vp, vp, C.c_int]
    L.ssw_state_align_search_finish.argtypes = [vp]
    L.ssw_state_align_search_n_frames.restype = i32
    L.ssw_state_align_search_n_frames.argtypes = [vp]
    L.ssw_state_align_search_states.restype = C.POINTER(SswAlignEntry)
    L.ssw_state_align_search_states.argtypes = [vp, C.POINTER(i32)]
    L.ssw_state_align_search_phones.restype = C.POINTER(SswAlignEntry)
    L.ssw_state_align_search_phones.argtypes = [vp, C.POINTER(i32)]
    L.ssw_state_align_search_free.argtypes = [vp]
    L.ssw_dict_load.restype = vp
    L.ssw_dict_load.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.ssw_dict_free.argtypes = [vp]
    L.ssw_dict_size.restype = i32
    L.ssw_dict_size.argtypes = [vp]
    L.ssw_dict_pron.restype = i32
    L.ssw_dict_pron.argtypes = [vp, C.c_char_p, vp, i32]
    L.ssw_dict_word.restype = C.c_char_p
    L.ssw_dict_word.argtypes = [vp, i32]
    L.ssw_dict_word_id.restype = i32
    L.ssw_dict_word_id.argtypes = [vp, C.c_char_p]
    L.ssw_first_pass_config_defaults.argtypes = [vp]
    L.ssw_first_pass_config_defaults.restype = None
    L.ssw_first_pass_graph.restype = i32
    L.ssw_first_pass_graph.argtypes = [vp, vp, vp, i32, vp, i32, vp, vp]
    L.ssw_first_pass_batch.restype = C.c_int
    L.ssw_first_pass_batch.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp]
    L.ssw_forced_align_batch.restype = vp
    L.ssw_forced_align_batch.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp, vp, vp]
    L.ssw_first_pass_prepare.restype = vp
    L.ssw_first_pass_prepare.argtypes = [vp, vp, vp, i32, vp, vp]
    L.ssw_first_pass_plan_free.argtypes = [vp]
    L.ssw_first_pass_plan_free.restype = None
    L.ssw_first_pass_run.restype = C.c_int
    L.ssw_first_pass_run.argtypes = [vp, vp, vp, i32, vp, i32, vp, vp, vp]
    L.ssw_forced_align_planned.restype = vp
    L.ssw_forced_align_planned.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    L.ssw_align_text_batch.restype = vp
    L.ssw_align_text_batch.argtypes = [vp, vp, vp, C.c_int, vp, i32, vp, i32, vp, vp, vp]
    L.ssw_align_text_batch_active.restype = vp
    L.ssw_align_text_batch_active.argtypes = [vp, vp, vp, C.c_int, vp, i32, vp, i32, vp, vp, vp]
    L.ssw_first_pass_batch_active.restype = C.c_int
    L.ssw_first_pass_batch_active.argtypes = [vp, vp, vp, C.c_int, vp, i32, vp, i32, vp, vp, i32,
                                              vp, vp, vp, vp, vp, vp]
    L.ssw_first_pass_active_carry.restype = C.c_int
    L.ssw_first_pass_active_carry.argtypes = [vp, i32, vp]
    L.ssw_first_pass_active_stats.restype = C.c_int
    L.ssw_first_pass_active_stats.argtypes = [vp, vp]
    L.ssw_alignment_set_status.restype = i32
    L.ssw_alignment_set_status.argtypes = [vp, i32]
    L.ssw_alignment_set_message.restype = C.c_char_p
    L.ssw_alignment_set_message.argtypes = [vp, i32]
    for fn in (L.ssw_alignment_set_words, L.ssw_alignment_set_states):
        fn.restype = i32
        fn.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp)]
    L.ssw_alignment_set_phones.restype = i32
    L.ssw_alignment_set_phones.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.ssw_alignment_set_json.restype = i32
    L.ssw_alignment_set_json.argtypes = [vp, i32, C.c_double, i32, i32, C.c_char_p, i32]
    L.ssw_dict_base_id.restype = i32
    L.ssw_dict_base_id.argtypes = [vp, i32]
    L.ssw_dict_is_filler.restype = i32
    L.ssw_dict_is_filler.argtypes = [vp, i32]
    L.ssw_alignment_set_free.argtypes = [vp]
    L.ssw_alignment_set_free.restype = None
    L.ssw_ciphone_name.restype = C.c_char_p
    L.ssw_ciphone_name.argtypes = [vp, i32]
    L.ssw_phone_id_nearest.restype = i32
    L.ssw_phone_id_nearest.argtypes = [vp, i32, i32, i32, i32]
    L.ssw_alignment_json.restype = i32
    L.ssw_alignment_json.argtypes = [vp, C.c_char_p, i32, C.c_double, i32, i32, i32, vp, vp, i32, vp,
                                     vp, vp, vp, vp, C.c_char_p, i32]
    L.ssw_alignment_populate.restype = i32
    L.ssw_alignment_populate.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.ssw_feat_batch.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp]
    L.ssw_comm_unique_id.argtypes = [C.c_char_p]
    L.ssw_comm_init.restype = vp
    L.ssw_comm_init.argtypes = [C.c_char_p, i32, i32, i32]
    L.ssw_comm_from_nccl.restype = vp
    L.ssw_comm_from_nccl.argtypes = [vp, i32, i32, i32]
    L.ssw_comm_from_transport.restype = vp
    L.ssw_comm_from_transport.argtypes = [vp, vp, i32, i32]
    L.ssw_comm_free.argtypes = [vp]
    L.ssw_comm_free.restype = None
    L.ssw_gather_alignments.argtypes = [vp, vp, i32, vp, vp, vp]
    L.ssw_comm_count.argtypes = [vp]
    L.ssw_comm_count.restype = i32
    L.ssw_stream_create.restype = vp
    L.ssw_stream_destroy.argtypes = [vp]
    L.ssw_stream_destroy.restype = None
    L.ssw_stream_synchronize.argtypes = [vp]
    L.ssw_device_malloc.restype = vp
    L.ssw_device_malloc.argtypes = [sz]
    L.ssw_device_free.argtypes = [vp]
    L.ssw_device_mem_info.argtypes = [vp, vp]
    L.ssw_memcpy_h2d.argtypes = [vp, vp, sz]
    L.ssw_memcpy_d2h.argtypes = [vp, vp, sz]
    _lib = L
    return L


def last_error() -> str:
    return lib().ssw_last_error().decode(errors="replace")
