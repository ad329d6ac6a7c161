"""Host-side mirror of the reference's interfaces for the accelerated path, over the C ABI.

Names follow the reference: `Model` stands for what `acmod_load_am` loads, `PtmMgau` for the
`mgau_t` returned by `ptm_mgau_init` (its `frame_eval` is the vtable slot `acmod_score`
calls), `StateAlignSearch` for `state_align_search_init/start/step/finish`.  Everything runs on
the GPU through `libssw_amd.so`; there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import SswAlignEntry, SswConfig, SswModelInfo

SCORER_PTM = 0
SCORER_MS = 1
INT_MAX = 2**31 - 1

_TABLES = {
    "mean": (0, np.float32), "var": (1, np.float32), "det": (2, np.float32),
    "ptm_mixw": (3, np.uint8), "ms_pdf": (4, np.uint8), "tp": (5, np.uint8),
    "sseq": (6, np.uint16), "sen2cb": (7, np.int16), "logadd8": (8, np.uint8),
    "phone_ssid": (9, np.int32), "phone_tmat": (10, np.int32),
    "rec": (11, np.float32), "scan_rec": (12, np.float32), "scan_d0": (13, np.float32),
    "scan_exact": (14, np.uint32), "scan_rec_mfma": (15, np.float32),
    "scan_exact_mfma": (16, np.uint32), "scan_wfrag": (17, np.uint16),
}


class SswError(RuntimeError):
    pass


class FirstPassConfig(C.Structure):
    """ssw_first_pass_config_t"""
    _fields_ = [("beam", C.c_double), ("pbeam", C.c_double), ("wbeam", C.c_double),
                ("wip", C.c_double), ("pip", C.c_double), ("lw", C.c_float),
                ("silprob", C.c_float), ("fillprob", C.c_float), ("use_filler", C.c_int32),
                ("use_altpron", C.c_int32), ("two_pass_history", C.c_int32)]


FP_NODE_DTYPE = np.dtype([("senid", np.uint16, 3), ("tmat", np.int16), ("pen", np.int32),
                          ("parent", np.int32), ("flags", np.uint32), ("ci_ext", np.int32),
                          ("state", np.int32), ("to_state", np.int32), ("wid", np.int32),
                          ("ctxt", np.uint64)], align=True)
WORD_SEG_DTYPE = np.dtype([("wid", np.int32), ("start", np.int32), ("duration", np.int32),
                           ("score", np.int32)])


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):  # torch tensor
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


def _check(rv, what):
    if rv is None or (isinstance(rv, int) and rv < 0):
        raise SswError(f"{what}: {_lib.last_error()}")
    return rv


def model_dir(name: str) -> str:
    """Path of a bundled acoustic model ("en-us", "fr-fr")."""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "model", name)


class Model:
    """Acoustic model tables on the GPU (ssw_model_load)."""

    def __init__(self, path=None, *, mdef=None, means=None, variances=None, sendump=None,
                 mixw=None, tmat=None, config=None):
        L = _lib.lib()
        if path is not None:
            j = lambda n: os.path.join(path, n)
            mdef = mdef or j("mdef")
            means = means or j("means")
            variances = variances or j("variances")
            tmat = tmat or j("transition_matrices")
            if sendump is None and mixw is None:
                if os.path.exists(j("sendump")):
                    sendump = j("sendump")
                else:
                    mixw = j("mixture_weights")
        cfg = SswConfig()
        L.ssw_config_defaults(C.byref(cfg))
        for k, v in (config or {}).items():
            setattr(cfg, k, v)
        enc = lambda s: None if s is None else os.fsencode(s)
        self._L = L
        self._m = L.ssw_model_load(enc(mdef), enc(means), enc(variances), enc(sendump),
                                   enc(mixw), enc(tmat), C.byref(cfg))
        if not self._m:
            raise SswError("ssw_model_load: " + _lib.last_error())
        info = SswModelInfo()
        L.ssw_model_info(self._m, C.byref(info))
        self.info = info
        for name, ctype in SswModelInfo._fields_:
            if name != "veclen":
                setattr(self, name, float(getattr(info, name)) if ctype is C.c_float
                        else int(getattr(info, name)))
        self.veclen = [int(info.veclen[i]) for i in range(self.n_feat)]
        # outcome of the matrix-core self-test at load (ssw_amd.h, ssw_model_info_t)
        self.selftest_message = (L.ssw_model_selftest_message(self._m) or b"").decode()

    def close(self):
        if getattr(self, "_m", None):
            self._L.ssw_model_free(self._m)
            self._m = None

    __del__ = close

    @property
    def tmat_n_emit(self) -> int:
        """Emitting states of the transition matrices (3 for both shipped models; 1, 2, 4 or 5
        take hmm_vit_eval's other branches, viterbi_align_any_kernel)."""
        if getattr(self, "_tmat_ne", None) is None:
            per = self.table("tp").size // max(self.n_tmat, 1)       # ne * (ne + 1) bytes a matrix
            self._tmat_ne = next((ne for ne in range(1, 17) if ne * (ne + 1) == per), 3)
        return self._tmat_ne

    def table(self, name: str) -> np.ndarray:
        """Host copy of a derived table (flat), for loader parity checks."""
        which, dtype = _TABLES[name]
        n = C.c_size_t(0)
        p = self._L.ssw_model_table(self._m, which, C.byref(n))
        if not p or n.value == 0:
            return np.zeros(0, dtype)
        buf = (C.c_char * n.value).from_address(p)
        return np.frombuffer(buf, dtype=dtype).copy()

    # ---- scoring ------------------------------------------------------------------
    def score_batch(self, feats, utt_off=None, scorer=SCORER_PTM, out=None) -> np.ndarray:
        """Score host features [n_frames][39]; returns int16 [n_frames][n_sen] (`out`, if given:
        a caller that scores batch after batch reuses one buffer instead of faulting in 10 KB
        of fresh pages per frame)."""
        feats = np.ascontiguousarray(feats, dtype=np.float32).reshape(-1, self.veclen_total)
        n = feats.shape[0]
        off = (np.array([0, n], np.int32) if utt_off is None
               else np.ascontiguousarray(utt_off, np.int32))
        if out is None:
            out = np.zeros((n, self.n_sen), np.int16)
        assert out.dtype == np.int16 and out.shape == (n, self.n_sen) and out.flags.c_contiguous
        _check(self._L.ssw_score_batch_host(self._m, scorer, _ptr(feats), n, _ptr(off),
                                            len(off) - 1, _ptr(out)), "ssw_score_batch_host")
        return out

    def score_batch_device(self, d_feats, n_frames, utt_off, d_out, stream=None,
                           scorer=SCORER_PTM, share_device=False):
        """Device pointers (ints or torch tensors); asynchronous on `stream`.  share_device:
        SSW_SCORE_SHARE_DEVICE, for callers that run another stream's kernels beside it."""
        off = np.ascontiguousarray(utt_off, np.int32)
        st = C.c_void_p(int(stream)) if stream else None
        if share_device:
            _check(self._L.ssw_score_batch_ex(self._m, scorer, _ptr(d_feats), int(n_frames),
                                              _ptr(off), len(off) - 1, _ptr(d_out), st, 2, None,
                                              None), "ssw_score_batch_ex")
            return
        _check(self._L.ssw_score_batch(self._m, scorer, _ptr(d_feats), int(n_frames), _ptr(off),
                                       len(off) - 1, _ptr(d_out), st),
               "ssw_score_batch")

    def score_batch_carry(self, feats, utt_off=None, carry_in=None, carry_utts=False,
                          scorer=SCORER_PTM, rewind=False):
        """ssw_score_batch_ex on host features: returns (scores int16 [n][n_sen], carry_out
        uint32 [n_cb * n_feat]); carry_in = the carry_out of an earlier call (or None).
        rewind = SSW_SCORE_CARRY_OUT_REWIND: carry_out is what the NEXT utterance, or the second
        pass over this one after acmod_rewind, starts from (history slot 1 of the reference's
        ring) instead of what a continuation of the same utterance starts from."""
        feats = np.ascontiguousarray(feats, dtype=np.float32).reshape(-1, self.veclen_total)
        n = feats.shape[0]
        off = (np.array([0, n], np.int32) if utt_off is None
               else np.ascontiguousarray(utt_off, np.int32))
        out = np.zeros((n, self.n_sen), np.int16)
        cin = None if carry_in is None else np.ascontiguousarray(carry_in, np.uint32)
        cout = np.zeros(self.n_cb * self.n_feat, np.uint32)
        if n == 0:
            return out, (cin.copy() if cin is not None else np.full_like(cout, 0x03020100))
        flags = (1 if carry_utts else 0) | (4 if rewind else 0)
        d_in = self.to_device(feats)
        d_out = self.device_malloc(out.nbytes)
        try:
            _check(self._L.ssw_score_batch_ex(self._m, scorer, d_in, n, _ptr(off), len(off) - 1,
                                              d_out, None, flags, _ptr(cin),
                                              _ptr(cout)), "ssw_score_batch_ex")
            _check(self._L.ssw_memcpy_d2h(_ptr(out), d_out, out.nbytes), "ssw_memcpy_d2h")
        finally:
            self.device_free(d_in)
            self.device_free(d_out)
        return out, cout

    def last_topn(self, n_frames):
        n_cbf = self.n_cb * self.n_feat
        cw = np.zeros((n_frames, self.n_cb, self.n_feat, self.topn), np.uint8)
        sc = np.zeros((n_frames, self.n_cb, self.n_feat, self.topn), np.int32)
        _check(self._L.ssw_score_batch_topn(self._m, n_frames, _ptr(cw), _ptr(sc)),
               "ssw_score_batch_topn")
        assert cw.size == n_frames * n_cbf * self.topn
        return cw, sc

    def debug_mfma_f16_tiles(self, A, B, C):
        """D = A B + C per tile on the matrix cores (one v_mfma_f32_32x32x16_f16 each):
        A float16 [n][32][16], B float16 [n][16][32], C float32 [n][32][32]."""
        A = np.ascontiguousarray(A, np.float16)
        B = np.ascontiguousarray(B, np.float16)
        C = np.ascontiguousarray(C, np.float32)
        n = len(A)
        assert A.shape == (n, 32, 16) and B.shape == (n, 16, 32) and C.shape == (n, 32, 32)
        D = np.zeros_like(C)
        _check(self._L.ssw_debug_mfma_f16_tiles(self._m, _ptr(A), _ptr(B), _ptr(C), _ptr(D), n),
               "ssw_debug_mfma_f16_tiles")
        return D

    def debug_scan_keys(self, feats, cbf):
        """Raw keys of the matrix-core scan for one codebook x stream: float32 [n_frames][128]."""
        feats = np.ascontiguousarray(feats, np.float32).reshape(-1, self.veclen_total)
        d = self.to_device(feats)
        out = np.zeros((len(feats), 128), np.float32)
        try:
            _check(self._L.ssw_debug_scan_keys(self._m, d, len(feats), int(cbf), _ptr(out)),
                   "ssw_debug_scan_keys")
        finally:
            self.device_free(d)
        return out

    def set_kernel_timing(self, enable=True):
        _check(self._L.ssw_set_kernel_timing(self._m, int(bool(enable))), "ssw_set_kernel_timing")

    def kernel_timing(self):
        """(topn_ms, senone_ms) of the last score_batch call, from HIP events on its stream.  For
        a batch scored in pieces (no split exists) the first number is the whole call and the
        second 0; `self.timing_split` says which it was."""
        ms = np.zeros(2, np.float32)
        n = _check(self._L.ssw_get_kernel_timing(self._m, _ptr(ms), 2), "ssw_get_kernel_timing")
        self.timing_split = n == 2
        return float(ms[0]), float(ms[1])

    def last_stats(self):
        st = np.zeros(2, np.int64)
        self._L.ssw_score_batch_stats(self._m, _ptr(st))
        return int(st[0]), int(st[1])

    def scan_audit_stats(self):
        """SSW_SCAN_AUDIT=k: (proven pairs redone exactly by audited waves, of those the pairs
        whose exact result differs from the scan's), running totals since the model was loaded."""
        st = np.zeros(2, np.int64)
        self._L.ssw_scan_audit_stats(self._m, _ptr(st))
        return int(st[0]), int(st[1])

    def align_stats(self):
        """(utterances through the byte-token alignment kernel, of those handed on to the
        full-token kernel), running totals"""
        st = np.zeros(2, np.int64)
        self._L.ssw_align_stats(self._m, _ptr(st))
        return int(st[0]), int(st[1])

    def first_pass_active_stats(self):
        """ssw_first_pass_active_stats: (utterances searched in the default configuration as a
        batch, their rounds summed, rounds of the last call, utterances that needed more than
        one round), running totals"""
        st = np.zeros(4, np.int64)
        self._L.ssw_first_pass_active_stats(self._m, _ptr(st))
        return tuple(int(x) for x in st)

    def first_pass_active_carry(self, n_utts):
        """ssw_first_pass_active_carry: uint8 [n_utts][n_cb][n_feat][4] codewords, best first --
        what the last align_text_batch_active call with two_pass_history handed on."""
        rows = np.zeros((n_utts, self.n_cb * self.n_feat), np.uint32)
        _check(self._L.ssw_first_pass_active_carry(self._m, n_utts, _ptr(rows)),
               "ssw_first_pass_active_carry")
        return rows.view(np.uint8).reshape(n_utts, self.n_cb, self.n_feat, 4)

    # ---- alignment ----------------------------------------------------------------
    def align_batch(self, d_senscr, frame_off, phone_off, senid, tmatid, sf=None, ef=None,
                    state_init=None, stream=None):
        """Viterbi forced alignment of a batch; returns (states[n,3] int32, status[n_utts]).
        senid: [total_phones][n_emit] (n_emit = the transition matrices', 3 unless the model says
        otherwise)."""
        ne = self.tmat_n_emit
        frame_off = np.ascontiguousarray(frame_off, np.int32)
        phone_off = np.ascontiguousarray(phone_off, np.int32)
        senid = np.ascontiguousarray(senid, np.uint16).reshape(-1, ne)
        n_ph = senid.shape[0]
        tmatid = np.ascontiguousarray(tmatid, np.int16)
        sf = np.zeros(n_ph, np.int32) if sf is None else np.ascontiguousarray(sf, np.int32)
        ef = (np.full(n_ph, INT_MAX, np.int32) if ef is None
              else np.ascontiguousarray(ef, np.int32))
        states = (np.zeros((n_ph * ne, 3), np.int32) if state_init is None
                  else np.ascontiguousarray(state_init, np.int32).copy())
        n_utts = len(frame_off) - 1
        status = np.zeros(n_utts, np.int32)
        _check(self._L.ssw_align_batch(self._m, _ptr(d_senscr), n_utts, _ptr(frame_off),
                                       _ptr(phone_off), _ptr(senid), _ptr(tmatid), _ptr(sf),
                                       _ptr(ef), _ptr(states), _ptr(status),
                                       C.c_void_p(int(stream)) if stream else None),
               "ssw_align_batch")
        return states, status

    def align_batch_active(self, d_feats, frame_off, phone_off, senid, tmatid, sf=None, ef=None,
                           seed_active=None, state_init=None, d_senscr=None, stream=None,
                           scorer=SCORER_PTM):
        """ssw_align_batch_active: scoring over the search's active senones (compallsen = no) +
        forced alignment; returns (states[n,3] int32, status[n_utts])."""
        frame_off = np.ascontiguousarray(frame_off, np.int32)
        phone_off = np.ascontiguousarray(phone_off, np.int32)
        senid = np.ascontiguousarray(senid, np.uint16).reshape(-1, 3)
        n_ph = senid.shape[0]
        tmatid = np.ascontiguousarray(tmatid, np.int16)
        sf = np.zeros(n_ph, np.int32) if sf is None else np.ascontiguousarray(sf, np.int32)
        ef = (np.full(n_ph, INT_MAX, np.int32) if ef is None
              else np.ascontiguousarray(ef, np.int32))
        states = (np.zeros((n_ph * 3, 3), np.int32) if state_init is None
                  else np.ascontiguousarray(state_init, np.int32).copy())
        n_utts = len(frame_off) - 1
        status = np.zeros(n_utts, np.int32)
        seed = None if seed_active is None else np.ascontiguousarray(seed_active, np.uint32)
        _check(self._L.ssw_align_batch_active_ex(
            self._m, int(scorer), _ptr(d_feats), n_utts, _ptr(frame_off), _ptr(phone_off), _ptr(senid),
            _ptr(tmatid), _ptr(sf), _ptr(ef), _ptr(seed), _ptr(states), _ptr(status),
            _ptr(d_senscr), C.c_void_p(int(stream)) if stream else None),
            "ssw_align_batch_active")
        return states, status

    # ---- compact score rows (ssw_amd.h, "Compact score rows") -----------------------
    def compact_plan(self, frame_off, phone_off, senid, stream=None) -> "CompactPlan":
        return CompactPlan(self, frame_off, phone_off, senid, stream)

    def score_batch_compact(self, d_feats, plan, d_compact, stream=None, scorer=SCORER_PTM,
                            share_device=False):
        """ssw_score_batch_compact: the plan's batch scored into rows that hold each
        utterance's own states only; asynchronous on `stream`."""
        _check(self._L.ssw_score_batch_compact(self._m, int(scorer), _ptr(d_feats), plan._p,
                                               _ptr(d_compact),
                                               C.c_void_p(int(stream)) if stream else None,
                                               2 if share_device else 0), "ssw_score_batch_compact")

    def align_batch_compact(self, plan, d_compact, tmatid, sf=None, ef=None, state_init=None,
                            stream=None, out=None):
        """ssw_align_batch_compact; returns (states[n,3] int32, status[n_utts]).  Without
        state_init the entries are output only (SSW_ALIGN_STATE_OUT_ONLY: nothing uploaded;
        `out`, if given, is the int32 [n,3] array they are written to, whatever it held)."""
        n_ph = plan.total_phones
        tmatid = np.ascontiguousarray(tmatid, np.int16)
        assert len(tmatid) == n_ph
        sf = np.zeros(n_ph, np.int32) if sf is None else np.ascontiguousarray(sf, np.int32)
        ef = (np.full(n_ph, INT_MAX, np.int32) if ef is None
              else np.ascontiguousarray(ef, np.int32))
        if state_init is not None:
            states = np.ascontiguousarray(state_init, np.int32).copy()
        elif out is not None:
            assert out.dtype == np.int32 and out.shape == (n_ph * 3, 3) and out.flags.c_contiguous
            states = out
        else:
            states = np.empty((n_ph * 3, 3), np.int32)
        status = np.zeros(plan.n_utts, np.int32)
        _check(self._L.ssw_align_batch_compact(self._m, plan._p, _ptr(d_compact), _ptr(tmatid),
                                               _ptr(sf), _ptr(ef), _ptr(states), _ptr(status),
                                               C.c_void_p(int(stream)) if stream else None,
                                               1 if state_init is None else 0),
               "ssw_align_batch_compact")
        return states, status

    def propagate(self, child, parent, n_parent):
        child = np.ascontiguousarray(child, np.int32)
        parent = np.ascontiguousarray(parent, np.int32)
        out = np.zeros((n_parent, 3), np.int32)
        _check(self._L.ssw_alignment_propagate(_ptr(child), _ptr(parent), len(parent), _ptr(out),
                                               n_parent), "ssw_alignment_propagate")
        return out

    # ---- dynamic features ------------------------------------------------------------
    def feat_batch(self, cep, utt_off=None) -> np.ndarray:
        """Batch CMN + 1s_c_d_dd on the GPU: host MFCC [n][ncep] -> host features [n][3*ncep]."""
        cep = np.ascontiguousarray(cep, np.float32)
        n, ncep = cep.shape
        off = (np.array([0, n], np.int32) if utt_off is None
               else np.ascontiguousarray(utt_off, np.int32))
        out = np.zeros((n, 3 * ncep), np.float32)
        if n == 0:
            return out
        d_in = self.to_device(cep)
        d_out = _check(self._L.ssw_device_malloc(out.nbytes), "ssw_device_malloc")
        try:
            _check(self._L.ssw_feat_batch(self._m, d_in, n, _ptr(off), len(off) - 1, ncep, d_out,
                                          None), "ssw_feat_batch")
            _check(self._L.ssw_memcpy_d2h(_ptr(out), d_out, out.nbytes), "ssw_memcpy_d2h")
        finally:
            self._L.ssw_device_free(d_in)
            self._L.ssw_device_free(d_out)
        return out

    # ---- device memory (no torch needed) --------------------------------------------
    def to_device(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = _check(self._L.ssw_device_malloc(arr.nbytes), "ssw_device_malloc")
        _check(self._L.ssw_memcpy_h2d(p, _ptr(arr), arr.nbytes), "ssw_memcpy_h2d")
        return p

    def device_malloc(self, nbytes: int) -> int:
        return _check(self._L.ssw_device_malloc(int(nbytes)), "ssw_device_malloc")

    def device_free(self, p):
        self._L.ssw_device_free(p)


class CompactPlan:
    """ssw_compact_plan_t: which states each utterance of a batch will be aligned against, so
    that the scorer stores their scores only (0.9 KB instead of 10 KB per frame of en-us)."""

    def __init__(self, model: Model, frame_off, phone_off, senid, stream=None):
        frame_off = np.ascontiguousarray(frame_off, np.int32)
        phone_off = np.ascontiguousarray(phone_off, np.int32)
        senid = np.ascontiguousarray(senid, np.uint16).reshape(-1, 3)
        self._L = model._L
        self.n_utts = len(frame_off) - 1
        self.total_phones = int(phone_off[-1])
        assert len(phone_off) == self.n_utts + 1 and len(senid) == self.total_phones
        self._p = self._L.ssw_compact_plan_create(model._m, self.n_utts, _ptr(frame_off),
                                                  _ptr(phone_off), _ptr(senid),
                                                  C.c_void_p(int(stream)) if stream else None)
        if not self._p:
            raise SswError("ssw_compact_plan_create: " + _lib.last_error())
        self.elems = int(self._L.ssw_compact_plan_elems(self._p))
        self.nbytes = 2 * self.elems

    def rows(self, u):
        """(offset in int16 elements, row length) of utterance u's rows in the compact buffer"""
        off, stride = C.c_longlong(0), C.c_int32(0)
        _check(self._L.ssw_compact_plan_rows(self._p, int(u), C.byref(off), C.byref(stride)),
               "ssw_compact_plan_rows")
        return int(off.value), int(stride.value)

    def free(self):
        if getattr(self, "_p", None):
            self._L.ssw_compact_plan_free(self._p)
            self._p = None

    __del__ = free


class PtmMgau:
    """`mgau_t` stand-in created by ssw_ptm_mgau_init; call through its vtable."""

    def __init__(self, model: Model):
        self._L = _lib.lib()
        self.model = model
        self._g = self._L.ssw_ptm_mgau_init(model._m)
        if not self._g:
            raise SswError("ssw_ptm_mgau_init: " + _lib.last_error())

    @property
    def name(self):
        return self._g.contents.vt.contents.name.decode()

    @property
    def frame_idx(self):
        return self._g.contents.frame_idx

    @frame_idx.setter
    def frame_idx(self, v):
        self._g.contents.frame_idx = int(v)  # acmod writes this field directly

    def reset_hist(self):
        self._L.ssw_mgau_reset_hist(self._g)

    def prescore(self, feats):
        feats = np.ascontiguousarray(feats, np.float32).reshape(-1, self.model.veclen_total)
        _check(self._L.ssw_mgau_prescore(self._g, _ptr(feats), feats.shape[0]),
               "ssw_mgau_prescore")

    def frame_eval(self, feat, frame, compallsen=True, senone_active=None):
        m = self.model
        feat = np.ascontiguousarray(feat, np.float32).reshape(-1)
        streams = (C.POINTER(C.c_float) * m.n_feat)()
        off = 0
        for f in range(m.n_feat):
            streams[f] = C.cast(feat[off:].ctypes.data, C.POINTER(C.c_float))
            off += m.veclen[f]
        out = np.zeros(m.n_sen, np.int16)
        act = None if senone_active is None else np.ascontiguousarray(senone_active, np.uint8)
        rv = self._g.contents.vt.contents.frame_eval(
            C.cast(self._g, C.c_void_p), _ptr(out), _ptr(act),
            0 if act is None else len(act), streams, int(frame), int(bool(compallsen)))
        _check(rv, "frame_eval")
        return out

    def transform(self, mllr=None):
        return self._g.contents.vt.contents.transform(C.cast(self._g, C.c_void_p), None)

    def free(self):
        if getattr(self, "_g", None):
            self._g.contents.vt.contents.free(C.cast(self._g, C.c_void_p))
            self._g = None

    __del__ = free


class StateAlignSearch:
    """state_align_search_init/start/step/finish over one utterance."""

    def __init__(self, model: Model, mgau: PtmMgau | None, ssid, tmatid, start=None,
                 duration=None):
        self._L = _lib.lib()
        self.model = model
        ssid = np.ascontiguousarray(ssid, np.int32)
        tmatid = np.ascontiguousarray(tmatid, np.int32)
        n = len(ssid)
        start = None if start is None else np.ascontiguousarray(start, np.int32)
        duration = None if duration is None else np.ascontiguousarray(duration, np.int32)
        self._s = self._L.ssw_state_align_search_init(
            model._m, mgau._g if mgau is not None else None, n, _ptr(ssid), _ptr(tmatid),
            _ptr(start), _ptr(duration))
        if not self._s:
            raise SswError("ssw_state_align_search_init: " + _lib.last_error())

    def start(self):
        return _check(self._L.ssw_state_align_search_start(self._s), "start")

    def step(self, feat, frame_idx):
        feat = np.ascontiguousarray(feat, np.float32).reshape(-1)
        return _check(self._L.ssw_state_align_search_step(self._s, _ptr(feat), int(frame_idx)),
                      "step")

    def finish(self):
        return _check(self._L.ssw_state_align_search_finish(self._s), "finish")

    def _entries(self, fn):
        n = C.c_int32(0)
        p = fn(self._s, C.byref(n))
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n.value, 3))
        return a.copy()

    def states(self):
        return self._entries(self._L.ssw_state_align_search_states)

    def phones(self):
        return self._entries(self._L.ssw_state_align_search_phones)

    def free(self):
        if getattr(self, "_s", None):
            self._L.ssw_state_align_search_free(self._s)
            self._s = None

    __del__ = free


class MsMgau(PtmMgau):
    """`mgau_t` stand-in created by ssw_ms_mgau_init (the "ms" scorer)."""

    def __init__(self, model: Model):
        self._L = _lib.lib()
        self.model = model
        self._g = self._L.ssw_ms_mgau_init(model._m)
        if not self._g:
            raise SswError("ssw_ms_mgau_init: " + _lib.last_error())


class Lexicon:
    """Pronunciation dictionary + alignment_populate on the host (ssw_dict_load,
    ssw_alignment_populate): words and their windows -> the per-phone rows the aligner takes."""

    def __init__(self, model: Model, dict_path=None, filler_path=None):
        self._L = _lib.lib()
        self.model = model
        enc = lambda s: None if s is None else os.fsencode(s)
        self._d = self._L.ssw_dict_load(model._m, enc(dict_path), enc(filler_path))
        if not self._d:
            raise SswError("ssw_dict_load: " + _lib.last_error())

    def __len__(self):
        return int(self._L.ssw_dict_size(self._d))

    def pron(self, word: str):
        buf = np.zeros(256, np.int32)
        n = self._L.ssw_dict_pron(self._d, word.encode(), _ptr(buf), 256)
        if n < 0:
            raise KeyError(word)
        return [self._L.ssw_ciphone_name(self.model._m, int(c)).decode() for c in buf[:n]]

    def phone_id_nearest(self, b, l, r, pos):
        return int(self._L.ssw_phone_id_nearest(self.model._m, int(b), int(l), int(r), int(pos)))

    def populate(self, words, start=None, duration=None, max_phones=21845):
        """Returns dict(ssid, tmatid, cipid, parent, start, duration) of int32 arrays."""
        n = len(words)
        arr = (C.c_char_p * n)(*[w.encode() for w in words])
        st = None if start is None else np.ascontiguousarray(start, np.int32)
        du = None if duration is None else np.ascontiguousarray(duration, np.int32)
        out = {k: np.zeros(max_phones, np.int32)
               for k in ("ssid", "tmatid", "cipid", "parent", "start", "duration")}
        k = self._L.ssw_alignment_populate(self.model._m, self._d, n, arr, _ptr(st), _ptr(du),
                                           max_phones, _ptr(out["ssid"]), _ptr(out["tmatid"]),
                                           _ptr(out["cipid"]), _ptr(out["parent"]),
                                           _ptr(out["start"]), _ptr(out["duration"]))
        _check(k, "ssw_alignment_populate")
        return {key: v[:k].copy() for key, v in out.items()}

    def alignment_json(self, hyp, words, word_al, cipid, parent, phone_al, n_frames,
                       state_senid=None, state_al=None, hyp_logprob=0, utt_start=0.0, frate=100):
        """decoder_result_json at align_level 1 (or 2 with the state level): the reference's
        one-line JSON for an alignment.  *_al: int32 [n][3] (start, duration, score)."""
        n = len(words)
        arr = (C.c_char_p * n)(*[w.encode() for w in words])
        wa = np.ascontiguousarray(word_al, np.int32).reshape(-1, 3)
        pa = np.ascontiguousarray(phone_al, np.int32).reshape(-1, 3)
        ci = np.ascontiguousarray(cipid, np.int32)
        par = np.ascontiguousarray(parent, np.int32)
        sid = None if state_senid is None else np.ascontiguousarray(state_senid, np.uint16)
        sa = None if state_al is None else np.ascontiguousarray(state_al, np.int32).reshape(-1, 3)
        args = (self.model._m, hyp.encode(), int(hyp_logprob), float(utt_start), int(frate),
                int(n_frames), n, arr, _ptr(wa), len(ci), _ptr(ci), _ptr(par), _ptr(pa), _ptr(sid),
                _ptr(sa))
        need = self._L.ssw_alignment_json(*args, None, 0)
        _check(need, "ssw_alignment_json")
        buf = C.create_string_buffer(need + 1)
        _check(self._L.ssw_alignment_json(*args, buf, need + 1), "ssw_alignment_json")
        return buf.value.decode()

    def word(self, wid: int):
        w = self._L.ssw_dict_word(self._d, int(wid))
        return None if w is None else w.decode()

    def word_id(self, word: str) -> int:
        return int(self._L.ssw_dict_word_id(self._d, word.encode()))

    def first_pass_config(self, **kw) -> FirstPassConfig:
        cfg = FirstPassConfig()
        self._L.ssw_first_pass_config_defaults(C.byref(cfg))
        for k, v in kw.items():
            setattr(cfg, k, v)
        return cfg

    def first_pass_graph(self, words, cfg=None, max_nodes=1 << 16):
        """The phone-tree HMMs the first pass searches for one text (host only): a structured
        array (FP_NODE_DTYPE) and the three beams."""
        arr = (C.c_char_p * len(words))(*[w.encode() for w in words])
        nodes = np.zeros(max_nodes, FP_NODE_DTYPE)
        beams = np.zeros(3, np.int32)
        n = self._L.ssw_first_pass_graph(self.model._m, self._d,
                                         None if cfg is None else C.byref(cfg), len(words), arr,
                                         max_nodes, _ptr(nodes), _ptr(beams))
        _check(n, "ssw_first_pass_graph")
        return nodes[:n].copy(), beams

    def first_pass(self, d_senscr, utt_off, texts, cfg=None, max_seg=None, stream=None):
        """ssw_first_pass_batch: device senone scores of a batch + one word list per utterance
        -> per utterance a list of (word, start, duration, score), or None when the grammar's
        final state is not reached."""
        n_seg, seg = self.first_pass_raw(d_senscr, utt_off, texts, cfg, max_seg, stream)
        out = []
        for u in range(len(n_seg)):
            if n_seg[u] <= -2:
                raise SswError(f"first_pass: utterance {u} needs max_seg >= {-n_seg[u] - 2}")
            if n_seg[u] < 0:
                out.append(None)
            else:
                out.append([(self.word(int(s["wid"])), int(s["start"]), int(s["duration"]),
                             int(s["score"])) for s in seg[u, :n_seg[u]]])
        return out

    def first_pass_raw(self, d_senscr, utt_off, texts, cfg=None, max_seg=None, stream=None):
        """The C call alone: (n_seg int32 [n_utts], seg WORD_SEG_DTYPE [n_utts][max_seg])."""
        off = np.ascontiguousarray(utt_off, np.int32)
        n_utts = len(off) - 1
        assert len(texts) == n_utts
        word_off = np.zeros(n_utts + 1, np.int32)
        word_off[1:] = np.cumsum([len(t) for t in texts])
        flat = [w.encode() for t in texts for w in t]
        arr = (C.c_char_p * max(1, len(flat)))(*flat)
        if max_seg is None:
            max_seg = 4 * max(len(t) for t in texts) + 8
        n_seg = np.zeros(n_utts, np.int32)
        seg = np.zeros((n_utts, max_seg), WORD_SEG_DTYPE)
        _check(self._L.ssw_first_pass_batch(self.model._m, self._d,
                                            None if cfg is None else C.byref(cfg),
                                            _ptr(d_senscr), int(off[-1]), _ptr(off), n_utts,
                                            _ptr(word_off), arr, max_seg, _ptr(n_seg), _ptr(seg),
                                            _ptr(stream)), "ssw_first_pass_batch")
        return n_seg, seg

    def first_pass_active(self, d_feats, utt_off, texts, cfg=None, max_seg=None, stream=None,
                          scorer=SCORER_PTM, d_senscr=None, want_seed=False):
        """ssw_first_pass_batch_active: the first pass in the reference's DEFAULT configuration
        (compallsen = no) for a batch, from feature rows in HBM.  Returns (segmentations as
        first_pass does, rounds int32 [n_utts]) and, with want_seed, the uint32
        [n_utts][(n_sen + 31) // 32] set acmod holds after each utterance's last frame; d_senscr
        (device int16 [n_frames][n_sen]) receives the rows as acmod's buffer would hold them."""
        n_seg, seg, rounds, seed = self.first_pass_active_raw(d_feats, utt_off, texts, cfg, max_seg,
                                                              stream, scorer, d_senscr, want_seed)
        out = []
        for u in range(len(n_seg)):
            if n_seg[u] <= -2:
                raise SswError(f"first_pass_active: utterance {u} needs max_seg >= {-n_seg[u] - 2}")
            if n_seg[u] < 0:
                out.append(None)
            else:
                out.append([(self.word(int(s["wid"])), int(s["start"]), int(s["duration"]),
                             int(s["score"])) for s in seg[u, :n_seg[u]]])
        return (out, rounds, seed) if want_seed else (out, rounds)

    def first_pass_active_raw(self, d_feats, utt_off, texts, cfg=None, max_seg=None, stream=None,
                              scorer=SCORER_PTM, d_senscr=None, want_seed=False):
        """The C call alone: (n_seg, seg WORD_SEG_DTYPE [n_utts][max_seg], rounds, seed or None);
        texts: list of word lists, or a Texts."""
        off = np.ascontiguousarray(utt_off, np.int32)
        n_utts = len(off) - 1
        tx = texts if isinstance(texts, Texts) else Texts(texts)
        assert tx.n_utts == n_utts
        if max_seg is None:
            max_seg = 4 * tx.max_words + 8
        n_seg = np.zeros(n_utts, np.int32)
        seg = np.zeros((n_utts, max_seg), WORD_SEG_DTYPE)
        rounds = np.zeros(n_utts, np.int32)
        seed = np.zeros((n_utts, (self.model.n_sen + 31) // 32), np.uint32) if want_seed else None
        _check(self._L.ssw_first_pass_batch_active(
            self.model._m, self._d, None if cfg is None else C.byref(cfg), int(scorer),
            _ptr(d_feats), int(off[-1]), _ptr(off), n_utts, _ptr(tx.word_off), tx.arr, max_seg,
            _ptr(n_seg), _ptr(seg), _ptr(seed), _ptr(d_senscr), _ptr(rounds), _ptr(stream)),
            "ssw_first_pass_batch_active")
        return n_seg, seg, rounds, seed

    def free(self):
        if getattr(self, "_d", None):
            self._L.ssw_dict_free(self._d)
            self._d = None

    __del__ = free


def _view(ptr, n, dtype, cols=None):
    if n <= 0 or not ptr:
        return np.zeros((0, cols) if cols else 0, dtype)
    count = n * (cols or 1)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dtype, count=count).copy()
    return arr.reshape(n, cols) if cols else arr


class AlignmentSet:
    """ssw_alignment_set_t: the alignment_t-shaped result of ssw_forced_align_batch."""

    def __init__(self, lib, handle, lex):
        self._L, self._a, self._lex = lib, handle, lex

    def status(self, u):
        return int(self._L.ssw_alignment_set_status(self._a, u))

    def message(self, u):
        """Why utterance u's text was rejected (status 3), else ''."""
        return self._L.ssw_alignment_set_message(self._a, u).decode()

    def utterance(self, u):
        """None unless status 0; else dict(words, word_al, cipid, parent, phone_al, senid,
        state_al) with *_al int32 [n][3] = (start, duration, score)."""
        if self.status(u) != 0:
            return None
        p1, p2, p3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = self._L.ssw_alignment_set_words(self._a, u, C.byref(p1), C.byref(p2))
        wid = _view(p1, n, np.int32)
        word_al = _view(p2, n, np.int32, 3)
        n = self._L.ssw_alignment_set_phones(self._a, u, C.byref(p1), C.byref(p2), C.byref(p3))
        cipid, parent, phone_al = _view(p1, n, np.int32), _view(p2, n, np.int32), _view(p3, n, np.int32, 3)
        n = self._L.ssw_alignment_set_states(self._a, u, C.byref(p1), C.byref(p2))
        senid, state_al = _view(p1, n, np.uint16), _view(p2, n, np.int32, 3)
        return {"wid": wid, "words": [self._lex.word(int(w)) for w in wid], "word_al": word_al,
                "cipid": cipid, "parent": parent, "phone_al": phone_al,
                "senid": senid.reshape(-1, 3), "state_al": state_al}

    def json(self, u, align_level=1, utt_start=0.0, frate=100):
        """ssw_alignment_set_json: the reference's one-line JSON for utterance u."""
        need = self._L.ssw_alignment_set_json(self._a, u, float(utt_start), frate, align_level,
                                              None, 0)
        _check(need, "ssw_alignment_set_json")
        buf = C.create_string_buffer(need + 1)
        _check(self._L.ssw_alignment_set_json(self._a, u, float(utt_start), frate, align_level,
                                              buf, need + 1), "ssw_alignment_set_json")
        return buf.value.decode()

    def free(self):
        if self._a:
            self._L.ssw_alignment_set_free(self._a)
            self._a = None

    __del__ = free


def forced_align_batch(model: Model, lex: Lexicon, d_senscr, utt_off, texts, cfg=None,
                       stream=None) -> AlignmentSet:
    """ssw_forced_align_batch: decoder_alignment (src/decoder.c:737-798) for a batch of
    utterances whose senone scores are in HBM -- first pass, populate with its word windows,
    constrained state alignment, propagate -- in one C call."""
    off = np.ascontiguousarray(utt_off, np.int32)
    n_utts = len(off) - 1
    word_off = np.zeros(n_utts + 1, np.int32)
    word_off[1:] = np.cumsum([len(t) for t in texts])
    flat = [w.encode() for t in texts for w in t]
    arr = (C.c_char_p * max(1, len(flat)))(*flat)
    L = _lib.lib()
    h = L.ssw_forced_align_batch(model._m, lex._d, None if cfg is None else C.byref(cfg),
                                 _ptr(d_senscr), int(off[-1]), _ptr(off), n_utts, _ptr(word_off),
                                 arr, _ptr(stream))
    if not h:
        raise SswError("ssw_forced_align_batch: " + _lib.last_error())
    return AlignmentSet(L, h, lex)


class FirstPassPlan:
    """ssw_first_pass_plan_t: the graphs of a batch of texts, built on the host without touching
    the device (ctypes releases the GIL: prepare the next batch on another thread while the GPU
    works on this one)."""

    def __init__(self, model: Model, lex: Lexicon, texts, cfg=None):
        self._L = _lib.lib()
        n_utts = len(texts)
        word_off = np.zeros(n_utts + 1, np.int32)
        word_off[1:] = np.cumsum([len(t) for t in texts])
        flat = [w.encode() for t in texts for w in t]
        arr = (C.c_char_p * max(1, len(flat)))(*flat)
        self.n_utts = n_utts
        self._p = self._L.ssw_first_pass_prepare(model._m, lex._d,
                                                 None if cfg is None else C.byref(cfg), n_utts,
                                                 _ptr(word_off), arr)
        if not self._p:
            raise SswError("ssw_first_pass_prepare: " + _lib.last_error())

    def free(self):
        if getattr(self, "_p", None):
            self._L.ssw_first_pass_plan_free(self._p)
            self._p = None

    __del__ = free


def forced_align_planned(model: Model, lex: Lexicon, plan: FirstPassPlan, d_senscr, utt_off,
                         stream=None) -> AlignmentSet:
    """ssw_forced_align_planned: forced_align_batch with the graphs prepared beforehand."""
    off = np.ascontiguousarray(utt_off, np.int32)
    assert len(off) - 1 == plan.n_utts
    L = _lib.lib()
    h = L.ssw_forced_align_planned(model._m, lex._d, plan._p, _ptr(d_senscr), int(off[-1]),
                                   _ptr(off), _ptr(stream))
    if not h:
        raise SswError("ssw_forced_align_planned: " + _lib.last_error())
    return AlignmentSet(L, h, lex)


class Texts:
    """The texts of a batch as the C calls take them (word_off + char **), marshalled once:
    turning a few thousand Python strings into a ctypes array costs about a millisecond, which a
    caller that aligns the same texts again (or a C host) does not pay."""

    def __init__(self, texts):
        self.n_utts = len(texts)
        self.word_off = np.zeros(self.n_utts + 1, np.int32)
        self.word_off[1:] = np.cumsum([len(t) for t in texts])
        self._flat = [w.encode() for t in texts for w in t]
        self.arr = (C.c_char_p * max(1, len(self._flat)))(*self._flat)
        self.max_words = max((len(t) for t in texts), default=0)

    def __len__(self):
        return self.n_utts


def align_text_batch(model: Model, lex: Lexicon, d_feats, utt_off, texts, cfg=None,
                     scorer=SCORER_PTM, stream=None) -> AlignmentSet:
    """ssw_align_text_batch: feature rows in HBM + texts (list of word lists, or a Texts) ->
    alignments (scoring, first pass, populate, constrained state alignment, propagate) in one C
    call."""
    off = np.ascontiguousarray(utt_off, np.int32)
    n_utts = len(off) - 1
    tx = texts if isinstance(texts, Texts) else Texts(texts)
    assert tx.n_utts == n_utts
    L = _lib.lib()
    h = L.ssw_align_text_batch(model._m, lex._d, None if cfg is None else C.byref(cfg), scorer,
                               _ptr(d_feats), int(off[-1]), _ptr(off), n_utts, _ptr(tx.word_off),
                               tx.arr, _ptr(stream))
    if not h:
        raise SswError("ssw_align_text_batch: " + _lib.last_error())
    return AlignmentSet(L, h, lex)


def align_text_batch_active(model: Model, lex: Lexicon, d_feats, utt_off, texts, cfg=None,
                            scorer=SCORER_PTM, stream=None) -> AlignmentSet:
    """ssw_align_text_batch_active: align_text_batch in the reference's DEFAULT configuration
    (compallsen = no): both passes score what their searches hold active."""
    off = np.ascontiguousarray(utt_off, np.int32)
    n_utts = len(off) - 1
    tx = texts if isinstance(texts, Texts) else Texts(texts)
    assert tx.n_utts == n_utts
    L = _lib.lib()
    h = L.ssw_align_text_batch_active(model._m, lex._d, None if cfg is None else C.byref(cfg),
                                      scorer, _ptr(d_feats), int(off[-1]), _ptr(off), n_utts,
                                      _ptr(tx.word_off), tx.arr, _ptr(stream))
    if not h:
        raise SswError("ssw_align_text_batch_active: " + _lib.last_error())
    return AlignmentSet(L, h, lex)


def forced_alignment(model: Model, lex: Lexicon, d_senscr, utt_off, texts, cfg=None, stream=None):
    """forced_align_batch, unpacked: one dict per utterance, None where it could not be aligned."""
    s = forced_align_batch(model, lex, d_senscr, utt_off, texts, cfg=cfg, stream=stream)
    out = [s.utterance(u) for u in range(len(texts))]
    s.free()
    return out
