"""ctypes front for the CPU oracle (oracle/ssw_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under soundswallower_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORC_NDIMS = 16
DIM_NAMES = ("n_cb", "n_feat", "n_density", "veclen_total", "n_sen", "n_ci_sen", "n_ciphone",
             "n_phone", "n_emit_state", "n_tmat", "n_sseq", "sil", "n_floored", "n_cd_tree",
             "has_ptm_mixw", "has_ms_pdf")


class OrcConfig(C.Structure):
    _fields_ = [("logbase", C.c_double), ("varfloor", C.c_double), ("mixwfloor", C.c_double),
                ("tmatfloor", C.c_double), ("topn", C.c_int32), ("ds", C.c_int32),
                ("aw", C.c_int32)]


class OrcLogmath(C.Structure):
    _fields_ = [("base", C.c_double), ("log_of_base", C.c_double), ("log10_of_base", C.c_double),
                ("inv_log_of_base", C.c_double), ("inv_log10_of_base", C.c_double),
                ("zero", C.c_int32), ("shift", C.c_int), ("width", C.c_int),
                ("table_size", C.c_uint32), ("table", C.c_void_p)]


def build(force: bool = False) -> str:
    """Compile the oracle if needed; returns the .so path."""
    so = os.path.join(_HERE, "libssw_oracle.so")
    src = os.path.join(_HERE, "ssw_oracle.c")
    hdr = os.path.join(_HERE, "ssw_oracle.h")
    fe = os.path.join(_HERE, "ssw_oracle_fe.c")
    stale = (not os.path.exists(so)
             or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr),
                                           os.path.getmtime(fe)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libssw_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.environ.get("SSW_ORACLE_LIB") or build()
    L = C.CDLL(so)
    vp, i32, f64 = C.c_void_p, C.c_int32, C.c_double
    L.orc_logmath_init.restype = C.POINTER(OrcLogmath)
    L.orc_logmath_init.argtypes = [f64, C.c_int, C.c_int]
    L.orc_logmath_free.argtypes = [C.POINTER(OrcLogmath)]
    L.orc_logmath_log.argtypes = [C.POINTER(OrcLogmath), f64]
    L.orc_logmath_ln_to_log.argtypes = [C.POINTER(OrcLogmath), f64]
    L.orc_logmath_exp.restype = f64
    L.orc_logmath_exp.argtypes = [C.POINTER(OrcLogmath), C.c_int]
    L.orc_logmath_add.argtypes = [C.POINTER(OrcLogmath), C.c_int, C.c_int]
    L.orc_logmath_table.restype = C.c_uint32
    L.orc_logmath_table.argtypes = [C.POINTER(OrcLogmath), vp, C.c_uint32]
    L.orc_config_defaults.argtypes = [C.POINTER(OrcConfig)]
    L.orc_model_load.restype = vp
    L.orc_model_load.argtypes = [C.c_char_p] * 6 + [C.POINTER(OrcConfig)]
    L.orc_model_free.argtypes = [vp]
    L.orc_last_error.restype = C.c_char_p
    L.orc_model_dims.argtypes = [vp, vp]
    for name in ("veclen", "mean", "var", "det", "ptm_mixw", "ms_pdf", "tp", "sseq", "sen2cimap",
                 "phone_ssid", "phone_tmat"):
        fn = getattr(L, "orc_model_" + name)
        fn.restype = vp
        fn.argtypes = [vp]
    L.orc_model_lmath.restype = C.POINTER(OrcLogmath)
    L.orc_model_lmath.argtypes = [vp]
    L.orc_model_lmath_8b.restype = C.POINTER(OrcLogmath)
    L.orc_model_lmath_8b.argtypes = [vp]
    L.orc_ptm_reset.argtypes = [vp]
    L.orc_ptm_set_frame_idx.argtypes = [vp, C.c_int]
    L.orc_ptm_frame_eval.argtypes = [vp, vp, vp, i32, vp, i32, i32]
    L.orc_ptm_get_topn.argtypes = [vp, C.c_int, vp, vp]
    L.orc_ptm_score_utt.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.orc_ms_frame_eval.argtypes = [vp, vp, vp, i32, vp, i32, i32]
    L.orc_ms_score_utt.argtypes = [vp, vp, C.c_int, vp]
    L.orc_flags2list.argtypes = [vp, C.c_int, vp]
    L.orc_state_align.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp,
                                  vp, vp, vp]
    L.orc_fe_mfcc.argtypes = [vp, C.c_size_t, C.c_int, f64, f64, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.orc_feat_1s_c_d_dd.argtypes = [vp, C.c_int, vp]
    L.orc_feat_1s_c_d_dd_ex.argtypes = [vp, C.c_int, vp, C.c_int]
    L.orc_hmm_vit_eval.restype = i32
    L.orc_scan_replay.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    L.orc_scan_replay.restype = None
    L.orc_hmm_vit_eval.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp]
    _LIB = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _view(addr, shape, dtype):
    n = int(np.prod(shape))
    if not addr or n == 0:
        return None
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


class Logmath:
    """orc_logmath_t wrapper (restates src/logmath.c)."""

    def __init__(self, base=1.0001, shift=0, use_table=True):
        self._l = lib()
        self._p = self._l.orc_logmath_init(base, shift, int(use_table))
        if not self._p:
            raise ValueError(self._l.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "_p", None):
            self._l.orc_logmath_free(self._p)
            self._p = None

    zero = property(lambda s: s._p.contents.zero)
    width = property(lambda s: s._p.contents.width)
    table_size = property(lambda s: s._p.contents.table_size)

    def log(self, p):
        return self._l.orc_logmath_log(self._p, float(p))

    def ln_to_log(self, p):
        return self._l.orc_logmath_ln_to_log(self._p, float(p))

    def exp(self, x):
        return self._l.orc_logmath_exp(self._p, int(x))

    def add(self, x, y):
        return self._l.orc_logmath_add(self._p, int(x), int(y))

    def table(self):
        out = np.zeros(self.table_size, dtype=np.uint32)
        self._l.orc_logmath_table(self._p, _ptr(out), out.size)
        return out


def _table_of(lm_ptr):
    out = np.zeros(lm_ptr.contents.table_size, dtype=np.uint32)
    lib().orc_logmath_table(lm_ptr, _ptr(out), out.size)
    return out


class Model:
    """A model loaded by the oracle's own loaders."""

    def __init__(self, model_dir=None, *, mdef=None, means=None, vars=None, sendump=None,
                 mixw=None, tmat=None, config=None):
        L = lib()
        if model_dir is not None:
            j = lambda n: os.path.join(model_dir, n)
            mdef = mdef or j("mdef")
            means = means or j("means")
            vars = vars or j("variances")
            tmat = tmat or j("transition_matrices")
            if sendump is None and mixw is None:
                if os.path.exists(j("sendump")):
                    sendump = j("sendump")
                else:
                    mixw = j("mixture_weights")
        cfg = OrcConfig()
        L.orc_config_defaults(C.byref(cfg))
        for k, v in (config or {}).items():
            setattr(cfg, k, v)
        enc = lambda s: None if s is None else os.fsencode(s)
        self._l = L
        self._m = L.orc_model_load(enc(mdef), enc(means), enc(vars), enc(sendump), enc(mixw),
                                   enc(tmat), C.byref(cfg))
        if not self._m:
            raise RuntimeError("oracle model load failed: " + L.orc_last_error().decode())
        self.topn = cfg.topn
        d = np.zeros(ORC_NDIMS, dtype=np.int32)
        L.orc_model_dims(self._m, _ptr(d))
        self.dims = dict(zip(DIM_NAMES, (int(x) for x in d)))
        for k, v in self.dims.items():
            setattr(self, k, v)

    def __del__(self):
        if getattr(self, "_m", None):
            self._l.orc_model_free(self._m)
            self._m = None

    # ---- tables -------------------------------------------------------------------
    @property
    def veclen(self):
        return _view(self._l.orc_model_veclen(self._m), (self.n_feat,), np.int32)

    def _gau(self, fn):
        return _view(fn(self._m), (self.n_cb * self.n_density * self.veclen_total,), np.float32)

    mean = property(lambda s: s._gau(s._l.orc_model_mean))
    var = property(lambda s: s._gau(s._l.orc_model_var))
    det = property(lambda s: _view(s._l.orc_model_det(s._m),
                                   (s.n_cb, s.n_feat, s.n_density), np.float32))
    ptm_mixw = property(lambda s: _view(s._l.orc_model_ptm_mixw(s._m),
                                        (s.n_feat, s.n_density, s.n_sen), np.uint8))
    ms_pdf = property(lambda s: _view(s._l.orc_model_ms_pdf(s._m),
                                      (s.n_sen, s.n_feat, s.n_density), np.uint8))
    tp = property(lambda s: _view(s._l.orc_model_tp(s._m),
                                  (s.n_tmat, s.n_emit_state, s.n_emit_state + 1), np.uint8))
    sseq = property(lambda s: _view(s._l.orc_model_sseq(s._m),
                                    (s.n_sseq, s.n_emit_state), np.uint16))
    sen2cimap = property(lambda s: _view(s._l.orc_model_sen2cimap(s._m), (s.n_sen,), np.int16))
    phone_ssid = property(lambda s: _view(s._l.orc_model_phone_ssid(s._m), (s.n_phone,), np.int32))
    phone_tmat = property(lambda s: _view(s._l.orc_model_phone_tmat(s._m), (s.n_phone,), np.int32))
    logadd_table = property(lambda s: _table_of(s._l.orc_model_lmath(s._m)))
    logadd_table_8b = property(lambda s: _table_of(s._l.orc_model_lmath_8b(s._m)))

    def mean4(self):
        """means as [cb][feat][density][veclen] (equal stream lengths only)."""
        vl = self.veclen
        assert len(set(vl.tolist())) == 1
        return self.mean.reshape(self.n_cb, self.n_feat, self.n_density, int(vl[0]))

    # ---- PTM ----------------------------------------------------------------------
    def ptm_reset(self):
        self._l.orc_ptm_reset(self._m)

    def ptm_set_frame_idx(self, i):
        self._l.orc_ptm_set_frame_idx(self._m, int(i))

    def ptm_frame_eval(self, feat, frame, compallsen=True, senone_active=None):
        feat = np.ascontiguousarray(feat, dtype=np.float32).reshape(-1)
        out = np.zeros(self.n_sen, dtype=np.int16)
        n_act = 0 if senone_active is None else len(senone_active)
        act = None if senone_active is None else np.ascontiguousarray(senone_active, np.uint8)
        self._l.orc_ptm_frame_eval(self._m, _ptr(out), _ptr(act), n_act, _ptr(feat), int(frame),
                                   int(bool(compallsen)))
        return out

    def ptm_get_topn(self, frame):
        shape = (self.n_cb, self.n_feat, self.topn)
        cw = np.zeros(shape, np.int32)
        sc = np.zeros(shape, np.int32)
        self._l.orc_ptm_get_topn(self._m, int(frame), _ptr(cw), _ptr(sc))
        return cw, sc

    def ptm_score_utt(self, feats, want_topn=False):
        feats = np.ascontiguousarray(feats, dtype=np.float32).reshape(-1, self.veclen_total)
        n = feats.shape[0]
        out = np.zeros((n, self.n_sen), np.int16)
        if want_topn:
            shape = (n, self.n_cb, self.n_feat, self.topn)
            cw = np.zeros(shape, np.int32)
            sc = np.zeros(shape, np.int32)
            self._l.orc_ptm_score_utt(self._m, _ptr(feats), n, _ptr(out), _ptr(cw), _ptr(sc))
            return out, cw, sc
        self._l.orc_ptm_score_utt(self._m, _ptr(feats), n, _ptr(out), None, None)
        return out

    def ptm_score_chain(self, feats, utt_off, reset=True):
        """Several utterances scored one after the other by ONE scorer, as a decoder does:
        acmod_start_utt puts frame_idx back to 0 and never resets the top-N history
        (src/acmod.c:367), so frame 0 of an utterance copies history slot 1 of the two-slot ring
        (src/ptm_mgau.c:425-437) -- the last odd-numbered frame scored before it."""
        feats = np.ascontiguousarray(feats, dtype=np.float32).reshape(-1, self.veclen_total)
        if reset:
            self.ptm_reset()
        out = np.zeros((len(feats), self.n_sen), np.int16)
        for u in range(len(utt_off) - 1):
            self.ptm_set_frame_idx(0)
            for t in range(int(utt_off[u]), int(utt_off[u + 1])):
                i = t - int(utt_off[u])
                out[t] = self.ptm_frame_eval(feats[t], i)
                self.ptm_set_frame_idx(i + 1)
        return out

    # ---- ms -----------------------------------------------------------------------
    def ms_score_utt(self, feats):
        feats = np.ascontiguousarray(feats, dtype=np.float32).reshape(-1, self.veclen_total)
        out = np.zeros((feats.shape[0], self.n_sen), np.int16)
        self._l.orc_ms_score_utt(self._m, _ptr(feats), feats.shape[0], _ptr(out))
        return out

    def ms_frame_eval(self, feat, frame=0, compallsen=True, senone_active=None):
        feat = np.ascontiguousarray(feat, dtype=np.float32).reshape(-1)
        out = np.zeros(self.n_sen, dtype=np.int16)
        n_act = 0 if senone_active is None else len(senone_active)
        act = None if senone_active is None else np.ascontiguousarray(senone_active, np.uint8)
        self._l.orc_ms_frame_eval(self._m, _ptr(out), _ptr(act), n_act, _ptr(feat), int(frame),
                                  int(bool(compallsen)))
        return out

    # ---- alignment ----------------------------------------------------------------
    def state_align(self, senscr, senid, tmatid, sf=None, ef=None, state_init=None, tp=None,
                    want_trace=False):
        """Returns (rv, states[n_states,3], phones[n_phones,3]) as int32 (start,duration,score)."""
        senscr = np.ascontiguousarray(senscr, dtype=np.int16)
        n_frames, n_sen = senscr.shape
        senid = np.ascontiguousarray(senid, dtype=np.uint16)
        n_phones, n_emit = senid.shape
        tmatid = np.ascontiguousarray(tmatid, dtype=np.int16)
        sf = np.zeros(n_phones, np.int32) if sf is None else np.ascontiguousarray(sf, np.int32)
        ef = (np.full(n_phones, 2**31 - 1, np.int32) if ef is None
              else np.ascontiguousarray(ef, np.int32))
        st = (np.zeros((n_phones * n_emit, 3), np.int32) if state_init is None
              else np.ascontiguousarray(state_init, np.int32).copy())
        ph = np.zeros((n_phones, 3), np.int32)
        tpv = None if tp is None else np.ascontiguousarray(tp, np.uint8)
        trace = np.zeros(max(n_frames, 1), np.int32) if want_trace else None
        rv = self._l.orc_state_align(self._m, _ptr(tpv), _ptr(senscr), n_sen, n_frames, n_phones,
                                     n_emit, _ptr(senid), _ptr(tmatid), _ptr(sf), _ptr(ef),
                                     _ptr(st), _ptr(ph), _ptr(trace))
        if want_trace:
            return rv, st, ph, trace
        return rv, st, ph


def flags2list(vec, n_sen):
    vec = np.ascontiguousarray(vec, dtype=np.uint32)
    out = np.zeros(n_sen + 64, np.uint8)
    n = lib().orc_flags2list(_ptr(vec), int(n_sen), _ptr(out))
    return out[:n].copy()


def hmm_vit_eval(tp, senscr, senid, score, history, out_score, out_history):
    tp = np.ascontiguousarray(tp, np.uint8)
    n_emit = tp.shape[0]
    senscr = np.ascontiguousarray(senscr, np.int16)
    senid = np.ascontiguousarray(senid, np.uint16)
    score = np.ascontiguousarray(score, np.int32).copy()
    history = np.ascontiguousarray(history, np.int32).copy()
    out = np.array([out_score, out_history], np.int32)
    best = lib().orc_hmm_vit_eval(n_emit, _ptr(tp), _ptr(senscr), _ptr(senid), _ptr(score),
                                  _ptr(history), _ptr(out))
    return best, score, history, int(out[0]), int(out[1])


def fe_mfcc(pcm, nfilt=40, lowerf=133.33334, upperf=6855.4976, lifter=0, remove_noise=False,
            transform="legacy"):
    """Whole-buffer MFCC of int16 PCM with the reference's default front end (16 kHz)."""
    pcm = np.ascontiguousarray(pcm, np.int16)
    max_fr = len(pcm) // 160 + 3
    cep = np.zeros((max_fr, 13), np.float32)
    n = lib().orc_fe_mfcc(_ptr(pcm), len(pcm), int(nfilt), float(lowerf), float(upperf),
                          int(lifter), int(bool(remove_noise)), int(transform == "legacy"),
                          _ptr(cep), max_fr)
    return cep[:n].copy()


def feat_1s_c_d_dd(cep, cmn=True):
    """Batch CMN + 1s_c_d_dd dynamic features: [n][13] -> [n][39].  cmn=False: "-cmn none"."""
    cep = np.ascontiguousarray(cep, np.float32).copy()
    out = np.zeros((cep.shape[0], 39), np.float32)
    lib().orc_feat_1s_c_d_dd_ex(_ptr(cep), cep.shape[0], _ptr(out), 1 if cmn else 0)
    return out


def _bind_mdef(L):
    L.orc_mdef_ciphone_id.argtypes = [C.c_void_p, C.c_char_p]
    L.orc_mdef_phone_id_nearest.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]


def ciphone_id(model: "Model", name: str) -> int:
    L = lib()
    _bind_mdef(L)
    return L.orc_mdef_ciphone_id(model._m, name.encode())


def phone_id_nearest(model: "Model", b: int, l: int, r: int, pos: int) -> int:
    """bin_mdef_phone_id_nearest; pos: 0 internal, 1 begin, 2 end, 3 single."""
    L = lib()
    _bind_mdef(L)
    return L.orc_mdef_phone_id_nearest(model._m, int(b), int(l), int(r), int(pos))


def scan_replay(rec, recq, x, veclen):
    """rec / recq: float32 [n_density][32] of one codebook-stream; x: float32 [n][>= veclen].
    Returns (ref, key), float32 [n][n_density]: the reference's density value and the GPU
    scan's quadratic-form key, each with the exact fp32 operation order."""
    rec = np.ascontiguousarray(rec, np.float32).reshape(-1, 32)
    recq = np.ascontiguousarray(recq, np.float32).reshape(-1, 32)
    x = np.ascontiguousarray(x, np.float32)
    n, nd = x.shape[0], rec.shape[0]
    ref = np.empty((n, nd), np.float32)
    key = np.empty((n, nd), np.float32)
    lib().orc_scan_replay(_ptr(rec), _ptr(recq), nd, veclen, _ptr(x), n, x.shape[1], _ptr(ref),
                          _ptr(key))
    return ref, key
