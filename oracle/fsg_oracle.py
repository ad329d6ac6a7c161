"""CPU restatement of SoundSwallower's FIRST pass for forced alignment -- TEST INFRASTRUCTURE ONLY.

`decoder_set_align_text` (src/decoder.c:686-735) turns the text into a linear FSG (state i
--word i--> state i+1); `fsg_search_init` (src/fsg_search.c:172-253) adds silence / filler
self-loops on every state and one parallel link per alternate pronunciation;
`fsg_lextree_init` (src/fsg_lextree.c:219-276) hangs a phone tree of HMMs off every state; the
frame loop `fsg_search_step` (src/fsg_search.c:665-745) is a beam-pruned Viterbi over those
trees with a word-exit history table (src/fsg_history.c); `fsg_search_seg_iter`
(src/fsg_search.c:1085-1143) backtraces the words with their frames, which `decoder_alignment`
(src/decoder.c:737-798) hands to the second pass as word windows.

This file follows the reference's data structures literally (lists in the reference's order,
the history table with its right-context subtraction) so that it can check the GPU kernel, which
is organised differently (dense arrays).  Small sizes only: pure-Python loops over the active
HMMs, with the HMM step itself in C (orc_hmm_vit_eval_many).

Exact score ties are real, not a corner: dictionaries list alternates with IDENTICAL
pronunciations (fr-fr does, often), whose HMMs score alike for ever, and the reference's answer
(which of them is reported) comes from the order of its lists.  So links are kept in the order
fsg_model_trans_add / fsg_model_add_alt leave them in (alternates w(2), w(3), ..., then w) and
the active lists are rebuilt by prepending exactly as fsg_search.c does.  What is NOT
reproduced is the iteration order of the reference's hash table over DESTINATION states
(fsg_model_arcs: the word links to state i+1 and the loops back to state i come in one order or
the other); that only orders HMMs that cannot tie (a word against a filler).

Pinned by tests/test_first_pass_oracle.py against the word segmentations the reference printed
for its two test recordings.
"""
import ctypes as C
import re

import numpy as np

from . import oracle as O

SENSCR_SHIFT = 10
WORST_SCORE = -(1 << 29)       # (int32)0xE0000000, hmm.h:81
POS_INTERNAL, POS_BEGIN, POS_END, POS_SINGLE = 0, 1, 2, 3


class Lexicon:
    """dict_init_s3file / dict_add_word (src/dict.c:72-134, 240-345): main dictionary, then the
    filler dictionary, then <s>, </s>, <sil> when missing; `word(2)` is linked into its base
    word's alternate list (newest first) and refused when the base is missing."""

    def __init__(self, model, dict_path, filler_path):
        self.pron = {}
        self.order = []
        self.alt = {}          # dict_nextalt
        self._read(model, dict_path)
        self.filler_start = len(self.order)
        self._read(model, filler_path)
        for w in ("<s>", "</s>", "<sil>"):
            self._add(w, [model.sil])
        self.fillers = set(self.order[self.filler_start:])

    def _add(self, w, ci):
        if w in self.pron:
            return
        m = re.match(r"^(.+)\((.*)\)$", w) if w.endswith(")") else None
        if m:
            # dict_word2basestr cuts at the LAST '(' that is not the first character
            i = w.rfind("(", 1, len(w) - 1)
            base = w[:i] if i > 0 else None
            if base is not None:
                if base not in self.pron:
                    return
                self.alt[w] = self.alt.get(base)
                self.alt[base] = w
        self.pron[w] = ci
        self.order.append(w)

    def _read(self, model, path):
        with open(path, encoding="utf-8") as fh:
            for line in fh:
                parts = line.split()
                if not parts or line.startswith("##") or line.startswith(";;"):
                    continue
                ci = [O.ciphone_id(model, p) for p in parts[1:]]
                if not ci or min(ci) < 0:
                    continue
                self._add(parts[0], ci)

    def alt_chain(self, w):
        """the word, then dict_nextalt until it ends (fsg_search_add_altpron)."""
        out = [w]
        while self.alt.get(out[-1]) is not None:
            out.append(self.alt[out[-1]])
        return out

    def is_filler(self, w):
        """dict_filler_word: in the filler range, <s> and </s> excepted (by base word)."""
        while w not in self.fillers and w.endswith(")") and w.rfind("(", 1) > 0 and w[:w.rfind("(", 1)] in self.pron:
            w = w[:w.rfind("(", 1)]
        return w in self.fillers and w not in ("<s>", "</s>")


class Config:
    """Defaults of include/soundswallower/config_defs.h for the first pass."""
    beam = 1e-48
    pbeam = 1e-48
    wbeam = 7e-29
    lw = 6.5
    wip = 0.65
    pip = 1.0
    silprob = 0.005
    fillprob = 1e-8
    fsgusefiller = True
    fsgusealtpron = True


class Node:
    __slots__ = ("ssid", "tmat", "logs2prob", "ci_ext", "ppos", "leaf", "sibling", "succ", "link",
                 "ctxt", "idx")

    def __init__(self):
        self.sibling = None
        self.succ = None
        self.link = None
        self.ctxt = 0      # fsg_pnode_ctxt_t: one bit per CI phone


class Link:
    __slots__ = ("frm", "to", "logs2prob", "word", "filler")

    def __init__(self, frm, to, logp, word, filler):
        self.frm, self.to, self.logs2prob, self.word, self.filler = frm, to, logp, word, filler


ALL_CTXT = (1 << 128) - 1   # fsg_pnode_add_all_ctxt: FSG_PNODE_CTXT_BVSZ (4) x 32 bits set


def _ssid(model, b, l, r, pos):
    return int(model.phone_ssid[O.phone_id_nearest(model, b, l, r, pos)])


def build_fsg(lex, words, lmath, cfg):
    """decoder_set_align_text + fsg_search_add_silences + fsg_search_add_altpron."""
    lw = np.float32(cfg.lw)
    n_state = len(words) + 1
    arcs = [[] for _ in range(n_state)]
    for i, w in enumerate(words):
        if w not in lex.pron:
            raise KeyError(f"Unknown word {w}")
        # fsg_model_add_alt PREPENDS each alternate's link to the state's list, in dict_nextalt
        # order (newest alternate first): the list ends up w(2), w(3), ..., w(k), w
        chain = lex.alt_chain(w) if cfg.fsgusealtpron else [w]
        for a in list(reversed(chain[1:])) + [w]:
            arcs[i].append(Link(i, i + 1, 0, a, False))
    logsil = int(np.float32(lmath.log(float(np.float32(cfg.silprob)))) * lw)
    logfil = int(np.float32(lmath.log(float(np.float32(cfg.fillprob)))) * lw)
    # src/fsg_search.c:107-116: `wid < dict_filler_end` leaves the LAST filler word out
    others = [f for f in lex.order[lex.filler_start:len(lex.order) - 1]
              if f not in ("<s>", "</s>", "<sil>")]
    for s in range(n_state if cfg.fsgusefiller else 0):
        # the loops are added after the text's links and before the alternates; each later
        # fsg_model_trans_add / add_alt prepends to the (state -> same state) list
        loops = [("<sil>", logsil)] + [(f, logfil) for f in others]
        order = []
        for f, lp in loops:
            order.insert(0, (f, lp))
        for f, lp in loops:
            for a in (lex.alt_chain(f)[1:] if cfg.fsgusealtpron else []):
                order.insert(0, (a, lp))
        for a, lp in order:
            arcs[s].append(Link(s, s, lp, a, True))
    return arcs


def context_lists(model, lex, arcs):
    """fsg_lextree_lc_rc (src/fsg_lextree.c:85-214): per state, sorted CI phone lists."""
    n_state = len(arcs)
    sil = model.sil
    lc = [set([sil]) for _ in range(n_state)]
    rc = [set([sil]) for _ in range(n_state)]
    for s in range(n_state):
        for l in arcs[s]:
            if l.filler:
                rc[l.frm].add(sil)
                lc[l.to].add(sil)
            else:
                p = lex.pron[l.word]
                rc[l.frm].add(p[0])
                lc[l.to].add(p[-1])
    return [sorted(x) for x in lc], [sorted(x) for x in rc]


def build_lextree(model, lex, arcs, wip, pip):
    """fsg_psubtree_init / psubtree_add_trans (src/fsg_lextree.c:589-660, 356-587), including
    its behaviours that look unintended: all left contexts of a word-initial phone share the
    HMM of the FIRST left context in the list (the ssid search at :496-503 never leaves pnode
    NULL once one node exists), and single-phone words take SIL as their right context."""
    sil = model.sil
    lcl, rcl = context_lists(model, lex, arcs)
    n_ci = model.n_ciphone
    roots = [None] * len(arcs)
    nodes = []

    def new(ssid, ci, logp, ci_ext, ppos, leaf):
        n = Node()
        n.ssid, n.tmat, n.logs2prob, n.ci_ext, n.ppos, n.leaf = ssid, int(model.phone_tmat[ci]), logp, ci_ext, ppos, leaf
        n.idx = len(nodes)
        nodes.append(n)
        return n

    for s in range(len(arcs)):
        root = None
        glists = {}     # (ci, rc) -> list of root nodes for that word-initial diphone
        for link in arcs[s]:
            pron = lex.pron[link.word]
            lp = link.logs2prob >> SENSCR_SHIFT
            lclist, rclist = lcl[s], rcl[link.to]
            if len(pron) == 1:
                ci = pron[0]
                if not lex.is_filler(link.word):
                    made = []
                    for lc in lclist:
                        ssid = _ssid(model, ci, lc, sil, POS_SINGLE)
                        for n in made:
                            if n.ssid == ssid:
                                n.ctxt |= 1 << lc
                                break
                        else:
                            n = new(ssid, ci, lp + wip + pip, ci, 0, True)
                            n.link = link
                            n.sibling = root
                            n.ctxt |= 1 << lc
                            root = n
                            made.insert(0, n)
                else:
                    n = new(int(model.phone_ssid[ci]), ci, lp + wip + pip, sil, 0, True)
                    n.link = link
                    n.sibling = root
                    n.ctxt = ALL_CTXT
                    root = n
                continue
            pred = None
            lc_nodes = None
            rc_nodes = []
            for p, ci in enumerate(pron):
                if p == 0:
                    rc = pron[1]
                    if (ci, rc) in glists:
                        lc_nodes = glists[(ci, rc)]
                        pred = lc_nodes[0]
                        continue
                    lc_nodes = []
                    first = None
                    for lc in lclist:
                        ssid = _ssid(model, ci, lc, rc, POS_BEGIN)
                        if first is None:
                            first = new(ssid, ci, wip + pip, ci, 0, False)
                            first.sibling = root
                            root = first
                            lc_nodes.insert(0, first)
                        # every later lc lands on the first node, whatever its own ssid
                        first.ctxt |= 1 << lc
                    glists[(ci, rc)] = lc_nodes
                    pred = root
                elif p != len(pron) - 1:
                    ssid = _ssid(model, ci, pron[p - 1], pron[p + 1], POS_INTERNAL)
                    n = pred.succ
                    youngest = n
                    while n is not None and (n.ssid != ssid or n.leaf):
                        n = n.sibling
                    if n is not None:
                        pred = n
                        continue
                    n = new(ssid, ci, pip, ci, p, False)
                    n.sibling = youngest
                    if p == 1:
                        for r in lc_nodes:
                            r.succ = n
                    else:
                        pred.succ = n
                    pred = n
                else:
                    lc = pron[p - 1]
                    by_ssid = {}
                    for rc in rclist:
                        ssid = _ssid(model, ci, lc, rc, POS_END)
                        n = by_ssid.get(ssid)
                        if n is None:
                            n = new(ssid, ci, lp + pip, ci, p, True)
                            n.sibling = rc_nodes[0] if rc_nodes else None
                            n.link = link
                            rc_nodes.insert(0, n)
                            by_ssid[ssid] = n
                        n.ctxt |= 1 << rc
                    if p == 1:
                        for r in lc_nodes:
                            if r.succ is None:
                                r.succ = rc_nodes[0]
                            else:
                                t = r.succ
                                while t.sibling is not None:
                                    t = t.sibling
                                t.sibling = rc_nodes[0]
                                break
                    else:
                        if pred.succ is None:
                            pred.succ = rc_nodes[0]
                        else:
                            t = pred.succ
                            while t.sibling is not None:
                                t = t.sibling
                            t.sibling = rc_nodes[0]
        roots[s] = root
    return nodes, roots


class Hist:
    __slots__ = ("link", "frame", "score", "pred", "lc", "rc")

    def __init__(self, link, frame, score, pred, lc, rc):
        self.link, self.frame, self.score, self.pred, self.lc, self.rc = link, frame, score, pred, lc, rc


def _in(ctxt, ci):
    return (ctxt >> ci) & 1


def first_pass(model, lex, words, senscr, cfg=Config, trace=None, n_frames=None):
    """Returns [(word, start_frame, end_frame, exit score)] as fsg_search_seg_iter yields them,
    or None when the final state is not reached in the last frame that has word exits.  senscr:
    int16 [T][n_sen], or a function (frame, senone ids [n_active][3]) -> int16 [n_sen] row with
    n_frames given (scoring interleaved with the search, as with compallsen=no)."""
    lmath = O.Logmath(1.0001, 0)
    lw = np.float32(cfg.lw)
    beam = int(lmath.log(cfg.beam)) >> SENSCR_SHIFT
    pbeam = int(lmath.log(cfg.pbeam)) >> SENSCR_SHIFT
    wbeam = int(lmath.log(cfg.wbeam)) >> SENSCR_SHIFT
    pip = int(np.float32(lmath.log(cfg.pip)) * lw) >> SENSCR_SHIFT
    wip = int(np.float32(lmath.log(cfg.wip)) * lw) >> SENSCR_SHIFT
    arcs = build_fsg(lex, words, lmath, cfg)
    nodes, roots = build_lextree(model, lex, arcs, wip, pip)
    n_state = len(arcs)
    final = n_state - 1
    N = len(nodes)
    senid = np.ascontiguousarray(model.sseq[[n.ssid for n in nodes]], np.uint16)
    tmat = np.array([n.tmat for n in nodes], np.int16)
    score = np.full((N, 3), WORST_SCORE, np.int32)
    hist = np.full((N, 3), -1, np.int32)
    out_score = np.full(N, WORST_SCORE, np.int32)
    out_hist = np.full(N, -1, np.int32)
    best = np.full(N, WORST_SCORE, np.int32)
    frame_of = np.full(N, -1, np.int64)
    L = O.lib()
    L.orc_hmm_vit_eval_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8
    L.orc_hmm_vit_eval_many.restype = None

    entries = []        # the history table (blkarray of fsg_hist_entry_t)
    frame_entries = {}  # (state, lc) -> list sorted by score, best first

    def entry_add(link, frame, sc, pred, lc, rc):
        """fsg_history_entry_add, src/fsg_history.c:129-205."""
        if frame < 0:
            entries.append(Hist(link, frame, sc, pred, lc, rc))
            return
        lst = frame_entries.setdefault((link.to, lc), [])
        pos = 0
        while pos < len(lst):
            e = lst[pos]
            if sc > e.score:
                break
            rc &= ~e.rc       # existing entry not worse: FSG_PNODE_CTXT_SUB(&rc, &entry->rc)
            if rc == 0:
                return
            pos += 1
        lst.insert(pos, Hist(link, frame, sc, pred, lc, rc))
        k = pos + 1
        while k < len(lst):
            lst[k].rc &= ~rc
            if lst[k].rc == 0:
                del lst[k]
            else:
                k += 1

    def end_frame():
        for key in sorted(frame_entries):
            entries.extend(frame_entries[key])
        frame_entries.clear()

    def enter(n, sc, h, nf, nxt):
        if frame_of[n.idx] < nf:
            nxt.insert(0, n)
        score[n.idx, 0] = sc
        hist[n.idx, 0] = h
        frame_of[n.idx] = nf

    def word_trans(bp_start, frame, bestscore, nxt):
        thresh = bestscore + beam
        nf = frame + 1
        for bp in range(bp_start, len(entries)):
            e = entries[bp]
            d = e.link.to if e.link is not None else 0
            r = roots[d]
            while r is not None:
                if _in(r.ctxt, e.lc) and _in(e.rc, r.ci_ext):
                    ns = e.score + r.logs2prob
                    if ns > thresh and ns > score[r.idx, 0]:
                        enter(r, ns, bp, nf, nxt)
                r = r.sibling

    # fsg_search_start
    active = []
    entry_add(None, -1, 0, -1, model.sil, ALL_CTXT)
    word_trans(0, -1, 0, active)
    T = len(senscr) if not callable(senscr) else n_frames
    for f in range(T):
        bp_start = len(entries)
        if not active:
            return None
        idx = np.array([n.idx for n in active], np.int32)
        # compallsen=no: acmod scores only the senones of the active HMMs
        # (fsg_search_sen_active, src/fsg_search.c:310-325), so the caller scores the frame
        scr = np.ascontiguousarray(senscr(f, senid[idx]) if callable(senscr) else senscr[f], np.int16)
        L.orc_hmm_vit_eval_many(model._m, scr.ctypes.data, len(idx), idx.ctypes.data,
                                senid.ctypes.data, tmat.ctypes.data, score.ctypes.data,
                                hist.ctypes.data, out_score.ctypes.data, out_hist.ctypes.data,
                                best.ctypes.data)
        bestscore = int(best[idx].max())
        bestscore = max(bestscore, WORST_SCORE)
        thresh, pth, wth = bestscore + beam, bestscore + pbeam, bestscore + wbeam
        nxt = []
        for n in active:
            i = n.idx
            if best[i] >= thresh:
                if frame_of[i] == f:
                    frame_of[i] = f + 1
                    nxt.insert(0, n)
                if not n.leaf:
                    if out_score[i] >= pth:
                        c = n.succ
                        while c is not None:
                            ns = int(out_score[i]) + c.logs2prob
                            if ns > thresh and ns > score[c.idx, 0]:
                                enter(c, ns, int(out_hist[i]), f + 1, nxt)
                            c = c.sibling
                elif out_score[i] >= wth:
                    single = n.link.filler or len(lex.pron[n.link.word]) == 1
                    entry_add(n.link, f, int(out_score[i]), int(out_hist[i]), n.ci_ext,
                              ALL_CTXT if single else n.ctxt)
        end_frame()
        # (no null transitions in a linear FSG: fsg_search_null_prop has nothing to do)
        word_trans(bp_start, f, bestscore, nxt)
        for n in active:
            i = n.idx
            if frame_of[i] == f:          # fsg_psubtree_pnode_deactivate -> hmm_clear
                score[i] = WORST_SCORE
                hist[i] = -1
                out_score[i] = WORST_SCORE
                out_hist[i] = -1
                best[i] = WORST_SCORE
                frame_of[i] = -1
        active = nxt
        if trace is not None:
            trace.append((f, bestscore, len(active), len(entries) - bp_start))

    # fsg_search_find_exit(frame_idx = n_frames, final = TRUE), src/fsg_search.c:854-925
    bp = len(entries) - 1
    if bp <= 0:
        return None
    last = entries[bp].frame
    bestsc, besthist = -(1 << 31), -1
    while bp > 0 and entries[bp].frame == last:
        e = entries[bp]
        if e.score == bestsc and e.link.to == final:
            besthist = bp
        elif e.score > bestsc and e.link.to == final:
            bestsc, besthist = e.score, bp
        bp -= 1
    if besthist == -1:
        return None
    seq = []
    bp = besthist
    while bp > 0:
        seq.append(entries[bp])
        bp = entries[bp].pred
    seq.reverse()
    out = []
    for e in seq:     # fsg_seg_bp2itor
        ph = entries[e.pred] if e.pred >= 0 else None
        sf = ph.frame + 1 if ph is not None else 0
        out.append((e.link.word, min(sf, e.frame), e.frame, e.score))
    return out
