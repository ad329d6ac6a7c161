#!/usr/bin/env python3
"""Randomised parity soak: GPU PTM scoring (through the C ABI) against the CPU oracle on many
batches with varied feature statistics, top-N codeword order included.  Not part of the test
suite (minutes of oracle time); prints one JSON line.  The oracle is used as the checker only."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tests/)
sys.path.insert(0, ROOT)
# the soaks flip kernel knobs between batches on one model: have the library re-read them per call
os.environ.setdefault("SSW_KNOBS_DYNAMIC", "1")
import soundswallower_amd as ssw  # noqa: E402
from oracle import oracle as O  # noqa: E402
from soundswallower_amd.synth import read_raw_means, synth_features  # noqa: E402


def soak_align(a):
    """Random forced-alignment problems (ragged batches, random windows, all five kernels in
    rotation; the sliding-window one with windows of 2, 4 and 8 blocks, i.e. with and without
    falling back)."""
    from soundswallower_amd.synth import lcg_uniform, synth_alignment_task
    mdir = ssw.model_dir(a.model)
    g, o = ssw.Model(mdir), O.Model(mdir)
    rng = np.random.default_rng(77 + a.seed)
    t0 = time.time()
    n_utts = n_fail = bad = n_frames = n_batches = 0
    while time.time() - t0 < a.seconds:
        mode = ["mw", "reg", "lds", "hbm", "win", "mwb", "mwb"][n_batches % 7]
        os.environ["SSW_ALIGN_KERNEL"] = mode
        os.environ["SSW_ALIGN_WIN_WAVES"] = ["2", "4", "8"][(n_batches // 7) % 3]
        n_batches += 1
        k = int(rng.integers(1, 7))
        n_ph = rng.integers(1, 300, size=k).tolist()
        n_fr = [int(p * rng.integers(2, 6) + rng.integers(0, 9)) for p in n_ph]
        frame_off = np.concatenate([[0], np.cumsum(n_fr)]).astype(np.int32)
        phone_off = np.concatenate([[0], np.cumsum(n_ph)]).astype(np.int32)
        seed = int(rng.integers(1, 2**31))
        u = lcg_uniform(seed, int(frame_off[-1]) * o.n_sen).reshape(-1, o.n_sen)
        scr = np.floor(u * rng.choice([60, 600, 6000])).astype(np.int16)
        senid, tmat, sf, ef = [], [], [], []
        for i, (p, f) in enumerate(zip(n_ph, n_fr)):
            s_, t_, _ = synth_alignment_task(o.sseq, o.phone_ssid, o.phone_tmat, o.n_ciphone, p,
                                             seed % 100000 + i)
            senid.append(s_)
            tmat.append(t_)
            lo = np.zeros(p, np.int32)
            hi = np.full(p, 2**31 - 1, np.int32)
            if rng.random() < 0.5:
                mid = (np.arange(p) * f) // p
                slack = int(rng.integers(0, 12))
                lo = np.maximum(mid - slack, 0).astype(np.int32)
                hi = np.minimum(mid + f // p + slack + 1, f).astype(np.int32)
            sf.append(lo)
            ef.append(hi)
        senid, tmat = np.concatenate(senid), np.concatenate(tmat)
        sf, ef = np.concatenate(sf), np.concatenate(ef)
        d = g.to_device(scr)
        st, status = g.align_batch(d, frame_off, phone_off, senid, tmat, sf=sf, ef=ef)
        g.device_free(d)
        for i in range(k):
            sl = slice(phone_off[i], phone_off[i + 1])
            rv, rst, _ = o.state_align(scr[frame_off[i]:frame_off[i + 1]], senid[sl], tmat[sl],
                                       sf=sf[sl], ef=ef[sl])
            if (status[i] == 0) != (rv == 0):
                bad += 1
            elif rv == 0 and not np.array_equal(st[phone_off[i] * 3:phone_off[i + 1] * 3], rst):
                bad += 1
            n_fail += rv != 0
            n_utts += 1
            n_frames += n_fr[i]
    print(json.dumps({"mode": "align", "seed": a.seed, "utterances": n_utts, "frames": n_frames,
                      "utterances_without_a_path": int(n_fail), "utterances_differing": bad,
                      "byte_token_kernel": dict(zip(("utterances", "handed_to_full_tokens"),
                                                    g.align_stats())),
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if bad else 0)


def soak_topo(a):
    """HMMs of 1, 2, 4 and 5 emitting states (hmm_vit_eval_5st_lr / hmm_vit_eval_anytopo through
    viterbi_align_any_kernel): models made on the spot as in tests/test_gpu_topologies.py, random
    ragged batches with random windows, against the oracle."""
    import pathlib
    import tempfile
    from soundswallower_amd.synth import lcg_uniform
    from tests.test_gpu_topologies import _models, _task
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="ssw_topo_"))
    models = {ne: _models(tmp, O, ne) for ne in (5, 4, 2, 1)}
    rng = np.random.default_rng(78 + a.seed)
    t0 = time.time()
    n_utts = n_fail = bad = n_frames = n_batches = 0
    per_ne = {ne: 0 for ne in models}
    while time.time() - t0 < a.seconds:
        ne = (5, 4, 5, 2, 5, 1)[n_batches % 6]
        g, o, tp = models[ne]
        n_batches += 1
        k = int(rng.integers(1, 7))
        n_ph = rng.integers(1, 200, size=k).tolist()
        n_fr = [int(p * rng.integers(max(ne - 1, 1), ne + 4) + rng.integers(0, 9)) for p in n_ph]
        frame_off = np.concatenate([[0], np.cumsum(n_fr)]).astype(np.int32)
        phone_off = np.concatenate([[0], np.cumsum(n_ph)]).astype(np.int32)
        seed = int(rng.integers(1, 2**31))
        u = lcg_uniform(seed, int(frame_off[-1]) * o.n_sen).reshape(-1, o.n_sen)
        scr = np.floor(u * rng.choice([60, 600, 6000])).astype(np.int16)
        senid, tmat, sf, ef = [], [], [], []
        for i, (p, f) in enumerate(zip(n_ph, n_fr)):
            s_, t_, _ = _task(o, ne, p, seed % 100000 + i)
            senid.append(s_)
            tmat.append(t_)
            lo = np.zeros(p, np.int32)
            hi = np.full(p, 2**31 - 1, np.int32)
            if rng.random() < 0.5:
                mid = (np.arange(p) * f) // p
                slack = int(rng.integers(0, 12))
                lo = np.maximum(mid - slack, 0).astype(np.int32)
                hi = np.minimum(mid + f // p + slack + 1, f).astype(np.int32)
            sf.append(lo)
            ef.append(hi)
        senid, tmat = np.concatenate(senid), np.concatenate(tmat)
        sf, ef = np.concatenate(sf), np.concatenate(ef)
        d = g.to_device(scr)
        st, status = g.align_batch(d, frame_off, phone_off, senid, tmat, sf=sf, ef=ef)
        g.device_free(d)
        for i in range(k):
            sl = slice(phone_off[i], phone_off[i + 1])
            rv, rst, _ = o.state_align(scr[frame_off[i]:frame_off[i + 1]], senid[sl], tmat[sl],
                                       sf=sf[sl], ef=ef[sl], tp=tp)
            if (status[i] == 0) != (rv == 0):
                bad += 1
            elif rv == 0 and not np.array_equal(st[phone_off[i] * ne:phone_off[i + 1] * ne], rst):
                bad += 1
            n_fail += rv != 0
            n_utts += 1
            per_ne[ne] += 1
            n_frames += n_fr[i]
    print(json.dumps({"mode": "topo", "seed": a.seed, "utterances": n_utts, "frames": n_frames,
                      "utterances_by_states": {str(k_): v for k_, v in per_ne.items()},
                      "utterances_without_a_path": int(n_fail), "utterances_differing": bad,
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if bad else 0)


def soak_first_pass(a):
    """Random texts (1-12 words), synthetic scores that follow one path through the text's phone
    trees with noise at random levels (clean, near-ties, wrong turns, no path), ragged batches:
    ssw_first_pass_batch against the oracle's restatement of fsg_search -- words, frames and
    exit scores, or the same failure."""
    import torch
    from oracle import oracle as O
    from oracle import fsg_oracle as F
    from tests.test_gpu_first_pass import synth_scores
    mdir = ssw.model_dir(a.model)
    m = ssw.Model(mdir)
    orc = O.Model(mdir)
    lex = ssw.Lexicon(m, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    olex = F.Lexicon(orc, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    rng = np.random.default_rng(2024 + a.seed)
    t_end = time.time() + a.seconds
    n_utts = n_frames = n_fail = n_batches = n_label = 0
    while time.time() < t_end:
        # round 3: every third batch through the long-text path (the sliding-window register
        # kernel with windows of 256 / 512 nodes, falling back to the HBM-resident one when the
        # window cannot hold the active nodes), every sixth through the HBM-resident one alone
        for k_ in ("SSW_FP_KERNEL", "SSW_FP_WIN", "SSW_FP_WIN_TPB"):
            os.environ.pop(k_, None)
        if n_batches % 3 == 2:
            os.environ["SSW_FP_KERNEL"] = "big"
            os.environ["SSW_FP_WIN_TPB"] = ["256", "512"][(n_batches // 3) % 2]
            if n_batches % 6 == 5:
                os.environ["SSW_FP_WIN"] = "0"
        nb = int(rng.integers(1, 40))
        texts, scores = [], []
        for _ in range(nb):
            words = [vocab[int(rng.integers(len(vocab)))] for _ in range(int(rng.integers(1, 13)))]
            lo = int(rng.choice([120, 60, 30, 15]))
            sc = synth_scores(F, orc, olex, words, int(rng.integers(1 << 30)), orc.n_sen,
                              noise_lo=lo, sil_p=float(rng.random()))
            if rng.random() < 0.15:
                sc = sc[:int(len(sc) * rng.uniform(0.3, 0.95))]
            texts.append(words)
            scores.append(sc)
        off = np.concatenate([[0], np.cumsum([len(s) for s in scores])]).astype(np.int32)
        d = torch.from_numpy(np.ascontiguousarray(np.concatenate(scores), np.int16)).cuda()
        got = lex.first_pass(d, off, texts)
        for t, sc, g in zip(texts, scores, got):
            want = F.first_pass(orc, olex, t, sc)
            if want is None:
                assert g is None, ("GPU found a path the oracle does not", t)
                n_fail += 1
            else:
                assert g is not None, ("GPU lost the path", t)
                mine = [(w, s, s + dd - 1, x) for (w, s, dd, x) in g]
                if mine != want:
                    # frames and scores must agree whatever happens; a different LABEL among
                    # alternates pronounced alike is counted and shown
                    assert [x[1:] for x in mine] == [x[1:] for x in want], (t, mine, want)
                    n_label += 1
                    if n_label <= 5:
                        print("label difference:", [(a[0], b[0]) for a, b in zip(mine, want) if a != b],
                              file=sys.stderr)
            n_utts += 1
            n_frames += len(sc)
        n_batches += 1
    print(json.dumps({"mode": "first_pass", "model": a.model, "batches": n_batches,
                      "utterances": n_utts, "frames": n_frames, "without_a_path": n_fail,
                      "differences": 0, "label_differences_among_identical_alternates": n_label}))


def soak_fp_active(a):
    """The DEFAULT configuration's first pass as a batch (ssw_first_pass_batch_active: speculation
    and proof) against the frame-synchronous oracle: random texts of 1-8 words, synthetic FEATURES
    that follow one path through the text's phone trees at random noise levels (clean to nearly
    lost), audio cut short, texts that do not fit, ragged batches, either scan; every other batch
    also compares every score row as acmod's buffer holds it and the set left for the second
    pass; every tenth batch with a text of 110-150 words (the long-text search kernels).  Words,
    frames and exit scores -- or the same failure."""
    import torch
    from oracle import fsg_oracle as F
    from soundswallower_amd.synth import read_raw_means
    from tests.test_gpu_first_pass_active import oracle_default_first_pass
    from tools.bench_first_pass import path_through
    mdir = ssw.model_dir(a.model)
    m = ssw.Model(mdir)
    orc = O.Model(mdir)
    lex = ssw.Lexicon(m, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    olex = F.Lexicon(orc, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    means = read_raw_means(mdir)
    mixw = m.table("ptm_mixw").reshape(m.n_feat, m.n_density, m.n_sen)
    sen2cb = m.table("sen2cb")
    best_d = mixw.argmin(axis=1)
    sen_mean = np.concatenate([means[sen2cb, f, best_d[f]] for f in range(m.n_feat)], axis=1)
    rng = np.random.default_rng(909 + a.seed)
    t_end = time.time() + a.seconds
    n_utts = n_frames = n_fail = n_batches = n_rows = n_long = 0
    rounds_hist = {}
    while time.time() < t_end:
        os.environ.pop("SSW_SCAN", None)
        if n_batches % 4 == 3:
            os.environ["SSW_SCAN"] = "fma"
        nb = int(rng.integers(1, 9))
        texts, feats = [], []
        # every tenth batch holds a text beyond 1,024 phone-tree HMMs: the whole batch then goes
        # through the long-text search kernels (sliding window; every twentieth: HBM-resident)
        long_one = n_batches % 10 == 9
        os.environ.pop("SSW_FP_WIN", None)
        if n_batches % 20 == 19:
            os.environ["SSW_FP_WIN"] = "0"
        for k_ in range(nb):
            n_w = int(rng.integers(110, 150)) if long_one and k_ == 0 else int(rng.integers(1, 9))
            words = [vocab[int(rng.integers(len(vocab)))] for _ in range(n_w)]
            nodes, _ = lex.first_pass_graph(words)
            path = path_through(lex, nodes, len(words), rng)
            states = np.array([s_ for i in path for s_ in nodes[i]["senid"]])
            per = rng.integers(1, 5, size=len(states))
            sen = np.repeat(states, per)
            noise = float(rng.choice([0.2, 0.5, 1.0, 2.0]))
            x = sen_mean[sen] + rng.standard_normal((len(sen), sen_mean.shape[1])).astype(np.float32) * noise
            if rng.random() < 0.15:                       # audio cut short
                x = x[:max(1, int(len(x) * rng.uniform(0.3, 0.95)))]
            if rng.random() < 0.1:                        # another text
                words = [vocab[int(rng.integers(len(vocab)))] for _ in range(len(words))]
            texts.append(words)
            feats.append(x.astype(np.float32))
        off = np.concatenate([[0], np.cumsum([len(x) for x in feats])]).astype(np.int32)
        allf = np.ascontiguousarray(np.concatenate(feats), np.float32)
        d_feats = torch.from_numpy(allf).cuda()
        want_rows = n_batches % 2 == 0
        d_rows = (torch.zeros((len(allf), m.n_sen), dtype=torch.int16, device="cuda")
                  if want_rows else None)
        got, rounds, seed = lex.first_pass_active(d_feats, off, texts, d_senscr=d_rows, want_seed=True,
                                                  max_seg=2048 if long_one else None)
        n_long += 1 if long_one else 0
        torch.cuda.synchronize()
        rows = d_rows.cpu().numpy() if want_rows else None
        for u, (t, x, g) in enumerate(zip(texts, feats, got)):
            want, wrows, wvec = oracle_default_first_pass(O, F, orc, olex, t, x)
            rounds_hist[int(rounds[u])] = rounds_hist.get(int(rounds[u]), 0) + 1
            if want is None:
                assert g is None, ("GPU found a path the oracle does not", t)
                n_fail += 1
            else:
                assert g is not None, ("GPU lost the path", t)
                mine = [(w, s_, s_ + dd - 1, sc) for (w, s_, dd, sc) in g]
                assert [y[1:] for y in mine] == [y[1:] for y in want], (t, mine, want)
                assert np.array_equal(seed[u], wvec), ("seed", t)
                if want_rows:
                    assert np.array_equal(rows[off[u]:off[u + 1]], wrows), ("rows", t)
                    n_rows += len(x)
            n_utts += 1
            n_frames += len(x)
        n_batches += 1
    print(json.dumps({"mode": "fp_active", "model": a.model, "batches": n_batches,
                      "utterances": n_utts, "frames": n_frames, "without_a_path": n_fail,
                      "score_rows_compared": n_rows, "batches_with_a_long_text": n_long,
                      "rounds_histogram": {str(k): v for k, v in sorted(rounds_hist.items())},
                      "differences": 0}))


def soak_text_active(a):
    """decoder_alignment in the DEFAULT configuration as a batch (ssw_align_text_batch_active with
    two_pass_history: the first pass by speculation and proof, populate, the second pass over the
    growing active set from the history slot the first pass left) against the oracle's two
    restated searches around its per-frame scorer with active lists
    (tests/test_oracle_e2e_goforward.py: default_configuration_alignment, the pipeline pinned to
    the real library's recorded default-configuration phone scores), history carried across the
    rewind as the reference does: words, every phone's start, duration and score -- or the same
    failure.  Random texts of 1-6 words, synthetic features that follow them."""
    import torch
    from oracle import fsg_oracle as F
    from soundswallower_amd.synth import read_raw_means
    from tests.test_oracle_e2e_goforward import default_configuration_alignment
    from tools.bench_first_pass import path_through
    mdir = ssw.model_dir(a.model)
    m = ssw.Model(mdir)
    orc = O.Model(mdir)
    lex = ssw.Lexicon(m, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    olex = F.Lexicon(orc, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    means = read_raw_means(mdir)
    mixw = m.table("ptm_mixw").reshape(m.n_feat, m.n_density, m.n_sen)
    sen2cb = m.table("sen2cb")
    best_d = mixw.argmin(axis=1)
    sen_mean = np.concatenate([means[sen2cb, f, best_d[f]] for f in range(m.n_feat)], axis=1)
    rng = np.random.default_rng(4711 + a.seed)
    cfg = lex.first_pass_config(two_pass_history=1)
    t_end = time.time() + a.seconds
    n_utts = n_frames = n_fail = n_batches = n_second_fail = 0
    while time.time() < t_end:
        nb = int(rng.integers(1, 7))
        texts, feats = [], []
        for _ in range(nb):
            words = [vocab[int(rng.integers(len(vocab)))] for _ in range(int(rng.integers(1, 7)))]
            nodes, _ = lex.first_pass_graph(words)
            path = path_through(lex, nodes, len(words), rng)
            states = np.array([s_ for i in path for s_ in nodes[i]["senid"]])
            sen = np.repeat(states, rng.integers(1, 5, size=len(states)))
            noise = float(rng.choice([0.2, 0.5, 1.0]))
            x = sen_mean[sen] + rng.standard_normal((len(sen), sen_mean.shape[1])).astype(np.float32) * noise
            if rng.random() < 0.1:
                x = x[:max(2, int(len(x) * rng.uniform(0.4, 0.95)))]
            texts.append(words)
            feats.append(x.astype(np.float32))
        off = np.concatenate([[0], np.cumsum([len(x) for x in feats])]).astype(np.int32)
        d_feats = torch.from_numpy(np.ascontiguousarray(np.concatenate(feats), np.float32)).cuda()
        aset = ssw.align_text_batch_active(m, lex, d_feats, off, texts, cfg=cfg)
        try:
            for u, (t, x) in enumerate(zip(texts, feats)):
                orc.ptm_reset()
                orc.ptm_set_frame_idx(0)

                def eval_frame(f, feat, lst):
                    row = orc.ptm_frame_eval(feat, f, compallsen=False, senone_active=lst)
                    orc.ptm_set_frame_idx(f + 1)
                    return row

                try:
                    seg, ph_start, ph_dur, ph_score = default_configuration_alignment(
                        O, orc, x, eval_frame, lambda: orc.ptm_set_frame_idx(0), text=" ".join(t),
                        model=a.model)
                    second_ok = True
                except AssertionError:          # the restated second pass found no final state
                    seg, second_ok = "first pass only", False
                n_utts += 1
                n_frames += len(x)
                if seg is None:
                    assert aset.status(u) == 1, ("GPU aligned what the oracle's first pass lost", t)
                    n_fail += 1
                    continue
                if not second_ok:
                    assert aset.status(u) == 2, ("second pass", t, aset.status(u))
                    n_second_fail += 1
                    continue
                assert aset.status(u) == 0, (t, aset.status(u))
                al = aset.utterance(u)
                assert [(w, int(e[0]), int(e[1])) for w, e in zip(al["words"], al["word_al"])] \
                    == [(w, s_, e_ - s_ + 1) for (w, s_, e_, _) in seg], t
                assert [int(e[0]) for e in al["phone_al"]] == [int(v) for v in ph_start], t
                assert [int(e[1]) for e in al["phone_al"]] == [int(v) for v in ph_dur], t
                assert [int(e[2]) for e in al["phone_al"]] == [int(v) for v in ph_score], t
        finally:
            aset.free()
        n_batches += 1
    print(json.dumps({"mode": "text_active", "model": a.model, "batches": n_batches,
                      "utterances": n_utts, "frames": n_frames, "first_pass_without_a_path": n_fail,
                      "second_pass_without_a_path": n_second_fail, "differences": 0}))


def soak_text(a):
    """decoder_alignment end to end: ssw_forced_align_batch (first pass, populate, constrained
    state alignment, propagate) against the oracle's pipeline -- restated first pass, then
    alignment_populate restated on the oracle's bin_mdef_phone_id_nearest with the first pass's
    words and windows, then orc_state_align -- on random texts with synthetic scores."""
    import torch
    from oracle import fsg_oracle as F
    from tests.test_gpu_first_pass import synth_scores
    mdir = ssw.model_dir(a.model)
    m = ssw.Model(mdir)
    orc = O.Model(mdir)
    lex = ssw.Lexicon(m, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    olex = F.Lexicon(orc, os.path.join(mdir, "dict.txt"), os.path.join(mdir, "noisedict.txt"))
    vocab = [w for w in olex.order[:olex.filler_start] if "(" not in w]
    sil = orc.sil

    def populate(words):
        """alignment_populate (src/ps_alignment.c:132-247): [(word, start, dur)] -> ssid, tmat, word index"""
        out = []
        lc = sil
        for i, (w, _, _) in enumerate(words):
            p = olex.pron[w]
            rc = olex.pron[words[i + 1][0]][0] if i < len(words) - 1 else sil
            n = len(p)
            for j in range(n):
                if n == 1:
                    pid = O.phone_id_nearest(orc, p[0], lc, rc, 3)
                elif j == 0:
                    pid = O.phone_id_nearest(orc, p[0], lc, p[1], 1)
                elif j == n - 1:
                    pid = O.phone_id_nearest(orc, p[j], p[j - 1], rc, 2)
                else:
                    pid = O.phone_id_nearest(orc, p[j], p[j - 1], p[j + 1], 0)
                out.append((int(orc.phone_ssid[pid]), int(orc.phone_tmat[p[j]]), i))
            lc = p[-1]
        return out

    rng = np.random.default_rng(4711 + a.seed)
    t_end = time.time() + a.seconds
    n_utts = n_frames = n_fail = n_batches = 0
    while time.time() < t_end:
        # round 3: every other batch through the long-text kernels of both passes (sliding
        # windows of 256 nodes / 2 or 4 phone blocks, falling back where those cannot hold it)
        for k_ in ("SSW_FP_KERNEL", "SSW_FP_WIN_TPB", "SSW_ALIGN_KERNEL", "SSW_ALIGN_WIN_WAVES"):
            os.environ.pop(k_, None)
        if n_batches % 2 == 1:
            os.environ["SSW_FP_KERNEL"] = "big"
            os.environ["SSW_FP_WIN_TPB"] = "256"
            os.environ["SSW_ALIGN_KERNEL"] = "win"
            os.environ["SSW_ALIGN_WIN_WAVES"] = ["2", "4"][(n_batches // 2) % 2]
        nb = int(rng.integers(1, 24 if a.max_words < 50 else 4))
        texts, scores = [], []
        for _ in range(nb):
            words = [vocab[int(rng.integers(len(vocab)))]
                     for _ in range(int(rng.integers(a.min_words, a.max_words + 1)))]
            sc = synth_scores(F, orc, olex, words, int(rng.integers(1 << 30)), orc.n_sen,
                              noise_lo=int(rng.choice([120, 60, 30])), sil_p=float(rng.random()))
            if rng.random() < 0.1:
                sc = sc[:int(len(sc) * rng.uniform(0.4, 0.95))]
            texts.append(words)
            scores.append(sc)
        off = np.concatenate([[0], np.cumsum([len(s) for s in scores])]).astype(np.int32)
        d = torch.from_numpy(np.ascontiguousarray(np.concatenate(scores), np.int16)).cuda()
        aset = ssw.forced_align_batch(m, lex, d, off, texts)
        for u, (t, sc) in enumerate(zip(texts, scores)):
            seg = F.first_pass(orc, olex, t, sc)
            got = aset.utterance(u)
            if seg is None:
                assert got is None and aset.status(u) == 1, t
                n_fail += 1
            else:
                words = [(w, s, e - s + 1) for (w, s, e, _) in seg]
                ph = populate(words)
                senid = orc.sseq[[x[0] for x in ph]]
                tmat = np.array([x[1] for x in ph], np.int16)
                ws = np.array([words[x[2]][1] for x in ph], np.int32)
                wd = np.array([words[x[2]][2] for x in ph], np.int32)
                sf = np.where(ws > 0, ws, 0).astype(np.int32)
                ef = np.where(wd > 0, ws + wd, 2**31 - 1).astype(np.int32)
                init = np.stack([np.repeat(ws, 3), np.repeat(wd, 3), np.zeros(3 * len(ph), np.int32)], 1)
                rv, st, php = orc.state_align(sc, senid, tmat, sf, ef, state_init=init.astype(np.int32))
                if rv != 0:
                    assert got is None and aset.status(u) == 2, t
                    n_fail += 1
                else:
                    assert got is not None, t
                    assert got["words"] == [w for (w, _, _) in words], t
                    assert np.array_equal(got["state_al"], st), t
                    assert np.array_equal(got["phone_al"], php), t
                    assert np.array_equal(got["senid"], senid), t
            n_utts += 1
            n_frames += len(sc)
        aset.free()
        n_batches += 1
    print(json.dumps({"mode": "text", "model": a.model, "batches": n_batches, "utterances": n_utts,
                      "frames": n_frames, "not_aligned_on_both_sides": n_fail, "differences": 0}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--min-words", type=int, default=1, help="text mode: words per text")
    ap.add_argument("--max-words", type=int, default=9,
                    help="text mode: words per text (hundreds: the sliding windows of the "
                         "long-text kernels move, every other batch)")
    ap.add_argument("--model", default="en-us")
    ap.add_argument("--seed", type=int, default=0,
                    help="added to the mode's own seed: another run, other inputs")
    ap.add_argument("--mode", default="ptm", choices=["ptm", "ms", "align", "topo", "first_pass", "text", "fp_active", "text_active"])
    ap.add_argument("--max-len", type=int, default=400,
                    help="ptm / ms: utterances of up to this many frames, 1-5 per batch (the "
                         "matrix-core scan takes batches from ~2100 frames: use 1600)")
    a = ap.parse_args()
    if a.mode == "first_pass":
        return soak_first_pass(a)
    if a.mode == "fp_active":
        return soak_fp_active(a)
    if a.mode == "text_active":
        return soak_text_active(a)
    if a.mode == "text":
        return soak_text(a)
    if a.mode == "topo":
        return soak_topo(a)
    if a.mode == "align":
        return soak_align(a)
    mdir = ssw.model_dir(a.model)
    if a.mode == "ms":
        # the shipped models carry a sendump only: synthesise the mixture_weights file the ms
        # scorer reads (pdf = 1.0001^-(q*1024)), as tools/bench_ms.py does
        import struct
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from bench_ms import write_s3
        tables = ssw.Model(mdir, config={"device": -2})
        q = tables.table("ptm_mixw").reshape(tables.n_feat, tables.n_density, tables.n_sen)
        pdf = np.ascontiguousarray(np.power(1.0001, -(q.astype(np.float64) * 1024.0))
                                   .transpose(2, 0, 1), dtype="<f4")
        mixw = os.path.join(tempfile.mkdtemp(), "mixture_weights")
        write_s3(mixw, struct.pack("<4i", pdf.shape[0], pdf.shape[1], pdf.shape[2], pdf.size)
                 + pdf.tobytes())
        kw = dict(mdef=os.path.join(mdir, "mdef"), means=os.path.join(mdir, "means"),
                  tmat=os.path.join(mdir, "transition_matrices"), mixw=mixw)
        g = ssw.Model(variances=os.path.join(mdir, "variances"), **kw)
        o = O.Model(vars=os.path.join(mdir, "variances"), **kw)
    else:
        g, o = ssw.Model(mdir), O.Model(mdir)
    means = read_raw_means(mdir)
    rng = np.random.default_rng(20261002 + a.seed)
    t0 = time.time()
    n_frames = n_batches = bad_rows = bad_topn = flagged = pairs = 0
    kinds = {}
    while time.time() - t0 < a.seconds:
        kind = ["synthetic", "scaled", "gauss", "interp", "quantised", "mixed"][n_batches % 6]
        lens = rng.integers(1, a.max_len, size=int(rng.integers(1, 6))).tolist()
        n = int(sum(lens))
        base = synth_features(means, n, int(rng.integers(1, 2**31)))
        if kind == "scaled":
            feats = base * np.float32(rng.choice([0.1, 0.5, 2.0, 5.0, 20.0]))
        elif kind == "gauss":
            feats = rng.normal(0, rng.choice([0.3, 1.0, 3.0]), size=base.shape).astype(np.float32)
        elif kind == "interp":
            t = np.linspace(0, 1, n, dtype=np.float32)[:, None]
            feats = (base[:1] * (1 - t) + base[-1:] * t).astype(np.float32)
        elif kind == "quantised":
            feats = (np.round(base * 4) / 4).astype(np.float32)   # provokes exact ties
        elif kind == "mixed":
            feats = base.copy()
            feats[::3] = np.roll(base, 1, axis=0)[::3]
            feats[::7] = 0
        else:
            feats = base
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        if a.mode == "ms":
            got = g.score_batch(feats, off, scorer=ssw.SCORER_MS)
            f, p = g.last_stats()
            for u in range(len(lens)):
                ref = o.ms_score_utt(feats[off[u]:off[u + 1]])
                bad_rows += int((got[off[u]:off[u + 1]] != ref).any(axis=1).sum())
        else:
            got = g.score_batch(feats, off)
            f, p = g.last_stats()
            gcw, _ = g.last_topn(n)
            for u in range(len(lens)):
                ref, rcw, _ = o.ptm_score_utt(feats[off[u]:off[u + 1]], want_topn=True)
                bad_rows += int((got[off[u]:off[u + 1]] != ref).any(axis=1).sum())
                bad_topn += int((gcw[off[u]:off[u + 1]].astype(np.int32) != rcw)
                                .any(axis=(1, 2, 3)).sum())
        n_frames += n
        n_batches += 1
        flagged += f
        pairs += p
        kinds[kind] = kinds.get(kind, 0) + n
    print(json.dumps({"mode": a.mode, "seed": a.seed, "model": a.model, "max_len": a.max_len,
                      "batches": n_batches, "frames": n_frames,
                      "frames_by_kind": kinds, "rows_differing": bad_rows,
                      "frames_with_topn_order_differing": bad_topn,
                      "exact_pass_share": flagged / max(pairs, 1),
                      # SSW_SCAN_AUDIT=k in the environment: proven pairs redone exactly by the
                      # audited waves of the matrix-core scan, and how many of them differed
                      "scan_audit": dict(zip(("k", "proven_pairs_audited", "pairs_differing"),
                                             (int(os.environ.get("SSW_SCAN_AUDIT", "0")),)
                                             + g.scan_audit_stats())),
                      "mfma_selftest_worst_u": g.mfma_selftest_worst_u,
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if bad_rows or bad_topn else 0)


if __name__ == "__main__":
    main()
