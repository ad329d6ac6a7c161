"""The proof obligation of the GPU's speculative top-N scan, replayed on the CPU.

The frames kernel (soundswallower_amd/csrc/ssw_k1a_frames.inc) ranks the 128 densities of a
codebook with a 26-FMA quadratic form instead of the reference's sub/mul/mul/sub chain
(src/ptm_mgau.c:63-68), and accepts the speculative top four only when the fifth key, widened,
is below the fourth exact score.  That is sound only if, for every density and every input,

    widen(key) >= the reference's fp32 density value,

with widen() the kernel's three fp32 statements.  The scan records are built on the host
(ssw_model.c:ssw_host_build_records), so the inequality can be checked here without a GPU:
oracle.scan_replay evaluates both sides with the exact operation order (fmaf for the kernel's
v_fma_f32), over the shipped models and over input families chosen to stress the bound."""
import os

import numpy as np
import pytest

import soundswallower_amd as ssw
from tests.conftest import MODEL_ROOT

def widen(key, d0):
    """The three statements after the scan (ssw_k1a_frames.inc, 'upper bound on the true
    value of every density outside the four'), in float32 like the kernel."""
    key = key.astype(np.float32)
    ub = key + np.abs(key) * np.float32(104.0 * 2.0 ** -24) + np.float32(1.0e-3)
    ub = ub + np.float32(d0)
    ub = ub + np.abs(ub) * np.float32(2.0 ** -22)
    return ub


def bucket_up(key):
    """The 5th key as the kernel sees it: low 7 mantissa bits were borrowed for the codeword
    and are pushed towards +inf."""
    b = key.astype(np.float32).view(np.uint32)
    neg = (b & np.uint32(0x80000000)) != 0
    return np.where(neg, b & np.uint32(0xFFFFFF80), b | np.uint32(127)).astype(np.uint32).view(np.float32)


def input_families(mean4, f, n, rng):
    """x for stream f, [n][veclen]: near the means (what speech looks like to the model), noise
    at several scales, points far out along single axes, midpoints between densities, and
    near-cancellation points x = mean (1 + eps) of the densities with the largest var mean^2."""
    n_cb, n_feat, n_den, vl = mean4.shape
    mu = mean4[:, f].reshape(-1, vl).astype(np.float32)
    out = []
    pick = mu[rng.integers(0, len(mu), n)]
    out.append(pick + ((rng.random((n, vl)) - 0.5) * 0.5).astype(np.float32))
    for sigma in (0.05, 1.0, 4.0, 25.0):
        out.append((rng.standard_normal((n, vl)) * sigma).astype(np.float32))
    far = pick.copy()
    far[np.arange(n), rng.integers(0, vl, n)] += (rng.choice([-1, 1], n) * rng.uniform(10, 300, n)).astype(np.float32)
    out.append(far)
    out.append(((pick + mu[rng.integers(0, len(mu), n)]) * np.float32(0.5)).astype(np.float32))
    out.append((pick * (1.0 + rng.uniform(-1e-3, 1e-3, (n, vl)))).astype(np.float32))
    out.append(np.zeros((4, vl), np.float32))
    return np.concatenate(out).astype(np.float32)


def m_exact_expected(name):
    """Densities left to the exact form in the shipped models."""
    return {"en-us": 114, "fr-fr": 115}[name]


@pytest.mark.parametrize("name", ["en-us", "fr-fr"])
def test_scan_key_bounds_the_reference_value(oracle_mod, name):
    m = ssw.Model(os.path.join(MODEL_ROOT, name), config={"device": -2})
    o = oracle_mod.Model(os.path.join(MODEL_ROOT, name))
    n_cbf, nd = m.n_cb * m.n_feat, m.n_density
    rec = m.table("rec").reshape(n_cbf, nd, 32)
    recq = m.table("scan_rec").reshape(n_cbf, nd, 32)
    d0 = m.table("scan_d0").reshape(n_cbf, 32)[:, 0]
    exl = m.table("scan_exact").reshape(n_cbf, -1)
    mean4 = o.mean4()
    vl = mean4.shape[3]
    rng = np.random.default_rng(20240611)
    n_pairs = 0
    n_exact = 0
    worst_margin = np.inf      # min of widen(key) - ref, in score units
    worst_use = 0.0            # max of (ref - d0 - key) / (104u|key| + 1e-3): share of the widening used
    # every codebook-stream of the model; several hundred inputs each (seconds on the CPU)
    for cbf in range(n_cbf):
        f = cbf % m.n_feat
        x = input_families(mean4, f, 96, rng)
        ref, key = oracle_mod.scan_replay(rec[cbf], recq[cbf], x, vl)
        on_list = np.zeros(nd, bool)
        on_list[exl[cbf, 1:1 + exl[cbf, 0]]] = True
        n_exact += int(on_list.sum())
        # exact-form densities: inert record, the key never competes
        assert np.all(key[:, on_list] == np.float32(-3.0e38))
        k, r = key[:, ~on_list], ref[:, ~on_list]
        assert np.isfinite(k).all() and np.isfinite(r).all()
        ub = widen(k, d0[cbf])
        assert np.all(ub >= r), (name, cbf, float((ub - r).min()))
        # what the kernel really uses is the bucketed 5th key, which is no smaller
        assert np.all(bucket_up(k) >= k)
        worst_margin = min(worst_margin, float((ub.astype(np.float64) - r).min()))
        over = r.astype(np.float64) - float(d0[cbf]) - k.astype(np.float64)
        room = 104.0 * 2.0 ** -24 * np.abs(k.astype(np.float64)) + 1.0e-3
        worst_use = max(worst_use, float((over / room).max()))
        n_pairs += k.size
    assert n_exact == m_exact_expected(name)
    assert n_pairs > 8_000_000
    # the analysis leaves slack: the key exceeded by at most this share of the widening
    assert worst_use < 1.0
    print(f"{name}: {n_pairs} (input, density) pairs, min margin {worst_margin:.4g}, "
          f"largest share of the widening used {worst_use:.3f}")


def test_widening_is_monotone():
    """The kernel widens only the 5th key and relies on widen() being non-decreasing, so that
    it bounds every smaller key's density as well."""
    rng = np.random.default_rng(5)
    mags = np.exp(rng.uniform(np.log(1e-6), np.log(3e9), 400_000)).astype(np.float32)
    v = np.sort(np.concatenate([mags, -mags, np.float32([0.0, -0.0])]))
    # neighbours in float order stress the rounding of each statement
    v = np.sort(np.concatenate([v, np.nextafter(v, np.float32(np.inf))]))
    for d0 in (0.0, -73000.0, 41000.0, -1.0e7):
        w = widen(v, d0)
        assert np.all(np.diff(w) >= 0)


def write_hostile_gaussians(tmp_path, n_cb=2, n_feat=3, nd=128, vl=13, seed=11,
                            scales=(1.0, 30.0, 300.0), var_lo=1e-5):
    """means / variances files (s3 gauden parameter layout, src/ms_gauden.c:105-216) with
    means up to a few hundred and variances from below the floor to 10."""
    from tests.test_cabi_host import _write_s3
    import struct
    rng = np.random.default_rng(seed)
    scale = rng.choice(list(scales), (n_cb, n_feat, nd, 1))
    mean = (rng.standard_normal((n_cb, n_feat, nd, vl)) * scale).astype("<f4")
    var = np.exp(rng.uniform(np.log(var_lo), np.log(10.0), (n_cb, n_feat, nd, vl))).astype("<f4")
    paths = {}
    for nm, arr in (("means", mean), ("variances", var)):
        payload = struct.pack("<3i", n_cb, n_feat, nd) + struct.pack(f"<{n_feat}i", *([vl] * n_feat))
        payload += struct.pack("<i", arr.size) + arr.tobytes()
        paths[nm] = str(tmp_path / nm)
        _write_s3(paths[nm], payload)
    return paths


def test_scan_bound_on_a_hostile_model(oracle_mod, tmp_path):
    """Records with large means and tight variances (big var mean^2, where the quadratic form
    cancels badly): those the bias cannot cover must land on the exact-form list, the rest must
    still satisfy the bound."""
    paths = write_hostile_gaussians(tmp_path)
    m = ssw.Model(config={"device": -2}, **paths)
    n_cbf, nd = m.n_cb * m.n_feat, m.n_density
    rec = m.table("rec").reshape(n_cbf, nd, 32)
    recq = m.table("scan_rec").reshape(n_cbf, nd, 32)
    d0 = m.table("scan_d0").reshape(n_cbf, 32)[:, 0]
    exl = m.table("scan_exact").reshape(n_cbf, -1)
    assert exl[:, 0].sum() > 0
    rng = np.random.default_rng(9)
    vl = 13
    for cbf in range(n_cbf):
        mu = rec[cbf, :, :vl]
        pick = mu[rng.integers(0, nd, 256)]
        x = np.concatenate([pick + rng.standard_normal((256, vl)).astype(np.float32) * 0.3,
                            pick * np.float32(1.0001),
                            rng.standard_normal((256, vl)).astype(np.float32) * 50]).astype(np.float32)
        ref, key = oracle_mod.scan_replay(rec[cbf], recq[cbf], x, vl)
        on_list = np.zeros(nd, bool)
        on_list[exl[cbf, 1:1 + exl[cbf, 0]]] = True
        k, r = key[:, ~on_list], ref[:, ~on_list]
        assert np.all(widen(k, d0[cbf]) >= r)


MFMA_WIDEN = np.float32(1.8537045e-05)   # SSW_MFMA_WIDEN, csrc/ssw_k1a_mfma.inc (K = 96)
MFMA_EPS = 34.0 * 2.0 ** -24             # the assumed accumulation error of one MFMA (ssw_model.c)


def mfma_widen(key, d0):
    """what ptm_topn_mfma_kernel does to the 5th key once it is back in score units"""
    key = key.astype(np.float32)
    ub = key + np.abs(key) * MFMA_WIDEN + np.float32(1.0e-3)
    ub = ub + np.float32(d0)
    return ub + np.abs(ub) * np.float32(2.384185791015625e-07)


def mfma_operands(m, name=None):
    """The matrix-core scan's tables as the kernel uses them: per codebook x stream the two
    binary16 parts of W [128][32] (scaled by 2^-s, the constant by 2^-(s + ec)), 2^s and 2^ec."""
    n_cbf = m.n_cb * m.n_feat
    wf = m.table("scan_wfrag").reshape(n_cbf, 4, 2, 2, 64, 8).view(np.float16)
    W = np.zeros((n_cbf, 2, 128, 32), np.float16)
    for rb in range(4):
        for kb in range(2):
            for lane in range(64):
                k0 = 16 * kb + 8 * (lane >> 5)
                W[:, :, 32 * rb + (lane & 31), k0:k0 + 8] = wf[:, rb, kb, :, lane, :]
    d0 = m.table("scan_d0").reshape(n_cbf, 32)
    return W, d0[:, 0], d0[:, 1], d0[:, 3]


def mfma_x_parts(x, xconst):
    """mfma_build_x + split2_f16 in numpy: X = (x, 32768, 32768, 2^ec | fl(x^2), 0, 0, 0) cut
    into two binary16 parts, the residual formed in float32 (exact)."""
    n = len(x)
    X = np.zeros((n, 32), np.float32)
    X[:, 0:13] = x
    X[:, 13] = X[:, 14] = 32768.0
    X[:, 15] = xconst
    X[:, 16:29] = x * x
    with np.errstate(over="ignore"):
        p1 = X.astype(np.float16)
        p2 = (X - p1.astype(np.float32)).astype(np.float16)
    return X, p1, p2


def test_mfma_scan_records_are_consistent_with_the_fma_ones():
    """The matrix-core scan's tables (csrc/ssw_model.c, ssw_host_build_mfma_records): the same
    quadratic form as SCAN_REC with its own error constant; the two binary16 parts of every
    record element add up to it within the split's residual (and the constant exactly: it is
    rounded UP to what its parts represent); exact-form densities have inert rows."""
    for name in ("en-us", "fr-fr"):
        m = ssw.Model(os.path.join(MODEL_ROOT, name), config={"device": -2})
        n_cbf = m.n_cb * m.n_feat
        rq = m.table("scan_rec").reshape(n_cbf, 128, 32)
        rm = m.table("scan_rec_mfma").reshape(n_cbf, 128, 32)
        exq = m.table("scan_exact").reshape(n_cbf, 132)
        exm = m.table("scan_exact_mfma").reshape(n_cbf, 132)
        W, d0, scale, xconst = mfma_operands(m)
        assert np.isfinite(W.astype(np.float32)).all()
        for cbf in range(n_cbf):
            live_q = np.ones(128, bool)
            live_q[exq[cbf, 1:1 + exq[cbf, 0]]] = False
            live_m = np.ones(128, bool)
            live_m[exm[cbf, 1:1 + exm[cbf, 0]]] = False
            both = live_q & live_m
            # same a and b as the vector-unit scan's records
            assert np.array_equal(rq[cbf, both][:, :13], rm[cbf, both][:, :13])
            assert np.array_equal(rq[cbf, both][:, 16:29], rm[cbf, both][:, 16:29])
            assert (rm[cbf, ~live_m, 15] < -1e37).all()
            s = float(scale[cbf])
            assert s == 2.0 ** round(np.log2(s)) and float(xconst[cbf]) == 2.0 ** round(np.log2(xconst[cbf]))
            back = (W[cbf, 0].astype(np.float64) + W[cbf, 1].astype(np.float64)) * s
            back[:, 15] *= float(xconst[cbf])
            live = rm[cbf, live_m].astype(np.float64)
            got = back[live_m]
            # the constant: exactly what the record says (its parts were chosen for that)
            assert np.array_equal(got[:, 15].astype(np.float32), rm[cbf, live_m, 15])
            ab = np.r_[0:13, 16:29]
            resid = np.abs(got[:, ab] - live[:, ab])
            assert (resid <= np.maximum(np.abs(live[:, ab]) * 2.0 ** -22 * 1.001, 2.0 ** -25 * s)).all()
            # inert rows: nothing but -65504 in the two spare slots, against X = 32768
            assert (W[cbf, 0, ~live_m][:, [13, 14]] == np.float16(-65504)).all()
            assert (np.delete(W[cbf, 0, ~live_m], [13, 14], axis=1) == 0).all()
            assert (W[cbf, 1, ~live_m] == 0).all() and (W[cbf, :, live_m][:, :, [13, 14]] == 0).all()


def _mfma_bound_without_the_adder(m, mean4, rng, tag, n_in=48, min_live=1):
    """the body of the two tests below: returns (pairs checked, min margin, largest share of
    the widening used)"""
    n_cbf, nd = m.n_cb * m.n_feat, m.n_density
    rec = m.table("rec").reshape(n_cbf, nd, 32)
    exm = m.table("scan_exact_mfma").reshape(n_cbf, 132)
    W, d0, scale, xconst = mfma_operands(m)
    worst_use, worst_margin, n_pairs = 0.0, np.inf, 0
    for cbf in range(n_cbf):
        f = cbf % m.n_feat
        x = input_families(mean4, f, n_in, rng)
        mu = rec[cbf, :, 0:13]
        R = (rec[cbf, :, 16:29] * mu * mu).sum(axis=1)
        worst_d = np.argsort(R)[-6:]
        x = np.concatenate([x, 2 * mu[worst_d], np.nextafter(2 * mu[worst_d], np.float32(np.inf)),
                            mu[worst_d], -2 * mu[worst_d],
                            (rng.standard_normal((32, 13)) * 1e-4).astype(np.float32)]).astype(np.float32)
        x = x[np.abs(x).max(axis=1) <= 255.0]          # beyond: the kernel does not trust the scan
        X, p1, p2 = mfma_x_parts(x, float(xconst[cbf]))
        W1, W2 = W[cbf, 0].astype(np.float64), W[cbf, 1].astype(np.float64)
        X1, X2 = p1.astype(np.float64), p2.astype(np.float64)
        kept = X1 @ (W1 + W2).T + X2 @ W1.T                      # (1,1) + (2,1) + (1,2), exact
        M = np.abs(X1) @ np.abs(W1).T * (1 + 2.0 ** -9)          # sum of |products| the MFMAs see
        live = np.ones(nd, bool)
        live[exm[cbf, 1:1 + exm[cbf, 0]]] = False
        assert (kept[:, ~live] < -4.0e9).all()                   # inert rows stay out of the way
        if live.sum() < min_live:
            continue
        s = float(scale[cbf])
        low = ((kept - 2.01 * MFMA_EPS * M) * s)[:, live]        # the least the device may return
        mean, var, det = rec[cbf, :, 0:13], rec[cbf, :, 16:29], rec[cbf, :, 15]
        ref = np.broadcast_to(det, (len(x), nd)).astype(np.float32).copy()
        for j in range(13):
            diff = x[:, None, j] - mean[None, :, j]
            ref = ref - (diff * diff) * var[None, :, j]
        ref = ref[:, live]
        ub = mfma_widen(low, d0[cbf]).astype(np.float64)
        # float32 rounding of `low` itself (the MFMA result is a float): one more ulp, downwards
        ub_min = ub - np.abs(ub) * 2.0 ** -23
        ok = np.isfinite(ref)
        assert (ub_min[ok] >= ref[ok]).all(), (tag, cbf, float((ub_min - ref)[ok].min()))
        worst_margin = min(worst_margin, float((ub_min - ref)[ok].min()))
        room = MFMA_WIDEN * np.abs(low) + 1.0e-3
        worst_use = max(worst_use, float(((ref - float(d0[cbf]) - low) / room)[ok].max()))
        n_pairs += int(ok.sum())
    return n_pairs, worst_margin, worst_use


@pytest.mark.parametrize("name", ["en-us", "fr-fr"])
def test_mfma_scan_bound_without_the_adder(oracle_mod, name):
    """Everything of the matrix-core scan's bound that does not depend on the matrix core's
    internal adder, replayed exactly: the operands the kernel builds (two binary16 parts of W from
    the model's own tables, two of X cut as mfma_build_x cuts them), the three part products it
    keeps, summed in float64.  Take away the most the six chained MFMAs may lose under the
    assumed eps (2.01 eps M, ssw_model.c) and the widened key must still be >= the reference's
    fp32 value -- for inputs near the means, noise, far-out points up to the +-255 the kernel
    accepts, midpoints and the cancelling points x = 2 mean.  What is left to the device test
    (tests/test_gpu_mfma_bound.py) is eps itself."""
    m = ssw.Model(os.path.join(MODEL_ROOT, name), config={"device": -2})
    o = oracle_mod.Model(os.path.join(MODEL_ROOT, name))
    rng = np.random.default_rng(20261003)
    n_pairs, worst_margin, worst_use = _mfma_bound_without_the_adder(m, o.mean4(), rng, name)
    assert n_pairs > 5_000_000 and worst_use < 1.0
    print(f"{name}: {n_pairs} pairs, min margin {worst_margin:.4g}, share of the widening used {worst_use:.3f}")


def test_mfma_scan_bound_on_a_hostile_model(tmp_path):
    """The same replay on records the shipped models do not contain: means up to a few hundred,
    variances from below the floor to 10 (scales of W from 2^-8 to 2^20 and beyond, elements that
    lose their second binary16 part, constants that need the 2^ec slot): what the host cannot
    cover lands on the exact-form list, the rest satisfies the bound."""
    paths = write_hostile_gaussians(tmp_path, n_cb=4, seed=23, scales=(0.3, 3.0, 20.0), var_lo=0.02)
    m = ssw.Model(config={"device": -2}, **paths)
    n_cbf = m.n_cb * m.n_feat
    mean4 = m.table("rec").reshape(m.n_cb, m.n_feat, m.n_density, 32)[:, :, :, 0:13]
    exm = m.table("scan_exact_mfma").reshape(n_cbf, 132)
    assert 0 < exm[:, 0].sum() < n_cbf * m.n_density        # some densities in, some out
    rng = np.random.default_rng(5)
    n_pairs, worst_margin, worst_use = _mfma_bound_without_the_adder(m, mean4, rng, "hostile", n_in=96)
    assert n_pairs > 20_000 and worst_use < 1.0
    print(f"hostile: {n_pairs} pairs, {int(exm[:, 0].sum())} of {n_cbf * m.n_density} densities exact-form, "
          f"min margin {worst_margin:.4g}, share of the widening used {worst_use:.3f}")


def test_mfma_constant_that_nearly_cancels(tmp_path):
    """a density whose constant c' = (det - d0 - sum var mean^2 + bias) 2^-(s + ec) is
    tiny -- one variance for every density (det = d0) and means of a few thousandths -- has a
    second binary16 part on the subnormal grid (2^-24), which 64 float steps upwards do not
    always reach; the host must then either hold the constant from above with its two parts or
    leave the density to the exact form.  Either way the replayed bound must hold."""
    from tests.test_cabi_host import _write_s3
    import struct
    rng = np.random.default_rng(41)
    n_cb, n_feat, nd, vl = 3, 3, 128, 13
    # the constant's scale 2^ec is per codebook x stream: half of every codebook's densities
    # share the reference variance and sit at the origin (constant = the bias alone), the other
    # half have constants in the hundreds and set the scale
    mean = (rng.standard_normal((n_cb, n_feat, nd, vl)) * 3.0e-3).astype("<f4")
    var = np.full((n_cb, n_feat, nd, vl), 0.7, "<f4")
    mean[:, :, nd // 2:] = (rng.standard_normal((n_cb, n_feat, nd // 2, vl)) * 4.0).astype("<f4")
    var[:, :, nd // 2:] = np.exp(rng.uniform(np.log(0.7), np.log(6.0),
                                             (n_cb, n_feat, nd // 2, vl))).astype("<f4")
    paths = {}
    for nm, arr in (("means", mean), ("variances", var)):
        payload = struct.pack("<3i", n_cb, n_feat, nd) + struct.pack(f"<{n_feat}i", *([vl] * n_feat))
        payload += struct.pack("<i", arr.size) + arr.tobytes()
        paths[nm] = str(tmp_path / nm)
        _write_s3(paths[nm], payload)
    m = ssw.Model(config={"device": -2}, **paths)
    n_cbf = m.n_cb * m.n_feat
    mean4 = m.table("rec").reshape(m.n_cb, m.n_feat, m.n_density, 32)[:, :, :, 0:13]
    W, d0, scale, xconst = mfma_operands(m)
    exm = m.table("scan_exact_mfma").reshape(n_cbf, 132)
    rm = m.table("scan_rec_mfma").reshape(n_cbf, nd, 32)
    # the constants the matrix cores see, back in score units, against what the record says
    small = 0
    for cbf in range(n_cbf):
        live = np.ones(nd, bool)
        live[exm[cbf, 1:1 + exm[cbf, 0]]] = False
        c_parts = (W[cbf, 0, :, 15].astype(np.float64) + W[cbf, 1, :, 15].astype(np.float64))
        small += int((np.abs(c_parts[live]) < 2.0 ** -7).sum())
        back = c_parts * float(scale[cbf]) * float(xconst[cbf])
        assert np.array_equal(back[live].astype(np.float32), rm[cbf, live, 15])
    rng = np.random.default_rng(6)
    n_pairs, worst_margin, worst_use = _mfma_bound_without_the_adder(m, mean4, rng, "tiny constants",
                                                                     n_in=96, min_live=0)
    print(f"tiny constants: {small} live constants below 2^-7 after scaling, {n_pairs} pairs, "
          f"{int(exm[:, 0].sum())} densities exact-form, min margin {worst_margin:.4g}")
    assert n_pairs > 20_000 and worst_margin >= 0.0
