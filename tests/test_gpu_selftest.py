"""The matrix-core scan's one hardware assumption as a checked invariant.

ssw_model_load measures the accumulation error of v_mfma_f32_32x32x16_f16 on the device it loads
the model on (adversarial alignment tiles + random / cancelling tiles against fp64 and closed-form
sums) and only then lets the matrix-core scan run there; a device that fails -- forced here through
the debug hook SSW_MFMA_SELFTEST_EPS, an accepted eps below what the hardware really does -- gets
the vector-unit scan, and must still produce the reference-confirmed checksums.  SSW_SCAN_AUDIT=k
checks the scan's claim itself in the product's process: audited waves redo proven pairs exactly
and count differences."""
import json
import os
import time

import numpy as np
import pytest

import soundswallower_amd as ssw
from soundswallower_amd.synth import synth_features
from tests.conftest import ROOT
from tests.test_golden_fixtures import crc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "synthetic_oracle.json")) as fh:
        return json.load(fh)


def _config2(means):
    feats = np.concatenate([synth_features(means, 256, 12345 + u) for u in range(16)])
    return feats, (np.arange(17) * 256).astype(np.int32)


def test_self_test_passes_on_this_device_and_is_cheap(gpu_en):
    assert gpu_en.scan_mode == 1 and gpu_en.mfma_selftest == 1
    # MI355X, round 4 measurements: 5-7 u; the bound assumes 34
    assert 0.4 < gpu_en.mfma_selftest_worst_u < 12.0, gpu_en.mfma_selftest_worst_u
    assert "enabled" in gpu_en.selftest_message
    # the cost: a model loaded when the process has already launched kernels (the first launch of
    # a process pays for loading the code object, whoever makes it)
    t0 = time.perf_counter()
    m = ssw.Model(ssw.model_dir("en-us"))
    load_ms = (time.perf_counter() - t0) * 1e3
    print("self-test: worst %.2f u, %.3f ms of a %.1f ms ssw_model_load"
          % (m.mfma_selftest_worst_u, m.mfma_selftest_ms, load_ms))
    assert m.mfma_selftest == 1 and m.mfma_selftest_ms < 1.0
    m.close()


def test_failed_self_test_falls_back_to_the_vector_unit_scan(golden, means_en, monkeypatch):
    monkeypatch.setenv("SSW_MFMA_SELFTEST_EPS", "0.25")      # the hardware does 5-7 u
    m = ssw.Model(ssw.model_dir("en-us"))
    monkeypatch.delenv("SSW_MFMA_SELFTEST_EPS")
    try:
        assert m.mfma_selftest == -1 and m.scan_mode == 0
        assert "FAILED" in m.selftest_message and "vector-unit scan" in m.selftest_message
        assert m.mfma_selftest_worst_u > 0.25
        feats, off = _config2(means_en)
        g = golden["config2_en_us_ptm"]
        assert crc(m.score_batch(feats, off)) == g["utt16x256"]["crc"]
        assert crc(m.score_batch(feats, np.array([0, 4096], np.int32))) == g["utt1x4096"]["crc"]
        # the matrix-core debug views say so instead of running
        with pytest.raises(ssw.SswError, match="no matrix-core scan tables"):
            m.debug_scan_keys(feats[:64], 0)
    finally:
        m.close()


def test_self_test_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("SSW_MFMA_SELFTEST", "off")
    m = ssw.Model(ssw.model_dir("en-us"))
    monkeypatch.delenv("SSW_MFMA_SELFTEST")
    assert m.mfma_selftest == 0 and m.scan_mode == 1 and m.selftest_message == ""
    m.close()


@pytest.mark.parametrize("k", [1, 3])
def test_scan_audit_finds_no_mismatch_and_changes_no_result(golden, means_en, monkeypatch, k):
    monkeypatch.setenv("SSW_SCAN_AUDIT", str(k))
    m = ssw.Model(ssw.model_dir("en-us"))
    try:
        feats, off = _config2(means_en)
        g = golden["config2_en_us_ptm"]
        assert crc(m.score_batch(feats, off)) == g["utt16x256"]["crc"]
        audited, differ = m.scan_audit_stats()
        flagged, pairs = m.last_stats()
        print("audit k=%d: %d proven pairs redone exactly, %d differ; %d of %d pairs unproven"
              % (k, audited, differ, flagged, pairs))
        assert differ == 0
        if k == 1:
            assert audited + flagged == pairs      # every pair went through the exact pass
        else:
            assert 0.2 * pairs < audited < 0.5 * pairs
        # real speech, both steps per wave, the other layout
        monkeypatch.setenv("SSW_MFMA_STEPS", "2")
        assert crc(m.score_batch(feats, np.array([0, 4096], np.int32))) == g["utt1x4096"]["crc"]
        cep = np.load(os.path.join(ROOT, "tests", "golden", "goforward_mfcc.npy"))
        f2 = m.feat_batch(cep)
        a = m.score_batch(f2)
        monkeypatch.setenv("SSW_SCAN_AUDIT", "0")
        assert np.array_equal(a, m.score_batch(f2))
        assert m.scan_audit_stats()[1] == 0
    finally:
        m.close()


def test_scan_audit_on_the_ms_scorer(gpu_fr, means_fr, monkeypatch):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_ms
    monkeypatch.setenv("SSW_SCAN_AUDIT", "2")
    m, means = bench_ms.build_model()
    try:
        feats = np.concatenate([synth_features(means, 256, 4242 + u) for u in range(8)])
        off = (np.arange(9) * 256).astype(np.int32)
        a = m.score_batch(feats, off, scorer=ssw.SCORER_MS)
        audited, differ = m.scan_audit_stats()
        assert audited > 0 and differ == 0
        monkeypatch.setenv("SSW_SCAN_AUDIT", "0")
        assert np.array_equal(a, m.score_batch(feats, off, scorer=ssw.SCORER_MS))
    finally:
        m.close()
