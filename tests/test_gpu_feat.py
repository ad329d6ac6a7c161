"""GPU parity: dynamic features (batch CMN + 1s_c_d_dd) vs the oracle, bit-exact float32."""
import os

import numpy as np
import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_feat_goforward_bit_exact(gpu_en, oracle_mod):
    pcm = np.fromfile(os.path.join(ROOT, "tests", "golden", "goforward.raw"), dtype="<i2")
    cep = oracle_mod.fe_mfcc(pcm, nfilt=20, lowerf=130, upperf=3700, lifter=22, remove_noise=True,
                             transform="dct")
    ref = oracle_mod.feat_1s_c_d_dd(cep)
    got = gpu_en.feat_batch(cep)
    assert got.tobytes() == ref.tobytes()


def test_feat_ragged_batch_and_edges(gpu_en, oracle_mod):
    rng = np.random.default_rng(4)
    lens = [1, 2, 3, 7, 300, 5]
    ceps = [rng.normal(0, 4, (n, 13)).astype(np.float32) for n in lens]
    for c in ceps:                    # an utterance with NO frame of c0 >= 0 divides 0 by 0:
        c[:, 0] = np.abs(c[:, 0])     # NaN on both sides, but NaN sign bits are platform lore
    ceps[4][::5, 0] = -1.0            # "zero energy" frames are left out of the mean
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    got = gpu_en.feat_batch(np.concatenate(ceps), off)
    ref = np.concatenate([oracle_mod.feat_1s_c_d_dd(c) for c in ceps])
    assert got.tobytes() == ref.tobytes()
