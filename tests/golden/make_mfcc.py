#!/usr/bin/env python3
"""Writes tests/golden/goforward_mfcc.npy and goforward_fr_mfcc.npy: the 13-dim cepstra of the
reference's two test recordings (tests/golden/goforward.raw = tests/data/goforward.wav without
its header; goforward_fr.raw = tests/data/goforward_fr.raw), computed by the CPU
oracle's restatement of the front end (oracle/ssw_oracle_fe.c, checked against the reference's
tests/_test_fe.res) with model/en-us/feat_params.json's settings.  A data fixture: bench.py's
`real_features` object and the GPU tests feed it to ssw_feat_batch, so that neither needs the
oracle (or a front end, which is out of scope for the product) to get real speech features.
Run from the repo root:  python tests/golden/make_mfcc.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

if __name__ == "__main__":
    pcm = np.fromfile(os.path.join(ROOT, "tests", "golden", "goforward.raw"), dtype="<i2")
    cep = O.fe_mfcc(pcm, nfilt=20, lowerf=130, upperf=3700, lifter=22, remove_noise=True,
                    transform="dct")
    assert cep.shape == (278, 13)
    np.save(os.path.join(ROOT, "tests", "golden", "goforward_mfcc.npy"), cep.astype(np.float32))
    print(cep.shape, float(cep[0, 0]))
    pcm = np.fromfile(os.path.join(ROOT, "tests", "golden", "goforward_fr.raw"), dtype="<i2")
    cep = O.fe_mfcc(pcm, nfilt=20, lowerf=130, upperf=3700, lifter=22, remove_noise=True,
                    transform="dct")       # model/fr-fr/feat_params.json: the same settings
    assert cep.shape == (239, 13)
    np.save(os.path.join(ROOT, "tests", "golden", "goforward_fr_mfcc.npy"), cep.astype(np.float32))
    print(cep.shape, float(cep[0, 0]))
