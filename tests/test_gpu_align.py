"""GPU parity: forced-alignment Viterbi through the C ABI vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

from soundswallower_amd.synth import lcg_uniform, synth_alignment_task

pytestmark = pytest.mark.gpu


def _random_senscr(n_frames, n_sen, seed):
    u = lcg_uniform(seed, n_frames * n_sen).reshape(n_frames, n_sen)
    scr = np.floor(u * 600).astype(np.int16)
    scr[np.arange(n_frames), np.floor(lcg_uniform(seed + 1, n_frames) * n_sen).astype(int)] = 0
    return scr


@pytest.mark.parametrize("n_phones,n_frames", [(1, 5), (5, 40), (64, 300), (65, 300), (150, 700)])
def test_align_matches_oracle(gpu_en, orc_en, n_phones, n_frames):
    senid, tmat, _ = synth_alignment_task(orc_en.sseq, orc_en.phone_ssid, orc_en.phone_tmat,
                                          orc_en.n_ciphone, n_phones, 99 + n_phones)
    scr = _random_senscr(n_frames, orc_en.n_sen, 7 + n_phones)
    rv, rst, rph = orc_en.state_align(scr, senid, tmat)
    d = gpu_en.to_device(scr)
    try:
        st, status = gpu_en.align_batch(d, [0, n_frames], [0, n_phones], senid, tmat)
    finally:
        gpu_en.device_free(d)
    assert (status[0] == 0) == (rv == 0)
    if rv == 0:
        assert np.array_equal(st, rst)
