"""Synthetic inputs shared by the oracle side and the GPU side.

64-bit LCG  s <- s * 6364136223846793005 + 1442695040888963407,  u = ((s >> 11) & (2^53-1)) / 2^53.
Features: per frame draw a codebook, per stream a codeword, x = mean_raw + (u - 0.5) * 0.5.
"""
from __future__ import annotations

import numpy as np

_A = np.uint64(6364136223846793005)
_C = np.uint64(1442695040888963407)
_MASK53 = np.uint64((1 << 53) - 1)


def lcg_uniform(seed: int, n: int) -> np.ndarray:
    """n successive uniforms in [0, 1) as float64 (vectorised by jumping the LCG)."""
    if n <= 0:
        return np.zeros(0, np.float64)
    # s_k = A^k s_0 + C (A^k - 1)/(A - 1); build by doubling so everything stays in uint64
    with np.errstate(over="ignore"):
        mul = np.empty(n, np.uint64)
        add = np.empty(n, np.uint64)
        mul[0] = _A
        add[0] = _C
        filled = 1
        while filled < n:
            m = min(filled, n - filled)
            # compose step (mul[filled-1], add[filled-1]) after steps 1..m
            mul[filled:filled + m] = mul[:m] * mul[filled - 1]
            add[filled:filled + m] = add[:m] * mul[filled - 1] + add[filled - 1]
            filled += m
        s = mul * np.uint64(seed) + add
    return ((s >> np.uint64(11)) & _MASK53).astype(np.float64) / float(1 << 53)


def synth_features(mean4: np.ndarray, n_frames: int, seed: int) -> np.ndarray:
    """mean4: raw means [n_cb][n_feat][n_density][veclen] float32.  Returns [n_frames][39]."""
    n_cb, n_feat, n_den, vl = mean4.shape
    per = 1 + n_feat * (1 + vl)
    u = lcg_uniform(seed, n_frames * per).reshape(n_frames, per)
    cb = np.floor(u[:, 0] * n_cb).astype(np.int64)
    out = np.empty((n_frames, n_feat * vl), np.float32)
    for k in range(n_feat):
        base = 1 + k * (1 + vl)
        cw = np.floor(u[:, base] * n_den).astype(np.int64)
        noise = ((u[:, base + 1:base + 1 + vl] - 0.5) * 0.5).astype(np.float32)
        out[:, k * vl:(k + 1) * vl] = (mean4[cb, k, cw, :].astype(np.float32) + noise).astype(
            np.float32)
    return out


def synth_alignment_task(sseq: np.ndarray, phone_ssid: np.ndarray, phone_tmat: np.ndarray,
                         n_ciphone: int, n_phones: int, seed: int):
    """A random phone string for the Viterbi kernel: triphone ids drawn from the mdef's phone
    table (so ssid/tmat pairs are real), first and last phone = a CI phone.  Returns
    (senid uint16 [n_phones][n_emit], tmatid int16 [n_phones], ssid int32 [n_phones])."""
    u = lcg_uniform(seed, n_phones)
    pid = np.floor(u * len(phone_ssid)).astype(np.int64)
    pid[0] = pid[0] % n_ciphone
    pid[-1] = pid[-1] % n_ciphone
    ssid = phone_ssid[pid].astype(np.int32)
    tmat = phone_tmat[pid].astype(np.int16)
    senid = sseq[ssid].astype(np.uint16)
    return senid, tmat, ssid


def read_raw_means(model_dir: str) -> np.ndarray:
    """float32 means exactly as stored in the s3 file: [cb][feat][density][veclen]
    (equal stream lengths only).  Input to synth_features on both the oracle and GPU sides."""
    import os
    with open(os.path.join(model_dir, "means"), "rb") as fh:
        blob = fh.read()
    end = blob.index(b"endhdr\n") + len(b"endhdr\n")
    magic = np.frombuffer(blob, dtype="<u4", count=1, offset=end)[0]
    bo = "<" if magic == 0x11223344 else ">"
    dims = np.frombuffer(blob, dtype=bo + "i4", count=3, offset=end + 4)
    n_cb, n_feat, n_den = (int(x) for x in dims)
    vl = np.frombuffer(blob, dtype=bo + "i4", count=n_feat, offset=end + 16)
    n = int(np.frombuffer(blob, dtype=bo + "i4", count=1, offset=end + 16 + 4 * n_feat)[0])
    data = np.frombuffer(blob, dtype=bo + "f4", count=n, offset=end + 20 + 4 * n_feat)
    if len(set(vl.tolist())) != 1:
        raise ValueError("streams of different lengths")
    return data.reshape(n_cb, n_feat, n_den, int(vl[0])).astype(np.float32)
