"""Per-wave phase time stamps of ptm_topn_frames_kernel (development aid).
Build the instrumented library first:  make -C soundswallower_amd/csrc timeline
then run this on the GPU box."""
import os, sys, shutil, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
os.environ["SSW_AMD_LIB"] = os.path.join(root, "gpurun_tl/libssw_amd_SSW_TIMELINE.so")
os.environ["SSW_TIMELINE_OUT"] = "/tmp/tl.bin"
sys.path.insert(0, root)
import torch
from soundswallower_amd import api, synth
m = api.Model(api.model_dir("en-us"))
raw = synth.read_raw_means(api.model_dir("en-us"))
feats = np.concatenate([synth.synth_features(raw, 256, 12345 + u) for u in range(16)])
off = np.arange(17, dtype=np.int32) * 256
for i in range(5):
    out = m.score_batch_host(feats, off) if hasattr(m, "score_batch_host") else m.score_batch(feats, off)
raw_tl = np.fromfile("/tmp/tl.bin", dtype=np.uint64)
tl2 = raw_tl[16384 * 6:].reshape(-1, 8) if len(raw_tl) > 16384 * 6 else None
tl = raw_tl[:16384 * 6].reshape(-1, 6)
if tl2 is not None:     # stamps inside the exact in-wave pass: entry, first pair's densities,
    m2 = tl2[:, 0] != 0     # first pair's exact step, end; [4] = pairs redone
    e = tl2[m2].astype(np.int64)
    if len(e):
        print("exact pass, %d waves: densities of the first pair p50 %d, its exact step p50 %d, "
              "whole pass per pair p50 %d (pairs per wave mean %.2f)"
              % (len(e), np.median(e[:, 1] - e[:, 0]), np.median(e[:, 2] - e[:, 1]),
                 np.median((e[:, 3] - e[:, 0]) / np.maximum(e[:, 4], 1)), e[:, 4].mean()))
        if len(sys.argv) > 1:
            np.save(sys.argv[1] + ".exact.npy", e)
tl = tl[tl[:, 0] != 0]
if len(sys.argv) > 1:
    np.save(sys.argv[1], tl)
xcc_all = tl[:, 5].astype(np.int64) & 0xf
hw_all = tl[:, 4].astype(np.int64)
grp = xcc_all * 100000 + ((hw_all >> 13) & 7) * 1000 + ((hw_all >> 8) & 15)
T = tl[:, :4].astype(np.int64)
for x in np.unique(grp):
    T[grp == x] -= T[grp == x, 0].min()
print("waves", len(tl), "clock units (s_memtime)")
# stamps of ptm_topn_mfma_kernel: 0 = LDS filled, 1 = first 64-frame step done, 2 = scan done
# (= stamp 1 with one step per wave), 3 = exact in-wave pass done
for k, nm in enumerate(("LDS filled", "1st step done", "scan done", "end")):
    print(f"{nm:11s} min {T[:,k].min():8d} p50 {int(np.median(T[:,k])):8d} p90 {int(np.percentile(T[:,k],90)):8d} max {T[:,k].max():8d}")
print("later steps p10/p50/p90/max", np.percentile(T[:,2]-T[:,1],[10,50,90,100]).astype(int))
print("first step p50/max", np.percentile(T[:,1]-T[:,0],[50,100]).astype(int), "exact pass p50/max", np.percentile(T[:,3]-T[:,2],[50,100]).astype(int))
hw = tl[:, 4].astype(np.int64); xcc = tl[:, 5].astype(np.int64) & 0xf
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7
key = xcc * 100000 + se * 1000 + cu * 10 + simd
u, c = np.unique(key, return_counts=True)
print("distinct (xcc,se,cu,simd):", len(u), "waves per SIMD histogram:", np.bincount(c))
cukey = xcc * 1000 + se * 100 + cu
u2, c2 = np.unique(cukey, return_counts=True)
print("distinct CUs:", len(u2), "waves per CU histogram:", np.bincount(c2))
late = T[:, 0] > 20000
print("waves starting late (>20000):", late.sum())
print("per-xcc end max:", [int(T[xcc_all == x, 3].max()) for x in np.unique(xcc_all)])
