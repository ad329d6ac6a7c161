"""Per-wave phase time stamps of ptm_senone_kernel (development aid).

Build the instrumented library first:  make -C soundswallower_amd/csrc timeline TLDEF=SSW_TIMELINE_SEN
then run this on the GPU box."""
import os, sys, shutil, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
os.environ["SSW_AMD_LIB"] = os.path.join(root, "gpurun_tl/libssw_amd_SSW_TIMELINE_SEN.so")
os.environ["SSW_TIMELINE_OUT"] = "/tmp/tl.bin"
sys.path.insert(0, root)
from soundswallower_amd import api, synth
m = api.Model(api.model_dir("en-us"))
raw = synth.read_raw_means(api.model_dir("en-us"))
feats = np.concatenate([synth.synth_features(raw, 256, 12345 + u) for u in range(16)])
off = np.arange(17, dtype=np.int32) * 256
for i in range(5):
    m.score_batch(feats, off)
tl = np.fromfile("/tmp/tl.bin", dtype=np.uint64).reshape(-1, 6)
tl = tl[tl[:, 0] != 0]
T = tl[:, :5].astype(np.int64)
d = np.diff(T, axis=1)
print("waves", len(tl))
for k, nm in enumerate(("prologue (top-N block -> LDS, normalise)", "main (mixw gathers + log-add chains)",
                        "block minimum (3 barriers)", "output scatter")):
    print(f"{nm:45s} p50 {int(np.median(d[:,k])):7d}  p90 {int(np.percentile(d[:,k],90)):7d}  max {d[:,k].max():7d}")
print("wave lifetime p50/max", int(np.median(T[:,4]-T[:,0])), (T[:,4]-T[:,0]).max())
