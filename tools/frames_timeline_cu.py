"""When does each CU finish, by the number of scan workgroups the dispatcher handed it?
(development aid; runs tools/frames_timeline.py first: build the instrumented library as it says)"""
import os, sys, subprocess, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
subprocess.run([sys.executable, os.path.join(root, "tools/frames_timeline.py"), "/tmp/tl.npy"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
tl = np.load("/tmp/tl.npy")
xcc = tl[:, 5].astype(np.int64) & 0xf
hw = tl[:, 4].astype(np.int64)
cu = (hw >> 8) & 15; se = (hw >> 13) & 7
T = tl[:, :4].astype(np.int64)
key = xcc * 1000 + se * 100 + cu
# normalise per xcc: min LDS-filled stamp... per (xcc): subtract min of T[:,0]
for k in np.unique(key):
    m = key == k
    T[m] -= T[m, 0].min()
ends = {}; cnt = {}
for k in np.unique(key):
    m = key == k
    ends[k] = T[m, 3].max(); cnt[k] = m.sum()
for n in sorted(set(cnt.values())):
    e = np.array([ends[k] for k in ends if cnt[k] == n])
    print("CUs with %d waves: %d, end mean %d min %d max %d" % (n, len(e), e.mean(), e.min(), e.max()))
for x in np.unique(xcc):
    ks = [k for k in ends if k // 1000 == x]
    late = max(ks, key=lambda k: ends[k])
    print("xcc", x, "end", ends[late], "latest CU has", cnt[late], "waves; CUs with 36:", sum(cnt[k] == 36 for k in ks))
