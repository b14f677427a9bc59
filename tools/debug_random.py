import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from scenes import rt
import test_gpu_parity as T
lib = rt._abi.load()
for seed in range(15):
    sc, cam, opt, nobj = T._random_scene(seed)
    gpu = T.hip_trace(lib, sc, opt, 40, 32, cam=cam)
    ref = O.trace(sc, opt, 40, 32, cam=cam)
    flips = gpu["hit"] != ref["hit"]
    st = gpu["status"] != ref["status"]
    same = ~flips & ~st
    d = np.abs(gpu["rgb"][:, same] - ref["rgb"][:, same])
    per = gpu["hit"][same].astype(np.float64) / max(nobj, 1)
    per = np.where(per > 0, per, 1.0)[None, :]
    e = np.minimum(d, np.abs(per - d))
    sd = np.abs((gpu["n_accept"] + gpu["n_reject"]).astype(np.int64) - (ref["n_accept"] + ref["n_reject"]).astype(np.int64))
    print("seed %2d metric %d a=%.1f nobj %d kinds %s: flips %d status-mismatch %d (gpu %s ref %s) rgberr %.2e maxstepdiff %d  nrej gpu %d ref %d  lam1 %.0f tol %.1e" % (
        seed, sc.metric, sc.a, nobj, [sc.obj[i].kind for i in range(nobj)], flips.sum(), st.sum(), np.bincount(gpu["status"], minlength=5), np.bincount(ref["status"], minlength=5),
        e.max(initial=0), sd[same].max(initial=0), gpu["n_reject"].sum(), ref["n_reject"].sum(), opt.lambda1, opt.reltol))
    if e.max(initial=0) > 1e-6:
        j = np.argmax(e.max(axis=0)); idx = np.flatnonzero(same)[j]
        print("    worst px", idx, "hit", gpu["hit"][idx], "rgb gpu", gpu["rgb"][:, idx], "ref", ref["rgb"][:, idx], "x gpu", gpu["state_end"][idx,:4], "ref", ref["state_end"][idx,:4])
