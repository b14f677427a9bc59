import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenes import scene_variant, rt
import test_gpu_parity as T
lib = rt._abi.load()
name = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sc, cam = scene_variant(name)
opt = rt.solver_defaults()
g = T.hip_trace(lib, sc, opt, n, n, cam=cam)
st = (g["n_accept"] + g["n_reject"]).astype(np.int64)
print(name, "steps mean %.1f median %.0f p99 %.0f max %d" % (st.mean(), np.median(st), np.percentile(st, 99), st.max()), "status", np.bincount(g["status"], minlength=5), "hit", np.bincount(g["hit"], minlength=4))
big = np.argsort(-st)[:6]
for i in big:
    print("  idx", i, "(i,j)=", i % n, i // n, "steps", st[i], "status", g["status"][i], "hit", g["hit"][i], "lam %.4f" % g["lambda_end"][i], "x_end", np.round(g["state_end"][i, :4], 4))
print("share of steps in rays with > 2000 steps: %.3f (%d rays)" % (st[st > 2000].sum() / st.sum(), (st > 2000).sum()))
