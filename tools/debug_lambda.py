import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from scenes import example, rt
import test_gpu_parity as T
lib = rt._abi.load()
sc, cam = example(2)
opt = rt.solver_defaults()
gpu = T.hip_trace(lib, sc, opt, 200, 200, cam=cam)
ref = O.trace(sc, opt, 200, 200, cam=cam)
d = np.abs(gpu["lambda_end"] - ref["lambda_end"])
print("n > 1e-9:", int((d > 1e-9).sum()), "max", d.max())
for i in np.argsort(-d)[:12]:
    print(i, "lam gpu %.12f ref %.12f  d %.3e  hit %d/%d steps %d/%d  xend gpu %s ref %s" % (
        gpu["lambda_end"][i], ref["lambda_end"][i], d[i], gpu["hit"][i], ref["hit"][i],
        gpu["n_accept"][i], ref["n_accept"][i], np.array2string(gpu["state_end"][i, :4], precision=6),
        np.array2string(ref["state_end"][i, :4], precision=6)))
