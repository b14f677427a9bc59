import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from scenes import rt
import test_gpu_parity as T
lib = rt._abi.load()
seed = int(sys.argv[1])
sc, cam, opt, nobj = T._random_scene(seed)
for o in range(nobj): print("obj", o+1, sc.obj[o].kind, list(sc.obj[o].p)[:9])
gpu = T.hip_trace(lib, sc, opt, 40, 32, cam=cam)
ref = O.trace(sc, opt, 40, 32, cam=cam)
ok = (gpu["status"] < 2) & (ref["status"] < 2) & (gpu["hit"] == ref["hit"])
d = np.abs(gpu["rgb"] - ref["rgb"])
per = np.where(gpu["hit"] > 0, gpu["hit"] / max(nobj, 1), 1.0)[None, :]
e = np.minimum(d, np.abs(per - d)).max(axis=0) * ok
np.set_printoptions(precision=9, suppress=True)
for idx in np.argsort(-e)[:8]:
    xg, xr = gpu["state_end"][idx], ref["state_end"][idx]
    print(idx, "err %.2e hit %d lam gpu %.9f ref %.9f steps %d/%d" % (e[idx], gpu["hit"][idx], gpu["lambda_end"][idx], ref["lambda_end"][idx], gpu["n_accept"][idx], ref["n_accept"][idx]))
    print("    x gpu", xg[:4], "rho_cyl %.9f" % np.hypot(xg[1], xg[2]), " u", xg[4:])
    print("    x ref", xr[:4], "rho_cyl %.9f" % np.hypot(xr[1], xr[2]), " u", xr[4:])
