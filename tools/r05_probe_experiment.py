#!/usr/bin/env python3
"""Round 5, GPU: what the load-time probe says about units that carry the compiler's EXEC-flip fault (built raw with `hipcc --genco`,
audit skipped), about the repaired example units, and how long it takes.  -> stdout (profiles/r05/unit_probe.log)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
from __graft_entry__ import load_package  # noqa: E402

rt = load_package()
import user_metrics as umx  # noqa: E402
import user_objects as uo  # noqa: E402

abi = rt._abi
um = sys.modules[rt.__name__ + ".user_metric"]
lib = abi.load()
abi.check(lib, lib.rtgr_init(-1))
out = os.path.join(ROOT, "gpurun_out", "r05", "probe")
os.makedirs(out, exist_ok=True)
REPEAT = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hh = f"-DRTGR_HEADER_HASH={um._build.header_hash():#x}ull"
for name, src in (("HELPER_ZOO", umx.HELPER_ZOO), ("KERR_BOYER_LINDQUIST", umx.KERR_BOYER_LINDQUIST)):
    unit = um.paste_source(open(um.TEMPLATE).read(), src)
    hip = os.path.join(out, name + ".hip")
    open(hip, "w").write(unit)
    for lvl, level in enumerate(um.LEVELS):
        raw = os.path.join(out, f"{name}_L{lvl}.hsaco")
        subprocess.check_call([um._build.HIPCC, "--genco", "--no-gpu-bundle-output", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-DRTGR_USER_NE=3", hh]
                              + level + ["-I", um.CSRC, "-o", raw, hip])
        n_audit = um.audit(raw)[0]
        for audit_on in (1, 0):
            t = time.time()
            try:
                with abi.options(lib, unit_audit=audit_on):
                    mid = um.load(raw)
                verdict = f"LOADED (probe_ok {um.unit_info(mid)['probe_ok']})"
                lib.rtgr_user_metric_unload(None, mid)
                um._ids.clear()
            except abi.RtgrError as e:
                verdict = "REFUSED: " + str(e).split("\n")[0][:230]
            print(f"{name} raw level {lvl}: audit finds {n_audit} block(s); load with unit_audit={audit_on}: {verdict}  [{time.time() - t:.2f} s]", flush=True)
        if n_audit:   # how RELIABLY the probe alone refuses the faulty unit: REPEAT loads with the audit off, by the probe's reason
            why = {}
            for _ in range(REPEAT):
                try:
                    with abi.options(lib, unit_audit=0):
                        mid = um.load(raw)
                    lib.rtgr_user_metric_unload(None, mid)
                    um._ids.clear()
                    k = "LOADED (not refused)"
                except abi.RtgrError as e:
                    k = str(e).split("probe — ")[-1].split(" (the symptom")[0][:110]
                why[k] = why.get(k, 0) + 1
            print(f"{name} raw level {lvl}: {REPEAT} loads with the audit off: {why}", flush=True)
# the repaired / sound units: probe cost
for name, src, st in (("KERR_SCHILD", umx.KERR_SCHILD, True), ("KERR_SCHILD_KS", umx.KERR_SCHILD_KS, True), ("HELPER_ZOO (repaired)", umx.HELPER_ZOO, True),
                      ("KERR_BOYER_LINDQUIST (repaired)", umx.KERR_BOYER_LINDQUIST, True), ("EXPANDING_ISOTROPIC", umx.EXPANDING_ISOTROPIC, False)):
    path = um.compile_user_metric(src, stationary=st)
    t = [0, 0]
    for k, probe in enumerate((1, 0)):
        t0 = time.time()
        with abi.options(lib, unit_probe=probe):
            mid = um.load(path)
        t[k] = time.time() - t0
        info = um.unit_info(mid)
        lib.rtgr_user_metric_unload(None, mid)
        um._ids.clear()
    print(f"{name}: load with probe {t[0] * 1e3:.0f} ms, without {t[1] * 1e3:.0f} ms; {info}", flush=True)
path = um.compile_user_metric(uo.SHAPES_WITH_REACH, built_for=(abi.KS_REF, False, False))
t0 = time.time(); mid = um.load(path)
print(f"SHAPES for ks_ref a=0: load with probe {(time.time() - t0) * 1e3:.0f} ms; {um.unit_info(mid)}", flush=True)
