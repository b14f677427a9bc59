"""Run-time units are self-verifying at load (round-4 review item 2): the EXEC-flip fault of ROCm 7.2's compiler
was a SILENT wrong answer.  Beside the textual audit of the code object there is a load-time PROBE that looks at the fault's own
symptom — a 32 x 32 frame traced through the FULL pass and the FAR + NEAR passes, under two schedules each — and needs no knowledge of the
instruction shape.  Here: a unit that carries the fault, with the audit switched off, must be refused by the probe alone; sound
units pass it; a unit compiled from other device headers than the library's kernels is refused before it can overrun a workspace
(ADVICE r4); an offload bundle — what a plain `hipcc --genco` writes — is audited like a bare code object."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from scenes import rt

sys.path.insert(0, os.path.join(ROOT, "examples"))
import user_metrics  # noqa: E402

abi = rt._abi
um = sys.modules[rt.__name__ + ".user_metric"]


def _raw_unit(tmp_path, source, name, extra=(), bundle=False, header_hash=None):
    """the unit as plain `hipcc --genco` builds it: no listing check, no repair (what a C / Julia user with hipcc would hand over)"""
    unit = um.paste_source(open(um.TEMPLATE).read(), source)
    hip, out = str(tmp_path / f"{name}.hip"), str(tmp_path / f"{name}.hsaco")
    with open(hip, "w") as fh:
        fh.write(unit)
    hh = um._build.header_hash() if header_hash is None else header_hash
    cmd = [um._build.HIPCC, "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-DRTGR_USER_NE=3", f"-DRTGR_HEADER_HASH={hh:#x}ull",
           *extra, "-I", um.CSRC, "-o", out, hip]
    if not bundle:
        cmd.insert(2, "--no-gpu-bundle-output")
    subprocess.check_call(cmd)
    return out


@pytest.fixture(scope="module")
def zoo_raw(tmp_path_factory):
    d = tmp_path_factory.mktemp("raw_units")
    level1 = um.LEVELS[1]          # the occupancy compile_user_metric settles on for this metric (no scratch)
    return _raw_unit(d, user_metrics.HELPER_ZOO, "zoo", level1), _raw_unit(d, user_metrics.HELPER_ZOO, "zoo_bundle", level1, bundle=True)


def test_raw_heavy_unit_carries_the_fault_and_the_audit_sees_it_in_bundles_too(zoo_raw):
    """(CPU) every occupancy level of the heavy example metric compiles, with plain hipcc, to code with the fault; the audit finds it
    in the bare code object AND in the clang offload bundle that `hipcc --genco` writes by default (round 4 loaded those unaudited)."""
    raw, bundle = zoo_raw
    n, report = um.audit(raw)
    assert n >= 1 and "stands BEFORE the EXEC flip" in report
    nb, report_b = um.audit(bundle)
    assert nb == n and "bundle at" in report_b
    assert open(bundle, "rb").read(24) == b"__CLANG_OFFLOAD_BUNDLE__"


@pytest.fixture(scope="module")
def lib():
    lib = abi.load()
    abi.check(lib, lib.rtgr_init(-1))
    return lib


@pytest.mark.gpu
def test_probe_alone_refuses_a_unit_with_the_fault(lib, zoo_raw):
    """The fault injected AFTER the audit (option unit_audit = 0 is the hook): the probe refuses the unit on its symptom, the unit
    is not left resident, and with the audit on both the bare code object and the bundle are refused by the audit.
    The symptom is a kernel reading state it never wrote for some lanes, so what it shows depends on what was there.  Round 6 (the
    kernels with object groups): the first probe of a process refused this unit and every later one let it pass — left-overs of the
    previous, identical probe run are the RIGHT values; and registers scrubbed with a NaN made every run end the same rays "NaN",
    reproducibly.  The probe now scrubs registers, LDS and workspace before every run, alternately with a NaN and an ordinary number:
    refused every time, here five times in a row."""
    raw, bundle = zoo_raw
    for path in (raw, bundle):
        with pytest.raises(abi.RtgrError, match="FLOW block"):
            um.load(path)
    for _ in range(5):
        with abi.options(lib, unit_audit=0):
            with pytest.raises(abi.RtgrError, match="refused by the load-time probe") as e:
                um.load(raw)
        assert "differ" in str(e.value) or "disagree" in str(e.value) or "apart" in str(e.value)
    assert lib.rtgr_user_metric_loaded(None, 0) in (0, 1)          # (other tests' units may be resident; this one must not be:)
    with abi.options(lib, unit_audit=0, unit_probe=0):             # both checks off: it loads — that is what the options mean
        mid = um.load(raw)
        assert um.unit_info(mid)["probe_ok"] == 0
        um._ids.clear()
    assert lib.rtgr_user_metric_loaded(None, mid) == 1
    # asked for again with the probe back on: a resident copy that was never probed is probed NOW — refused, and gone
    with abi.options(lib, unit_audit=0):
        with pytest.raises(abi.RtgrError, match="refused by the load-time probe"):
            um.load(raw)
    um._ids.clear()
    assert lib.rtgr_user_metric_loaded(None, mid) == 0


@pytest.mark.gpu
def test_sound_units_pass_the_probe_and_say_so(lib):
    for src, st in ((user_metrics.SCHWARZSCHILD_ISOTROPIC, True), (user_metrics.HELPER_ZOO, True), (user_metrics.EXPANDING_ISOTROPIC, False)):
        mid = um.load(um.compile_user_metric(src, stationary=st))
        info = um.unit_info(mid)
        assert info["probe_ok"] == 1 and info["metric"] == abi.USER and info["has_objects"] == 0, info


@pytest.mark.gpu
def test_a_unit_built_from_other_headers_is_refused(lib, tmp_path):
    """The record layouts and argument blocks a unit shares with the library are not part of the C ABI; a unit kept on disk from an
    earlier build of the headers must not load (it would write records of another shape into the workspace)."""
    stale = _raw_unit(tmp_path, user_metrics.SCHWARZSCHILD_ISOTROPIC, "stale", header_hash=0x1234)
    with pytest.raises(abi.RtgrError, match="header hash differs"):
        um.load(stale)
    unrecorded = _raw_unit(tmp_path, user_metrics.SCHWARZSCHILD_ISOTROPIC, "by_hand", header_hash=0)   # built by hand: ABI version only
    mid = um.load(unrecorded)
    assert um.unit_info(mid)["probe_ok"] == 1


@pytest.mark.gpu
def test_loading_and_probing_a_unit_does_not_disturb_frames_traced_meanwhile(lib):
    """The probe picks the pass structure of ITS ten small traces per call (thread-local), not through the device's `split` option:
    another host thread that traces a scene of another unit while units are loaded, probed and unloaded gets the same frame every
    time, and the option is what it was."""
    import ctypes as C
    import threading

    import numpy as np
    from scenes import scene_variant
    from test_gpu_parity import hip_trace
    sc, cam = scene_variant("ks_ref0_shapes")
    opt = rt.solver_defaults()
    ref = hip_trace(lib, sc, opt, 48, 48, cam=cam)
    split0 = C.c_long(0)
    abi.check(lib, lib.rtgr_get_option(None, b"split", C.byref(split0)))
    stop, errors, frames = threading.Event(), [], [0]

    def tracer():
        try:
            while not stop.is_set():
                got = hip_trace(lib, sc, opt, 48, 48, cam=cam)
                for k in ("rgb", "hit", "n_accept", "n_reject"):
                    assert np.array_equal(got[k], ref[k]), k
                frames[0] += 1
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=tracer)
    th.start()
    try:
        path = um.compile_user_metric(user_metrics.SCHWARZSCHILD_ISOTROPIC, stationary=True)
        for _ in range(3):
            mid = um.load(path)
            assert um.unit_info(mid)["probe_ok"] == 1
            abi.check(lib, lib.rtgr_user_metric_unload(None, mid))
            um._ids.clear()
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    assert frames[0] >= 3
    now = C.c_long(0)
    abi.check(lib, lib.rtgr_get_option(None, b"split", C.byref(now)))
    assert now.value == split0.value
