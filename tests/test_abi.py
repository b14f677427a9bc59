"""CPU: the C-ABI library builds, loads and exports every symbol include/rtgr.h declares; argument validation that
does not need a GPU; the product never falls back to a CPU path."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_package

rt = load_package()
abi = rt._abi


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(abi.LIB_PATH):
        subprocess.check_call(["python", os.path.join(ROOT, "raytracegr.jl_amd", "build.py")])
    return abi.load()


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "rtgr.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rtgr_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_ctypes_agree_on_exports():
    assert _declared_symbols() == sorted(abi.EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for s in _declared_symbols():
        assert hasattr(lib, s), s
    assert lib.rtgr_abi_version() == abi.RTGR_ABI_VERSION


def test_struct_sizes_match_the_header(lib):
    # layouts in include/rtgr.h: object 8+72, scene 8+16+8+16*80+8, solver 8*8+8, camera 128, counters 64, outputs 64
    assert C.sizeof(abi.rtgr_object) == 80
    assert C.sizeof(abi.rtgr_scene) == 32 + 16 * 80 + 8
    assert C.sizeof(abi.rtgr_solver) == 72
    assert C.sizeof(abi.rtgr_camera) == 128
    assert C.sizeof(abi.rtgr_counters) == 64
    assert C.sizeof(abi.rtgr_ray_outputs) == 64


def test_solver_defaults_are_the_reference_constants(lib):
    s = abi.rtgr_solver()
    assert lib.rtgr_solver_defaults(C.byref(s), 0) == 0
    assert s.reltol == s.abstol == 2.0 ** -39                     # eps(Float64)^(3/4)  src/RayTraceGR.jl:485
    assert (s.lambda0, s.lambda1, s.hit_threshold) == (0.0, 100.0, 0.01)
    assert list(s.miss_rgb) == [1.0, 0.0, 0.0]
    assert s.interp_points == 10
    p = rt.solver_defaults()
    assert (p.reltol, p.lambda1, p.hit_threshold, p.interp_points) == (s.reltol, s.lambda1, s.hit_threshold, 10)
    assert lib.rtgr_solver_defaults(C.byref(s), 1) == 0
    assert abs(s.reltol - float(np.finfo(np.float32).eps) ** 0.75) < 1e-18


def test_no_cpu_fallback_without_a_gpu(lib):
    """Without a HIP device every compute entry point must FAIL (RTGR_ERR_NO_DEVICE), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sc = rt.make_scene(rt.minkowski, [])
    opt = rt.solver_defaults()
    s0 = np.zeros((1, 8))
    rgb = np.zeros(3)
    rc = lib.rtgr_trace_f64(None, C.byref(sc), C.byref(opt), s0.ctypes.data, None, 1, 1, 0, 1, rgb.ctypes.data, None, None)
    assert rc == abi.ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.rtgr_last_error()
    with pytest.raises(abi.RtgrError):
        rt.christoffel(rt.kerr_schild, [0, 2, 0, 0])


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under raytracegr.jl_amd/, include/, examples/, julia/ or tools/ may load
    it — directly or through a test module (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do)."""
    for base in ("raytracegr.jl_amd", "include", "examples", "julia", "tools"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".sh", ".jl", ".c")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "librtgr_oracle" not in txt and "oracle_lib" not in txt and "rtgr_oracle_" not in txt, f
                    assert "test_gpu_parity" not in txt and "test_truth" not in txt, f   # (those modules import the oracle)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        abi.load(str(tmp_path / "nope.so"))


def test_arbitrary_metric_callable_is_refused():
    with pytest.raises(TypeError):
        rt.make_scene(lambda x: np.eye(4), [])


# ---- a compiled C caller: the layout a Julia ccall depends on ---------------------------------------------------------
def _build_c_caller(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_layout")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c", "abi_layout.c"),
                           "-o", exe, "-ldl", "-lm"])   # every offsetof / sizeof is a _Static_assert: checked right here
    return exe


def test_c_caller_layout_and_symbols(lib, tmp_path):
    """include/rtgr.h compiled as C11 by gcc: offsets and sizes of every struct equal the table the Julia stub is written
    against (julia/RayTraceGRHIP.jl "fieldoffset table"), ctypes agrees, and every entry point the stub binds resolves."""
    import subprocess
    exe = _build_c_caller(tmp_path)
    out = subprocess.check_output([exe, "--symbols", abi.LIB_PATH], text=True).split()
    sizes = dict(zip(out[0::2], map(int, out[1::2])))
    assert sizes == {"scene": C.sizeof(abi.rtgr_scene), "solver": C.sizeof(abi.rtgr_solver), "camera": C.sizeof(abi.rtgr_camera),
                     "counters": C.sizeof(abi.rtgr_counters), "outputs": C.sizeof(abi.rtgr_ray_outputs),
                     "object": C.sizeof(abi.rtgr_object), "pixel": rt.pixel_dtype().itemsize}
    assert abi.rtgr_scene.user_metric.offset == 24 and abi.rtgr_scene.obj.offset == 32 and abi.rtgr_scene.objects.offset == 1312
    assert abi.rtgr_solver.max_steps.offset == 64 and abi.rtgr_ray_outputs.redshift.offset == 48 and abi.rtgr_ray_outputs.hit32.offset == 56
    jl = open(os.path.join(ROOT, "julia", "RayTraceGRHIP.jl")).read()
    for name, size in (("RtgrScene", 1320), ("RtgrSolver", 72), ("RtgrCamera", 128), ("RtgrRayOutputs", 64), (r"Pixel\{Float64\}", 88)):
        assert re.search(r"^#\s+" + name + r"\s+" + str(size) + r"\s", jl, re.M), name


@pytest.mark.gpu
@pytest.mark.parametrize("ndev", [0, 3])
def test_c_caller_renders_example2(lib, tmp_path, ndev):
    """The same C program, on the GPU: example2() through rtgr_make_canvas_f64 -> an 88-byte Pixel array laid out as
    src/RayTraceGR.jl:446-450 -> rtgr_trace_pixels_f64 (+ rtgr_trace_one_f64), image bytes == the reference's sphere2.png.
    ndev = 3: through an explicit context that lists the GPU three times — the drop-in entry then deals the rows to all
    three (logical) devices, as it does to the 8 GPUs of a node."""
    import subprocess
    from raytracegr_jl_amd.png import read_png
    exe = _build_c_caller(tmp_path)
    out = str(tmp_path / "img.bin")
    res = subprocess.run([exe, "--render", abi.LIB_PATH, out] + ([str(ndev)] if ndev else []), capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    img = np.fromfile(out, np.uint8).reshape(200, 200, 3)
    gold = read_png(os.path.join(ROOT, "tests", "golden", "sphere2.png"))
    assert int((img != gold).any(axis=2).sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("ndev", [0, 2])
def test_c_caller_renders_the_kerr_disk_scene(lib, tmp_path, ndev):
    """BASELINE config 5's scene through the bytes julia/RayTraceGRHIP.jl's `render(KerrSchild(1.0, 0.998), [caelum, frustum,
    Disk(0.05, 2, 4)], cam...; details = true)` passes: RTGR_KS_TRUE with (M, a), an RTGR_DISK object, the camera struct, an
    rtgr_ray_outputs block — compiled C, rtgr_trace_f64, 64 x 64 — against the oracle: hit map, statuses, step counts, RGB at
    1e-6 (the same bytes proven for the configurations the reference's own knob cannot reach)."""
    import subprocess
    import oracle_lib as O
    from scenes import wrap_aware_rgb_err
    exe = _build_c_caller(tmp_path)
    out = str(tmp_path / "disk.bin")
    res = subprocess.run([exe, "--render-disk", abi.LIB_PATH, out] + ([str(ndev)] if ndev else []), capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    n = 64 * 64
    raw = open(out, "rb").read()
    rgb = np.frombuffer(raw, np.float64, 3 * n).reshape(3, n)
    hit = np.frombuffer(raw, np.uint8, n, 24 * n)
    status = np.frombuffer(raw, np.uint8, n, 25 * n)
    nacc = np.frombuffer(raw, np.uint32, n, 26 * n)
    nrej = np.frombuffer(raw, np.uint32, n, 30 * n)
    _, objs, cam = rt.example2_scene()
    sc = rt.make_scene(rt.KerrSchild(1.0, 0.998), objs[:2] + [rt.Disk(0.05, 2.0, 4.0)])
    ref = O.trace(sc, rt.solver_defaults(), 64, 64, cam=rt.make_camera(**cam))
    flips = hit != ref["hit"]
    assert int(flips.sum()) <= 2, int(flips.sum())
    assert (hit == 3).sum() > 50                      # the disk is in the picture
    same = ~flips
    assert (status[same] == ref["status"][same]).all()
    assert np.abs((nacc + nrej).astype(np.int64) - (ref["n_accept"] + ref["n_reject"]).astype(np.int64))[same].max() <= 2
    assert wrap_aware_rgb_err(rgb[:, same], ref["rgb"][:, same], hit[same], sc=sc) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("nobj", [16, 17, 40])
def test_c_caller_traces_an_object_list_of_any_length(lib, tmp_path, nobj):
    """`objs::Vector{Object{T}}` (src/RayTraceGR.jl:433-441, :483) from a plain C process: N objects in a caller array behind
    rtgr_scene.objects — the bytes julia/RayTraceGRHIP.jl passes as `pointer(packed)` —, the 32-bit hit map; against the oracle on the
    scene rebuilt from the objects the C program wrote.  16: the array route with a list the inline slots would hold; 17: one beyond."""
    import subprocess
    import oracle_lib as O
    from scenes import wrap_aware_rgb_err
    exe = _build_c_caller(tmp_path)
    out = str(tmp_path / "many.bin")
    res = subprocess.run([exe, "--render-many", abi.LIB_PATH, out, str(nobj)], capture_output=True, text=True)
    assert res.returncode == 0, (res.returncode, res.stderr)
    n = 48 * 48
    raw = open(out, "rb").read()
    rgb = np.frombuffer(raw, np.float64, 3 * n).reshape(3, n)
    hit = np.frombuffer(raw, np.uint32, n, 24 * n)
    status = np.frombuffer(raw, np.uint8, n, 28 * n)
    nacc = np.frombuffer(raw, np.uint32, n, 29 * n)
    nrej = np.frombuffer(raw, np.uint32, n, 33 * n)
    arr = (abi.rtgr_object * nobj).from_buffer_copy(raw[37 * n:37 * n + 80 * nobj])
    sc = rt.make_scene(rt.kerr_schild, [])
    sc.nobj, sc._keep = nobj, arr
    sc.objects = C.cast(arr, C.POINTER(abi.rtgr_object))
    _, _, cam = rt.example2_scene()
    ref = O.trace(sc, rt.solver_defaults(), 48, 48, cam=rt.make_camera(**cam))
    flips = hit != ref["hit"]
    assert int(flips.sum()) <= 4, int(flips.sum())
    assert len(np.unique(hit)) >= 6 and hit.max() == nobj                   # several spheres in view, the last of the list among them
    same = ~flips
    assert (status[same] == ref["status"][same]).all()
    sd = np.abs((nacc + nrej).astype(np.int64) - (ref["n_accept"] + ref["n_reject"]).astype(np.int64))
    steps = (ref["n_accept"] + ref["n_reject"]).astype(np.int64)
    bad = np.nonzero(same & (sd > np.maximum(2, steps * 15 // 1000)))[0]   # (captured rays, ~1300 steps until the far plane: counts within 1.5 %)
    assert len(bad) == 0, [(int(k), int(hit[k]), int(status[k]), int(nacc[k] + nrej[k]), int(ref["n_accept"][k] + ref["n_reject"][k])) for k in bad[:8]]
    assert wrap_aware_rgb_err(rgb[:, same], ref["rgb"][:, same], hit[same], sc=sc) <= 1e-6


@pytest.mark.gpu
def test_c_caller_compiles_and_traces_user_objects_of_two_sources(lib, tmp_path):
    """The reference's second extension point from a plain C process — no Python, no torch-bundled HIP runtime, no hipcc: two object
    sources joined (rtgr_user_source_join), the unit built in-process (rtgr_user_unit_compile: hiprtc + comgr resolved by the
    library), audited and probed at load, the scene checked (rtgr_scene_check) and traced (rtgr_trace_f64) — against the oracle's
    torus twin and the built-in Sphere in the ball's place.  What a Julia `trace_rays(kerr_schild, [caelum, frustum,
    DeviceObject(torus…), DeviceObject(ball…)], canvas)` does, byte for byte."""
    import subprocess
    import oracle_lib as O
    from scenes import wrap_aware_rgb_err
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import user_objects
    exe = _build_c_caller(tmp_path)
    out = str(tmp_path / "objs.bin")
    res = subprocess.run([exe, "--render-user-objects", abi.LIB_PATH, out], capture_output=True, text=True)
    assert res.returncode == 0, (res.returncode, res.stderr)
    n = 64 * 64
    raw = open(out, "rb").read()
    rgb = np.frombuffer(raw, np.float64, 3 * n).reshape(3, n)
    hit = np.frombuffer(raw, np.uint8, n, 24 * n)
    status = np.frombuffer(raw, np.uint8, n, 25 * n)
    nacc = np.frombuffer(raw, np.uint32, n, 26 * n)
    nrej = np.frombuffer(raw, np.uint32, n, 30 * n)
    _, objs, cam = rt.example2_scene()
    shapes = rt.UserObjects(user_objects.SHAPES)
    sc = rt.make_scene(rt.kerr_schild, objs[:2] + [shapes(user_objects.TORUS, [4.0, 0.0, 0.0, 0.9, 0.3]),
                                                    rt.Sphere([0.0, 4.6, -0.9, 0.9], [1.0, 0.0, 0.0, 0.0], 0.35)], units=False)
    ref = O.trace(sc, rt.solver_defaults(), 64, 64, cam=rt.make_camera(**cam))
    flips = hit != ref["hit"]
    assert int(flips.sum()) <= 2, int(flips.sum())     # (the ball's distance is written without the built-in's fused operations)
    assert (hit == 3).sum() > 100 and (hit == 4).sum() > 30                                            # torus and ball in the picture
    same = ~flips
    assert (status[same] == ref["status"][same]).all()
    assert np.abs((nacc + nrej).astype(np.int64) - (ref["n_accept"] + ref["n_reject"]).astype(np.int64))[same].max() <= 1
    assert wrap_aware_rgb_err(rgb[:, same], ref["rgb"][:, same], hit[same], sc=sc) <= 1e-6
