"""Build-time checks on the generated gfx950 ISA (CPU only: hipcc cross-compiles without a GPU).

The one-instruction min / max / clamp helpers of the hot loop are raw inline asm (rtgr_physics.hpp rmin / rmax / rmaxabs,
rtgr_persistent.hpp fmax1 / fmin1).  LLVM's hazard recogniser treats inline asm as opaque, so a transcendental result
(v_rcp / v_rsq / v_exp / v_log …) consumed by the very next VALU instruction would get no s_nop for the gfx940-family
"trans use" hazard.  tools/isa_mix.py --hazards scans the final ISA of every integrate / prepare kernel for that
adjacency; this test fails the suite when one appears."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "raytracegr.jl_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.parametrize("unit", ["tu_f64_ksref", "tu_f64_kstrue", "tu_f32_closed"])
def test_no_trans_use_hazard_next_to_inline_asm(unit, tmp_path):
    asm = str(tmp_path / f"{unit}.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only",
                           "-Wno-unused-command-line-argument", "-o", asm, os.path.join(CSRC, unit + ".hip")],
                          stderr=subprocess.DEVNULL)
    kernels = [l.split(":")[0] for l in open(asm) if l.startswith("_ZN4rtgr") and l.rstrip().split(";")[0].rstrip().endswith(":")]
    kernels = [k for k in kernels if "integrate" in k or "prepare_kernel" in k or "resolve_kernel" in k]
    assert len(kernels) >= 8, kernels
    for k in kernels:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), asm, k, "--hazards"],
                           capture_output=True, text=True)
        assert r.returncode == 0, (k, r.stdout[-2000:])
    # the same listing through the EXEC-flip check: the strict rule, and the wider heuristic net that would point
    # at a misplaced allocator copy even where the compiler has dropped the skip branch
    ie = _isa_exec()
    lines = open(asm).read().split("\n")
    assert ie.find(lines) == [] and ie.suspicious(lines) == [], (ie.find(lines), ie.suspicious(lines))


def test_plain_build_relinks_after_an_experiment_build(tmp_path):
    """build.py records which flag set librtgr_hip.so was linked from (librtgr_hip.so.tag).  Sequence of ADVICE r2: a plain
    build, an experiment build (-DRTGR_ROOT_STATS), a plain build again — the last one finds its own objects fresh AND older
    than the library, and must still relink, or tests and bench would run the experiment binary.  (A stand-in compiler
    that only creates its output file: the logic under test is the driver's, not hipcc's.)"""
    import importlib.util
    fake = tmp_path / "fake_hipcc"
    log = tmp_path / "calls.log"
    fake.write_text("#!/bin/sh\nprev=\nfor a in \"$@\"; do [ \"$prev\" = -o ] && out=\"$a\"; prev=\"$a\"; done\n"
                    f"echo \"$@\" >> {log}\n: > \"$out\"\n")
    fake.chmod(0o755)
    spec = importlib.util.spec_from_file_location("rtgr_build_t", os.path.join(ROOT, "raytracegr.jl_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.HIPCC = str(fake)
    out, obj = str(tmp_path / "lib.so"), str(tmp_path / "obj")

    def links():
        return sum(1 for l in open(log) if " -shared " in l) if log.exists() else 0

    m.build(out=out, obj_dir=obj, verbose=False)
    assert links() == 1 and m.linked_tag(out) == "std"
    m.build(out=out, obj_dir=obj, verbose=False)
    assert links() == 1                                   # nothing to do
    m.build(out=out, obj_dir=obj, verbose=False, extra=("-DRTGR_ROOT_STATS",))
    assert links() == 2 and m.linked_tag(out) not in (None, "std")
    m.build(out=out, obj_dir=obj, verbose=False)          # objects of the plain build are fresh and OLDER than the library
    assert links() == 3 and m.linked_tag(out) == "std"


# ---------------------------------------------------------------------------------------------------------------------
# The EXEC-flip fault of ROCm 7.2's LLVM (raytracegr.jl_amd/isa_exec.py, csrc/rtgr_isa_audit.hpp)
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
FAULTY_LISTING = """\
	.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
	.text
	.globl	probe
	.p2align	8
	.type	probe,@function
probe:
	v_cmp_gt_u32_e32 vcc, 7, v0
	s_and_saveexec_b64 s[6:7], vcc
	s_xor_b64 s[2:3], exec, s[6:7]
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_mov_b32_e32 v1, 1
.LBB0_2:                                ; the FLOW block: what the allocator put ahead of the flip runs for the `then` lanes only
	v_accvgpr_write_b32 a0, v2
	s_mov_b32 s8, 0x54442d18
	scratch_store_dwordx2 off, v[4:5], off offset:16
	s_andn2_saveexec_b64 s[2:3], s[2:3]
	s_cbranch_execz .LBB0_4
; %bb.3:
	v_mov_b32_e32 v1, 2
.LBB0_4:
	s_or_b64 exec, exec, s[2:3]
	v_accvgpr_read_b32 v3, a0
	s_endpgm
.Lfunc_end0:
	.size	probe, .Lfunc_end0-probe
"""


def _isa_exec():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rtgr_isa_exec_t", os.path.join(ROOT, "raytracegr.jl_amd", "isa_exec.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _assemble(listing, out):
    asm = out + ".s"
    with open(asm, "w") as fh:
        fh.write(listing)
    subprocess.check_call([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm, "-o", out + ".o"])
    subprocess.check_call([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "-shared", "-o", out, out + ".o"])
    return out


def test_exec_flip_fault_is_found_and_repaired_in_a_listing_and_in_a_code_object(tmp_path):
    """The shape in a hand-written listing: isa_exec.find reports it, isa_exec.repair rewrites the flip into s_or_saveexec … s_xor
    around the misplaced instructions (and refuses a block it is not proven for); the library's load-time audit
    (rtgr_code_object_audit: the same rule on the DISASSEMBLY) sees it in the assembled code object and not in the repaired one."""
    ie = _isa_exec()
    lines = FAULTY_LISTING.split("\n")
    hits = ie.find(lines)
    assert len(hits) == 1 and hits[0].label == ".LBB0_2" and [s.split()[0] for _, s in hits[0].early] == ["v_accvgpr_write_b32", "scratch_store_dwordx2"]
    assert [c.split()[0] for _, c, _ in ie.suspicious(lines)] == ["scratch_store_dwordx2"]     # (the wider net sees the last one)
    fixed, n = ie.repair(lines)
    assert n == 1 and ie.find(fixed) == [] and ie.suspicious(fixed) == []
    body = [l.split(";")[0].strip() for l in fixed]
    at = body.index(".LBB0_2:")
    assert body[at + 1] == "s_or_saveexec_b64 s[2:3], s[2:3]" and body[at + 5] == "s_xor_b64 exec, exec, s[2:3]"
    assert "s_andn2_saveexec_b64 s[2:3], s[2:3]" not in body
    # a clean listing is returned as it is; a block with anything but copies / spills / constants ahead of the flip is refused
    assert ie.repair(fixed) == (fixed, 0)
    odd = FAULTY_LISTING.replace("v_accvgpr_write_b32 a0, v2", "v_add_f32_e32 v2, v2, v2")
    with pytest.raises(ie.RepairError):
        ie.repair(odd.split("\n"))
    masks = FAULTY_LISTING.replace("s_mov_b32 s8, 0x54442d18", "s_mov_b32 s2, 0x54442d18")
    with pytest.raises(ie.RepairError):
        ie.repair(masks.split("\n"))
    # the `then` side's own instructions ahead of a flip in a block nothing branches to are NOT the shape
    merged = FAULTY_LISTING.replace("\ts_cbranch_execz .LBB0_2\n", "")
    assert ie.find(merged.split("\n")) == []
    from scenes import rt
    um = sys.modules[rt.__name__ + ".user_metric"]
    bad = _assemble(FAULTY_LISTING, str(tmp_path / "faulty.hsaco"))
    good = _assemble("\n".join(fixed), str(tmp_path / "fixed.hsaco"))
    n_bad, report = um.audit(bad)
    assert n_bad == 1 and "v_accvgpr_write_b32 a0, v2" in report and "scratch_store_dwordx2" in report and "s_andn2_saveexec_b64" in report
    assert um.audit(good) == (0, "")
    with pytest.raises(RuntimeError, match="cannot be audited"):
        um.audit(os.path.join(ROOT, "include", "rtgr.h"))


def test_the_librarys_listing_repair_gives_the_python_modules_answers(tmp_path):
    """rtgr_listing_repair (csrc/rtgr_isa_repair.hpp, the step of the in-process build) against raytracegr.jl_amd/isa_exec.py on the
    same listings: the hand-written one and its variants, and the real listing of the heavy example metric at the occupancy level
    where this compiler produces the fault — same count, same rewritten text."""
    import ctypes as C
    from scenes import rt
    ie = _isa_exec()
    lib = rt._abi.load()

    def both(listing, name):
        src, dst = str(tmp_path / f"{name}.s"), str(tmp_path / f"{name}.fixed.s")
        with open(src, "w") as fh:
            fh.write(listing)
        n_found, n_fixed = C.c_int32(-1), C.c_int32(-1)
        assert lib.rtgr_listing_repair(src.encode(), None, C.byref(n_found)) == 0
        lines = listing.split("\n")
        assert n_found.value == len(ie.find(lines)), name
        try:
            want, n = ie.repair(lines)
        except ie.RepairError:
            assert lib.rtgr_listing_repair(src.encode(), dst.encode(), C.byref(n_fixed)) == rt._abi.ERR_BAD_ARG and not os.path.exists(dst), name
            return None
        assert lib.rtgr_listing_repair(src.encode(), dst.encode(), C.byref(n_fixed)) == 0 and n_fixed.value == n, name
        strip = lambda ls: [l.split(" ; isa_")[0].rstrip() for l in ls]      # (the two sign their inserted line differently)
        assert strip(open(dst).read().split("\n")) == strip(want), name
        return n

    assert both(FAULTY_LISTING, "faulty") == 1
    assert both(FAULTY_LISTING.replace("\ts_cbranch_execz .LBB0_2\n", ""), "merged") == 0
    assert both(FAULTY_LISTING.replace("v_accvgpr_write_b32 a0, v2", "v_add_f32_e32 v2, v2, v2"), "odd") is None
    assert both(FAULTY_LISTING.replace("s_mov_b32 s8, 0x54442d18", "s_mov_b32 s2, 0x54442d18"), "masks") is None
    unfused = FAULTY_LISTING.replace("s_andn2_saveexec_b64 s[2:3], s[2:3]", "s_or_saveexec_b64 s[2:3], s[2:3]\n\ts_xor_b64 exec, exec, s[2:3]")
    assert both(unfused, "unfused") == 1
    no_else = FAULTY_LISTING.replace("s_andn2_saveexec_b64 s[2:3], s[2:3]", "s_or_b64 exec, exec, s[2:3]")
    assert both(no_else, "no_else") == 1
    # the real thing: the heavy metric's unit at one wave per SIMD
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import user_metrics
    um = sys.modules[rt.__name__ + ".user_metric"]
    unit = um.compile_user_metric(user_metrics.HELPER_ZOO, stationary=True)[:-len(".hsaco")] + ".hip"
    raw = str(tmp_path / "zoo.s")
    subprocess.check_call([HIPCC, "--cuda-device-only", "--no-gpu-bundle-output", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "-DRTGR_USER_NE=3",
                           "-DRTGR_WAVES_PER_SIMD_GENERIC=1", "-DRTGR_WAVES_PER_SIMD_GENERIC_F32=2", "-I", CSRC, "-o", raw, unit], stderr=subprocess.DEVNULL)
    n = both(open(raw).read(), "zoo")
    assert n is not None      # (n > 0 with ROCm 7.2.0; a compiler that no longer produces the fault gives 0, and the test still holds)


def test_the_kernels_the_library_ships_are_free_of_the_exec_flip_fault():
    """Every gfx950 code object embedded in librtgr_hip.so (one offload bundle per translation unit), audited by the library itself."""
    from scenes import rt
    um = sys.modules[rt.__name__ + ".user_metric"]
    n, report = um.audit(rt._abi.LIB_PATH)
    assert n == 0, report


def test_the_library_in_the_tree_was_built_from_the_headers_in_the_tree():
    """librtgr_hip.so is git-ignored but travels to the GPU box as it lies; a library left over from before an edit of a device header
    (or of include/rtgr.h, which they include) refuses every run-time unit built afterwards ("built from other device headers") — on the
    GPU box, minutes later.  Say so here: the hash compiled into the library equals the hash of the headers on disk."""
    import ctypes as C
    from scenes import rt
    lib = rt._abi.load()
    fn = lib.rtgr_testhook_header_hash
    fn.restype = C.c_ulonglong
    assert fn() == _build_module("rtgr_build_t3").header_hash(), "stale librtgr_hip.so: python -c 'import __graft_entry__ as g; g.build()'"


def test_no_kernel_of_the_library_keeps_its_arguments_in_scratch(tmp_path):
    """A kernel's argument block (ResolveArgs / IntegrateArgs with the scene's 16 inline objects: 1.4 KB) is read with scalar loads
    from the kernarg segment.  One unlucky access pattern — a per-lane index into the inline objects next to a walk over the device
    table — and the compiler copies the whole block into every lane's scratch memory first: the resolve kernel of a three-object
    scene went from 0.38 to 2.5 ms that way (round 6, caught by the cost table only).  So: no kernel the library ships holds more
    than the few spill slots the hot kernels are known to have (<= 64 bytes per lane, RTGR_USER_MAX_SCRATCH's bar for units)."""
    import re
    import struct
    from scenes import rt
    blob = open(rt._abi.LIB_PATH, "rb").read()
    sizes = {}
    for k, m in enumerate(re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)):
        o = m.start()
        n = struct.unpack_from("<Q", blob, o + 24)[0]
        p = o + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size:
                co = tmp_path / f"{k}.hsaco"
                co.write_bytes(blob[o + off:o + off + size])
                txt = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True, check=True).stdout
                cur = None
                for line in txt.splitlines():
                    mm = re.match(r"\s+\.name:\s+(\S+)", line)
                    if mm:
                        cur = mm.group(1)
                    mm = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
                    if mm and cur:
                        sizes[cur] = int(mm.group(1))
    assert len(sizes) > 40 and any("resolve_kernel" in k for k in sizes) and any("integrate" in k for k in sizes)
    fat = {k: v for k, v in sizes.items() if v > 64}
    assert not fat, fat


def _build_module(name="rtgr_build_t2"):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "raytracegr.jl_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _embedded_code(path):
    """disassembly (instruction text only) of the gfx950 code objects bundled into an object file / library"""
    import re
    import struct
    blob, out = open(path, "rb").read(), []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        o = m.start()
        n = struct.unpack_from("<Q", blob, o + 24)[0]
        p = o + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size:
                co = path + f".{len(out)}.hsaco"
                with open(co, "wb") as fh:
                    fh.write(blob[o + off:o + off + size])
                dis = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
                out.append([l.split("//")[0].rstrip() for l in dis.splitlines() if "file format" not in l])
    return out


def test_a_library_unit_built_through_its_listing_is_the_same_code(tmp_path):
    """build.py's fallback route for the library's own kernels (taken when the post-link audit finds the EXEC-flip fault, or with
    --via-listing): device half to a listing, check / repair, assembler, lld, bundler, host half with the fat binary included —
    for a unit that needs no repair the object carries the same device instructions and the same host symbols as `hipcc -c`."""
    b = _build_module()
    src = os.path.join(CSRC, "tu_f64_mink.hip")
    via, std = str(tmp_path / "via.o"), str(tmp_path / "std.o")
    assert b.compile_via_listing(src, via, verbose=False) == 0
    subprocess.check_call([b.HIPCC] + b.FLAGS + ["-c", "-o", std, src], stderr=subprocess.DEVNULL)
    code_via, code_std = _embedded_code(via), _embedded_code(std)
    assert len(code_via) == len(code_std) == 1 and code_via == code_std and len(code_via[0]) > 1000
    syms = lambda p: sorted(l.split()[-1] for l in subprocess.run(["nm", p], capture_output=True, text=True, check=True).stdout.splitlines()
                            if "__hip_cuid_" not in l and "__hip_gpubin_handle_" not in l)   # (named after the compilation-unit id: the listing route fixes its own)
    assert syms(via) == syms(std)
    assert not [f for f in os.listdir(tmp_path) if f.endswith((".hipfb", ".dev.o"))]      # intermediates are removed


def test_build_falls_back_to_the_listing_route_when_the_audit_finds_the_fault(tmp_path):
    """The driver logic with stand-in tools (every tool just creates its output file): a post-link audit that reports the fault makes
    build() compile the kernel units again through compile_via_listing — not the kernel-free host units (rtgr_context.hip …) — and link again; an
    audit that still reports it after that is an error."""
    b = _build_module("rtgr_build_t3")
    log = tmp_path / "calls.log"
    tool = ("#!/bin/sh\nprev=\nfor a in \"$@\"; do [ \"$prev\" = -o ] && out=\"$a\"; case \"$a\" in -output=*) out=\"${a#-output=}\";; esac; prev=\"$a\"; done\n"
            f"echo \"$(basename $0) $@\" >> {log}\n: > \"$out\"\n")
    bindir = tmp_path / "bin"
    bindir.mkdir()
    for name in ("hipcc", "clang", "lld", "clang-offload-bundler"):
        (bindir / name).write_text(tool)
        (bindir / name).chmod(0o755)
    b.HIPCC, b.LLVM_BIN = str(bindir / "hipcc"), str(bindir)
    answers = [(2, "  .text+0x10: `v_accvgpr_write_b32 a0, v2` stands BEFORE the EXEC flip"), (0, "")]
    b.audit = lambda lib: answers.pop(0)
    out, obj = str(tmp_path / "lib.so"), str(tmp_path / "obj")
    b.build(out=out, obj_dir=obj, verbose=False)
    calls = open(log).read().splitlines()
    links = [c for c in calls if " -shared " in c and c.startswith("hipcc")]
    device_listings = [c for c in calls if "--cuda-device-only" in c]
    assert len(links) == 2 and len(device_listings) == len(b.UNITS) - len(b.HOST_ONLY_UNITS)
    assert not any(h in c for c in device_listings for h in b.HOST_ONLY_UNITS) and answers == []
    assert sum(1 for c in calls if c.startswith("clang-offload-bundler")) == len(device_listings)
    answers.extend([(1, "x"), (1, "x")])
    with pytest.raises(RuntimeError, match="survive the listing route"):
        b.build(out=out, obj_dir=obj, verbose=False, force=True)


def test_the_audit_survives_truncated_and_corrupted_files(tmp_path):
    """rtgr_code_object_audit (and through it rtgr_user_metric_load) reads UNTRUSTED files: truncated images, section tables and
    bundle entries whose offsets / sizes lie — including values near 2^64, where `off + size > bytes` wraps — must come
    back as RTGR_ERR_BAD_ARG or a count, never as a crash.  Run in a child process so that a crash is a test failure, not the end
    of the test run: 12 targeted corruptions of a real unit and of a real offload bundle + 300 seeded random ones."""
    code = r'''
import ctypes, os, random, struct, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "examples"))
from __graft_entry__ import load_package
rt = load_package()
import user_metrics
um = sys.modules[rt.__name__ + ".user_metric"]
lib = rt._abi.load()
good = open(um.compile_user_metric(user_metrics.SCHWARZSCHILD_ISOTROPIC, stationary=True), "rb").read()
tmp = sys.argv[2]
def audit(blob):
    p = os.path.join(tmp, "case.bin")
    open(p, "wb").write(blob)
    n = ctypes.c_int32(-7); buf = ctypes.create_string_buffer(4096)
    rc = lib.rtgr_code_object_audit(p.encode(), ctypes.byref(n), buf, 4096)
    assert rc in (0, -1), rc
    return rc, n.value
assert audit(good) == (0, 0)
U64 = 0xFFFFFFFFFFFFFFFF
e_shoff, e_shnum, e_shstrndx = struct.unpack_from("<Q", good, 0x28)[0], struct.unpack_from("<H", good, 0x3C)[0], struct.unpack_from("<H", good, 0x3E)[0]
cases = [good[:n] for n in (0, 1, 16, 63, 64, 200, e_shoff, e_shoff + 17, len(good) - 1)]
def patch(off, fmt, val):
    b = bytearray(good); struct.pack_into(fmt, b, off, val); return bytes(b)
cases += [patch(0x28, "<Q", v) for v in (U64, U64 - 63, len(good) - 8, 1 << 63)]                       # e_shoff
cases += [patch(0x3C, "<H", 0xFFFF), patch(0x3E, "<H", 0xFFFF), patch(0x3A, "<H", 1)]                   # e_shnum, e_shstrndx, e_shentsize
for k in range(e_shnum):                                                                                # every section's offset / size
    base = e_shoff + 64 * k
    cases += [patch(base + 0x18, "<Q", U64 - 7), patch(base + 0x20, "<Q", U64), patch(base + 0x18, "<Q", len(good) - 3), patch(base + 0x00, "<I", 0xFFFFFFF0)]
# an offload bundle around the good image, then with lying entries
triple = b"hipv4-amdgcn-amd-amdhsa--gfx950"
def bundle(off, size, tl=None, n=1):
    head = b"__CLANG_OFFLOAD_BUNDLE__" + struct.pack("<Q", n) + struct.pack("<QQQ", off, size, len(triple) if tl is None else tl) + triple
    return head + b"\0" * (4096 - len(head)) + good
assert audit(bundle(4096, len(good))) == (0, 0)
cases += [bundle(U64 - 5, 64), bundle(4096, U64), bundle(U64, U64), bundle(4096, len(good) + 1), bundle(4096, len(good), tl=U64), bundle(4096, len(good), tl=300),
          bundle(4096, len(good), n=U64), bundle(1 << 62, 1 << 62)]
rng = random.Random(5)
for _ in range(300):
    b = bytearray(good if rng.random() < 0.7 else bundle(4096, len(good)))
    for _ in range(rng.randint(1, 6)):
        at = rng.randrange(0, min(len(b), e_shoff + 64 * e_shnum if rng.random() < 0.5 else 4200))
        b[at] = rng.randrange(256)
    if rng.random() < 0.3:
        b = b[:rng.randrange(1, len(b))]
    cases.append(bytes(b))
seen = {0: 0, -1: 0}
for c in cases:
    rc, n = audit(c)
    seen[rc] += 1
print("cases", len(cases), "ok", seen[0], "refused", seen[-1])
'''
    r = subprocess.run([sys.executable, "-c", code, ROOT, str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "cases" in r.stdout and "refused" in r.stdout
