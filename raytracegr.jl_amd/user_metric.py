"""Run-time compiled metrics: the host side of rtgr_user_metric_load (include/rtgr.h).

The reference takes the metric as ANY Julia callable `x -> SMatrix{4,4}` and lets the compiler specialise dmetric /
geodesic / solve on it (src/RayTraceGR.jl:302-309, :358-370, :457-511).  The MI355X counterpart keeps that freedom
without a tracing compiler: the metric is written once as a C++ function template over the scalar type, pasted into
csrc/rtgr_user_unit.hip.in, compiled for gfx950 with `hipcc --genco`, and the code object — canvas, FAR / NEAR / FULL
integrate passes and the evaluation hooks specialised on that metric — is loaded into the running library.

    m = UserMetric('''
        template <class S> __device__ void rtgr_user_metric(const S x[4], double M, double a, S g[4][4]) {
            ...fill all 16 entries of the symmetric g...
        }''', M=1.0)
    trace_rays(m, objs, canvas)        # every entry point that takes a metric accepts it

Code objects are cached by content hash (source + the headers they were built from) under
raytracegr.jl_amd/build/user/ (override: RTGR_USER_CACHE), so a metric is compiled once.
"""
import hashlib
import re
import os
import subprocess

from . import _abi
from . import build as _build
from . import isa_exec as _isa

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
TEMPLATE = os.path.join(CSRC, "rtgr_user_unit.hip.in")
_HEADERS = [os.path.join(CSRC, f) for f in ("rtgr_args.hpp", "rtgr_physics.hpp", "rtgr_integrator.hpp",
                                            "rtgr_persistent.hpp", "rtgr_tsit5_tables.hpp")] + \
           [os.path.join(HERE, "..", "include", "rtgr.h")]
FLAGS = ["--cuda-device-only", "--no-gpu-bundle-output", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wall", "-Wno-unused-function"]
LLVM_BIN = os.path.join(os.path.dirname(os.path.realpath(_build.HIPCC)), "..", "lib", "llvm", "bin")
_ids = {}  # (context handle value or None, code object path) -> module id returned by rtgr_user_metric_load


def cache_dir():
    d = os.environ.get("RTGR_USER_CACHE") or os.path.join(HERE, "build", "user")
    os.makedirs(d, exist_ok=True)
    return d


def _env_digest():
    """what EVERY unit is built from besides its own source: the template, the device headers, the listing check / repair, the flags.
    It leads a unit's file name (metric_<env>_<source>.hsaco), so the units of an earlier state of those files can be told by name."""
    h = hashlib.sha256()
    for f in [TEMPLATE, _isa.__file__] + _HEADERS:   # (the listing check / repair is part of how a unit is built)
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:8]


# the files compile_user_metric writes into the cache directory: metric_<env 8 hex>_<source 20 hex> + .hip / .hsaco / .L<level>….s,
# each possibly with a .tmp<pid> part while it is being written (and the names of earlier rounds, without the <env> part)
_UNIT_FILE = re.compile(r"^metric_(?:([0-9a-f]{8})_)?[0-9a-f]{20}(?:\.L\d+)?(?:\.tmp\d+)?\.(?:hsaco|hip|s|o)(?:\.tmp\d+)?(?:\.o)?$")


def prune_stale_units(verbose=False):
    """Delete cached units that can no longer be named: a unit's file name leads with the digest of the template, the device headers
    and the listing tool (_env_digest), so a file whose name carries ANOTHER digest belongs to a state of those files nothing computes
    any more (each header edit used to leave ~40 files, 8 MB, behind — all of which travel to the GPU box with every gpurun call).
    Decided by NAME and content hash only: files that are not units (RTGR_USER_CACHE may point at a shared directory) are never
    touched, and neither a `touch` nor a `git checkout` makes a valid unit look stale.  -> files removed"""
    cur = _env_digest()
    d, gone = cache_dir(), 0
    for name in os.listdir(d):
        m = _UNIT_FILE.match(name)
        f = os.path.join(d, name)
        if m and m.group(1) != cur and os.path.isfile(f):
            os.unlink(f)
            gone += 1
    if verbose and gone:
        print(f"{d}: removed {gone} unit file(s) built from other device headers than the current ones")
    return gone


def _digest(source, extra_flags=()):
    h = hashlib.sha256()
    h.update(source.encode())
    h.update(" ".join(extra_flags).encode())
    return _env_digest() + "_" + h.hexdigest()[:20]


def _code_only(source):
    """the source without its comments and string literals (what the compiler sees of it)"""
    return re.sub(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\\n])*"', " ", source, flags=re.S)


def defines(source, name):
    """True when `source` declares or defines the function `name` — the identifier followed by `(`, outside comments.  (A comment
    that merely mentions rtgr_user_reach used to switch -DRTGR_USER_REACH=1 on and the build then failed on an undefined template,
    ADVICE r5.)  The same rule as rtgr_units.hip: source_defines."""
    return re.search(r"(?<![A-Za-z0-9_])" + re.escape(name) + r"\s*\(", _code_only(source)) is not None


def paste_source(template, source):
    """the unit template with the caller's source in place.  The `#line` directive behind the source names the TEMPLATE line that
    follows it — computed here from where the directive stands, not written into the template by hand (a hard-coded number drifted
    with every edit of the template's header comment).  The same rule as rtgr_units.hip: build_unit_image."""
    lines = template.split("\n")
    at = next(i for i, l in enumerate(lines) if "@RTGR_TEMPLATE_LINE@" in l)
    lines[at] = lines[at].replace("@RTGR_TEMPLATE_LINE@", str(at + 2))    # (the directive is line at + 1: the line after it is at + 2)
    return "\n".join(lines).replace("@RTGR_USER_SOURCE@", source)


def unit_defines(source, stationary=False, built_for=None):
    """What a unit is made of, read off its source text and the metric variant it is meant for — the same rules as rtgr_units.hip:
    plan_unit.  built_for = (metric kind, generic, spin) of a built-in metric for a source that defines OBJECTS only."""
    ks_form = defines(source, "rtgr_user_ks")    # the metric given in Kerr–Schild form: f and k instead of the 16 entries
    metric = ks_form or defines(source, "rtgr_user_metric")
    dist, colr = defines(source, "rtgr_user_distance"), defines(source, "rtgr_user_objcolor")
    if dist != colr:
        raise ValueError("objects need both methods of the reference's Object (src/RayTraceGR.jl:377-389): rtgr_user_distance AND rtgr_user_objcolor")
    if not metric and not dist:
        raise ValueError("the source must define `template <class S> __device__ void rtgr_user_metric(const S x[4], "
                         "double M, double a, S g[4][4])` (or rtgr_user_ks(const S x[4], double M, double a, S& f, S k[3]) "
                         "for a metric of Kerr-Schild form) and / or the object methods rtgr_user_distance / rtgr_user_objcolor")
    extra = []
    if metric:
        if built_for is not None:
            raise ValueError("the source defines a metric of its own: built_for must be None")
        if stationary or ks_form:
            extra.append("-DRTGR_USER_NE=3")
        if ks_form:
            extra.append("-DRTGR_USER_KS=1")
    else:
        if built_for is None:
            raise ValueError("a unit of objects alone is built for ONE built-in metric variant: built_for = (kind, generic, spin)")
        kind, generic, spin = int(built_for[0]), bool(built_for[1]), bool(built_for[2])
        if kind == _abi.MINKOWSKI:
            generic = spin = False
        if not 0 <= kind < _abi.USER:
            raise ValueError("built_for names no built-in metric")
        extra += [f"-DRTGR_UNIT_BUILTIN_METRIC={kind}", f"-DRTGR_UNIT_GENERIC={int(generic)}", f"-DRTGR_UNIT_SPIN={int(spin and not generic)}"]
    if dist:
        extra.append("-DRTGR_USER_OBJECTS=1")
        if defines(source, "rtgr_user_reach"):
            extra.append("-DRTGR_USER_REACH=1")
        if defines(source, "rtgr_user_sample"):
            extra.append("-DRTGR_USER_SAMPLE=1")
    elif defines(source, "rtgr_user_reach"):
        raise ValueError("rtgr_user_reach without rtgr_user_distance / rtgr_user_objcolor")
    return extra, (stationary or ks_form) if metric else False


def compile_user_metric(source, verbose=False, stationary=False, built_for=None):
    """Build (or fetch from the cache) the code object of a run-time unit — a user metric, user objects, or both; returns its
    path.  Needs hipcc, no GPU.
    stationary=True declares that the metric does not depend on t: the integrate kernels then carry the three spatial
    partials only (-DRTGR_USER_NE=3), a quarter less dual arithmetic.  built_for: see unit_defines."""
    extra, stat = unit_defines(source, stationary, built_for)
    d = cache_dir()
    tag = _digest(source, extra + [f"max_scratch={MAX_SCRATCH}"])
    extra = extra + [f"-DRTGR_HEADER_HASH={_build.header_hash():#x}ull"]   # (the headers are part of the digest already)
    out = os.path.join(d, f"metric_{tag}.hsaco")
    if os.path.exists(out):
        return out
    if os.environ.get("RTGR_NO_COMPILE") == "1":
        # set by multi-rank and profiled runs (bench.py, tools/prof.sh): N ranks must not all start hipcc on a cold cache, and
        # a child process must not be spawned from inside a rocprofv3 --pmc session (the preloaded tool has initialised the GPU)
        raise RuntimeError(f"user metric {tag} is not in the cache ({d}) and RTGR_NO_COMPILE=1: build it first in a single "
                           f"plain process (__graft_entry__.build() precompiles the example metrics; "
                           f"user_metric.compile_user_metric(source) any other)")
    if not os.path.exists(_build.HIPCC):
        # no hipcc on this box (a runtime-only ROCm): the library builds the same unit in-process (hiprtc + libamd_comgr, listing
        # checked and repaired the same way) — cached here under the same content hash
        tmp = out + f".tmp{os.getpid()}"
        build_in_process(source, tmp, stationary=bool(stat), built_for=built_for)
        os.replace(tmp, out)
        return out
    unit = paste_source(open(TEMPLATE).read(), source)
    src = os.path.join(d, f"metric_{tag}.hip")
    tmp_src = src + f".tmp{os.getpid()}"   # concurrent ranks may generate the same unit: never expose a partial file
    with open(tmp_src, "w") as fh:
        fh.write(unit)
    os.replace(tmp_src, src)
    tmp = out + f".tmp{os.getpid()}"
    # Two steps, hipcc -S then assemble + link (the same code object as `hipcc --genco`, instruction for instruction: checked
    # in tests/test_user_metric.py), because the LISTING is looked at in between: isa_exec finds — and rewrites — register
    # copies / spills that this LLVM places ahead of a FLOW block's EXEC flip, which is what the Float64 FULL
    # pass of the heavy example metric was wrong from in round 4.
    # Occupancy levels: the unit's generic-RHS kernels are built for 2 (Float64) / 3 (Float32) waves per SIMD; a metric whose
    # integrate kernels SPILL there (more than MAX_SCRATCH bytes per lane) is rebuilt with more registers per lane — 1 / 2, then
    # 1 / 1 (512 registers), the last level taken as it comes: the heavy example metrics spill 200-400 registers per step at two
    # waves per SIMD, which costs more than the second wave returns.  rtgr_user_metric_compile does the same in-process.
    usable, problems = [], []          # [(listing path, scratch bytes per lane)] of the levels whose code is sound
    for n_level, level in enumerate(LEVELS):
        asm = f"{src[:-4]}.L{n_level}.tmp{os.getpid()}.s"
        cmd = [_build.HIPCC] + FLAGS + extra + level + ["-S", "-Rpass-analysis=kernel-resource-usage", "-I", CSRC, "-o", asm, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on the unit's source ({src}):\n{r.stderr[-4000:]}")
        worst = max(integrate_kernel_scratch(r.stderr).values(), default=0)
        with open(asm) as fh:
            lines = fh.read().split("\n")
        try:
            lines, repaired = _isa.repair(lines)
        except _isa.RepairError as e:      # this level's code carries the fault in a form the rewrite is not proven for
            problems.append(f"level {n_level}: {e}")
            os.unlink(asm)
            continue
        if verbose:
            print(f"   level {n_level}: scratch {worst} B per lane, {repaired} FLOW block(s) rewritten", flush=True)
        if repaired:
            with open(asm, "w") as fh:
                fh.write("\n".join(lines))
        usable.append((asm, worst))
        if worst <= MAX_SCRATCH:
            break
    if not usable:
        raise RuntimeError(f"user metric {tag}: every occupancy level compiles to code with vector instructions ahead of an EXEC "
                           f"flip that cannot be repaired (raytracegr.jl_amd/isa_exec.py):\n" + "\n".join(problems))
    best = usable[-1]                  # the level without spills if there is one (the loop stops there), else the last sound one
    for asm, _ in usable[:-1]:
        os.unlink(asm)
    _assemble(best[0], tmp)
    os.unlink(best[0])
    os.replace(tmp, out)  # atomic: concurrent ranks may compile the same metric
    return out


def _assemble(asm, out):
    """gfx950 listing -> code object: what `hipcc --genco` runs after code generation (clang as the assembler, lld -shared)"""
    obj = out + ".o"
    try:
        subprocess.run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm, "-o", obj],
                       check=True, capture_output=True, text=True)
        subprocess.run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", out, obj],
                       check=True, capture_output=True, text=True)
    except subprocess.CalledProcessError as e:
        raise RuntimeError(f"assembling {asm} failed:\n{e.stderr[-3000:]}")
    finally:
        if os.path.exists(obj):
            os.unlink(obj)


MAX_SCRATCH = 64   # bytes per lane; == RTGR_USER_MAX_SCRATCH of rtgr_unit_build.hpp
# (generic-RHS kernels | closed-form kernels of a built-in metric in a unit of objects: whichever the unit instantiates reads its own)
LEVELS = [[], ["-DRTGR_WAVES_PER_SIMD_GENERIC=1", "-DRTGR_WAVES_PER_SIMD_GENERIC_F32=2",
               "-DRTGR_WAVES_PER_SIMD_FAR=2", "-DRTGR_WAVES_PER_SIMD=1", "-DRTGR_WAVES_PER_SIMD_F32=2"],
          ["-DRTGR_WAVES_PER_SIMD_GENERIC=1", "-DRTGR_WAVES_PER_SIMD_GENERIC_F32=1",
           "-DRTGR_WAVES_PER_SIMD_FAR=1", "-DRTGR_WAVES_PER_SIMD=1", "-DRTGR_WAVES_PER_SIMD_F32=1"]]


def code_object_scratch(path):
    """{kernel: private_segment_fixed_size} of a code object's integrate kernels, from its metadata notes (llvm-readelf; tests, tools)"""
    import re
    txt = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", path],
                         capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            cur = m.group(1)
        m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
        if m and cur and cur.startswith("rtgr_user_integrate"):
            out[cur] = int(m.group(1))
    return out


def integrate_kernel_scratch(remarks):
    """{kernel: scratch bytes per lane} of the unit's integrate kernels, from hipcc's -Rpass-analysis=kernel-resource-usage remarks"""
    import re
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        m = re.search(r"remark:\s+ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and cur and cur.startswith("rtgr_user_integrate"):
            out[cur] = int(m.group(1))
    return out


def load(path, ctx=None):
    """Load the code object into the context (default: the process's default context) and return its module id — the
    value a scene carries in rtgr_scene.user_metric.  Loading the same file again is a no-op that returns the same id;
    several metrics may be resident at once."""
    import ctypes as C
    lib = _abi.load()
    key = (getattr(ctx, "value", ctx), path)
    mid = _ids.get(key)
    if mid is not None and lib.rtgr_user_metric_loaded(ctx, mid) == 1:
        return mid
    out = C.c_uint64(0)
    _abi.check(lib, lib.rtgr_user_metric_load(ctx, path.encode(), C.byref(out)))
    _ids[key] = out.value
    return out.value


class UserMetric:
    """A metric given as source text; duck-types api.Metric (kind / M / a / generic) so make_scene accepts it."""
    kind = _abi.USER
    generic = True

    def __init__(self, source, M=1.0, a=0.0, name="user_metric", verbose=False, stationary=False, jit=False):
        """jit=False: the unit is built with `hipcc --genco` here and now (needs hipcc, not a GPU; cached on disk by content
        hash) and loaded with rtgr_user_metric_load on first use.  jit=True: the SOURCE is handed to the library, which
        builds it in-process (hiprtc + libamd_comgr) when the metric is first used in a context (rtgr_user_metric_compile: one call,
        no hipcc on the box, ~5 s, not cached across processes)."""
        self.source, self.M, self.a, self.name = source, float(M), float(a), name
        self.stationary, self.jit = bool(stationary), bool(jit)
        self.code_object = None if jit else compile_user_metric(source, verbose=verbose, stationary=stationary)
        self._jit_ids = {}

    def module_id(self, ctx=None):
        """id of this metric's module in the context (loads the code object — or compiles the source — on first use)"""
        if not self.jit:
            return load(self.code_object, ctx)
        import ctypes as C
        lib = _abi.load()
        key = getattr(ctx, "value", ctx)
        mid = self._jit_ids.get(key)
        if mid is not None and lib.rtgr_user_metric_loaded(ctx, mid) == 1:
            return mid
        out = C.c_uint64(0)
        _abi.check(lib, lib.rtgr_user_metric_compile(ctx, self.source.encode(), 1 if self.stationary else 0, C.byref(out)))
        self._jit_ids[key] = out.value
        return out.value

    def __call__(self, x):
        from . import api
        return api._eval_metric(self, x, want=(True, False, False))[0]

    def __repr__(self):
        return f"UserMetric({self.name}, M={self.M}, a={self.a}, {'hiprtc' if self.jit else os.path.basename(self.code_object)})"


class UserObjects:
    """A family of NEW `Object{T}` subtypes (src/RayTraceGR.jl:374-389) given as device source: the reference's two methods

        template <class S> __device__ S    rtgr_user_distance(unsigned type, const S x[4], const S p[9]);
        template <class S> __device__ void rtgr_user_objcolor(unsigned type, const S x[4], const S p[9], S rgb[3]);

    (+ optionally rtgr_user_reach, include/rtgr.h "user objects").  Calling the family makes an object:

        shapes = UserObjects(SOURCE)
        torus = shapes(0, [R, r])            # type 0 of the source, fields p[0..8]
        trace_rays(kerr_schild, [sky, torus, plane], canvas)

    The unit that carries the objects' kernels is built for the metric it is traced with (make_scene: one unit per (metric
    variant, source), cached on disk by content hash like user metrics)."""

    def __init__(self, source, name="user_objects", jit=False, ntypes=None):
        if not (defines(source, "rtgr_user_distance") and defines(source, "rtgr_user_objcolor")):
            raise ValueError("the source must define rtgr_user_distance and rtgr_user_objcolor (the two methods of the reference's Object)")
        if defines(source, "rtgr_user_metric") or defines(source, "rtgr_user_ks"):
            raise ValueError("a UserObjects source defines objects only; give the metric as a UserMetric (make_scene joins the two sources)")
        self.source, self.name, self.jit = source, name, bool(jit)
        self.ntypes = None if ntypes is None else int(ntypes)     # how many object types the source defines: needed to join families
        self._jit_ids = {}

    def __call__(self, type, params=()):
        if self.ntypes is not None and not 0 <= int(type) < self.ntypes:
            raise ValueError(f"{self.name} defines the object types 0..{self.ntypes - 1}, not {type}")
        return UserObject(self, type, params)

    _joined = {}

    @classmethod
    def join(cls, families):
        """Objects of several families in one scene: (joined family, type base of each family).  Compiled code holds a scene's
        objects in one unit, so the sources become ONE source (rtgr_user_source_join: a namespace per family under dispatchers on
        the renumbered type tag); make_scene does this by itself when the objects of a scene come from more than one family."""
        import ctypes as C
        families = list(families)
        for f in families:
            if f.ntypes is None:
                raise ValueError(f"{f.name}: joining families needs the number of object types each source defines — UserObjects(source, ntypes=...)")
        key = tuple((f.source, f.ntypes, f.jit) for f in families)
        if key not in cls._joined:
            lib = _abi.load()
            srcs = (C.c_char_p * len(families))(*[f.source.encode() for f in families])
            nt = (C.c_uint32 * len(families))(*[f.ntypes for f in families])
            need = C.c_uint64(0)
            _abi.check(lib, lib.rtgr_user_source_join(srcs, nt, len(families), None, 0, C.byref(need)))
            buf = C.create_string_buffer(need.value)
            _abi.check(lib, lib.rtgr_user_source_join(srcs, nt, len(families), buf, need.value, C.byref(need)))
            cls._joined[key] = cls(buf.value.decode(), name="+".join(f.name for f in families), jit=any(f.jit for f in families),
                                   ntypes=sum(f.ntypes for f in families))
        bases, b = [], 0
        for f in families:
            bases.append(b)
            b += f.ntypes
        return cls._joined[key], bases

    def unit_id(self, metric, ctx=None):
        """id (in ctx) of the unit carrying this family's kernels for `metric` — a built-in Metric or a UserMetric, whose source
        is then compiled into the same unit.  Builds (hipcc, cached; or in-process when jit) and loads on first use."""
        import ctypes as C
        if isinstance(metric, UserMetric):
            source, stationary, built_for = metric.source + "\n" + self.source, metric.stationary, None
            jit = self.jit or metric.jit
        else:
            kind = int(metric.kind)
            generic = bool(metric.generic) and kind != _abi.MINKOWSKI
            source, stationary = self.source, False
            built_for = (kind, generic, kind != _abi.MINKOWSKI and (generic or metric.a != 0.0))
            jit = self.jit
        if not jit:
            return load(compile_user_metric(source, stationary=stationary, built_for=built_for), ctx)
        lib = _abi.load()
        key = (getattr(ctx, "value", ctx), source, built_for)
        mid = self._jit_ids.get(key)
        if mid is not None and lib.rtgr_user_metric_loaded(ctx, mid) == 1:
            return mid
        out = C.c_uint64(0)
        sc = _built_for_scene(built_for)
        _abi.check(lib, lib.rtgr_user_unit_compile(ctx, source.encode(), 1 if stationary else 0, C.byref(sc) if sc is not None else None, C.byref(out)))
        self._jit_ids[key] = out.value
        return out.value

    def __repr__(self):
        return f"UserObjects({self.name})"


class UserObject:
    """One object of a UserObjects family: (type tag, up to 9 fields) — what rtgr_object carries for RTGR_USER_OBJECT."""
    kind = _abi.USER_OBJECT

    def __init__(self, family, type, params=()):
        self.family, self.type = family, int(type)
        self.params = [float(v) for v in params]
        if len(self.params) > 9:
            raise ValueError("an object has at most 9 scalar fields (rtgr_object.p)")

    def _pack(self):
        return self.params + [0.0] * (9 - len(self.params))

    def __repr__(self):
        return f"UserObject({self.family.name}, type={self.type}, p={self.params})"


def unit_info(mid, ctx=None):
    """rtgr_user_unit_info as a dict: what a resident unit was built for"""
    import ctypes as C
    lib = _abi.load()
    info = _abi.rtgr_unit_info()
    _abi.check(lib, lib.rtgr_user_unit_info(ctx, mid, C.byref(info)))
    return info.as_dict()


def audit(path):
    """(number of FLOW blocks with vector instructions ahead of their EXEC flip, report) of a code object — or of a library that
    embeds code objects — through rtgr_code_object_audit (include/rtgr.h; no GPU needed)"""
    import ctypes as C
    lib = _abi.load()
    n, buf = C.c_int32(0), C.create_string_buffer(8192)
    _abi.check(lib, lib.rtgr_code_object_audit(path.encode(), C.byref(n), buf, len(buf)))
    return n.value, buf.value.decode()


def _built_for_scene(built_for):
    """(kind, generic, spin) -> the rtgr_scene rtgr_user_unit_compile / _build read the metric variant from (or None)"""
    if built_for is None:
        return None
    sc = _abi.rtgr_scene()
    sc.metric = int(built_for[0]) | (_abi.METRIC_GENERIC if built_for[1] else 0)
    sc.M, sc.a = 1.0, (0.5 if built_for[2] else 0.0)
    return sc


def build_in_process(source, path, stationary=False, built_for=None):
    """rtgr_user_unit_build: the unit built by the LIBRARY (hiprtc + libamd_comgr, listing checked and repaired in between) into
    `path` — what a C or Julia caller gets without hipcc; no GPU needed"""
    import ctypes as C
    lib = _abi.load()
    sc = _built_for_scene(built_for)
    _abi.check(lib, lib.rtgr_user_unit_build(source.encode(), 1 if stationary else 0, C.byref(sc) if sc is not None else None, path.encode()))
    return path


if __name__ == "__main__":
    # python -m raytracegr.jl_amd.user_metric metric.hip [--stationary] [-o unit.hsaco]: build a unit for rtgr_user_metric_load —
    # the route for C / Julia callers whose metric rtgr_user_metric_compile cannot build soundly in-process (include/rtgr.h)
    import shutil
    import sys
    args = sys.argv[1:]
    if not args or args[0].startswith("-"):
        raise SystemExit("usage: python -m raytracegr.jl_amd.user_metric metric.hip [--stationary] [-o unit.hsaco]")
    with open(args[0]) as fh:
        built = compile_user_metric(fh.read(), verbose=True, stationary="--stationary" in args)
    if "-o" in args:
        shutil.copyfile(built, args[args.index("-o") + 1])
        built = args[args.index("-o") + 1]
    print(built)
