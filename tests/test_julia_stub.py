"""Static cross-check of julia/RayTraceGRHIP.jl against include/rtgr.h.

The image has no Julia, so the stub cannot be executed here (tests/c/abi_layout.c is the compiled stand-in for the bytes
it passes).  What CAN be checked without Julia is everything a typo would break first: every `ccall` of the stub names a
function the header declares and the library exports, passes as many argument types and as many arguments as the C
prototype has parameters, with Julia types of the right KIND (pointer / 32-bit int / 64-bit unsigned) and the right return
type; the struct declarations carry the header's fields in the header's order; and the block structure of the file is
balanced (every function / struct / if / begin / module has its `end`)."""
import os
import re

import pytest

from __graft_entry__ import load_package

rt = load_package()
abi = rt._abi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "RayTraceGRHIP.jl")
HDR = os.path.join(ROOT, "include", "rtgr.h")


def split_top(text):
    """split at top-level commas (no nesting inside (), {}, [])"""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def matching_paren(text, start):
    depth = 0
    for k in range(start, len(text)):
        if text[k] == "(":
            depth += 1
        elif text[k] == ")":
            depth -= 1
            if depth == 0:
                return k
    raise AssertionError("unbalanced parenthesis")


def strip_julia(text):
    """code only: docstrings / strings blanked (kept as \"\" so that argument counts survive), comments removed"""
    text = re.sub(r'"""(?:.|\n)*?"""', '""', text)
    text = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', text)
    return re.sub(r"#[^\n]*", "", text)


def c_prototypes():
    """name -> (return type, [parameter types]) of every function include/rtgr.h declares"""
    text = open(HDR).read()
    text = re.sub(r"/\*(?:.|\n)*?\*/", " ", text)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_][A-Za-z_0-9]*(?:\s*\*)?)\s+(rtgr_[a-z0-9_]+)\s*\(", text):
        end = matching_paren(text, m.end() - 1)
        params = text[m.end():end].strip()
        plist = [] if params in ("", "void") else split_top(params)
        protos[m.group(2)] = (re.sub(r"\s+", " ", m.group(1)).strip(), plist)
    return protos


def c_kind(t):
    t = t.strip()
    if "*" in t or "[" in t:
        return "ptr"
    base = re.sub(r"\b(const|unsigned)\b", "", t).split()[0]
    return {"int": "i32", "uint64_t": "u64", "uint32_t": "u32", "double": "f64", "float": "f32", "long": "i64"}[base]


def jl_kind(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring", "Ctx"):
        return "ptr"
    return {"Cint": "i32", "UInt64": "u64", "UInt32": "u32", "Float64": "f64", "Float32": "f32", "Clong": "i64"}[t]


def ccalls():
    code = strip_julia(open(JL).read())
    calls = []
    for m in re.finditer(r"ccall\(", code):
        end = matching_paren(code, m.end() - 1)
        parts = split_top(code[m.end():end])
        sym = re.match(r"\(:([a-z0-9_]+),\s*librtgr\)", parts[0])
        assert sym, parts[0]
        types = parts[2].strip()
        assert types.startswith("(") and types.endswith(")"), types
        tl = split_top(types[1:-1])
        calls.append((sym.group(1), parts[1].strip(), tl, parts[3:]))
    return calls


def test_every_ccall_matches_its_c_prototype():
    protos = c_prototypes()
    assert len(protos) > 40 and "rtgr_trace_pixels_f64" in protos   # (the header parser sees the header)
    calls = ccalls()
    names = {c[0] for c in calls}
    # the entry points the stub is there for
    assert {"rtgr_create", "rtgr_destroy", "rtgr_last_error", "rtgr_solver_defaults", "rtgr_trace_pixels_f64",
            "rtgr_trace_pixels_f32", "rtgr_trace_one_f64", "rtgr_user_metric_load", "rtgr_user_metric_compile"} <= names
    for sym, ret, types, args in calls:
        assert sym in protos, f"{sym}: not declared in include/rtgr.h"
        cret, cparams = protos[sym]
        assert len(types) == len(cparams), f"{sym}: {len(types)} Julia argument types, {len(cparams)} C parameters"
        assert len(args) == len(cparams), f"{sym}: {len(args)} arguments passed, {len(cparams)} C parameters"
        for k, (jt, ct) in enumerate(zip(types, cparams)):
            assert jl_kind(jt) == c_kind(ct), f"{sym} argument {k + 1}: Julia {jt} vs C `{ct}`"
        want = "Cstring" if "char" in cret else {"int": "Cint", "void": "Cvoid"}[cret]
        assert ret == want, f"{sym}: return type {ret}, C `{cret}`"
        if cparams and "rtgr_context" in cparams[0]:
            assert types[0] == "Ctx" and args[0].strip() in ("handle(ctx)", "c.handle"), (sym, types[0], args[0])


def test_every_bound_symbol_is_exported():
    """(needs the built library, not a GPU)"""
    import ctypes
    lib = ctypes.CDLL(abi.LIB_PATH)
    for sym, *_ in ccalls():
        assert hasattr(lib, sym), sym


def test_struct_fields_follow_the_header():
    """Field names and order of the stub's structs == the header's (sizes and offsets: tests/c/abi_layout.c)."""
    code = strip_julia(open(JL).read())
    hdr = re.sub(r"/\*(?:.|\n)*?\*/", " ", open(HDR).read())
    for jname, cname in (("RtgrObject", "rtgr_object"), ("RtgrScene", "rtgr_scene"), ("RtgrSolver", "rtgr_solver"),
                         ("RtgrCounters", "rtgr_counters")):
        body = re.search(r"struct " + jname + r"\n((?:.|\n)*?)\nend", code).group(1)
        jfields = re.findall(r"([A-Za-z_0-9]+)::", body)
        cbody = re.search(r"typedef struct[^{]*\{((?:[^{}]|\{[^{}]*\})*)\}\s*" + cname + r"\s*;", hdr).group(1)
        cfields = [re.sub(r"\[.*", "", d.strip().split()[-1]).lstrip("*") for d in cbody.split(";") if d.strip()]
        assert jfields == cfields, (jname, jfields, cfields)


def test_blocks_are_balanced():
    code = strip_julia(open(JL).read())
    openers = 0
    for line in code.split("\n"):
        s = line.strip()
        toks = re.findall(r"[A-Za-z_@][A-Za-z_0-9!]*", s)
        for k, t in enumerate(toks):
            if t in ("function", "struct", "if", "begin", "module", "for", "while", "let", "try", "do", "quote"):
                if t == "struct" and k > 0 and toks[k - 1] == "mutable":
                    openers += 1
                elif t == "if" and k > 0 and toks[k - 1] == "else":     # (no `else if` in Julia, but be strict)
                    raise AssertionError(line)
                else:
                    openers += 1
            elif t == "end":
                openers -= 1
            assert openers >= 0, line
    assert openers == 0
    for a, b in ("()", "[]", "{}"):
        assert code.count(a) == code.count(b), (a, code.count(a), code.count(b))
