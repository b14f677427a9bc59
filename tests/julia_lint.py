"""A small static checker for the Julia files of this repository (julia/*.jl) — test infrastructure.

There is no Julia in this image, so julia/RayTraceGRHIP.jl and julia/runtests_hip.jl have never been parsed by the language
itself.  Regular expressions over the text (tests/test_julia_stub.py) catch a wrong `ccall`; they do not catch
what a parser or a first run would: an unterminated string, brackets that close in the wrong order, an `end` too many inside a
comprehension, a name that is used and defined nowhere.  This module is a tokenizer for the subset of Julia those files use
and three checks on the token stream:

  * `check_structure(tokens)`  — brackets nest properly (a stack, with the kind of every bracket), every block opener
    (function / struct / if / for / while / let / begin / try / do / module / quote / macro) has its `end`, `elseif` / `else` /
    `catch` / `finally` stand inside the right kind of block, `end` inside `[...]` is an index;
  * `check_names(tokens, known)` — every identifier a function body reads is an argument, a local it assigns, a loop or closure
    variable, a type parameter, a name the file defines at top level, or one of `known` (what Base / StaticArrays / RayTraceGR /
    Test provide and these files use): a typo in a variable or function name is an undefined name;
  * `string_literals_terminate` is implicit: the tokenizer raises on an unterminated string, comment or character literal.

It is NOT a Julia parser: operators are not given precedence, macros are not expanded, types are not checked.  It is written
to err on the side of reporting (an unknown construct raises LintError) so that what it passes has at least been tokenized
end to end.
"""
import re

KEYWORDS = {"function", "struct", "mutable", "if", "elseif", "else", "for", "while", "let", "begin", "end", "try", "catch", "finally",
            "do", "module", "baremodule", "quote", "macro", "return", "break", "continue", "const", "global", "local", "using", "import",
            "export", "where", "in", "isa", "true", "false", "nothing", "abstract", "type", "primitive"}
OPENERS = {"function", "struct", "if", "for", "while", "let", "begin", "try", "do", "module", "baremodule", "quote", "macro"}
CLOSE = {")": "(", "]": "[", "}": "{"}
ID_START = re.compile(r"[^\W\d]|[∇∂αβγδεζηθλμνξπρστφχψωΓΔΘΛΞΠΣΦΨΩϕϱϖ]", re.UNICODE)
ID_REST = re.compile(r"[\w!′⁰¹²³⁴⁵⁶⁷⁸⁹₀₁₂₃₄₅₆₇₈₉]", re.UNICODE)
NUMBER = re.compile(r"0x[0-9a-fA-F_]+|\d[\d_]*(?:\.\d[\d_]*)?(?:[eEf][+-]?\d+)?|\.\d+(?:[eEf][+-]?\d+)?")
OPERATORS = ["...", "===", "!==", "->", "=>", "::", "<:", ">:", "==", "!=", "<=", ">=", "&&", "||", "+=", "-=", "*=", "/=", "^=", "|=", "&=",
             ".+", ".-", ".*", "./", ".^", ".==", ".<=", ".>=", ".<", ".>", ".=", "<<", ">>", "|>", "≤", "≥", "≠", "∈", "÷", "⊻", "≈", "∉", "∘"]


class LintError(AssertionError):
    pass


class Tok:
    __slots__ = ("kind", "val", "line")

    def __init__(self, kind, val, line):
        self.kind, self.val, self.line = kind, val, line

    def __repr__(self):
        return f"{self.kind}:{self.val!r}@{self.line}"


def tokenize(text, line0=1):
    """-> [Tok]; kinds: id, kw, num, str, char, sym (:name), macro (@name), op, punct.  String interpolations `$(…)` / `$name` are
    tokenized recursively and appended to the string token's `.val` as a list (kind 'str', val = [parts…])."""
    toks, i, line, n = [], 0, line0, len(text)

    def err(msg):
        raise LintError(f"line {line}: {msg}")

    def prev_is_value():
        for t in reversed(toks):
            return t.kind in ("id", "num", "str", "char") or t.val in (")", "]", "}", "end", "'") or (t.kind == "kw" and t.val in ("end", "true", "false", "nothing"))
        return False

    while i < n:
        c = text[i]
        if c == "\n":
            toks.append(Tok("nl", "\n", line))
            line += 1
            i += 1
        elif c in " \t\r":
            i += 1
        elif text.startswith("#=", i):
            depth, j = 1, i + 2
            while depth and j < n:
                if text.startswith("#=", j):
                    depth, j = depth + 1, j + 2
                elif text.startswith("=#", j):
                    depth, j = depth - 1, j + 2
                else:
                    line += text[j] == "\n"
                    j += 1
            if depth:
                err("unterminated #= comment")
            i = j
        elif c == "#":
            while i < n and text[i] != "\n":
                i += 1
        elif c == '"':
            triple = text.startswith('"""', i)
            q = '"""' if triple else '"'
            j, parts, start_line = i + len(q), [], line
            while True:
                if j >= n:
                    line = start_line
                    err("unterminated string literal")
                if text.startswith(q, j):
                    j += len(q)
                    break
                ch = text[j]
                if ch == "\\":
                    j += 2
                    continue
                if ch == "\n":
                    if not triple:
                        line = start_line
                        err("newline inside a \"...\" string literal")
                    line += 1
                if ch == "$":
                    if j + 1 < n and text[j + 1] == "(":
                        depth, k = 1, j + 2
                        while depth and k < n:
                            depth += (text[k] == "(") - (text[k] == ")")
                            k += 1
                        if depth:
                            err("unterminated $( interpolation")
                        parts.append(tokenize(text[j + 2:k - 1], line))
                        j = k
                        continue
                    m = ID_START.match(text, j + 1)
                    if m:
                        k = j + 1
                        while k < n and ID_REST.match(text, k):
                            k += 1
                        parts.append([Tok("id", text[j + 1:k], line)])
                        j = k
                        continue
                j += 1
            toks.append(Tok("str", parts, start_line))
            i = j
        elif c == "'" and prev_is_value():
            toks.append(Tok("op", "'", line))          # postfix transpose / adjoint
            i += 1
        elif c == "'":
            m = re.compile(r"'(?:\\.|\\x[0-9a-fA-F]{2}|\\u[0-9a-fA-F]{1,4}|[^'\\\n])'").match(text, i)
            if not m:
                err("bad character literal")
            toks.append(Tok("char", m.group(0), line))
            i = m.end()
        elif c == "@":
            j = i + 1
            while j < n and (ID_REST.match(text, j) or text[j] == "."):
                j += 1
            if j == i + 1:
                err("stray @")
            toks.append(Tok("macro", text[i:j], line))
            i = j
        elif c == ":" and i + 1 < n and ID_START.match(text, i + 1) and not prev_is_value() and not text.startswith("::", i):
            j = i + 1
            while j < n and ID_REST.match(text, j):
                j += 1
            toks.append(Tok("sym", text[i:j], line))
            i = j
        elif ID_START.match(text, i):
            j = i
            while j < n and ID_REST.match(text, j):
                j += 1
            w = text[i:j]
            toks.append(Tok("kw" if w in KEYWORDS else "id", w, line))
            i = j
        elif c.isdigit() or (c == "." and i + 1 < n and text[i + 1].isdigit() and not prev_is_value()):
            m = NUMBER.match(text, i)
            toks.append(Tok("num", m.group(0), line))
            i = m.end()
        else:
            for op in OPERATORS:
                if text.startswith(op, i):
                    toks.append(Tok("op", op, line))
                    i += len(op)
                    break
            else:
                if c in "()[]{}":
                    toks.append(Tok("punct", c, line))
                elif c in ",;":
                    toks.append(Tok("punct", c, line))
                elif c in "+-*/\\^%<>=!&|~?:.$√×·":
                    toks.append(Tok("op", c, line))
                else:
                    err(f"character {c!r} this checker does not know")
                i += 1
    return toks


def _code(tokens):
    return [t for t in tokens if t.kind != "nl"]


def check_structure(tokens):
    """bracket nesting + block openers / `end` (see module docstring); returns the number of blocks seen"""
    stack, blocks = [], 0          # entries: ("(", line) | ("[", line) | ("{", line) | (opener keyword, line)
    toks = tokens
    for k, t in enumerate(toks):
        if t.kind == "str":
            for part in t.val:
                check_structure(part)
            continue
        if t.kind == "punct" and t.val in "([{":
            stack.append((t.val, t.line))
        elif t.kind == "punct" and t.val in ")]}":
            if not stack or stack[-1][0] != CLOSE[t.val]:
                raise LintError(f"line {t.line}: `{t.val}` closes {stack[-1] if stack else 'nothing'}")
            stack.pop()
        elif t.kind == "kw":
            inside_bracket = next((s[0] for s in reversed(stack) if s[0] in "([{" or s[0] in OPENERS), None) in ("(", "[", "{")
            if t.val in ("for", "if") and inside_bracket:
                continue                                           # generator / comprehension / ternary-free filter: opens nothing
            if t.val == "end" and inside_bracket:
                if stack and next(s[0] for s in reversed(stack) if s[0] in "([{") == "[":
                    continue                                       # a[end]: lastindex
                raise LintError(f"line {t.line}: `end` inside ( ) or {{ }}")
            if t.val == "type" and not (k and toks[k - 1].kind == "kw" and toks[k - 1].val in ("abstract", "primitive")):
                continue
            if t.val == "type":
                stack.append(("abstract type", t.line))
                blocks += 1
                continue
            if t.val in OPENERS:                                   # (`mutable struct` is ONE block: `mutable` opens nothing)
                stack.append((t.val, t.line))
                blocks += 1
            elif t.val in ("elseif", "else"):
                if not stack or stack[-1][0] not in ("if", "try"):      # (try … catch … else … end exists since Julia 1.8)
                    raise LintError(f"line {t.line}: `{t.val}` outside an if block (innermost open: {stack[-1] if stack else None})")
            elif t.val in ("catch", "finally"):
                if not stack or stack[-1][0] != "try":
                    raise LintError(f"line {t.line}: `{t.val}` outside a try block")
            elif t.val == "end":
                if not stack or stack[-1][0] in "([{":
                    raise LintError(f"line {t.line}: `end` without an open block (innermost open: {stack[-1] if stack else None})")
                stack.pop()
    if stack:
        raise LintError(f"unclosed {stack[-1][0]!r} opened at line {stack[-1][1]}")
    return blocks


def toplevel_definitions(tokens):
    """names a file defines at depth 0 or directly inside its `module`: functions (both forms), structs, consts, plain assignments,
    abstract types, macros"""
    toks, names, depth_stack = _code(tokens), set(), []
    k = 0
    while k < len(toks):
        t = toks[k]
        top = all(s in ("module",) for s in depth_stack)
        if t.kind == "punct" and t.val in "([{":
            depth_stack.append(t.val)
        elif t.kind == "punct" and t.val in ")]}":
            depth_stack.pop()
        elif t.kind == "kw":
            inside_bracket = any(s in "([{" for s in depth_stack)
            if t.val in OPENERS and not (t.val in ("for", "if") and inside_bracket):
                if top and t.val in ("function", "macro", "struct", "module"):
                    j = k + 1
                    while toks[j].kind == "id" and toks[j + 1].kind == "op" and toks[j + 1].val == ".":   # function Base.close(...)
                        j += 2
                    if toks[j].kind == "id":
                        names.add(toks[j].val)
                depth_stack.append(t.val)
            elif t.val == "type" and k and toks[k - 1].kind == "kw" and toks[k - 1].val == "abstract":
                if top:
                    names.add(toks[k + 1].val)
                depth_stack.append("abstract type")
            elif t.val == "end" and not inside_bracket:
                depth_stack.pop()
            elif t.val == "const" and top and toks[k + 1].kind == "id":
                names.add(toks[k + 1].val)
        elif t.kind == "id" and top and k + 1 < len(toks):
            nxt = toks[k + 1]
            if nxt.kind == "op" and nxt.val == "=" and (k == 0 or toks[k - 1].line != t.line or toks[k - 1].val in (";", "const")):
                names.add(t.val)                                                        # NAME = …
            elif nxt.kind == "punct" and nxt.val in "({" and (k == 0 or toks[k - 1].line != t.line):
                j, d = k + 1, 0                                                          # f(args) [where {T}] = …   (short form)
                while j < len(toks):
                    d += (toks[j].val in ("(", "{", "[")) - (toks[j].val in (")", "}", "]")) if toks[j].kind == "punct" else 0
                    j += 1
                    if d == 0 and not (j < len(toks) and toks[j].kind == "punct" and toks[j].val in "({"):
                        break
                if j < len(toks) and toks[j].kind == "kw" and toks[j].val == "where":
                    j += 1
                    if toks[j].kind == "punct" and toks[j].val == "{":
                        d = 0
                        while True:
                            d += (toks[j].val == "{") - (toks[j].val == "}") if toks[j].kind == "punct" else 0
                            j += 1
                            if d == 0:
                                break
                    else:
                        j += 1
                if j < len(toks) and toks[j].kind == "op" and toks[j].val == "=":
                    names.add(t.val)
        k += 1
    return names


def function_bodies(tokens):
    """[(name, line, header tokens, body tokens)] of every `function … end` (long form), nested ones included as part of their parent"""
    toks, out = _code(tokens), []
    k = 0
    while k < len(toks):
        t = toks[k]
        if t.kind == "kw" and t.val == "function":
            # header: everything on the `function` line(s) — it ends at the first token, outside every bracket, that is the last of
            # its line (`function f(a,\n b) where {T}` spans lines only inside brackets; `function (m::K)(x) where {T}` and
            # `function S{T}(a) where {T}` have several bracket groups)
            j, d = k + 1, 0
            while j < len(toks):
                tt = toks[j]
                if tt.kind == "punct":
                    d += (tt.val in "([{") - (tt.val in ")]}")
                j += 1
                if d == 0 and (j >= len(toks) or toks[j].line > tt.line):
                    break
            header = toks[k + 1:j]
            depth, m = 1, j
            bracket = 0
            while m < len(toks) and depth:
                tt = toks[m]
                if tt.kind == "punct" and tt.val in "([{":
                    bracket += 1
                elif tt.kind == "punct" and tt.val in ")]}":
                    bracket -= 1
                elif tt.kind == "kw":
                    if tt.val in OPENERS and not (tt.val in ("for", "if") and bracket):
                        depth += 1
                    elif tt.val == "end" and not bracket:
                        depth -= 1
                m += 1
            name = next((h.val for h in header if h.kind == "id"), "?")
            out.append((name, t.line, header, toks[j:m - 1]))
        k += 1
    return out


def _declared_in(tokens):
    """identifiers a token sequence binds: `x = …`, `x, y = …`, `(a, b) = …`, `for x in`, `for (a, b) in`, `x -> …`, `(a, b) -> …`,
    `do x, y`, `local x`, `catch e`, comprehension variables, `x::T` declarations in argument lists, keyword names before `=`"""
    toks, bound = tokens, set()
    for k, t in enumerate(toks):
        if t.kind == "str":
            continue
        if t.kind == "op" and t.val in ("=", "+=", "-=", "*=", "/=", "|=", "&=", "^=", ".="):
            # inside the parentheses of a call (`f(a, b; key = v)`, `f(a, key = v)`) the `=` names a KEYWORD: it binds at most the
            # one identifier before it (a default value in a signature), never the positional arguments further left
            d, j, in_call = 0, k - 1, False
            while j >= 0 and toks[j].line == t.line:
                if toks[j].kind == "punct":
                    d += (toks[j].val in ")]}") - (toks[j].val in "([{")
                    if d < 0:
                        in_call = True
                        break
                j -= 1
            if in_call:
                if k and toks[k - 1].kind == "id":
                    bound.add(toks[k - 1].val)
                continue
            j = k - 1                                         # names on the left-hand side: ids separated by commas / parentheses, back to
            while j >= 0 and toks[j].line == t.line:          # the start of the statement (a left-hand side does not span lines here)
                p = toks[j]
                if p.kind == "id" and not (j and toks[j - 1].kind == "op" and toks[j - 1].val == "."):
                    bound.add(p.val)
                    j -= 1
                elif p.kind == "punct" and p.val in ",()":
                    j -= 1
                elif p.kind == "op" and p.val == "::":        # x::T = …
                    j -= 1
                elif p.kind == "punct" and p.val in "]}":     # a[i] = … / T{…}: skip the bracket, it binds nothing new
                    d = 0
                    while j >= 0:
                        d += (toks[j].val in ")]}") - (toks[j].val in "([{") if toks[j].kind == "punct" else 0
                        j -= 1
                        if d == 0:
                            break
                    # the indexed name itself is an existing variable, not a new binding
                    if j >= 0 and toks[j].kind == "id":
                        j -= 1
                    break
                else:
                    break
        elif t.kind == "kw" and t.val in ("for", "local", "global", "catch", "do"):
            j = k + 1
            while j < len(toks) and toks[j].line == t.line:
                p = toks[j]
                if p.kind == "id":
                    bound.add(p.val)
                elif p.kind == "kw" and p.val == "in" or (p.kind == "op" and p.val in ("=", "∈")):
                    if t.val == "for":
                        # further `, y in …` clauses of the same for
                        d, m = 0, j + 1
                        while m < len(toks) and toks[m].line == t.line:
                            q = toks[m]
                            d += (q.val in "([{") - (q.val in ")]}") if q.kind == "punct" else 0
                            if d < 0:
                                break
                            if d == 0 and q.kind == "punct" and q.val == "," and m + 1 < len(toks) and toks[m + 1].kind == "id":
                                bound.add(toks[m + 1].val)
                            m += 1
                    break
                elif not (p.kind == "punct" and p.val in ",()"):
                    break
                j += 1
        elif t.kind == "op" and t.val == "->":
            j = k - 1
            if j >= 0 and toks[j].kind == "id":
                bound.add(toks[j].val)
            elif j >= 0 and toks[j].kind == "punct" and toks[j].val == ")":
                while j >= 0 and not (toks[j].kind == "punct" and toks[j].val == "("):
                    if toks[j].kind == "id":
                        bound.add(toks[j].val)
                    j -= 1
    return bound


def _header_names(header):
    """argument names, keyword names and type parameters of a function header"""
    names, d = set(), 0
    for k, t in enumerate(header):
        if t.kind == "punct":
            d += (t.val in "([{") - (t.val in ")]}")
        if t.kind == "id":
            prev = header[k - 1] if k else None
            nxt = header[k + 1] if k + 1 < len(header) else None
            if prev is not None and prev.kind == "op" and prev.val in ("::", "<:", "."):
                continue                                       # a type, or a qualified name's tail
            if nxt is not None and nxt.kind == "op" and nxt.val == "." and d == 0:
                continue                                       # Module.f: the module
            names.add(t.val)
    return names


def check_names(tokens, known):
    """[(function, line, name)] of identifiers read in function bodies that nothing binds (see module docstring)"""
    top = toplevel_definitions(tokens)
    problems = []
    for name, line, header, body in function_bodies(tokens):
        bound = _header_names(header) | _declared_in(body) | top | known
        for part in [body]:
            for k, t in enumerate(part):
                ids = []
                if t.kind == "id":
                    ids = [(t, part, k)]
                elif t.kind == "str":
                    for sub in t.val:
                        bound |= _declared_in(sub)
                        ids += [(s, sub, m) for m, s in enumerate(sub) if s.kind == "id"]
                for tok, seq, m in ids:
                    prev = seq[m - 1] if m else None
                    nxt = seq[m + 1] if m + 1 < len(seq) else None
                    if prev is not None and prev.kind == "op" and prev.val == ".":
                        continue                               # field or qualified name: x.field, Module.name
                    if nxt is not None and nxt.kind == "op" and nxt.val == "=" and _in_call_parens(seq, m):
                        continue                               # keyword argument name in a call: f(x; name = value)
                    if tok.val not in bound:
                        problems.append((name, tok.line, tok.val))
    return problems


def _in_call_parens(seq, m):
    d = 0
    for j in range(m - 1, -1, -1):
        t = seq[j]
        if t.kind == "punct":
            if t.val in ")]}":
                d += 1
            elif t.val in "([{":
                if d == 0:
                    return t.val == "(" and j > 0 and (seq[j - 1].kind in ("id", "macro") or seq[j - 1].val in ("}", ")"))
                d -= 1
    return False


def check_toplevel_names(tokens, known):
    """the same for the code OUTSIDE long-form function bodies: top-level statements, `@testset … begin … end` blocks, struct
    bodies, and the short-form definitions `f(args) = expr` (whose argument names are taken as bound for the whole pass: the price of
    not parsing statements — a typo is still a name bound nowhere)"""
    toks = _code(tokens)
    inside = [False] * len(toks)
    index = {id(t): k for k, t in enumerate(toks)}
    for _, _, header, body in function_bodies(tokens):
        for t in header + body:
            inside[index[id(t)]] = True
    outer = [t for k, t in enumerate(toks) if not inside[k]]
    bound = toplevel_definitions(tokens) | known | _declared_in(outer)
    # short forms and struct fields: every identifier directly followed by `::`, and the arguments of `name(args) =` definitions
    for k, t in enumerate(outer):
        if t.kind == "id" and k + 1 < len(outer) and outer[k + 1].kind == "op" and outer[k + 1].val == "::":
            bound.add(t.val)
    lines = {}
    for t in outer:
        lines.setdefault(t.line, []).append(t)
    for ln, ts in lines.items():
        if len(ts) > 3 and ts[0].kind in ("id",) and ts[1].kind == "punct" and ts[1].val in "({":
            d = 0
            for k, t in enumerate(ts):
                if t.kind == "punct":
                    d += (t.val in "([{") - (t.val in ")]}")
                if d == 0 and k and t.kind == "op" and t.val == "=":
                    bound |= _header_names(ts[:k])
                    break
        if ts and ts[0].kind == "kw" and ts[0].val in ("struct", "mutable", "abstract"):
            bound |= {t.val for t in ts if t.kind == "id"}                   # the declared type and its parameters
        for k, t in enumerate(ts):
            if t.kind == "kw" and t.val == "where":
                bound |= {q.val for q in ts[k:] if q.kind == "id"}            # type parameters of short-form methods
    problems = []
    for k, t in enumerate(outer):
        cands = [(t, outer, k)] if t.kind == "id" else []
        if t.kind == "str":
            for sub in t.val:
                cands += [(s, sub, m) for m, s in enumerate(sub) if s.kind == "id"]
        for tok, seq, m in cands:
            prev = seq[m - 1] if m else None
            nxt = seq[m + 1] if m + 1 < len(seq) else None
            if prev is not None and prev.kind == "op" and prev.val == ".":
                continue
            if prev is not None and prev.kind == "kw" and prev.val in ("using", "import", "export", "module"):
                continue
            if nxt is not None and nxt.kind == "op" and nxt.val == "=" and _in_call_parens(seq, m):
                continue
            if tok.val not in bound:
                problems.append(("<top level>", tok.line, tok.val))
    return problems


# ---- call arity: a call of a function the file defines must fit one of its methods' positional-argument counts ---------------------
def _split_args(seq):
    """tokens between a call's / header's parentheses -> (positional groups, keyword groups): split at depth-0 commas; everything after
    a depth-0 `;` is keyword; in a CALL a group `name = value` is a keyword argument too (caller tells which by `is_call`)"""
    groups, cur, d, after_semi, marks = [], [], 0, False, []
    for t in seq:
        if t.kind == "punct" and t.val in "([{":
            d += 1
        elif t.kind == "punct" and t.val in ")]}":
            d -= 1
        if d == 0 and t.kind == "punct" and t.val in ",;":
            if cur:
                groups.append(cur)
                marks.append(after_semi)
            cur = []
            if t.val == ";":
                after_semi = True
            continue
        cur.append(t)
    if cur:
        groups.append(cur)
        marks.append(after_semi)
    return [g for g, kw in zip(groups, marks) if not kw], [g for g, kw in zip(groups, marks) if kw]


def _has_top_level(group, kind, val):
    d = 0
    for t in group:
        if t.kind == "punct" and t.val in "([{":
            d += 1
        elif t.kind == "punct" and t.val in ")]}":
            d -= 1
        elif d == 0 and t.kind == kind and t.val == val:
            return True
    return False


def _paren_group(toks, k):
    """toks[k] is `(`: (tokens inside, index of the matching `)`)"""
    d, j = 0, k
    while j < len(toks):
        if toks[j].kind == "punct" and toks[j].val in "([{":
            d += 1
        elif toks[j].kind == "punct" and toks[j].val in ")]}":
            d -= 1
            if d == 0:
                return toks[k + 1:j], j
        j += 1
    raise LintError(f"line {toks[k].line}: unclosed (")


def _skip_curly(toks, j):
    if j < len(toks) and toks[j].kind == "punct" and toks[j].val == "{":
        d = 0
        while j < len(toks):
            d += (toks[j].val == "{") - (toks[j].val == "}") if toks[j].kind == "punct" else 0
            j += 1
            if d == 0:
                break
    return j


def method_arities(tokens):
    """{name: [(min positional, max positional or None)]} for the functions a file defines under an UNQUALIFIED name (a method added
    to another module's function — `Base.close(c) = …`, `RayTraceGR.distance(o, x) = …` — says nothing about that function's other
    methods) and for its structs (the default constructor takes one argument per field)"""
    toks, out, headers = _code(tokens), {}, []
    for name, line, header, body in function_bodies(tokens):
        headers.append(header)
    lines = {}
    for t in toks:
        lines.setdefault(t.line, []).append(t)
    inside_fn = set()
    for _, _, header, body in function_bodies(tokens):
        inside_fn |= {id(t) for t in body}
    for ln, ts in lines.items():                                  # short forms at top level: `name(args) [where …] = …`
        if ts[0].kind == "id" and id(ts[0]) not in inside_fn and len(ts) > 3:
            j = _skip_curly(ts, 1)
            if j < len(ts) and ts[j].kind == "punct" and ts[j].val == "(":
                try:
                    inner, close = _paren_group(ts, j)
                except LintError:
                    continue                                   # (the argument list continues on the next line: long enough to be a long form)
                m = close + 1
                if m < len(ts) and ts[m].kind == "op" and ts[m].val == "::":       # f(x)::T = …
                    m += 2
                    m = _skip_curly(ts, m)
                if m < len(ts) and ts[m].kind == "kw" and ts[m].val == "where":
                    m += 1
                    m = _skip_curly(ts, m) if ts[m].kind == "punct" else m + 1
                if m < len(ts) and ts[m].kind == "op" and ts[m].val == "=":
                    headers.append(ts[:close + 1])
    for header in headers:
        if not header or header[0].kind != "id":
            continue                                              # `function (m::K)(x)`: a callable object, no name to call
        if len(header) > 1 and header[1].kind == "op" and header[1].val == ".":
            continue                                              # Module.f: another module's function
        j = _skip_curly(header, 1)
        if j >= len(header) or not (header[j].kind == "punct" and header[j].val == "("):
            continue
        inner, _ = _paren_group(header, j)
        pos, _ = _split_args(inner)
        var = any(g[-1].kind == "op" and g[-1].val == "..." for g in pos)
        need = sum(1 for g in pos if not _has_top_level(g, "op", "=") and not (g[-1].kind == "op" and g[-1].val == "..."))
        out.setdefault(header[0].val, []).append((need, None if var else len(pos)))
    k = 0
    while k < len(toks):                                          # structs: one positional argument per field
        t = toks[k]
        if t.kind == "kw" and t.val == "struct":
            name = toks[k + 1].val
            depth, j, fields, bracket = 1, k + 2, 0, 0
            while j < len(toks) and depth:
                tt = toks[j]
                if tt.kind == "punct" and tt.val in "([{":
                    bracket += 1
                elif tt.kind == "punct" and tt.val in ")]}":
                    bracket -= 1
                elif tt.kind == "kw" and tt.val in OPENERS and not bracket:
                    depth += 1
                elif tt.kind == "kw" and tt.val == "end" and not bracket:
                    depth -= 1
                elif depth == 1 and not bracket and tt.line != t.line and (tt.kind == "id" or (tt.kind == "kw" and tt.val == "type")):   # (`type` is a legal field name)
                    # a field: an identifier that starts a statement (first of its line, or right behind a `;`)
                    prev, nxt = toks[j - 1], toks[j + 1]
                    starts = prev.line != tt.line or (prev.kind == "punct" and prev.val == ";")
                    if starts and ((nxt.kind == "op" and nxt.val == "::") or nxt.line != tt.line or (nxt.kind == "punct" and nxt.val == ";")):
                        fields += 1
                j += 1
            out.setdefault(name, []).append((fields, fields))
            k = j
            continue
        k += 1
    return out


def check_arity(tokens, arities, qualifier=None):
    """[(line, name, positional arguments given, what the definitions take)] for calls `name(…)` (or `qualifier.name(…)`) whose number
    of positional arguments fits no method of `arities`.  Calls with a splat are skipped; a `do` block adds its function as first
    argument; `name = value` inside the call's parentheses and everything behind `;` are keyword arguments."""
    toks, problems = _code(tokens), []
    defs = set()
    for _, _, header, _ in function_bodies(tokens):
        defs |= {id(t) for t in header}
    k = 0
    while k < len(toks):
        t = toks[k]
        if t.kind == "id" and t.val in arities and id(t) not in defs:
            prev = toks[k - 1] if k else None
            qualified = prev is not None and prev.kind == "op" and prev.val == "."
            if qualified and not (qualifier and k >= 2 and toks[k - 2].kind == "id" and toks[k - 2].val == qualifier):
                k += 1
                continue                                          # x.name(…): a field or another module's function
            if not qualified and qualifier:
                k += 1
                continue                                          # (checking a client file: only Module.name(…) is the module's)
            j = _skip_curly(toks, k + 1)
            if j < len(toks) and toks[j].kind == "punct" and toks[j].val == "(" and toks[j].line == t.line:
                inner, close = _paren_group(toks, j)
                after = toks[close + 1] if close + 1 < len(toks) else None
                # a definition in short form is not a call: `name(args) [where …] =` at the start of a line
                first_on_line = k == 0 or toks[k - 1].line != t.line
                if first_on_line and after is not None and ((after.kind == "op" and after.val in ("=", "::")) or (after.kind == "kw" and after.val == "where")):
                    k = close + 1
                    continue
                pos, _ = _split_args(inner)
                if any(_has_top_level(g, "op", "...") for g in pos):
                    k += 1
                    continue
                n = sum(1 for g in pos if not (_has_top_level(g, "op", "=") and g[0].kind == "id" and len(g) > 1 and g[1].kind == "op" and g[1].val == "="))
                if after is not None and after.kind == "kw" and after.val == "do":
                    n += 1
                if not any(lo <= n and (hi is None or n <= hi) for lo, hi in arities[t.val]):
                    problems.append((t.line, t.val, n, arities[t.val]))
        k += 1
    return problems


# ---- field access: `x.name` must name a field of SOME struct in sight (a misspelled field is a first-run error too) -----------------
def struct_fields(tokens):
    """{field names} of every struct the file defines"""
    toks, out = _code(tokens), set()
    k = 0
    while k < len(toks):
        t = toks[k]
        if t.kind == "kw" and t.val == "struct":
            depth, j, bracket = 1, k + 2, 0
            while j < len(toks) and depth:
                tt = toks[j]
                if tt.kind == "punct" and tt.val in "([{":
                    bracket += 1
                elif tt.kind == "punct" and tt.val in ")]}":
                    bracket -= 1
                elif tt.kind == "kw" and tt.val in OPENERS and not bracket:
                    depth += 1
                elif tt.kind == "kw" and tt.val == "end" and not bracket:
                    depth -= 1
                elif depth == 1 and not bracket and tt.line != t.line and (tt.kind == "id" or (tt.kind == "kw" and tt.val == "type")):
                    prev, nxt = toks[j - 1], toks[j + 1]
                    starts = prev.line != tt.line or (prev.kind == "punct" and prev.val == ";")
                    if starts and ((nxt.kind == "op" and nxt.val == "::") or nxt.line != tt.line or (nxt.kind == "punct" and nxt.val == ";")):
                        out.add(tt.val)
                j += 1
            k = j
            continue
        k += 1
    return out


def check_fields(tokens, fields, modules):
    """[(line, name)] of `x.name` accesses (x: an identifier that is no module, or the result of an index / call) whose `name` is no
    field of any known struct.  `name(` directly behind the dot is a qualified call (Module.f(…)) and is skipped with its module."""
    problems = []

    def scan(toks):
        for k, t in enumerate(toks):
            if t.kind == "str":
                for sub in t.val:
                    scan(_code(sub))
                continue
            if not (t.kind == "op" and t.val == "." and 0 < k < len(toks) - 1):
                continue
            prev, nxt = toks[k - 1], toks[k + 1]
            if not (nxt.kind == "id" or (nxt.kind == "kw" and nxt.val == "type")) or nxt.line != t.line:
                continue                                          # broadcasting dot, `.+`, `f.(x)`
            if prev.kind == "id":
                if prev.val in modules:
                    continue
                # a chain Module.Sub.name: walk to the head
                j, head = k - 1, prev
                while j >= 2 and toks[j - 1].kind == "op" and toks[j - 1].val == "." and toks[j - 2].kind == "id":
                    j -= 2
                    head = toks[j]
                if head.val in modules:
                    continue
            elif not (prev.kind == "punct" and prev.val in ")]"):
                continue
            if nxt.val not in fields:
                problems.append((nxt.line, nxt.val))

    scan(_code(tokens))
    return problems
