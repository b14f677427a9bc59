"""Check — and repair — of one code-generation fault of ROCm 7.2's LLVM (AMD clang 22.0.0git, roc-7.2.0) in gfx950 assembly
listings.  Found in round 4 in the FULL pass of a heavy run-time metric; the library's own kernels are checked
for it at build time (tests/test_build_checks.py), every run-time unit when it is compiled (user_metric.compile_user_metric), and
every code object when it is loaded (rtgr_api.hip: audit_code_object, the same rule on the disassembly).

The fault.  A divergent `if / else` is laid out as  header → then → FLOW → else → join.  The FLOW block starts with the
instruction that switches EXEC from the `then` lanes to the `else` lanes; while the register allocator works, that is
`s_or_saveexec_b64` (EXEC = then ∪ else) at the top of the block and `s_xor_b64 exec, exec, saved` at its end, so that a
live-range split copy or spill placed "at the top of the block, after its prologue" runs for ALL lanes of the if.  When SALU
instructions (constants hoisted into the block) come to stand before the `s_or_saveexec_b64`, the allocator no longer recognises
it as the block's prologue and puts its copies BEFORE it — under the `then` lanes' EXEC.  A later pass fuses the pair into
`s_andn2_saveexec_b64`, and the result reads

        s_cbranch_execz .LBB5_166               ; no `then` lane: skip to FLOW
        …then side…
    .LBB5_166:
        v_accvgpr_write_b32 a106, v174          ; <- split copy (h², on its way to an AGPR): runs for the THEN lanes only,
        s_mov_b32 s18, 0xb42fdfa7               ;    in the usual case for NO lane at all
        v_accvgpr_write_b32 a107, v175
        s_andn2_saveexec_b64 s[2:3], s[2:3]     ; EXEC = the `else` lanes, only now
        …
        v_accvgpr_read_b32 v123, a107           ; every lane reads the AGPR back: stale for the lanes that took `else`

(observed: the stale pair held another ray's λ-offset, and x^t of the Float64 FULL pass came out shifted by it times ½·k₀ᵗ).

The rule (find): a label that an `s_cbranch_execz` targets, then vector instructions, then `s_andn2_saveexec_b64` /
`s_or_saveexec_b64` (or, for an `if` without `else`, the join's `s_or_b64 exec, exec, saved`: the same mechanism, not seen so far),
with no other label, branch or EXEC write in between.  Instructions of the `then` side cannot stand there
(they are before the label), so whatever vector instruction does is the join's own code under the wrong mask.

The repair: restore the pre-fusion form around the misplaced instructions,

    .LBB5_166:
        s_or_saveexec_b64 s[2:3], s[2:3]        ; saved = then lanes, EXEC = then ∪ else
        v_accvgpr_write_b32 a106, v174          ; the copies run for every lane of the if, as the allocator meant
        s_mov_b32 s18, 0xb42fdfa7
        v_accvgpr_write_b32 a107, v175
        s_xor_b64 exec, exec, s[2:3]            ; EXEC = else lanes

which is only done when the instructions in between are of the kinds the allocator inserts (AGPR / VGPR copies, scratch spills
and reloads) plus `s_mov` constants, and none of them touches the two mask registers; anything else raises RepairError.
"""
import re

FLIP = re.compile(r"^(s_andn2_saveexec_b64|s_or_saveexec_b64)\s+(\S+?),\s*(\S+)$")
END_CF = re.compile(r"^(s_or_b64)\s+(exec),\s*exec,\s*(\S+)$")     # the join of an if WITHOUT else: the same fault can arise ahead of it
EXEC_WRITE = re.compile(r"^(s_\w+saveexec_b64\b|s_\w+\s+exec(_lo|_hi)?\s*,)")
VECTOR = re.compile(r"^(v_|scratch_|global_|flat_|buffer_|ds_)")
LANE_SGPR_IO = ("v_readlane", "v_readfirstlane", "v_writelane")   # SGPR spill traffic: does not depend on EXEC
BRANCH = re.compile(r"^(s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc|s_call)")
ALLOCATOR_VECTOR = re.compile(r"^(v_accvgpr_write_b32|v_accvgpr_read_b32|v_accvgpr_mov_b32|v_mov_b32(_e32)?|v_mov_b64(_e32)?|"
                              r"scratch_store_\w+|scratch_load_\w+)\s")
# (s_movk_i32: a 16-bit constant into an SGPR — the same move as s_mov_b32 with a literal, reads neither EXEC nor SCC; the compiler
#  picks it for small constants such as scratch offsets: round 6's headers made one appear ahead of a flip.  s_waitcnt: no register
#  operand, indifferent to EXEC)
PLAIN_SALU = re.compile(r"^(s_mov_b32|s_mov_b64|s_movk_i32|s_nop|s_waitcnt)\s")


class RepairError(RuntimeError):
    pass


class Hit:
    def __init__(self, function, label, label_at, early, flip_at, flip):
        self.function, self.label, self.label_at, self.early, self.flip_at, self.flip = function, label, label_at, early, flip_at, flip

    def __str__(self):
        return "\n".join(f"{self.function}: line {n + 1}: `{s}` stands BEFORE the EXEC flip `{self.flip}` (line {self.flip_at + 1}) of {self.label}"
                         for n, s in self.early)


def _code(line):
    return line.split(";")[0].strip()


def _functions(lines):
    """[(name, first, last)] line-index spans of the functions of a listing (`name:` … `.Lfunc_end…`)"""
    out, name, start = [], None, 0
    for i, l in enumerate(lines):
        c = _code(l)
        if name is None and c.endswith(":") and not c.startswith(".") and " " not in c:
            name, start = c[:-1], i
        elif name is not None and l.startswith(".Lfunc_end"):
            out.append((name, start, i))
            name = None
    return out


def find(lines):
    """[Hit] of a listing given as a list of lines"""
    hits = []
    for fname, a, b in _functions(lines):
        insts = [(i, _code(lines[i])) for i in range(a + 1, b)]
        insts = [(i, c) for i, c in insts if c and not (c.startswith(".") and not c.endswith(":"))]
        targets = {c.split()[-1] for _, c in insts if c.startswith("s_cbranch_execz")}
        for k, (i, c) in enumerate(insts):
            if not (c.endswith(":") and c[:-1] in targets):
                continue
            early = []
            for j, c2 in insts[k + 1:]:
                if c2.endswith(":") or BRANCH.match(c2):
                    break
                if FLIP.match(c2) or END_CF.match(c2):
                    if early:
                        hits.append(Hit(fname, c[:-1], i, early, j, c2))
                    break
                if EXEC_WRITE.match(c2):
                    break
                if VECTOR.match(c2) and not c2.startswith(LANE_SGPR_IO):
                    early.append((j, c2))
    return hits


def suspicious(lines):
    """[(line index, instruction, flip)] — a WIDER, heuristic net for listings one can look at by hand (build checks of the library's
    own kernels): an allocator-kind instruction (AGPR copy, scratch spill / reload) DIRECTLY ahead of a flip or of a join's
    `s_or_b64 exec`, with nothing but `s_mov` constants in between, whether or not a label precedes it.  Where the compiler has dropped
    the `s_cbranch_execz` that skips the `then` side, find() cannot tell a misplaced copy from the `then` side's own last
    instruction; neither can this — it only says where to look.  Never repaired automatically, and NOT a criterion for run-time
    units: tried as one in round 4, it rejected every occupancy level of the Boyer–Lindquist example metric (35-53 places: the
    `then` sides of the short branches inside acos / atan2 legitimately END in the AGPR copy that defines the joined value), a unit
    that traces true geodesics to 5e-12.  It reports nothing in the library's own listings, which is what the build checks use it for."""
    code = [_code(l) for l in lines]
    out = []
    for i, c in enumerate(code):
        if not (FLIP.match(c) or END_CF.match(c)):
            continue
        j = i - 1
        while j > 0 and (not code[j] or PLAIN_SALU.match(code[j]) or (code[j].startswith(".") and not code[j].endswith(":"))):
            j -= 1
        if ALLOCATOR_VECTOR.match(code[j]) and not code[j].startswith("v_mov"):
            out.append((j, code[j], c))
    return out


def _sgprs(operand_text):
    """the set of SGPR numbers an instruction's text mentions (s5, s[2:3]; vcc / exec are reported as -1 / -2)"""
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", operand_text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", operand_text):
        out.add(int(m.group(1)))
    if re.search(r"\bvcc(_lo|_hi)?\b", operand_text):
        out.add(-1)
    if re.search(r"\bexec(_lo|_hi)?\b", operand_text):
        out.add(-2)
    return out


def repair(lines):
    """(new lines, number of FLOW blocks rewritten).  Raises RepairError when a hit is not of the shape the rewrite is proven for."""
    hits = find(lines)
    if not hits:
        return lines, 0
    out = list(lines)
    for h in sorted(hits, key=lambda h: -h.label_at):       # bottom-up: line indices above stay valid
        m = FLIP.match(h.flip) or END_CF.match(h.flip)
        op, saved, src = m.group(1), m.group(2), m.group(3)
        masks = (_sgprs(saved) | _sgprs(src)) - {-2}
        between = [(i, _code(out[i])) for i in range(h.label_at + 1, h.flip_at)]
        between = [(i, c) for i, c in between if c and not c.startswith(".")]
        for i, c in between:
            if VECTOR.match(c) and not c.startswith(LANE_SGPR_IO):
                if not ALLOCATOR_VECTOR.match(c):
                    raise RepairError(f"{h.function}: `{c}` ahead of `{h.flip}` is not a copy, spill or reload")
                if _sgprs(c):
                    raise RepairError(f"{h.function}: `{c}` ahead of `{h.flip}` uses a scalar register")
            elif not (PLAIN_SALU.match(c) or c.startswith(LANE_SGPR_IO)):
                raise RepairError(f"{h.function}: `{c}` ahead of `{h.flip}`: only s_mov constants may stand there")
            if c.startswith(LANE_SGPR_IO) or PLAIN_SALU.match(c):
                if _sgprs(c) & (masks | {-2}):
                    raise RepairError(f"{h.function}: `{c}` ahead of `{h.flip}` touches the mask registers")
        indent = re.match(r"\s*", out[h.flip_at]).group(0)
        note = " ; isa_exec.repair: EXEC = then + else lanes for the allocator's copies below"
        if op == "s_andn2_saveexec_b64":
            out[h.flip_at] = f"{indent}s_xor_b64 exec, exec, {saved}"
            out.insert(h.label_at + 1, f"{indent}s_or_saveexec_b64 {saved}, {src}{note}")
        else:   # s_or_saveexec_b64 not fused with its s_xor, or the s_or_b64 exec, exec, saved of an if without else: only move it up
            flip_text = _code(out[h.flip_at])
            del out[h.flip_at]
            out.insert(h.label_at + 1, f"{indent}{flip_text}{note}")
    left = find(out)
    if left:
        raise RepairError("the rewrite left the shape in place:\n" + "\n".join(map(str, left)))
    return out, len(hits)
