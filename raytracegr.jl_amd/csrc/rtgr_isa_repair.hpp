// The check and the repair of raytracegr.jl_amd/isa_exec.py on a gfx950 assembly LISTING, in C++, for the in-process build of run-time
// metrics (rtgr_unit_build.hpp): vector instructions that ROCm 7.2's register allocator has placed at the top of a FLOW block AHEAD of
// the instruction that switches EXEC to the block's lanes (the Python module has the full story and is the version the
// CPU tests exercise line by line — tests/test_build_checks.py holds the two to the same answers on the same listings).
//
//     .LBBn_m:                                   <- an s_cbranch_execz targets it
//         v_accvgpr_write_b32 a106, v174         <- meant for every lane of the if; runs for the `then` lanes only
//         s_mov_b32 s18, 0xb42fdfa7
//         s_andn2_saveexec_b64 s[2:3], s[2:3]    <- the flip
// becomes
//     .LBBn_m:
//         s_or_saveexec_b64 s[2:3], s[2:3]       ; saved = then lanes, EXEC = then + else lanes
//         v_accvgpr_write_b32 a106, v174
//         s_mov_b32 s18, 0xb42fdfa7
//         s_xor_b64 exec, exec, s[2:3]           ; EXEC = else lanes
// only if what stands in between is of the kinds the allocator inserts (AGPR / VGPR copies, scratch spills and reloads) plus s_mov
// constants, none touching the mask registers.  An unfused s_or_saveexec_b64, or the s_or_b64 exec, exec, saved of an if without else,
// is moved up to the label.  Host code only.
#pragma once
#include <cctype>
#include <cstring>
#include <cstdlib>
#include <set>
#include <string>
#include <unordered_set>
#include <vector>

namespace rtgr {
namespace isa_repair {

inline bool starts(const std::string& s, const char* p) { return s.compare(0, std::strlen(p), p) == 0; }
inline std::string code_of(const std::string& line) {   // the instruction / label / directive of a line, without comment and blanks
    size_t e = line.find(';');
    if (e == std::string::npos) e = line.size();
    size_t b = 0;
    while (b < e && (line[b] == ' ' || line[b] == '\t')) b++;
    while (e > b && (line[e - 1] == ' ' || line[e - 1] == '\t' || line[e - 1] == '\r')) e--;
    return line.substr(b, e - b);
}
inline std::string mnemonic(const std::string& c) { return c.substr(0, c.find_first_of(" \t")); }
inline bool is_label(const std::string& c) { return !c.empty() && c.back() == ':'; }
inline bool is_lane_sgpr_io(const std::string& c) { return starts(c, "v_readlane") || starts(c, "v_readfirstlane") || starts(c, "v_writelane"); }
inline bool is_vector(const std::string& c) {
    return starts(c, "v_") || starts(c, "scratch_") || starts(c, "global_") || starts(c, "flat_") || starts(c, "buffer_") || starts(c, "ds_");
}
inline bool is_branch(const std::string& c) {
    return starts(c, "s_cbranch") || starts(c, "s_branch") || starts(c, "s_endpgm") || starts(c, "s_setpc") || starts(c, "s_swappc") || starts(c, "s_call");
}
// operands of `mnemonic a, b, c` (blanks trimmed)
inline std::vector<std::string> operands(const std::string& c) {
    std::vector<std::string> out;
    size_t p = c.find_first_of(" \t");
    if (p == std::string::npos) return out;
    std::string cur;
    int depth = 0;
    for (; p < c.size(); p++) {
        const char ch = c[p];
        if (ch == '[') depth++;
        if (ch == ']') depth--;
        if (ch == ',' && depth == 0) { out.push_back(cur); cur.clear(); continue; }
        if ((ch == ' ' || ch == '\t') && cur.empty()) continue;
        cur += ch;
    }
    while (!cur.empty() && (cur.back() == ' ' || cur.back() == '\t')) cur.pop_back();
    if (!cur.empty()) out.push_back(cur);
    return out;
}
struct Flip { bool ok = false; std::string op, saved, src; };
inline Flip parse_flip(const std::string& c) {   // the three instructions that switch EXEC to a join / FLOW block's lanes
    Flip f;
    const std::string m = mnemonic(c);
    const std::vector<std::string> o = operands(c);
    if ((m == "s_andn2_saveexec_b64" || m == "s_or_saveexec_b64") && o.size() == 2) { f.ok = true; f.op = m; f.saved = o[0]; f.src = o[1]; }
    else if (m == "s_or_b64" && o.size() == 3 && o[0] == "exec" && o[1] == "exec") { f.ok = true; f.op = m; f.saved = "exec"; f.src = o[2]; }
    return f;
}
inline bool writes_exec(const std::string& c) {
    if (!starts(c, "s_")) return false;
    if (mnemonic(c).find("saveexec") != std::string::npos) return true;
    const std::vector<std::string> o = operands(c);
    return !o.empty() && starts(o[0], "exec");
}
// SGPR numbers a text mentions (s5, s[2:3]); vcc = -1, exec = -2
inline std::set<int> sgprs(const std::string& t) {
    std::set<int> out;
    auto word_start = [&](size_t i) { return i == 0 || !(std::isalnum((unsigned char)t[i - 1]) || t[i - 1] == '_' || t[i - 1] == '.'); };
    for (size_t i = 0; i < t.size(); i++) {
        if (!word_start(i)) continue;
        if (t[i] == 's' && i + 1 < t.size()) {
            if (t[i + 1] == '[') {
                char* e1 = nullptr;
                const long a = std::strtol(t.c_str() + i + 2, &e1, 10);
                if (e1 && *e1 == ':') {
                    char* e2 = nullptr;
                    const long b = std::strtol(e1 + 1, &e2, 10);
                    if (e2 && *e2 == ']') for (long k = a; k <= b; k++) out.insert((int)k);
                }
            } else if (std::isdigit((unsigned char)t[i + 1])) {
                char* e = nullptr;
                const long a = std::strtol(t.c_str() + i + 1, &e, 10);
                if (e && !(std::isalnum((unsigned char)*e) || *e == '_')) out.insert((int)a);
            }
        }
        if (t.compare(i, 3, "vcc") == 0) out.insert(-1);
        if (t.compare(i, 4, "exec") == 0) out.insert(-2);
    }
    return out;
}
inline bool allocator_vector(const std::string& c) {
    const std::string m = mnemonic(c);
    return m == "v_accvgpr_write_b32" || m == "v_accvgpr_read_b32" || m == "v_accvgpr_mov_b32" || m == "v_mov_b32" || m == "v_mov_b32_e32" ||
           m == "v_mov_b64" || m == "v_mov_b64_e32" || starts(m, "scratch_store_") || starts(m, "scratch_load_");
}
inline bool plain_salu(const std::string& c) {
    const std::string m = mnemonic(c);
    // (s_movk_i32: a 16-bit constant into an SGPR, the same move as s_mov_b32 with a literal; s_waitcnt: no register operand, indifferent to EXEC)
    return m == "s_mov_b32" || m == "s_mov_b64" || m == "s_movk_i32" || m == "s_nop" || m == "s_waitcnt";
}

struct Hit {
    std::string function, label, flip;
    size_t label_at = 0, flip_at = 0;
    std::vector<size_t> early;   // line indices of the vector instructions ahead of the flip
};

inline std::vector<Hit> find(const std::vector<std::string>& lines) {
    std::vector<Hit> hits;
    size_t i = 0;
    while (i < lines.size()) {
        // a function: `name:` … `.Lfunc_end`
        const std::string c = code_of(lines[i]);
        if (!(is_label(c) && c[0] != '.' && c.find_first_of(" \t") == std::string::npos)) { i++; continue; }
        const std::string fname = c.substr(0, c.size() - 1);
        size_t end = i + 1;
        while (end < lines.size() && !starts(lines[end], ".Lfunc_end")) end++;
        std::vector<std::pair<size_t, std::string>> insts;
        for (size_t k = i + 1; k < end; k++) {
            std::string ck = code_of(lines[k]);
            if (ck.empty() || (ck[0] == '.' && !is_label(ck))) continue;
            insts.emplace_back(k, std::move(ck));
        }
        std::unordered_set<std::string> targets;
        for (auto& in : insts)
            if (starts(in.second, "s_cbranch_execz")) { const auto o = operands(in.second); if (!o.empty()) targets.insert(o.back()); }
        for (size_t k = 0; k < insts.size(); k++) {
            const std::string& lc = insts[k].second;
            if (!is_label(lc) || !targets.count(lc.substr(0, lc.size() - 1))) continue;
            Hit h;
            for (size_t j = k + 1; j < insts.size(); j++) {
                const std::string& c2 = insts[j].second;
                if (is_label(c2) || is_branch(c2)) break;
                if (parse_flip(c2).ok) {
                    if (!h.early.empty()) {
                        h.function = fname; h.label = lc.substr(0, lc.size() - 1); h.label_at = insts[k].first; h.flip_at = insts[j].first; h.flip = c2;
                        hits.push_back(h);
                    }
                    break;
                }
                if (writes_exec(c2)) break;
                if (is_vector(c2) && !is_lane_sgpr_io(c2)) h.early.push_back(insts[j].first);
            }
        }
        i = end + 1;
    }
    return hits;
}

// rewrites `lines` in place; returns the number of blocks rewritten, or -1 with the reason in *why when a block is not of the shape
// the rewrite is proven for (lines are then left partly rewritten: the caller discards them)
inline int repair(std::vector<std::string>& lines, std::string* why) {
    std::vector<Hit> hits = find(lines);
    if (hits.empty()) return 0;
    auto refuse = [&](const Hit& h, const std::string& c, const char* what) {
        if (why) *why = h.function + ": `" + c + "` ahead of `" + h.flip + "` " + what;
        return -1;
    };
    for (size_t n = hits.size(); n-- > 0;) {   // bottom-up: the line indices above stay valid (find() reports in listing order)
        const Hit& h = hits[n];
        const Flip f = parse_flip(h.flip);
        std::set<int> masks = sgprs(f.saved), s2 = sgprs(f.src);
        masks.insert(s2.begin(), s2.end());
        masks.erase(-2);
        for (size_t k = h.label_at + 1; k < h.flip_at; k++) {
            const std::string c = code_of(lines[k]);
            if (c.empty() || c[0] == '.') continue;
            const std::set<int> used = sgprs(c);
            if (is_vector(c) && !is_lane_sgpr_io(c)) {
                if (!allocator_vector(c)) return refuse(h, c, "is not a copy, spill or reload");
                if (!used.empty()) return refuse(h, c, "uses a scalar register");
            } else {
                if (!(plain_salu(c) || is_lane_sgpr_io(c))) return refuse(h, c, ": only s_mov constants may stand there");
                if (used.count(-2)) return refuse(h, c, "touches EXEC");
                for (int m : masks) if (used.count(m)) return refuse(h, c, "touches the mask registers");
            }
        }
        const std::string& fl = lines[h.flip_at];
        const std::string indent = fl.substr(0, fl.find_first_not_of(" \t"));
        const std::string note = " ; isa_repair: EXEC = then + else lanes for the allocator's copies below";
        if (f.op == "s_andn2_saveexec_b64") {
            lines[h.flip_at] = indent + "s_xor_b64 exec, exec, " + f.saved;
            lines.insert(lines.begin() + (long)h.label_at + 1, indent + "s_or_saveexec_b64 " + f.saved + ", " + f.src + note);
        } else {
            const std::string flip_text = code_of(fl);
            lines.erase(lines.begin() + (long)h.flip_at);
            lines.insert(lines.begin() + (long)h.label_at + 1, indent + flip_text + note);
        }
    }
    if (!find(lines).empty()) { if (why) *why = "the rewrite left the shape in place"; return -1; }
    return (int)hits.size();
}

}  // namespace isa_repair
}  // namespace rtgr
