// Load-time audit of a gfx950 code object for ONE code-generation fault of ROCm 7.2's LLVM: a register copy,
// spill or reload that the register allocator has put at the top of the FLOW block of a divergent if / else, BEFORE the
// instruction that switches EXEC to the lanes of the `else` side — so that it runs for the `then` lanes only (in the usual case:
// for no lane at all) and every other lane later reads a stale register.  Found in round 4 in the Float64 FULL pass of a heavy
// run-time metric; raytracegr.jl_amd/isa_exec.py has the story, the same rule on assembly listings, and the repair that
// user_metric.compile_user_metric applies.  This file is the rule on a code object's DISASSEMBLY, so that every image handed to
// rtgr_user_metric_load / produced by rtgr_user_metric_compile is looked at whatever built it:
//
//     <address an s_cbranch_execz targets>:
//         vector instruction(s)                      <- reported
//         s_andn2_saveexec_b64 / s_or_saveexec_b64   <- the EXEC flip (or s_or_b64 exec, exec, saved: the join of an if without else)
//     with no other branch target, branch or EXEC write in between.
//
// Host code only.  The disassembler is libamd_comgr's (the library hiprtc and the HIP runtime compile with), resolved with
// dlopen at first use: no link-time dependency.
#pragma once
#include <dlfcn.h>
#include <elf.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

namespace rtgr {
namespace isa_audit {

struct Comgr {
    void* h = nullptr;
    struct info_t { uint64_t handle; };
    int (*create)(const char*, uint64_t (*)(uint64_t, char*, uint64_t, void*), void (*)(const char*, void*), void (*)(uint64_t, void*), info_t*) = nullptr;
    int (*destroy)(info_t) = nullptr;
    int (*disassemble)(info_t, uint64_t, void*, uint64_t*) = nullptr;
    bool ok() const { return create && destroy && disassemble; }
};
inline Comgr& comgr() {
    static Comgr c;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"libamd_comgr.so.3", "libamd_comgr.so", "/opt/rocm/lib/libamd_comgr.so"}) {
            c.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (c.h) break;
        }
        if (!c.h) return;
        c.create = (decltype(c.create))dlsym(c.h, "amd_comgr_create_disassembly_info");
        c.destroy = (decltype(c.destroy))dlsym(c.h, "amd_comgr_destroy_disassembly_info");
        c.disassemble = (decltype(c.disassemble))dlsym(c.h, "amd_comgr_disassemble_instruction");
    });
    return c;
}

struct Inst {
    uint64_t addr;
    std::string text;      // mnemonic + operands, no leading blanks
    uint64_t target;       // branch target (valid when has_target)
    bool has_target;
};
struct Reader {
    const char* image;
    uint64_t addr, off, size;   // the .text section: address, file offset, bytes
    Inst cur;
};
inline uint64_t read_cb(uint64_t from, char* to, uint64_t size, void* u) {
    const Reader* r = (const Reader*)u;
    if (from < r->addr || from >= r->addr + r->size) return 0;
    const uint64_t k = std::min<uint64_t>(size, r->addr + r->size - from);
    std::memcpy(to, r->image + r->off + (from - r->addr), k);
    return k;
}
inline void inst_cb(const char* s, void* u) {
    while (*s == ' ' || *s == '\t') s++;
    ((Reader*)u)->cur.text = s;
}
inline void addr_cb(uint64_t a, void* u) {
    Reader* r = (Reader*)u;
    r->cur.target = a;
    r->cur.has_target = true;
}

inline bool starts(const std::string& s, const char* p) { return s.compare(0, std::strlen(p), p) == 0; }
inline bool is_vector(const std::string& s) {
    if (starts(s, "v_readlane") || starts(s, "v_readfirstlane") || starts(s, "v_writelane")) return false;   // SGPR spill traffic: EXEC-independent
    return starts(s, "v_") || starts(s, "scratch_") || starts(s, "global_") || starts(s, "flat_") || starts(s, "buffer_") || starts(s, "ds_");
}
inline bool is_flip(const std::string& s) {   // SI_ELSE's flip, or the join of an if without else (s_or_b64 exec, exec, saved)
    return starts(s, "s_andn2_saveexec_b64") || starts(s, "s_or_saveexec_b64") || starts(s, "s_or_b64 exec, exec,");
}
inline bool is_branch(const std::string& s) {
    return starts(s, "s_cbranch") || starts(s, "s_branch") || starts(s, "s_endpgm") || starts(s, "s_setpc") || starts(s, "s_swappc") || starts(s, "s_call");
}
inline bool writes_exec(const std::string& s) {
    if (!starts(s, "s_")) return false;
    if (s.find("saveexec") != std::string::npos) return true;
    const size_t sp = s.find_first_of(" \t");
    if (sp == std::string::npos) return false;
    const size_t op = s.find_first_not_of(" \t", sp);
    return op != std::string::npos && s.compare(op, 4, "exec") == 0;
}

// [off, off + size) lies inside a file of `bytes` bytes — written in subtraction form: the operands come from an untrusted file, and
// `off + size > bytes` wraps for offsets near 2^64
inline bool in_file(uint64_t off, uint64_t size, uint64_t bytes) { return off <= bytes && size <= bytes - off; }

// number of FLOW blocks of `image` with vector instructions ahead of their EXEC flip (their description appended to *report);
// NOT_UNDERSTOOD (-1) when the image is not an ELF64 code object with a .text section this code can walk, NO_DISASSEMBLER (-2) when
// libamd_comgr is not on the box — the reason in *report.  The two are different findings: a box without the disassembler cannot
// audit anything, a file that is not understood may be a container the runtime WOULD load (rtgr_units.hip: load_module_image).
constexpr int NOT_UNDERSTOOD = -1, NO_DISASSEMBLER = -2;
inline int audit(const char* image, size_t bytes, std::string* report) {
    auto note = [&](const std::string& s) { if (report) *report += s; };
    if (bytes < sizeof(Elf64_Ehdr) || std::memcmp(image, ELFMAG, SELFMAG) != 0 || image[EI_CLASS] != ELFCLASS64) { note("not an ELF64 image"); return NOT_UNDERSTOOD; }
    const Elf64_Ehdr* eh = (const Elf64_Ehdr*)image;
    if (eh->e_shoff == 0 || eh->e_shentsize != sizeof(Elf64_Shdr) || !in_file(eh->e_shoff, (uint64_t)eh->e_shnum * sizeof(Elf64_Shdr), bytes) ||
        eh->e_shstrndx >= eh->e_shnum) { note("no section table"); return NOT_UNDERSTOOD; }
    const Elf64_Shdr* sh = (const Elf64_Shdr*)(image + eh->e_shoff);
    const Elf64_Shdr& names = sh[eh->e_shstrndx];
    if (names.sh_size == 0 || !in_file(names.sh_offset, names.sh_size, bytes) || image[names.sh_offset + names.sh_size - 1] != 0) { note("bad section names"); return NOT_UNDERSTOOD; }
    Reader r{image, 0, 0, 0, {}};
    for (int i = 0; i < eh->e_shnum; i++) {
        if (sh[i].sh_name >= names.sh_size) continue;
        if (std::strcmp(image + names.sh_offset + sh[i].sh_name, ".text") == 0 && sh[i].sh_type == SHT_PROGBITS) {
            if (!in_file(sh[i].sh_offset, sh[i].sh_size, bytes)) { note("bad .text section"); return NOT_UNDERSTOOD; }
            r.addr = sh[i].sh_addr; r.off = sh[i].sh_offset; r.size = sh[i].sh_size;
        }
    }
    if (r.size == 0 || r.addr > UINT64_MAX - r.size) { note("no .text section"); return NOT_UNDERSTOOD; }
    Comgr& C = comgr();
    if (!C.ok()) { note("libamd_comgr not found"); return NO_DISASSEMBLER; }
    Comgr::info_t info{0};
    if (C.create("amdgcn-amd-amdhsa--gfx950", read_cb, inst_cb, addr_cb, &info) != 0) { note("amd_comgr_create_disassembly_info failed"); return NO_DISASSEMBLER; }
    std::vector<Inst> insts;
    insts.reserve(r.size / 6);
    for (uint64_t a = r.addr; a < r.addr + r.size;) {
        uint64_t sz = 0;
        r.cur = Inst{a, std::string(), 0, false};
        if (C.disassemble(info, a, &r, &sz) != 0 || sz == 0) { a += 4; continue; }   // padding between functions
        insts.push_back(std::move(r.cur));
        a += sz;
    }
    (void)C.destroy(info);
    std::unordered_set<uint64_t> labels, execz_targets;
    for (const Inst& i : insts)
        if (i.has_target && is_branch(i.text)) {
            labels.insert(i.target);
            if (starts(i.text, "s_cbranch_execz")) execz_targets.insert(i.target);
        }
    int found = 0;
    char buf[64];
    for (size_t k = 0; k < insts.size(); k++) {
        if (!execz_targets.count(insts[k].addr)) continue;
        std::vector<size_t> early;
        for (size_t j = k; j < insts.size(); j++) {
            const std::string& s = insts[j].text;
            if (j > k && labels.count(insts[j].addr)) break;
            if (is_branch(s)) break;
            if (is_flip(s)) {
                if (!early.empty()) {
                    found++;
                    for (size_t e : early) {
                        std::snprintf(buf, sizeof buf, "  .text+0x%llx: ", (unsigned long long)(insts[e].addr - r.addr));
                        note(std::string(buf) + "`" + insts[e].text + "` stands BEFORE the EXEC flip `" + s + "`\n");
                    }
                }
                break;
            }
            if (writes_exec(s)) break;
            if (is_vector(s)) early.push_back(j);
        }
    }
    return found;
}

// … of a code object, OR of a host library / executable that embeds code objects (clang offload bundles in its .hip_fatbin section:
// librtgr_hip.so itself — tests/test_build_checks.py audits the kernels the library ships): the sum over every gfx950 entry
inline int audit_any(const char* image, size_t bytes, std::string* report) {
    if (bytes >= sizeof(Elf64_Ehdr) && std::memcmp(image, ELFMAG, SELFMAG) == 0 && ((const Elf64_Ehdr*)image)->e_machine == 224 /* EM_AMDGPU */)
        return audit(image, bytes, report);
    static const char magic[] = "__CLANG_OFFLOAD_BUNDLE__";
    const size_t ml = sizeof magic - 1;
    int total = -1;
    for (size_t o = 0; o + ml + 8 <= bytes;) {
        const void* hit = memmem(image + o, bytes - o, magic, ml);
        if (!hit) break;
        const size_t b = (size_t)((const char*)hit - image);
        o = b + ml;
        if (b + ml + 8 > bytes) break;   // (a truncated header at the very end of the file)
        uint64_t n = 0;
        std::memcpy(&n, image + b + ml, 8);
        size_t p = b + ml + 8;
        for (uint64_t k = 0; k < n && k < 64 && p + 24 <= bytes; k++) {
            uint64_t off, size, tl;
            std::memcpy(&off, image + p, 8); std::memcpy(&size, image + p + 8, 8); std::memcpy(&tl, image + p + 16, 8);
            p += 24;
            if (tl > 256 || tl > bytes - p) break;
            const std::string triple(image + p, (size_t)tl);
            p += tl;
            if (triple.find("amdgcn") == std::string::npos || size == 0 || off > bytes - b || size > bytes - b - off) continue;
            std::string sub;
            const int f = audit(image + b + off, (size_t)size, &sub);
            if (f < 0) { if (report) *report += triple + ": " + sub + "\n"; return f; }
            if (f > 0 && report) *report += triple + " (bundle at 0x" + std::to_string(b) + "):\n" + sub;
            total = (total < 0 ? 0 : total) + f;
        }
    }
    if (total < 0 && report) *report += "neither a gfx950 code object nor a file with embedded (uncompressed) offload bundles";
    return total;
}

}  // namespace isa_audit
}  // namespace rtgr
