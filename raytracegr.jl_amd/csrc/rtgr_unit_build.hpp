// In-process build of a run-time metric's translation unit (rtgr_user_metric_compile / rtgr_user_metric_build, include/rtgr.h): source
// text -> gfx950 code object with no hipcc on the box and no child process.
//
//   hiprtc (-fgpu-rdc)            source + the device headers next to the library -> optimised LLVM bitcode, device libraries linked in
//   libamd_comgr  CODEGEN         bitcode -> assembly LISTING
//   rtgr_isa_repair.hpp           the listing is checked for — and cleared of — register copies that ROCm 7.2's register allocator
//                                 places ahead of a FLOW block's EXEC flip: the reason for this detour; plain hiprtc
//                                 hands back a code object, and the heavy example metric comes out of it faulty at every occupancy
//   libamd_comgr  ASSEMBLE, LINK  listing -> relocatable -> code object
//
// walked over the occupancy levels of the generic-RHS kernels (2 / 3 waves per SIMD for Float64 / Float32, then 1 / 2, then 1 / 1)
// until the integrate kernels no longer spill more than RTGR_USER_MAX_SCRATCH bytes per lane — read from the listing's kernel
// metadata, so no GPU is needed anywhere in this file.  Both libraries are resolved with dlopen at first use (they are what the HIP
// runtime itself compiles with): no link-time dependency.  Host code only.
#pragma once
#include <amd_comgr/amd_comgr.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rtgr_isa_repair.hpp"

namespace rtgr {
namespace unit_build {

constexpr int RTGR_USER_MAX_SCRATCH = 64;   // bytes per lane above which a unit is rebuilt at a lower occupancy (== user_metric.MAX_SCRATCH)

struct Hiprtc {
    void* h = nullptr;
    int (*create)(void**, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*compile)(void*, int, const char* const*) = nullptr;
    int (*log_size)(void*, size_t*) = nullptr;
    int (*log)(void*, char*) = nullptr;
    int (*bitcode_size)(void*, size_t*) = nullptr;
    int (*bitcode)(void*, char*) = nullptr;
    int (*destroy)(void**) = nullptr;
    bool ok() const { return create && compile && log_size && log && bitcode_size && bitcode && destroy; }
};
inline Hiprtc& hiprtc() {
    static Hiprtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
        }
        if (!r.h) return;
        r.create = (decltype(r.create))dlsym(r.h, "hiprtcCreateProgram");
        r.compile = (decltype(r.compile))dlsym(r.h, "hiprtcCompileProgram");
        r.log_size = (decltype(r.log_size))dlsym(r.h, "hiprtcGetProgramLogSize");
        r.log = (decltype(r.log))dlsym(r.h, "hiprtcGetProgramLog");
        r.bitcode_size = (decltype(r.bitcode_size))dlsym(r.h, "hiprtcGetBitcodeSize");
        r.bitcode = (decltype(r.bitcode))dlsym(r.h, "hiprtcGetBitcode");
        r.destroy = (decltype(r.destroy))dlsym(r.h, "hiprtcDestroyProgram");
    });
    return r;
}

struct Comgr {
    void* h = nullptr;
    decltype(&amd_comgr_create_data) create_data = nullptr;
    decltype(&amd_comgr_set_data) set_data = nullptr;
    decltype(&amd_comgr_set_data_name) set_data_name = nullptr;
    decltype(&amd_comgr_get_data) get_data = nullptr;
    decltype(&amd_comgr_release_data) release_data = nullptr;
    decltype(&amd_comgr_create_data_set) create_data_set = nullptr;
    decltype(&amd_comgr_destroy_data_set) destroy_data_set = nullptr;
    decltype(&amd_comgr_data_set_add) data_set_add = nullptr;
    decltype(&amd_comgr_create_action_info) create_action_info = nullptr;
    decltype(&amd_comgr_destroy_action_info) destroy_action_info = nullptr;
    decltype(&amd_comgr_action_info_set_isa_name) set_isa_name = nullptr;
    decltype(&amd_comgr_action_info_set_option_list) set_option_list = nullptr;
    decltype(&amd_comgr_do_action) do_action = nullptr;
    decltype(&amd_comgr_action_data_count) data_count = nullptr;
    decltype(&amd_comgr_action_data_get_data) data_get = nullptr;
    bool ok() const {
        return create_data && set_data && set_data_name && get_data && release_data && create_data_set && destroy_data_set && data_set_add &&
               create_action_info && destroy_action_info && set_isa_name && set_option_list && do_action && data_count && data_get;
    }
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
#define RTGR_COMGR_SYM(field, sym) c.field = (decltype(c.field))dlsym(c.h, #sym)
        RTGR_COMGR_SYM(create_data, amd_comgr_create_data);
        RTGR_COMGR_SYM(set_data, amd_comgr_set_data);
        RTGR_COMGR_SYM(set_data_name, amd_comgr_set_data_name);
        RTGR_COMGR_SYM(get_data, amd_comgr_get_data);
        RTGR_COMGR_SYM(release_data, amd_comgr_release_data);
        RTGR_COMGR_SYM(create_data_set, amd_comgr_create_data_set);
        RTGR_COMGR_SYM(destroy_data_set, amd_comgr_destroy_data_set);
        RTGR_COMGR_SYM(data_set_add, amd_comgr_data_set_add);
        RTGR_COMGR_SYM(create_action_info, amd_comgr_create_action_info);
        RTGR_COMGR_SYM(destroy_action_info, amd_comgr_destroy_action_info);
        RTGR_COMGR_SYM(set_isa_name, amd_comgr_action_info_set_isa_name);
        RTGR_COMGR_SYM(set_option_list, amd_comgr_action_info_set_option_list);
        RTGR_COMGR_SYM(do_action, amd_comgr_do_action);
        RTGR_COMGR_SYM(data_count, amd_comgr_action_data_count);
        RTGR_COMGR_SYM(data_get, amd_comgr_action_data_get_data);
#undef RTGR_COMGR_SYM
    });
    return c;
}

// one comgr action on one input: `in` of kind in_kind -> the single output of kind out_kind; false with *why on failure
inline bool comgr_action(amd_comgr_action_kind_t kind, amd_comgr_data_kind_t in_kind, const std::string& in, const char* in_name,
                         amd_comgr_data_kind_t out_kind, const std::vector<const char*>& opts, std::string& out, std::string* why) {
    Comgr& C = comgr();
    amd_comgr_data_t d{0}, o{0};
    amd_comgr_data_set_t is{0}, os{0};
    amd_comgr_action_info_t ai{0};
    bool have_d = false, have_o = false, have_is = false, have_os = false, have_ai = false, done = false;
    auto S = [](amd_comgr_status_t s) { return s == AMD_COMGR_STATUS_SUCCESS; };
    do {
        if (!S(C.create_data(in_kind, &d))) break;
        have_d = true;
        if (!S(C.set_data(d, in.size(), in.data())) || !S(C.set_data_name(d, in_name))) break;
        if (!S(C.create_data_set(&is))) break;
        have_is = true;
        if (!S(C.create_data_set(&os))) break;
        have_os = true;
        if (!S(C.data_set_add(is, d))) break;
        if (!S(C.create_action_info(&ai))) break;
        have_ai = true;
        if (!S(C.set_isa_name(ai, "amdgcn-amd-amdhsa--gfx950"))) break;
        if (!opts.empty() && !S(C.set_option_list(ai, const_cast<const char**>(opts.data()), opts.size()))) break;
        if (!S(C.do_action(kind, ai, is, os))) break;
        size_t cnt = 0;
        if (!S(C.data_count(os, out_kind, &cnt)) || cnt != 1) break;
        if (!S(C.data_get(os, out_kind, 0, &o))) break;
        have_o = true;
        size_t n = 0;
        if (!S(C.get_data(o, &n, nullptr))) break;
        out.resize(n);
        if (!S(C.get_data(o, &n, &out[0]))) break;
        done = true;
    } while (false);
    if (have_o) (void)C.release_data(o);
    if (have_ai) (void)C.destroy_action_info(ai);
    if (have_os) (void)C.destroy_data_set(os);
    if (have_is) (void)C.destroy_data_set(is);
    if (have_d) (void)C.release_data(d);
    if (!done && why) *why = std::string("libamd_comgr action ") + std::to_string((int)kind) + " on " + in_name + " failed";
    return done;
}

// worst `.private_segment_fixed_size` of the unit's integrate kernels, from the kernel metadata at the end of a listing (YAML: a
// kernel's own keys stand at an indentation of four, `.name` before `.private_segment_fixed_size`; its arguments' keys deeper)
inline int integrate_scratch(const std::vector<std::string>& lines) {
    int worst = 0;
    bool inside = false;
    std::string name;
    for (const std::string& l : lines) {
        if (l.find(".amdgpu_metadata") != std::string::npos) { inside = l.find(".end_amdgpu_metadata") == std::string::npos; continue; }
        if (!inside || l.size() < 6 || l.compare(0, 4, "    ") != 0 || l[4] != '.') continue;
        if (l.compare(4, 6, ".name:") == 0) {
            const size_t v = l.find_first_not_of(" \t", 10);
            name = v == std::string::npos ? std::string() : l.substr(v);
        } else if (l.compare(4, 28, ".private_segment_fixed_size:") == 0 && name.compare(0, 19, "rtgr_user_integrate") == 0) {
            const long v = std::strtol(l.c_str() + 32, nullptr, 10);
            if (v > worst) worst = (int)v;
        }
    }
    return worst;
}

struct Built {
    std::string image;     // the code object
    int level = -1;        // occupancy level it was built at
    int scratch = 0;       // bytes per lane of its integrate kernels
    int repaired = 0;      // FLOW blocks rewritten in its listing
};

// 0 = ok; 1 = the user's source does not compile (log in *why); 2 = anything else (*why)
// `defines`: what the unit is made of (-DRTGR_USER_NE=3, -DRTGR_USER_KS=1, -DRTGR_USER_OBJECTS=1, -DRTGR_UNIT_BUILTIN_METRIC=…,
// -DRTGR_HEADER_HASH=…: rtgr_units.hip plan_unit, the same rules as user_metric.py unit_defines)
inline int build(const std::string& unit, const std::string& include_dir, const std::vector<std::string>& defines, Built* out, std::string* why) {
    static std::mutex one_build_at_a_time;   // (the compiler libraries keep process-wide state; builds are seconds long and rare)
    std::lock_guard<std::mutex> lock(one_build_at_a_time);
    Hiprtc& R = hiprtc();
    if (!R.ok()) { *why = "libhiprtc not found: build the unit with `python -m raytracegr.jl_amd.user_metric` and use rtgr_user_metric_load"; return 2; }
    if (!comgr().ok()) { *why = "libamd_comgr not found: build the unit with `python -m raytracegr.jl_amd.user_metric` and use rtgr_user_metric_load"; return 2; }
    // hiprtc pre-includes the HIP device API and has no system headers: the two the units ask for are given in memory
    static const char* const hdr_src[] = {
        "#pragma once\n",
        "#pragma once\ntypedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
        "typedef int int32_t; typedef unsigned int uint32_t; typedef long int64_t; typedef unsigned long uint64_t;\n"};
    static const char* const hdr_name[] = {"hip/hip_runtime.h", "stdint.h"};
    // occupancy levels (generic-RHS kernels | closed-form kernels of a built-in metric: whichever the unit instantiates reads its own)
    static const char* const LEVELS[][5] = {{nullptr, nullptr, nullptr, nullptr, nullptr},
                                            {"-DRTGR_WAVES_PER_SIMD_GENERIC=1", "-DRTGR_WAVES_PER_SIMD_GENERIC_F32=2",
                                             "-DRTGR_WAVES_PER_SIMD_FAR=2", "-DRTGR_WAVES_PER_SIMD=1", "-DRTGR_WAVES_PER_SIMD_F32=2"},
                                            {"-DRTGR_WAVES_PER_SIMD_GENERIC=1", "-DRTGR_WAVES_PER_SIMD_GENERIC_F32=1",
                                             "-DRTGR_WAVES_PER_SIMD_FAR=1", "-DRTGR_WAVES_PER_SIMD=1", "-DRTGR_WAVES_PER_SIMD_F32=1"}};
    const std::string inc = "-I" + include_dir;
    std::string problems;
    Built best;
    for (int level = 0; level < 3; level++) {
        void* prog = nullptr;
        if (R.create(&prog, unit.c_str(), "rtgr_user_unit.hip", 2, hdr_src, hdr_name) != 0) { *why = "hiprtcCreateProgram failed"; return 2; }
        // -fgpu-rdc: hiprtc then keeps the optimised bitcode (hiprtcGetBitcode) instead of going on to a code object
        std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fgpu-rdc", inc.c_str()};
        for (const std::string& d : defines) opts.push_back(d.c_str());
        for (const char* o : LEVELS[level]) if (o) opts.push_back(o);
        const int cr = R.compile(prog, (int)opts.size(), opts.data());
        if (cr != 0) {
            size_t ls = 0;
            std::string log;
            if (R.log_size(prog, &ls) == 0 && ls > 1) { log.resize(ls); (void)R.log(prog, &log[0]); }
            (void)R.destroy(&prog);
            if (log.size() > 6000) log.resize(6000);
            *why = "hiprtc failed on the unit's source:\n" + log;
            return 1;
        }
        size_t bs = 0;
        std::string bc;
        if (R.bitcode_size(prog, &bs) != 0 || bs == 0) { (void)R.destroy(&prog); *why = "hiprtcGetBitcodeSize failed"; return 2; }
        bc.resize(bs);
        const int gr = R.bitcode(prog, &bc[0]);
        (void)R.destroy(&prog);
        if (gr != 0) { *why = "hiprtcGetBitcode failed"; return 2; }
        std::string listing;
        if (!comgr_action(AMD_COMGR_ACTION_CODEGEN_BC_TO_ASSEMBLY, AMD_COMGR_DATA_KIND_BC, bc, "rtgr_user_unit.bc", AMD_COMGR_DATA_KIND_SOURCE, {"-O3"},
                          listing, why)) return 2;
        std::vector<std::string> lines;
        for (size_t p = 0; p <= listing.size();) {
            const size_t e = listing.find('\n', p);
            lines.push_back(listing.substr(p, (e == std::string::npos ? listing.size() : e) - p));
            if (e == std::string::npos) break;
            p = e + 1;
        }
        std::string reason;
        const int repaired = isa_repair::repair(lines, &reason);
        if (repaired < 0) {   // this level's code carries the fault in a form the rewrite is not proven for: the next level is other code
            problems += "level " + std::to_string(level) + ": " + reason + "\n";
            continue;
        }
        const int scratch = integrate_scratch(lines);
        if (best.level >= 0 && scratch > RTGR_USER_MAX_SCRATCH && level < 2) continue;   // no better than what is kept
        if (repaired > 0) {
            listing.clear();
            for (size_t k = 0; k < lines.size(); k++) { listing += lines[k]; if (k + 1 < lines.size()) listing += '\n'; }
        }
        std::string obj;
        Built b;
        if (!comgr_action(AMD_COMGR_ACTION_ASSEMBLE_SOURCE_TO_RELOCATABLE, AMD_COMGR_DATA_KIND_SOURCE, listing, "rtgr_user_unit.s",
                          AMD_COMGR_DATA_KIND_RELOCATABLE, {}, obj, why)) return 2;
        if (!comgr_action(AMD_COMGR_ACTION_LINK_RELOCATABLE_TO_EXECUTABLE, AMD_COMGR_DATA_KIND_RELOCATABLE, obj, "rtgr_user_unit.o",
                          AMD_COMGR_DATA_KIND_EXECUTABLE, {}, b.image, why)) return 2;
        b.level = level; b.scratch = scratch; b.repaired = repaired;
        best = std::move(b);
        if (scratch <= RTGR_USER_MAX_SCRATCH) break;   // fits: done.  Otherwise the next level has more registers per lane
    }
    if (best.level < 0) {
        *why = "every occupancy level of this metric compiles to code with vector instructions ahead of an EXEC flip that cannot be repaired "
               "(rtgr_isa_repair.hpp):\n" + problems;
        return 2;
    }
    *out = std::move(best);
    return 0;
}

}  // namespace unit_build
}  // namespace rtgr
