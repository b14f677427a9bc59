#!/usr/bin/env python3
"""gpurun_out/prof_<round>_<config>/ (tools/collect_profiles.sh) -> profiles/<round>/<config>/{summary.txt, entry.json,
kernel_stats.csv, bench_line.json} and profiles/<round>/flops.json, which records the kernel-source hash the counters were
collected from (bench.py only uses an entry when that hash equals the current sources').

    python tools/merge_flops.py r02
"""
import glob
import importlib.util
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "raytracegr.jl_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
out = {"kernel_source_hash": b.kernel_source_hash(), "hash_of": "sha256 of csrc/{%s} + compile flags" % ", ".join(b.KERNEL_HEADERS),
       "how": "tools/collect_profiles.sh on one MI355X; rocprofv3 --pmc passes separate from the --kernel-trace pass", "entries": {}}
rows_out = {"kernel_source_hash": out["kernel_source_hash"],
            "what": "per-image-row checksums (sum of the RGB bit patterns of a row) of the N = 1 device-entry frames; bench.py compares the "
                    "frame every other run delivers against them and names rows and ranks on a mismatch"}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_*"))):
    cfg = os.path.basename(d)[len(f"prof_{rnd}_"):]
    ej = os.path.join(d, "entry.json")
    if not os.path.exists(ej):
        print("no entry for", cfg)
        continue
    line = None
    for l in open(os.path.join(d, "bench_plain.log")):
        if l.startswith("{"):
            line = json.loads(l)
    e = json.load(open(ej))
    key = f'{line["config"]["variant"]}/{line["dtype"]}/{line["config"]["rhs"]}'
    if cfg.endswith("_scalar"):   # the same configuration through the scalar Float32 kernel (RTGR_PACK=0): kept beside, never used by bench.py
        key += "/scalar_kernel"
    e["config_dir"] = f"profiles/{rnd}/{cfg}"
    out["entries"][key] = e
    # the N = 1 device-entry frame of this configuration from THESE kernel sources: bench.py asserts that every other way of
    # delivering the frame (N ranks, the sharded / host / pixels entries) yields the same order-independent bit-level checksum
    if line.get("frame_checksum") is not None and line["n_gpus"] == 1 and line["config"].get("entry") == "device" and not cfg.endswith("_scalar"):
        out.setdefault("frame_checksums", {})[f'{key}/{line["config"]["size"]}'] = line["frame_checksum"]
        if line.get("row_checksums"):   # one checksum per image row (bench.py --emit-row-checksums): names the rows of a mismatch at N > 1
            rows_out.setdefault("rows", {})[f'{key}/{line["config"]["size"]}'] = line.pop("row_checksums")
    dst = os.path.join(ROOT, "profiles", rnd, cfg)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(d, "summary.txt"), dst)
    shutil.copy(ej, dst)
    stats = sorted(glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if stats:   # (gpurun MERGES result directories: an earlier collection's file may still lie beside the new one)
        shutil.copy(stats[-1], os.path.join(dst, "kernel_stats.csv"))
    json.dump(line, open(os.path.join(dst, "bench_line.json"), "w"), indent=1)
    print(key, "flop/attempt %.1f" % e["flop_per_step_attempt"], "->", dst)
# the measured issue ceiling of a kernel's instruction mix (tools/micro/mix_replay.hip; profiles/<round>/mix_replay.log), taken
# at the waves-per-SIMD the kernel runs with
replay = os.path.join(ROOT, "profiles", rnd, "mix_replay.log")
OCCUPANCY = {"ks_ref0/f64/closed": 4, "ks_true08/f64/closed": 3, "ks_ref0/f64/generic": 2, "ks_true08/f64/user_ks": 3}
if os.path.exists(replay):
    import re
    table, key, pure = {}, None, False
    for l in open(replay):
        if l.startswith("--"):
            m = re.search(r"instruction mix of (\S+)", l)
            pure = "v_fma_f64 per iteration" in l
            if m:
                key = m.group(1)
            elif not pure:
                key = "ks_ref0/f64/closed"          # (the first run's header did not name its mix)
        m = re.match(r"(\d) wave\(s\)/SIMD:.* \(([0-9.]+) of 78.6\)", l)
        if m and key:
            table.setdefault(key, {}).setdefault("pure_fma" if pure else "mix", {})[int(m.group(1))] = float(m.group(2))
    for key, t in table.items():
        if key in out["entries"] and "mix" in t:
            w = OCCUPANCY.get(key, 4)
            out["entries"][key]["issue_ceiling"] = {
                "waves_per_simd": w, "frac_of_peak_this_mix_can_issue": t["mix"].get(w), "by_waves_per_simd": t["mix"],
                "frac_of_peak_pure_fma_attains": t.get("pure_fma", {}).get(w),
                "how": "tools/micro/mix_replay.hip: the per-wave-step instruction counts of this entry as INDEPENDENT "
                       "instructions, chip-wide, wall clock; " + os.path.relpath(replay, ROOT)}
json.dump(out, open(os.path.join(ROOT, "profiles", rnd, "flops.json"), "w"), indent=1)
if rows_out.get("rows"):
    json.dump(rows_out, open(os.path.join(ROOT, "profiles", rnd, "row_checksums.json"), "w"))
print("wrote profiles/%s/flops.json for kernel sources %s" % (rnd, out["kernel_source_hash"]))
