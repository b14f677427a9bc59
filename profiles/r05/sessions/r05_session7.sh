#!/bin/bash
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
( time python bench.py ) > $O/bench_default.log 2>&1; echo "bench rc=$?"
grep "^{" $O/bench_default.log | tail -n 1 > $O/bench_default_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_default_line.json"))
r = d["roofline"]
print("headline", d["value"], d["ms_per_step"], "frac", r["frac"], r.get("counters"), "traffic_over_alg", r.get("traffic_over_algorithmic"), "hbm", r.get("hbm_measured_GBps"), r.get("hbm_algorithmic_GBps"))
print("checksum ok", d.get("frame_checksum_ok"), d.get("frame_checksum_source"))
for k, v in d.get("variants", {}).items():
    rr = v.get("roofline", {})
    print(k, {kk: v.get(kk) for kk in ("ms_per_pass", "step_attempts_per_s", "user_over_builtin", "builtin_ms_per_pass", "same_frame", "error")}, "frac", rr.get("frac"), rr.get("counters"), rr.get("frac_of_scalar_issue_peak"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:60])
PY
tail -n 4 $O/bench_default.log | cut -c1-300
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_gpu_full2.log 2>&1; echo "pytest rc=$?"
tail -n 12 $O/pytest_gpu_full2.log
