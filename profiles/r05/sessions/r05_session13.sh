#!/bin/bash
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
( time python bench.py ) > $O/bench_default2.log 2>&1; echo "bench rc=$?"
grep "^{" $O/bench_default2.log | tail -n 1 > $O/bench_default_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/bench_default_line.json"))
r = d["roofline"]
print("headline", d["value"], d["ms_per_step"], "frac", r["frac"], r.get("counters"), "checksum", d.get("frame_checksum_ok"))
for k, v in d.get("variants", {}).items():
    rr = v.get("roofline", {})
    print(k, v.get("ms_per_pass"), "frac", rr.get("frac"), rr.get("counters"), v.get("two_frames_in_flight"), v.get("user_over_builtin"), v.get("same_frame"), v.get("error"))
PY
tail -n 4 $O/bench_default2.log | cut -c1-200
