#!/bin/bash
set -o pipefail
O=gpurun_out/r05
mkdir -p $O
bash tools/collect_profiles.sh r05 c
timeout -k 10 900 python -m pytest tests/test_user_objects.py tests/test_unit_probe.py tests/test_user_metric.py -q -m gpu > $O/pytest_units3.log 2>&1; echo "pytest rc=$?"
tail -n 6 $O/pytest_units3.log
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --size 1024 --steps 5 --warmup 1 --cpu-sample 0 --extras 0 --live-counters 0 --emit-row-checksums > $O/n1_1024.json 2> /dev/null
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --backend gloo --size 1024 --steps 5 --warmup 1 --checksum-reference $O/n1_1024.json 2> /dev/null | grep "^{" > $O/bench_gloo_2rank_rehearsal_line.json; echo "2-rank rc=$?"
RTGR_BENCH_CORRUPT_RANK=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 2 --backend gloo --size 1024 --steps 5 --warmup 1 --checksum-reference $O/n1_1024.json 2> $O/corrupt.err | grep "^{" > $O/bench_gloo_2rank_corrupted_rank_line.json; echo "corrupted 2-rank rc=${PIPESTATUS[0]}"
python bench.py --size 1024 --steps 3 --warmup 1 --cpu-sample 0 --extras 0 --live-counters 0 --entry sharded --ctx-devices 3 --checksum-reference $O/n1_1024.json 2> /dev/null | grep "^{" > $O/bench_sharded_3dev_rehearsal_line.json; echo "sharded rc=$?"
python - <<'PY'
import json
for f in ("bench_gloo_2rank_rehearsal_line", "bench_gloo_2rank_corrupted_rank_line", "bench_sharded_3dev_rehearsal_line"):
    d = json.load(open(f"gpurun_out/r05/{f}.json"))
    print(f, d["ms_per_step"], d.get("frame_checksum_ok"), d.get("frame_checksum_bad_rows"), d.get("row_checksums_sha256"))
PY
