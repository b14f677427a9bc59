"""bench.py's contract on the GPU: the JSON line's schema, and the N > 1 code path (process group on RCCL, exchange on a
side stream, two frames in flight) run with a ONE-rank group — the only way its RCCL calls can be exercised on a one-GPU
box (several ranks may not share a GPU under RCCL; the multi-rank logic itself is rehearsed with gloo, tests/test_sharded_gloo.py)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*args):
    env = dict(os.environ, MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


def test_bench_line_schema_and_one_rank_exchange_path():
    base = _bench("--size", "512", "--steps", "3", "--warmup", "1", "--cpu-sample", "48", "--extras", "0")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "rays_per_s", "frame_checksum"):
        assert k in base, k
    assert base["n_gpus"] == 1 and base["steps"] == 3 and base["warmup"] == 1 and base["dtype"] == "f64"
    assert base["higher_is_better"] is True and base["vs_baseline"] is None and base["data"] == "synthetic"
    assert "workload" in base["config"] and "model" not in base["config"]
    roof = base["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["frac"] is None or 0.0 < roof["frac"] <= 1.0          # None: no profile of the current kernel sources
    assert roof["frac"] is not None or "stale_profile" in roof
    cpu = base["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and "sample" in cpu
    assert abs(base["value"] * base["ms_per_step"] * 1e-3 - base["step_attempts_per_pass"]) <= 1e-6 * base["step_attempts_per_pass"]
    # the N > 1 path on a one-rank RCCL group: same frame, bit for bit; status bytes gathered; two frames in flight
    ex = _bench("--size", "512", "--steps", "4", "--warmup", "1", "--exchange-at-n1")
    assert ex["frame_checksum"] == base["frame_checksum"]
    assert ex["exchange"] == "overlapped with the next pass" and ex["frames_in_flight"] == 2
    assert ex["gathered_status_not_event"] == 0 and ex["step_attempts_per_pass"] == base["step_attempts_per_pass"]
    assert ex["cpu_baseline"] is None                                  # the CPU leg is N = 1 only
    serial = _bench("--size", "512", "--steps", "2", "--warmup", "1", "--exchange-at-n1", "--no-overlap")
    assert serial["frame_checksum"] == base["frame_checksum"] and serial["exchange"] == "in turn"


def test_bench_host_entry_points():
    """--entry host / pixels: the timed passes go through rtgr_trace_f64 / rtgr_trace_pixels_f64 (PCIe-inclusive)."""
    base = _bench("--size", "256", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--extras", "0")
    for entry in ("host", "pixels"):
        d = _bench("--size", "256", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--extras", "0", "--entry", entry)
        assert d["config"]["entry"] == entry and d["step_attempts_per_pass"] == base["step_attempts_per_pass"]
        assert d["value"] > 0 and d["roofline"]["launches"] >= 2


def test_bench_single_process_multi_device_entries():
    """--entry sharded (rtgr_trace_sharded_device_f64, frame gathered on device 0) and --entry pixels / host with
    --ctx-devices N (the drop-in host entries dealing rows to every device of the context): the delivered frame is the
    single-device frame bit for bit (frame_checksum) and the counters are the same job's.  One GPU: the context lists it N
    times, which the line says (ctx_note) — a rehearsal of the code path, not a scaling figure."""
    base = _bench("--size", "256", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--extras", "0")
    for args in (("--entry", "sharded", "--ctx-devices", "3"), ("--entry", "pixels", "--ctx-devices", "2"),
                 ("--entry", "host", "--ctx-devices", "4"), ("--entry", "host")):
        d = _bench("--size", "256", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--extras", "0", *args)
        assert d["frame_checksum"] == base["frame_checksum"], args
        assert d["step_attempts_per_pass"] == base["step_attempts_per_pass"] and d["n_gpus"] == 1
        if "--ctx-devices" in args:
            n = int(args[-1])
            assert d["ctx_devices"] == [0] * n and "rehearsal" in d["ctx_note"]
            assert len(d["per_device_kernel_ms"]) == n and all(v["far"] > 0 for v in d["per_device_kernel_ms"])
    for k in ("contract_8d_frac", "contract_8d_note"):
        assert k in base["roofline"]
