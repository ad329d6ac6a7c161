"""CPU: the committed bench line of the round (profiles/rNN_bench_line.json, plain `python
bench.py` on the GPU box) against the contract the driver parses: the fields, the two objects the
scope table asks for (`roofline`, `cpu_baseline`), and the arithmetic that ties them together --
`value` x `ms_per_step` is the step's 4096 frames, `roofline.frac` x peak x `ms_per_step` is the
step's algorithmic bytes (within 1 %)."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_line.json")))
    assert lines, "no committed bench line"
    with open(lines[-1]) as fh:
        return lines[-1], json.loads(fh.read().strip().splitlines()[-1])


def test_fields_of_the_contract():
    path, b = _latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in b, (path, k)
    assert b["higher_is_better"] is True and b["scaling"] == "weak" and b["data"] == "synthetic"
    assert b["vs_baseline"] is None            # BASELINE.md holds no published number for it
    assert "workload" in b["config"] and "model" not in b["config"]
    with open(os.path.join(ROOT, "BASELINE.json")) as fh:
        base = json.load(fh)
    assert b["metric"].split(" (")[0] in base["metric"]      # senone-frames/sec
    r, c = b["roofline"], b["cpu_baseline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_the_line_is_consistent_with_itself():
    path, b = _latest()
    r = b["roofline"]
    frames = 4096 * b["n_gpus"]
    # value = frames per step / wall time per step
    assert b["value"] * b["ms_per_step"] * 1e-3 == pytest.approx(frames, rel=1e-3)
    # frac on the WALL time: frac x peak x ms_per_step = algorithmic bytes per frame x frames
    got = r["frac"] * r["peak"] * 1e9 * b["ms_per_step"] * 1e-3
    assert got == pytest.approx(r["algorithmic_bytes_per_frame"] * frames, rel=0.01)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-6)
    # the event-based figure cannot be worse than the wall's (the events sit inside the region)
    assert r["frac_kernels"] >= r["frac"] * 0.999
    # counters, when quoted, are those of the sources the line ran on; traffic per step is below
    # the algorithmic bytes here (the 2 MB mixture-weight table is served by the L2s)
    if r["traffic"] is not None:
        assert r["kernel_src_sha"] in r["traffic_source"]
        assert 0 < r["traffic"] < r["algorithmic_bytes_per_frame"] * frames
        assert r["hbm_traffic_frac"] == pytest.approx(
            r["traffic"] / (r["kernel_ms"] * 1e-3) / (r["peak"] * 1e9), rel=0.01)


def test_the_profile_behind_the_line_agrees_with_it():
    """rocprofv3's per-kernel averages (same command under the profiler, the summary committed
    beside the line) add up to the line's device time per step within the pool's spread."""
    import csv
    path, b = _latest()
    stats = path.replace("_bench_line.json", "_kernel_stats.csv")
    assert os.path.exists(stats), stats
    with open(stats) as fh:
        rows = list(csv.DictReader(fh))
    pick = lambda key: next(float(x["AverageNs"]) for x in rows if key in x["Name"])
    per_step_us = (pick("ptm_topn_mfma_kernel") + pick("ptm_senone_kernel")) * 1e-3
    assert per_step_us == pytest.approx(b["roofline"]["kernel_ms"] * 1e3, rel=0.06)
