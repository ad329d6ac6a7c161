"""CPU: `bench.py --gpus N` without a launcher -- the parent's side of it.
The children here are stand-ins started through the same code path (subprocess.Popen is given
another command line): what is checked is the environment each rank gets, that ONE JSON line
reaches stdout whatever else rank 0 printed, and that a failing rank fails the parent."""
import json
import subprocess
import sys

import pytest

import bench


def _patched_popen(monkeypatch, script_of_rank):
    real = subprocess.Popen
    seen = []

    def fake(cmd, env=None, **kw):
        r = int(env["RANK"])
        seen.append(dict(env))
        return real([sys.executable, "-c", script_of_rank(r)], env=env, **kw)

    monkeypatch.setattr(subprocess, "Popen", fake)
    return seen


def test_parent_relays_one_json_line_and_sets_the_ranks_up(monkeypatch, capsys):
    def script(r):
        if r == 0:
            return ("import os, json; print('[Gloo] Rank 0 is connected to 2 peer ranks'); "
                    "print(json.dumps({'n_gpus': int(os.environ['WORLD_SIZE']), "
                    "'rank': os.environ['RANK'], 'port': os.environ['MASTER_PORT']}))")
        return "print('noise from another rank')"
    seen = _patched_popen(monkeypatch, script)
    bench.self_launch(3)
    out, err = capsys.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 3 and line["rank"] == "0"
    assert "Gloo" in err and "noise" not in out
    assert sorted(int(e["RANK"]) for e in seen) == [0, 1, 2]
    for e in seen:
        assert e["LOCAL_RANK"] == e["RANK"] and e["WORLD_SIZE"] == "3"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == line["port"]
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_failing_rank_fails_the_parent(monkeypatch, capsys):
    def script(r):
        if r == 1:
            return "import sys; sys.exit(7)"
        return "print('{\"ok\": true}')"
    _patched_popen(monkeypatch, script)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(2)
    assert "code 7" in str(e.value)


def test_no_line_from_rank_zero_is_an_error(monkeypatch, capsys):
    _patched_popen(monkeypatch, lambda r: "print('not json')")
    with pytest.raises(SystemExit) as e:
        bench.self_launch(2)
    assert "no line" in str(e.value)


def test_an_early_death_does_not_wait_for_rank_zero(monkeypatch, capsys):
    """rank 1 dies at once while rank 0 sits in a rendezvous that will never
    complete (here: a sleep of ten minutes).  The parent must notice the dead rank within
    moments, stop rank 0 and fail -- not block in rank 0's pipe until a process-group timeout."""
    import time

    def script(r):
        if r == 1:
            return "import sys; sys.exit(3)"
        return "import time; print('waiting', flush=True); time.sleep(600)"
    _patched_popen(monkeypatch, script)
    t0 = time.time()
    with pytest.raises(SystemExit) as e:
        bench.self_launch(2)
    assert time.time() - t0 < 30
    assert "code 3" in str(e.value)


def test_a_hung_job_ends_at_the_parents_limit(monkeypatch, capsys):
    monkeypatch.setenv("SSW_BENCH_LAUNCH_TIMEOUT", "1")
    _patched_popen(monkeypatch, lambda r: "import time; time.sleep(600)")
    with pytest.raises(SystemExit):
        bench.self_launch(2)
