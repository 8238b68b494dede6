"""bench.py --gpus N, the per-rank SUPERVISOR's protocol, on the CPU: the launcher starts two ranks whose workers are test doubles
($RAPIDNET_BENCH_FAKE_WORKER: they sleep and print what a worker prints; no GPU, no solver), so what is checked is the part that
has never run on a multi-GPU node -- the wall-clock budget, the partial result line, an optional job that is slow or hangs, a
worker that dies after the headline.  Whatever the optional parts do, rank 0's ONE line must be on stdout within the budget."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
START_UP = 90      # allowance for the launcher's start-up on a cold container (six launchers at once, torch paged in for the first time)


@pytest.fixture(scope="module", autouse=True)
def _paged_in():
    """one untimed start of the launcher's imports, so that the timed runs below do not carry a fresh container's first `import torch`"""
    subprocess.run([sys.executable, "-c", "import torch.distributed.run"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(fake, budget, tmp, tag, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--time-budget", str(budget)] + list(extra)
    env = dict(os.environ, RAPIDNET_BENCH_FAKE_WORKER=fake, RAPIDNET_BENCH_LINE_FILE=os.path.join(str(tmp), "line_%s.json" % tag), OMP_NUM_THREADS="1")
    return time.time(), subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _finish(t0, proc, limit):
    """(elapsed, exit code, JSON lines on stdout, stderr); the pipes are drained by threads, so several runs can be awaited in any order"""
    import threading

    box = {}

    def wait():
        out, err = proc.communicate()
        box.update(out=out, err=err, t=time.time())

    th = threading.Thread(target=wait, daemon=True)
    th.start()
    return th, box, t0, proc, limit


def _result(handle):
    th, box, t0, proc, limit = handle
    th.join(limit)
    if th.is_alive():
        proc.kill()
        th.join(10)
        raise AssertionError("the run did not end within %d s" % limit)
    lines = [l for l in box["out"].splitlines() if l.strip().startswith("{")]
    return box["t"] - t0, proc.returncode, lines, box["err"]


def _supervisor_clock(err):
    """rank 0's supervisor's own clock when it said "done" (the budget runs from the supervisor's start: the launcher's start-up and the
    first import of torch in a fresh container -- a minute or two while the image pages in -- are not the supervisor's to keep)"""
    import re
    m = re.findall(r"supervisor rank 0 \[\s*([0-9.]+) s of [0-9.]+\]: done", err)
    assert m, err[-1500:]
    return float(m[-1])


def test_supervisor_keeps_the_headline_whatever_the_optional_parts_do(tmp_path):
    runs = {
        # everything in time: the complete line, with the second worker's result merged in
        "plain": _launch("headline=0.5,total=1,alt=0.5", 120, tmp_path, "plain"),
        # the optional one-shot job hangs: killed at ITS limit (min(120, left - 15) = ~36 s of a 55 s budget), line printed within the budget
        "alt_hangs": _launch("headline=0.5,total=1,alt=3600", 55, tmp_path, "alt_hangs"),
        # too little budget left for the optional job: skipped by rank 0's decision on every rank
        "alt_skipped": _launch("headline=0.5,total=1,alt=3600", 32, tmp_path, "alt_skipped"),
        # rank 0's worker hangs AFTER the headline was measured: killed at the budget, the partial line is printed, exit code 0
        "worker_hangs": _launch("headline=0.5,total=1,alt=0.5,hang_rank=0", 30, tmp_path, "worker_hangs"),
        # rank 0's worker dies after the headline: the partial line is printed
        "worker_dies": _launch("headline=0.5,total=2,alt=0.5,fail=3,fail_rank=0", 120, tmp_path, "worker_dies"),
        # the optional job is switched off
        "no_alt": _launch("headline=0.5,total=1,alt=3600", 120, tmp_path, "no_alt", ("--no-alt-exchange",)),
    }
    handles = {k: _finish(t0, p, 200) for k, (t0, p) in runs.items()}
    res = {k: _result(h) for k, h in handles.items()}
    for k, (elapsed, rc, lines, err) in res.items():
        assert rc == 0, (k, rc, err[-1500:])
        assert len(lines) == 1, (k, lines, err[-1500:])
    d = {k: json.loads(v[2][0]) for k, v in res.items()}
    assert d["plain"].get("complete") and d["plain"]["alt_exchange"] == {"value": 1.0, "fake": True} and "partial" not in d["plain"]
    assert d["alt_hangs"].get("complete") and "did not finish within its" in d["alt_hangs"]["alt_exchange"]["error"]
    assert _supervisor_clock(res["alt_hangs"][3]) < 55 and res["alt_hangs"][0] < 55 + START_UP, res["alt_hangs"][:1]
    assert d["alt_skipped"].get("complete") and d["alt_skipped"]["alt_exchange"]["error"].startswith("skipped: budget")
    assert _supervisor_clock(res["alt_skipped"][3]) < 32 and res["alt_skipped"][0] < 32 + START_UP, res["alt_skipped"][:1]   # nobody waited for the optional job
    assert "complete" not in d["worker_hangs"] and "did not finish within the time budget" in d["worker_hangs"]["partial"] and d["worker_hangs"]["value"] == 123.0
    assert res["worker_hangs"][0] < 30 + 25 + START_UP, res["worker_hangs"][0]
    assert "complete" not in d["worker_dies"] and "ended with code 3" in d["worker_dies"]["partial"]
    assert d["no_alt"].get("complete") and "alt_exchange" not in d["no_alt"]
    # the line was on file from the moment it existed
    for k in runs:
        assert json.load(open(os.path.join(str(tmp_path), "line_%s.json" % k)))["value"] == 123.0


def test_a_worker_that_fails_before_any_headline_fails_the_job(tmp_path):
    """rank 1's worker dies (code 7) while rank 0 has no headline yet: no line, a non-zero exit code, and nobody waits for the budget"""
    t0, p = _launch("headline=20,total=21,alt=0.5,fail=7,fail_rank=1,fail_at=2", 120, tmp_path, "early")
    elapsed, rc, lines, err = _result(_finish(t0, p, 120))
    assert rc != 0 and not lines, (rc, lines, err[-1500:])
    assert elapsed < 20 + 60, elapsed          # (budget: 120 s)


def test_merge_reports_the_better_exchange_as_value():
    """N > 1: rank 0's supervisor merges the first worker's line (RCCL defaults) with the second worker's result (the context chose its
    exchange itself, RCCL under the hint set): `value` is the better timed region, `exchange` names it and keeps every candidate's figure."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    line = {"metric": "apg_iterations_per_sec", "value": 4000.0, "ms_per_step": 0.25, "exchange_us": 41.0, "config": {"parallelism": "subtree sharding below stage 2, 1 RCCL all-reduce/iteration"},
            "shard_ceiling_same_box": {"ms_per_step": 0.13}}
    tune = {"transport": 1, "candidates": 3, "collective_us": 170.0, "oneshot_us": 150.0, "iterations": 200}
    better = bench.merge_exchange(dict(line, config=dict(line["config"])), {"value": 5000.0, "ms_per_step": 0.2, "chosen": "one-shot", "tune": tune, "rccl_hints": dict(bench.RCCL_HINTS)})
    assert better["value"] == 5000.0 and better["ms_per_step"] == 0.2 and better["exchange"]["chosen"] == "auto:one-shot"
    assert better["exchange"]["value_rccl_default"] == 4000.0 and better["exchange"]["candidates"]["rccl_default"]["exchange_us"] == 41.0
    assert better["exchange"]["candidates"]["auto"]["tune_us_per_iteration"] == {"collective_hinted": 170.0, "one_shot": 150.0}
    assert abs(better["speedup_vs_shard_ceiling"] - 0.13 / 0.2) < 1e-12 and "auto:one-shot" in better["config"]["parallelism"]
    worse = bench.merge_exchange(dict(line, config=dict(line["config"])), {"value": 3000.0, "ms_per_step": 1 / 3.0, "chosen": "collective", "tune": tune})
    assert worse["value"] == 4000.0 and worse["exchange"]["chosen"] == "rccl_default" and worse["exchange"]["candidates"]["auto"]["value"] == 3000.0
    failed = bench.merge_exchange(dict(line, config=dict(line["config"])), {"error": "the one-shot exchange job did not finish"})
    assert failed["value"] == 4000.0 and "error" in failed["exchange"]["candidates"]["auto"] and failed["alt_exchange"]["error"]
    assert set(bench.RCCL_HINTS) == {"NCCL_PROTO", "NCCL_ALGO", "NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS"}
