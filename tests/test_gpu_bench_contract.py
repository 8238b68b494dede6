"""bench.py's output contract, checked on the GPU box: exactly ONE JSON line on stdout carrying the driver's fields, the
`roofline` object of the dominant kernel and the `cpu_baseline` object (a short run: 20 timed steps)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    # (the default line minus what has tests of its own and takes long here: the 160 GB wide network -- test_gpu_baseline_configs.py -- and the
    #  quasi-Newton loops at full size -- test_gpu_fullsize.py; the round's collection runs the default line as the driver does)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--profile-steps", "10", "--cpu-iterations", "2",
                        "--other-configs", "barcelona31,barcelona493:f32", "--no-quasi-newton"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %d" % len(lines)
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "apg_iterations_per_sec" and d["unit"] == "iterations/s" and d["higher_is_better"] is True
    # where the run's wall-clock went: every phase of a second or more is named
    assert d["phases_s"] and all(isinstance(n_, str) and t_ >= 0.05 for n_, t_ in d["phases_s"]) and any("PMC" in n_ for n_, _ in d["phases_s"])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "barcelona493" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "k_stream_gemv"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.5 < r["frac"] < 1.0, "the streaming kernel should sit between 50 % and 100 % of the HBM peak"
    # achieved = algorithmic bytes per launch / average launch time; the PMC traffic may not be far above the algorithmic bytes
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-3 * r["achieved"]
    # traffic comes from the PMC counters of THIS run (two rocprofv3 --pmc child passes started before the GPU is touched);
    # only if the profiler cannot run it is carried over from profiles/traffic.json, and dropped when the kernels have
    # changed since that file was collected.  Either way the line says which.
    ts = r["traffic_source"]
    if ts["measured_in_this_run"]:
        assert r["traffic"] is not None and "FETCH_SIZE" in ts["how"]
        # the streaming kernel reads every byte once: PMC traffic within 3 % of the algorithmic bytes
        assert 0.97 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.03, r["traffic"] / r["algorithmic_bytes_per_launch"]
        du = r["dual_update"]
        assert du["traffic"] is not None and 0.9 < du["traffic"] / du["algorithmic_bytes_per_launch"] < 1.1
        assert r["dual_update_traffic"] == du["traffic"]
    else:
        assert (r["traffic"] is not None) == bool(ts.get("matches_current_kernels"))
    # round 6: the kernel north_star's 60 % names has FLAT figures beside the streaming kernel's (a record that keeps scalars only keeps them)
    assert r["dual_update_frac"] == r["dual_update"]["frac"] and r["dual_update_us"] == r["dual_update"]["avg_launch_us"] and r["dual_update_kernel"] == "k_dual_stage"
    assert 0.3 < r["dual_update_frac"] < 1.0 and 5.0 < r["dual_update_us"] < 60.0
    # the timed region is repeated: spread over the regions, and the other single-GPU configs of BASELINE.json ride along
    sp = d["timing_spread"]
    assert sp["regions"] >= 5 and sp["ms_per_step_min"] <= sp["ms_per_step_median"] <= sp["ms_per_step_max"]
    names = [(c.get("workload"), c.get("dtype")) for c in d.get("configs", [])]
    assert names == [("barcelona31", "f64"), ("barcelona493", "f32")], names      # the last: the headline tree in the reference's only precision
    for c in d["configs"]:
        assert "error" not in c, c
        assert c["roofline"]["kernel"] == "k_stream_gemv" and 0.3 < c["roofline"]["frac"] < 1.0 and c["cpu_baseline"]["value"] > 0
        # PMC traffic is only ever attached to the (workload, precision) it was collected on: the fp32 leg of the headline tree has passes of
        # its own (round 5 printed the fp64 figure there: a bogus 2.0 x), the others carry none
        cr = c["roofline"]
        if (c["workload"], c["dtype"]) == ("barcelona493", "f32") and ts["measured_in_this_run"]:
            assert cr["traffic"] is not None and 0.97 < cr["traffic"] / cr["algorithmic_bytes_per_launch"] < 1.03, cr["traffic"] / cr["algorithmic_bytes_per_launch"]
            assert 0.9 < cr["dual_update_traffic"] / cr["dual_update_bytes"] < 1.1
        elif c["workload"] != "barcelona493":
            assert cr["traffic"] is None
    # round 5: the one-GPU ceiling of the multi-GPU run and the quasi-Newton loops are part of the driver's line
    sc = d["shard_ceiling"]
    assert [r["world"] for r in sc["shards"]] == [2, 4, 8]
    for r in sc["shards"]:
        assert "error" not in r, r
        for k in ("rccl_one_rank", "one_shot"):
            assert 0 < r[k]["ms_per_step"] < d["ms_per_step"] and r[k]["speedup_before_wire"] > 1.0, r
    assert sc["shards"][0]["local_nodes"] in (5452, 5430) and sc["shards"][2]["local_nodes"] in (1382, 1360)
    assert sc["shards"][2]["rccl_one_rank"]["speedup_before_wire"] > 3.5 and sc["shards"][2]["exchange_budget_us_for_3p5x"] > 0
    fw = d["fused_walk_dual"]      # forward walk + dual update: one launch by shape (the default here) against the two launches, same context
    assert "error" not in fw and 0.9 * d["value"] < fw["two_launches_same_context"]["value"] < 1.1 * d["value"] and fw["speedup"] > 0.97, fw
    # the exact fast path a drop-in caller gets (RN_OPS_AUTO): timed beside the headline, with its matrix-unit use from this run's SQ counters
    sm = d["structured_mode"]
    assert "error" not in sm and sm["value"] > 4 * d["value"], sm.get("value")
    if ts["measured_in_this_run"]:
        assert "error" not in sm["mfma"], sm["mfma"]
        assert 0.05 < sm["mfma_busy_frac"] < 1.0 and any(k.startswith("k_gemm_comp") for k in sm["mfma"]["kernels"]), sm["mfma"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "iterations/s" and c["value"] > 0
    assert c["scaled_by_nodes"] is False and "10864 nodes" in c["sample"]     # timed on the bench workload itself
    assert d["value"] > 10 * c["value"]                  # north_star: >= 10x the host-CPU baseline at 1 GPU


def _devices():
    # asked in a child process: importing torch into the process that also drives librapidnet_hip contexts (every other GPU test
    # of the suite) ends in "double free or corruption" at interpreter exit
    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE, timeout=300)
    return int(out.stdout.decode().strip().splitlines()[-1])


def test_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts its two ranks itself (a child `torch.distributed.run`, before the
    parent touches the GPU).  On a box with fewer GPUs than ranks the run must die INSIDE RCCL -- which refuses two ranks on one
    device -- with a non-zero exit code and no JSON line; with --allow-oversubscribe it rehearses the whole launcher path (two
    processes, gloo rendezvous, rn_create_sharded on both, agreed fallback exchange, max-over-ranks timing) and labels the
    line as a fallback.  With two or more GPUs the plain form simply has to produce a sharded result."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--profile-steps", "10", "--repeats", "1",
            "--cpu-iterations", "2"]
    if _devices() >= 2:
        p = subprocess.run(base, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        d = json.loads([l for l in p.stdout.decode().splitlines() if l.strip()][-1])
        assert d["n_gpus"] == 2 and d["rccl"]["ranks_seen_by_rccl"] == 2 and "FALLBACK" not in d["config"]["parallelism"]
        return
    p = subprocess.run(base + ["--no-traffic", "--other-configs", ""], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0
    assert not [l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")], "no result line for a run that could not create its communicator"
    err = p.stderr.decode()
    assert "ncclCommInitRank failed" in err and "RCCL could not create the communicator" in err, err[-2000:]
    # (the rehearsal leaves out what other tests cover and what takes minutes on an oversubscribed card: the PMC pre-pass -- the N = 1
    #  contract test checks the counters -- and the wide fp32 network's 160 GB on one device)
    p = subprocess.run(base + ["--allow-oversubscribe", "--no-traffic", "--other-configs", "", "--time-budget", "300"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "partial" not in d, d["partial"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert "FALLBACK" in d["config"]["parallelism"] and d["rccl"]["ranks"] == 2 and d["rccl"]["ranks_seen_by_rccl"] == 0
    assert d["local_nodes"] in (5452, 5430)        # half of the 493 chains + the 18 replicated crown nodes
    # the N > 1 line explains itself: who held how many nodes, every rank's own step time, the CPU leg
    pr = d["per_rank"]
    assert sorted(pr["local_nodes"]) == [5430, 5452] and pr["local_nodes_min"] == 5430 and pr["local_nodes_max"] == 5452
    assert len(pr["ms_per_step_own"]) == 2 and 0 < pr["ms_per_step_own_min"] <= pr["ms_per_step_own_max"] <= d["ms_per_step"] * 1.001
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "10864 nodes" in c["sample"]
    # the second worker (the context chooses its exchange itself) is attempted last and reported either way (here: no RCCL communicator, so it
    # says why it did not run); `exchange` names what `value` was measured on, the per-iteration collective has a top-level figure, and the
    # same-box ceiling of a rank's shard gives speedup_vs_shard_ceiling
    assert "alt_exchange" in d and ("error" in d["alt_exchange"] or d["alt_exchange"]["value"] > 0), d.get("alt_exchange")
    ex = d["exchange"]
    assert ex["chosen"] == "rccl_default" or ex["chosen"].startswith("auto:"), ex
    assert set(ex["candidates"]) == {"rccl_default", "auto"} and ex["candidates"]["rccl_default"]["value"] == ex["value_rccl_default"]
    assert d["value"] >= ex["value_rccl_default"]
    sb = d["shard_ceiling_same_box"]
    assert "error" not in sb and sb["world"] == 2 and sb["local_nodes"] in (5452, 5430) and 0 < d["speedup_vs_shard_ceiling"] < 1.5, (sb, d.get("speedup_vs_shard_ceiling"))
    # the supervisor's phase log: the optional job was decided once, by rank 0, and the run stayed inside its budget
    err = p.stderr.decode()
    assert "one-shot exchange job: run" in err or "one-shot exchange job: skip" in err, err[-1500:]


def test_a_rank_that_dies_before_the_communicator_set_up_ends_the_job_within_seconds():
    """The first real multi-GPU run must not be able to hang: every rank creates its shard context WITHOUT a communicator, the
    ranks agree over gloo that all succeeded, and only then call rn_comm_init.  Here rank 1 leaves abruptly right before the
    communicator set-up ($RAPIDNET_BENCH_FAULT): rank 0 must notice (its gloo collective fails, or the launcher ends it), nobody
    may be left inside ncclCommInitRank, and the run must end with a non-zero code and no result line -- within seconds, not at
    somebody's lease timeout."""
    import time

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--profile-steps", "0", "--repeats", "0",
           "--no-cpu-baseline", "--other-configs", "", "--allow-oversubscribe"]
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, RAPIDNET_BENCH_FAULT="before_comm_init:1"))
    elapsed = time.time() - t0
    err = p.stderr.decode()
    assert p.returncode != 0, err[-2000:]
    assert "injected fault at 'before_comm_init'" in err, err[-2000:]
    assert not [l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")]
    assert "ncclCommInitRank" not in err.replace("rn_comm_init / ncclCommInitRank", ""), "a rank entered the communicator set-up although a peer had failed"
    assert elapsed < 120, "took %.0f s" % elapsed          # set-up (problem data, two contexts) + a few seconds; no lease-long wait
