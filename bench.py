#!/usr/bin/env python3
"""bench.py -- APG iterations/s of the scenario-tree SMPC solve path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE accelerated-proximal-gradient iteration (extrapolation, backward+forward tree sweep, prox,
residual, dual update -- the loop body of SmpcController::algorithmApg, SmpcController.cu:1512-1522) on the
headline workload of BASELINE.json: Barcelona-style DWN (63 states / 114 inputs / nv = 97), N = 24, 493 scenarios
(17 x 29 tree, 10 864 nodes), fp64, synthetic data (rapidnet_amd/synth.py, seed 20260103), all inputs resident in HBM.
For N > 1 the SAME tree is sharded by subtree below stage 2 (18 crown nodes replicated, 493 chains dealt round-robin)
with one RCCL all-reduce of the cut parents' children sums per iteration => strong scaling.  torch.distributed (gloo) only
carries the rendezvous, the ncclUniqueId and the timing barriers; the data path is the library's own ncclAllReduce.

Rank 0 prints one JSON line (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL's cross-process buffers); must be set before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="barcelona493", help="named config of rapidnet_amd.synth.CONFIGS")
    ap.add_argument("--precision", default=None, help="f64 | f32 (default: f64, f32 for wide4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iterations", type=int, default=20, help="timed iterations of the CPU baseline (after 2 warm-ups)")
    ap.add_argument("--dense-only", action="store_true", help="skip the structured-mode pass (tuning sweeps)")
    ap.add_argument("--force-shard", action="store_true", help="debug: run the sharded code path (partition, RCCL communicator, cut all-reduce) even with one rank")
    ap.add_argument("--emulate-world", type=int, default=0, help="debug/timing only: with one rank, run rank 0's shard of an N-rank partition through the sharded code path (one-rank communicator; iterates are NOT the solution, the other ranks' sums are missing)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "torch"], help="sharded runs: 'rccl' = the library's own ncclAllReduce on the solver's stream (default); 'torch' = step-wise fallback, the cut payload is all-reduced through torch.distributed (slow; used automatically if the library's communicator cannot be created)")
    ap.add_argument("--structured", action="store_true", help="RN_OPS_STRUCTURED: no per-node operator blocks (see DESIGN.md)")
    ap.add_argument("--profile-steps", type=int, default=40, help="steps of the per-launch hipEvent pass (0 = skip)")
    ap.add_argument("--repeats", type=int, default=7, help="further timed regions of --steps steps after the contract's one (median / min / max in the JSON line)")
    ap.add_argument("--other-configs", default="barcelona31,wide4096,barcelona493:f32", help="comma list of further BASELINE.json configs timed in the same run on 1 GPU "
                    "(their own roofline objects, in the `configs` array of the JSON line); 'name:f32' / 'name:f64' picks the precision "
                    "(barcelona493:f32 = the headline tree in the reference's only precision, Configuration.h:31); '' = none")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure the HBM traffic of the dominant kernels in this run (two short rocprofv3 --pmc child runs, "
                    "started before this process touches the GPU); roofline.traffic then falls back to profiles/traffic.json if its kernel fingerprint matches")
    ap.add_argument("--traffic-probe", action="store_true", help=argparse.SUPPRESS)   # the child run the counters are collected on
    ap.add_argument("--one-shot", action="store_true", help="debug/timing only, with --force-shard / --emulate-world: the one-shot exchange at the cut (the rank writes to and reads "
                    "from its own inbox) instead of the one-rank ncclAllReduce -- what a rank executes except the wire, for both transports")
    ap.add_argument("--alt-exchange-only", action="store_true", help=argparse.SUPPRESS)   # the second worker of a rank: times the one-shot exchange (see supervise)
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)              # N > 1: the process that does a rank's GPU work (started by the rank's supervisor)
    ap.add_argument("--no-alt-exchange", action="store_true", help="N > 1: skip the extra pass that times the one-shot exchange at the cut beside the RCCL one")
    ap.add_argument("--time-budget", type=float, default=480.0, help="N > 1: wall-clock budget (seconds) of a rank's supervisor, covering the PMC pre-pass, the worker that produces the "
                    "JSON line and the optional one-shot-exchange worker; rank 0's line is on stdout no later than this, whatever the optional parts do")
    ap.add_argument("--no-shard-ceiling", action="store_true", help="1 GPU: skip the `shard_ceiling` object (rank 0's shard of a 2 / 4 / 8-rank partition through the whole sharded path on this one GPU)")
    ap.add_argument("--no-quasi-newton", action="store_true", help="1 GPU: skip the `quasi_newton` object (global-FBE / NAMA loops, dense and structured)")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE", help="A/B runs: rn_debug_set_knob on every context of the run (dual_trips, dual_pipe, vlv_wide, slab_pipe, "
                    "slab_frag, unscaled_walk, stream_two_per_cu, stream_split, nama_pair, ls_sequential, value_mfma); the JSON line lists them")
    ap.add_argument("--tune-iterations", type=int, default=200, help="N > 1, second worker: iterations per candidate of rn_exchange_autotune")
    ap.add_argument("--allow-oversubscribe", action="store_true", help="rehearsal only: with fewer GPUs than ranks the ranks share devices; RCCL refuses "
                    "that (duplicate GPU), so the exchange falls back to torch.distributed/gloo and the JSON line says so")
    return ap.parse_args()


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _pmc_medians(directory, counter):
    """kernel name -> median counter value per dispatch, from a rocprofv3 --pmc run's counter_collection.csv files."""
    import csv
    import glob
    import statistics

    out = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            d = out.setdefault(row["Kernel_Name"], {})
            d[row.get("Dispatch_Id")] = d.get(row.get("Dispatch_Id"), 0.0) + float(row["Counter_Value"])
    return {k: (statistics.median(v.values()), len(v)) for k, v in out.items()}


def measure_traffic(args, extra_probe_args=(), launcher_env=False, limit=150.0, precision=None):
    """HBM bytes per launch of k_stream_gemv and of the fused dual update FROM THE PMC COUNTERS OF THIS RUN, collected as
    /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3) prescribes: FETCH_SIZE and WRITE_SIZE in separate passes
    (`rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --traffic-probe ...`, the program itself behind `--`),
    FETCH_SIZE doubled (gfx950 tallies a 128-byte request of a wide streaming read at 64 B), WRITE_SIZE as reported, KiB x 1024,
    medians over the dispatches.  The two child runs are started BEFORE this process makes any GPU call.  Returns
    (traffic dict, source dict); on any failure the dict is empty and the source says why."""
    import shutil
    import subprocess
    import tempfile

    src = {"measured_in_this_run": False, "how": None}
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        src["why_not"] = "rocprofv3 not found"
        return {}, src
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        src["why_not"] = "this run is itself being profiled"
        return {}, src
    tmp = tempfile.mkdtemp(prefix="rn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    if launcher_env:     # called by a rank's supervisor: the probe is a one-process run of its own, not a rank of this job
        env = {k: v for k, v in env.items() if not (k.startswith("TORCHELASTIC_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE",
                                                                                                "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "RAPIDNET_BENCH_FAULT"))}
    probe = ["python3", os.path.abspath(__file__), "--traffic-probe", "--workload", args.workload, "--steps", "40", "--warmup", "20"] + list(extra_probe_args)
    if precision or args.precision:
        probe += ["--precision", precision or args.precision]
    for kv in args.knob:
        probe += ["--knob", kv]
    med = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--"] + probe,
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=limit)
            if r.returncode != 0:
                src["why_not"] = "rocprofv3 --pmc %s exited with %d: %s" % (counter, r.returncode, r.stderr[-300:])
                return {}, src
            med[counter] = _pmc_medians(out, counter)
    except Exception as e:   # noqa: BLE001 -- a missing profiler must not fail the benchmark
        src["why_not"] = "%s: %s" % (type(e).__name__, e)
        return {}, src
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    def total(prefix):
        # a kernel template has several instantiations (k_dual_stage: first / inner / last iteration of a batch): the one with the
        # most dispatches is the launch the per-launch figures of the JSON line are about
        best = max((k for k, (_, n) in med["FETCH_SIZE"].items() if k.startswith(prefix) and n >= 4), key=lambda k: med["FETCH_SIZE"][k][1], default=None)
        if best is None:
            return None, 0
        f, n = med["FETCH_SIZE"][best]
        return 1024.0 * (2.0 * f + med["WRITE_SIZE"].get(best, (0.0, 0))[0]), n

    t = {}
    for key, prefix in (("k_stream_gemv_bytes_per_launch", "void rn::k_stream_gemv<"), ("k_dual_stage_bytes_per_launch", "void rn::k_dual_stage<"),
                        ("k_dual_fused_bytes_per_launch", "void rn::k_dual_fused<")):
        v, n = total(prefix)
        if v is not None:
            t[key] = v
            t[key.replace("bytes_per_launch", "dispatches")] = n
    src.update({"measured_in_this_run": True,
                "how": "two child runs of this bench.py under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, batches of 20 iterations of the same "
                       "workload as the timed region runs them), median per dispatch, bytes = 1024 x (2 x FETCH_SIZE + WRITE_SIZE): FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM) prescribes for "
                       "16-byte-per-lane streaming reads on gfx950"})
    return t, src


def measure_mfma(args, limit=150.0):
    """Matrix-unit use of the STRUCTURED operator mode's shared-operator products from the SQ counters of this run (SURVEY.md section 8(d):
    "report MFMA utilisation" for the shared-operator model): one child run of this bench.py in structured mode under
    `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES` (counters only; started before this process touches the GPU).
    SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 1 024 SIMDs, the cycles a SIMD's matrix pipe was busy: busy fraction = that / (1 024 x
    launch duration x 2.4 GHz), median per launch; the durations are those of the counter-collecting run itself."""
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return {"error": "rocprofv3 not available (or this run is itself being profiled)"}
    tmp = tempfile.mkdtemp(prefix="rn_sq_", dir="/tmp")
    probe = ["python3", os.path.abspath(__file__), "--traffic-probe", "--structured", "--workload", args.workload, "--steps", "40", "--warmup", "20"]
    for kv in args.knob:
        probe += ["--knob", kv]
    try:
        r = subprocess.run([exe, "--kernel-trace", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "--output-format", "csv", "-d", tmp, "-o", "sq", "--"] + probe,
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=limit)
        if r.returncode != 0:
            return {"error": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES exited with %d: %s" % (r.returncode, r.stderr[-300:])}
        busy, durs = {}, {}
        short = lambda n: n.replace("void rn::", "").split("(")[0]
        for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if row.get("Counter_Name") == "SQ_VALU_MFMA_BUSY_CYCLES":
                    d = busy.setdefault(short(row["Kernel_Name"]), {})
                    d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
        for path in glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                durs.setdefault(short(row["Kernel_Name"]), []).append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
    except Exception as e:   # noqa: BLE001 -- a missing profiler must not fail the benchmark
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out, tot_busy, tot_us = {}, 0.0, 0.0
    for k, v in busy.items():
        if not k.startswith(("k_gemm_", "k_value_mfma")) or len(v) < 4 or k not in durs:
            continue
        b, d_us = statistics.median(v.values()), statistics.median(durs[k])
        floor_us = b / 1024.0 / 2400.0
        out[k] = {"launches": len(v), "busy_cycles_per_launch": b, "launch_us": d_us, "mfma_floor_us": floor_us, "busy_frac": floor_us / d_us if d_us > 0 else None}
        tot_busy += floor_us; tot_us += d_us
    if not out:
        return {"error": "no MFMA kernel in the counter files"}
    return {"kernels": out, "busy_frac": tot_busy / tot_us if tot_us > 0 else None, "unit": "fraction of the 1 024 matrix pipes' cycles (2.4 GHz) inside the launches",
            "how": "child run of this bench.py (--structured) under `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES`, medians per launch; "
                   "dense peak of v_mfma_f64_16x16x4: one per 64 cycles and SIMD"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (torch.distributed.run, one rank
    per GPU) before this process has made any GPU / HIP call, relay rank 0's JSON line and return the launcher's exit code.
    Nothing here imports torch or loads the HIP library; a process that has touched the GPU is never re-exec'ed."""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:            # the ranks send everything but rank 0's result to stderr
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr)
    rc = proc.wait()
    if rc == 0 and line is not None:
        print(line, flush=True)
        return 0
    print("bench.py --gpus %d: the launcher exited with code %d%s" % (args.gpus, rc, "" if line is None else " after a result line (discarded)"), file=sys.stderr)
    return rc if rc != 0 else 1


class AgreedFailure(RuntimeError):
    """N > 1: a set-up step failed on some rank and ALL ranks know (they raise this together, so a caller may catch it on every
    rank and go on -- used for the secondary configs; for the headline the job ends instead)."""


class Watchdog:
    """N > 1 only: a rank that makes no progress for `limit` seconds (a peer died inside a collective, ncclCommInitRank never
    returned ...) prints what it was doing and leaves with a non-zero code, so that the launcher tears the job down instead of
    the run hanging until somebody's lease ends.  beat(what) is called at every phase boundary; the library calls in between
    release the GIL, so this thread keeps running while the main thread is stuck in one."""

    def __init__(self, rank, limit):
        import threading

        self.rank, self.limit, self.what, self.t = rank, limit, "start", time.time()
        self.stop = threading.Event()
        threading.Thread(target=self._run, daemon=True).start()

    def beat(self, what, limit=None):
        self.what, self.t = what, time.time()
        if limit is not None:
            self.limit = limit

    def _run(self):
        while not self.stop.wait(1.0):
            if time.time() - self.t > self.limit:
                print("bench.py rank %d: no progress for %.0f s in phase '%s' -- giving up (exit code 4)" % (self.rank, self.limit, self.what), file=sys.stderr, flush=True)
                os._exit(4)


def _fault(point, rank):
    """Test hook ($RAPIDNET_BENCH_FAULT = '<point>:<rank>'): the named rank leaves abruptly at the named point, as a rank whose
    GPU or library call failed would (tests/test_gpu_bench_contract.py checks that its peers exit within seconds)."""
    spec = os.environ.get("RAPIDNET_BENCH_FAULT", "")
    if spec and spec.split(":")[0] == point and int(spec.split(":")[1]) == rank:
        print("bench.py rank %d: injected fault at '%s'" % (rank, point), file=sys.stderr, flush=True)
        os._exit(17)


def _job_tag():
    """One name per launch, the same in every rank's supervisor (all on one node): the rendezvous port + the launcher's run id."""
    return "%s_%s" % (os.environ.get("MASTER_PORT", "0"), "".join(c for c in os.environ.get("TORCHELASTIC_RUN_ID", "none") if c.isalnum())[:24])


def _line_file(world):
    """Where rank 0's result line is kept from the moment it exists (besides stderr): $RAPIDNET_BENCH_LINE_FILE, else
    gpurun_out/bench_line_n<N>.json under the repository (the driver pulls gpurun_out/), else /tmp."""
    path = os.environ.get("RAPIDNET_BENCH_LINE_FILE")
    if path:
        return path
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        if os.access(d, os.W_OK):
            return os.path.join(d, "bench_line_n%d.json" % world)
    except OSError:
        pass
    return "/tmp/rapidnet_bench_line_n%d.json" % world


def _fake_worker(args, rank):
    """Test double of a rank's worker ($RAPIDNET_BENCH_FAKE_WORKER = 'headline=<s>,total=<s>,alt=<s>[,fail=<code>][,hang_rank=<r>]'):
    no GPU, no torch -- it sleeps and prints what a worker prints, so that the SUPERVISOR's protocol (time budget, partial
    lines, a slow or hanging optional job, a worker that dies after the headline) is checked on the CPU in seconds
    (tests/test_bench_supervisor.py).  Never reached without that variable."""
    spec = dict(kv.split("=") for kv in os.environ["RAPIDNET_BENCH_FAKE_WORKER"].split(",") if "=" in kv)
    f = lambda k, d: float(spec.get(k, d))
    if args.alt_exchange_only:
        time.sleep(f("alt", 0.5))
        if rank == 0:
            print(json.dumps({"alt_exchange": {"value": 1.0, "fake": True}}), flush=True)
        os._exit(0)
    line = {"metric": "apg_iterations_per_sec", "value": 123.0, "unit": "iterations/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "fake": True}
    if "fail_at" in spec and int(f("fail_rank", 0)) == rank:      # dies before any headline exists
        time.sleep(f("fail_at", 1.0))
        os._exit(int(f("fail", 1)))
    time.sleep(f("headline", 0.5))
    if rank == 0:
        print(json.dumps(dict(line, partial="headline measured; secondary configs and CPU baseline pending")), flush=True)
    if int(f("hang_rank", -1)) == rank:
        time.sleep(3600)
    time.sleep(max(0.0, f("total", 1.0) - f("headline", 0.5)))
    if int(f("fail", 0)) and int(f("fail_rank", 0)) == rank:
        os._exit(int(f("fail", 0)))
    if rank == 0:
        print(json.dumps(dict(line, complete=True)), flush=True)
    os._exit(0)


def merge_exchange(out, alt):
    """N > 1, rank 0's supervisor: the line of the first worker (communicator's all-reduce, RCCL defaults) and the result of the second (the context
    chose its exchange itself, RCCL under the hint set): `value` is the better of the two, `exchange` says which it was and what each candidate took."""
    out["alt_exchange"] = alt
    first = {"value": out.get("value"), "ms_per_step": out.get("ms_per_step"), "exchange_us": out.get("exchange_us")}
    ex = {"candidates": {"rccl_default": dict(first, what="ncclAllReduce on the solver's stream, RCCL's own defaults: the first worker's timed region")},
          "chosen": "rccl_default", "value_rccl_default": out.get("value")}
    if isinstance(alt, dict) and "error" not in alt and alt.get("value") and alt.get("ms_per_step"):
        t = alt.get("tune") or {}
        ex["candidates"]["auto"] = {"value": alt["value"], "ms_per_step": alt["ms_per_step"], "chosen_transport": alt.get("chosen"), "rccl_hints": alt.get("rccl_hints"),
                                    "tune_us_per_iteration": {"collective_hinted": t.get("collective_us"), "one_shot": t.get("oneshot_us")},
                                    "what": "second worker: RN_EXCHANGE_AUTO under the RCCL hint set; its timed region ran on the transport the tuner kept"}
        if alt["value"] > (out.get("value") or 0.0):
            ex["chosen"] = "auto:" + str(alt.get("chosen"))
            out["value"], out["ms_per_step"] = alt["value"], alt["ms_per_step"]
            if alt.get("timing_spread"):
                out["timing_spread"] = alt["timing_spread"]
            if alt.get("per_rank"):
                out["per_rank"] = alt["per_rank"]
            if isinstance(out.get("config"), dict):
                out["config"]["ms_per_controlStep_500it"] = 500.0 * alt["ms_per_step"]
                out["config"]["parallelism"] = str(out["config"].get("parallelism", "")).replace("1 RCCL all-reduce/iteration", "1 exchange/iteration (%s)" % ex["chosen"])
    else:
        ex["candidates"]["auto"] = {"error": (alt or {}).get("error", "no result") if isinstance(alt, dict) else "no result"}
    sc = out.get("shard_ceiling_same_box")
    if isinstance(sc, dict) and sc.get("ms_per_step") and out.get("ms_per_step"):
        out["speedup_vs_shard_ceiling"] = sc["ms_per_step"] / out["ms_per_step"]
    out["exchange"] = ex
    return out


def supervise(args):
    """N > 1: what the launcher starts for a rank is this SUPERVISOR, which never touches the GPU.  It runs the rank's work as child
    processes, one after the other, so that at no time more than one process per rank holds the device:
      0. rank 0 only: HBM traffic of its shard from the PMC counters (two rocprofv3 passes of a one-process run);
      1. the worker (`--worker`): everything the JSON line reports, RCCL exchange; rank 0's worker prints the line -- a first,
         PARTIAL one as soon as the headline is measured, the complete one at its end;
      2. unless --no-alt-exchange: a second worker (`--worker --alt-exchange-only`) that times the one-shot exchange at the cut.
    Everything runs against ONE wall-clock budget (--time-budget, 480 s; the driver's limit is 600 s): steps 0 and 2 are optional and
    get what the budget leaves (step 0: at most 150 s per pass and only while 330 s remain for the rest; step 2: at most 120 s,
    skipped below 30 s -- rank 0 decides and tells the other supervisors through a file, so that either all ranks start it or
    none), step 1 gets the rest.  Rank 0's line is written to a file (gpurun_out/bench_line_n<N>.json) and echoed to stderr the
    moment it exists, and printed on stdout when step 2 has ended or the budget is used up, whichever comes first -- also when this
    supervisor is told to go (SIGTERM) and when the worker dies after the headline was measured (the line then says `partial`).
    A worker that fails BEFORE any headline exists ends this process with its code at once (the launcher then ends the other
    ranks); a supervisor that is ended takes its worker with it."""
    import ctypes
    import signal
    import subprocess
    import threading

    t_start = time.time()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    budget = max(30.0, float(args.time_budget))
    remaining = lambda: budget - (time.time() - t_start)
    child = [None]
    best = [None]          # rank 0: the most complete result line seen so far (text)
    printed = [False]
    line_path = _line_file(world)
    flag_path = "/tmp/rapidnet_bench_%s" % _job_tag()       # + ".headline": rank 0 has a line; + ".alt": rank 0's decision on step 2
    lock = threading.RLock()      # (re-entrant: the SIGTERM handler runs on the main thread, possibly inside keep / emit)

    def _die_with_parent():                      # the worker gets SIGKILL if this process disappears without a word
        try:
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)
        except Exception:   # noqa: BLE001
            pass

    def keep(text):
        """rank 0: a (more complete) result line exists -- file + stderr at once"""
        with lock:
            first = best[0] is None
            best[0] = text
        try:
            with open(line_path + ".tmp", "w") as f:
                f.write(text + "\n")
            os.replace(line_path + ".tmp", line_path)
            if first:
                open(flag_path + ".headline", "w").close()
        except OSError:
            pass
        print("bench.py: result line so far (also in %s): %s" % (line_path, text), file=sys.stderr, flush=True)

    def emit(note=None):
        """rank 0: print the best line on stdout, ONCE"""
        with lock:
            if printed[0] or best[0] is None:
                return best[0] is not None
            printed[0] = True
            text = best[0]
        if note:
            try:
                d = json.loads(text)
                d["partial"] = (d.get("partial", "") + "; " if d.get("partial") else "") + note
                text = json.dumps(d)
            except ValueError:
                pass
        print(text, flush=True)
        return True

    def _on_term(signum, _frame):
        if child[0] is not None and child[0].poll() is None:
            child[0].kill()
        if rank == 0 and emit("the supervisor was ended by signal %d before the run was complete" % signum):
            os._exit(0)
        os._exit(128 + signum)

    signal.signal(signal.SIGTERM, _on_term)
    signal.signal(signal.SIGINT, _on_term)
    base = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--worker"]

    def run_worker(cmd, env, key, limit, on_line=None):
        """(exit code or None after a time-out, the last stdout line that is a JSON object holding `key`)"""
        child[0] = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, preexec_fn=_die_with_parent)
        found = [None]

        def pump():
            for ln in child[0].stdout:            # everything but the result goes to stderr
                t = ln.strip()
                if t.startswith("{") and key in t:
                    found[0] = t
                    if on_line:
                        on_line(t)
                elif t:
                    print(t, file=sys.stderr, flush=True)

        th = threading.Thread(target=pump, daemon=True)
        th.start()
        try:
            rc = child[0].wait(timeout=max(1.0, limit))
        except subprocess.TimeoutExpired:
            child[0].kill()
            child[0].wait()
            rc = None
        th.join(10.0)
        return rc, found[0]

    def say(what):
        print("bench.py supervisor rank %d [%5.1f s of %.0f]: %s" % (rank, time.time() - t_start, budget, what), file=sys.stderr, flush=True)

    for suffix in (".headline", ".alt"):          # a stale file of an earlier job on the same port
        try:
            if rank == 0:
                os.remove(flag_path + suffix)
        except OSError:
            pass
    wenv = dict(os.environ)
    if rank == 0 and not (args.no_traffic or args.structured or args.alt_exchange_only or args.traffic_probe or os.environ.get("RAPIDNET_BENCH_FAKE_WORKER")):
        # HBM traffic of rank 0's shard from the PMC counters: two one-process runs of the sharded path on this rank's device
        # (`--emulate-world N`: rank 0's shard, one-rank communicator) under rocprofv3, before the rank's worker starts -- the other
        # ranks' workers wait for it in the rendezvous.  Handed to the worker, which puts it into `roofline.traffic`.
        per_pass = min(150.0, (remaining() - 330.0) / 2.0)
        say("PMC pre-pass: %s" % ("2 passes of at most %.0f s" % per_pass if per_pass >= 40.0 else "skipped (budget)"))
        if per_pass >= 40.0:
            t, src = measure_traffic(args, ("--emulate-world", str(world)), launcher_env=True, limit=per_pass)
            if isinstance(src, dict) and src.get("how"):
                src["how"] = "rank 0's shard, measured by one-process runs of the sharded path (--emulate-world %d) on rank 0's device before the job: " % world + src["how"]
        else:
            t, src = {}, {"measured_in_this_run": False, "how": None, "why_not": "skipped: budget (%.0f s left of %.0f)" % (remaining(), budget)}
        wenv["RAPIDNET_BENCH_TRAFFIC_JSON"] = json.dumps({"traffic": t, "source": src})
    # ---- step 1: the worker.  It may use what is left of the budget (a margin for printing aside).
    say("starting the worker (limit %.0f s)" % (remaining() - 5.0))
    rc, line = run_worker(base, wenv, '"metric"', remaining() - 5.0, on_line=keep if rank == 0 else None)
    if rc != 0:
        why = "did not finish within the time budget of %.0f s (killed)" % budget if rc is None else "ended with code %s" % rc
        print("bench.py rank %d: the worker %s" % (rank, why), file=sys.stderr, flush=True)
        if rank == 0:
            if emit("rank 0's worker %s after the headline was measured: what it had not finished is missing" % why):
                os._exit(0)
            os._exit(rc if rc and rc > 0 else 1)
        # another rank: if rank 0 already holds a headline the job has a result -- leave quietly (a non-zero code would make the
        # launcher end rank 0's supervisor before it prints); otherwise fail the job at once
        os._exit(0 if (rc is None or os.path.exists(flag_path + ".headline")) else (rc if rc > 0 else 1))
    # ---- step 2: the optional one-shot-exchange worker, within what the budget leaves; ONE decision for all ranks (rank 0's)
    alt = None
    if not (args.structured or args.no_alt_exchange or args.alt_exchange_only or args.traffic_probe):
        if rank == 0:
            left = remaining()
            limit = min(120.0, left - 15.0)
            decision = "run %.1f" % limit if limit >= 30.0 else "skip"
            try:
                with open(flag_path + ".alt.tmp", "w") as f:
                    f.write(decision)
                os.replace(flag_path + ".alt.tmp", flag_path + ".alt")
            except OSError:
                decision = "skip"
        else:
            decision, t_wait = None, time.time()
            while decision is None and time.time() - t_wait < 20.0:
                try:
                    decision = open(flag_path + ".alt").read().strip() or None
                except OSError:
                    time.sleep(0.1)
            decision = decision or "skip"
        say("worker done; one-shot exchange job: %s" % decision)
        if decision.startswith("run"):
            limit = float(decision.split()[1])
            env = {k: v for k, v in os.environ.items() if k != "RAPIDNET_BENCH_FAULT"}
            for hk, hv in RCCL_HINTS.items():
                env.setdefault(hk, hv)           # (a value the caller exported wins)
            env["RAPIDNET_BENCH_STORE_PREFIX"] = "alt_exchange"      # the launcher's store is shared with step 1: keys of its own
            rc2, aline = run_worker(base + ["--alt-exchange-only"], env, '"alt_exchange"', min(limit, max(1.0, remaining() - 5.0)))
            if aline is not None:
                try:
                    alt = json.loads(aline)["alt_exchange"]
                except ValueError:
                    alt = None
            if alt is None:
                alt = {"error": "the one-shot exchange job did not finish within its %.0f s (killed)" % limit if rc2 is None
                       else "the one-shot exchange job ended with code %d and no result (see stderr)" % rc2}
        else:
            alt = {"error": "skipped: budget (%.0f s of %.0f left after the headline run)" % (remaining(), budget)}
    say("done")
    if rank == 0:
        if line is None:
            print("bench.py: rank 0's worker ended without a result line", file=sys.stderr, flush=True)
            os._exit(1)
        if alt is not None:
            try:
                out = merge_exchange(json.loads(line), alt)
            except Exception as e:   # noqa: BLE001 -- whatever the second worker printed cannot take the first worker's line with it
                out = json.loads(line)
                out["alt_exchange"] = {"error": "could not be merged (%s: %s)" % (type(e).__name__, e), "raw": alt}
            with lock:
                best[0] = json.dumps(out)
        emit()
        for suffix in (".headline", ".alt"):
            try:
                os.remove(flag_path + suffix)
            except OSError:
                pass
    os._exit(0)


# The documented RCCL hint set, tried as a candidate of its own: the per-iteration payload is 30 KB (17 cut parents x 223 doubles + 2), far
# below the sizes RCCL's tuner is built around.  LL = the low-latency protocol (flag-in-data, no separate synchronisation), Tree = log-depth
# instead of a ring over 8 point-to-point xGMI hops, one channel = one workgroup of the collective kernel instead of several that each carry a
# sliver.  The library never sets NCCL_* itself; bench.py's SECOND worker -- whose communicator the auto-tuner times against the one-shot
# exchange -- runs under them, the first (the plain headline) under RCCL's own defaults, and the line reports the better one as `value`.
RCCL_HINTS = {"NCCL_PROTO": "LL", "NCCL_ALGO": "Tree", "NCCL_MAX_NCHANNELS": "1", "NCCL_MIN_NCHANNELS": "1"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(problem_name, problem, precision, timed_iterations=20, sample_levels=None):
    """The CPU oracle ("port" of the reference's sweep: oracle/apg_oracle.c, 1 thread pinned to one core) timed on THIS
    workload: the same network, tree, forecasts, step size and precision as the GPU run (SURVEY.md section 8(d),
    BASELINE.md section 2): 2 warm-up iterations, then `timed_iterations` iterations timed one by one; value = 1 / median.
    A workload whose dense per-node blocks do not fit in host memory (wide4096: 160 GB in fp32) is timed on a bounded
    sub-tree of the same network instead and scaled by node count -- the JSON says so."""
    from oracle import oracle as oracle_mod
    from oracle.oracle import Oracle
    from rapidnet_amd import synth

    oracle_mod.build(force=True, march="native", variant="_native")   # the host's full ISA, in a file of its own
    nx, nu, nd = (int(problem["network"][k][0]) for k in ("nx", "nu", "nd"))
    nv, nodes_full = int(problem["config"]["nv"][0]), int(problem["tree"]["nodes"][0])
    s = 8 if precision == "f64" else 4
    oracle_bytes = float(nodes_full) * s * (4.0 * nv * nx + 2.0 * nv * nu + 2.0 * nx * nx + nu * nu + 12.0 * (2 * nx + nu))
    try:
        host_mem = os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        host_mem = 64 << 30
    # secondary configs of a run (sample_levels given) are always timed on a bounded sample once their blocks exceed 4 GB:
    # the full wide4096 workload takes the host 126 s to factor and 7 s per iteration
    scaled = oracle_bytes > (min(0.45 * host_mem, 4e9) if sample_levels else 0.45 * host_mem)
    p, note = problem, ""
    if scaled:
        idx, _, _, _, ne, N, branching = synth.CONFIGS[problem_name]
        sample_branching = list(branching[:-1]) if len(branching) > 1 else [2]
        if sample_levels:   # the secondary configs of a run: a smaller sample keeps the whole bench within minutes
            sample_branching = list(branching[:sample_levels])
        synth.CONFIGS["_cpu_sample"] = (idx, nx, nu, nd, ne, N, sample_branching)
        p = synth.make_problem("_cpu_sample", step_size=float(problem["config"]["stepSize"][0]), feasible=problem_name in synth.FEASIBLE)
        note = "; the full tree's blocks (%.0f GB) are too large for a bounded CPU sample: timed on the %s sub-tree and scaled by nodes" % (
            oracle_bytes / 1e9, "x".join(map(str, sample_branching)))
    cores = sorted(os.sched_getaffinity(0))
    core = cores[len(cores) // 2]
    os.sched_setaffinity(0, {core})
    try:
        t0 = time.perf_counter()
        o = Oracle(p["network"], p["tree"], p["config"], precision=precision, variant="_native")
        dh, ah = synth.forecast_at(p["forecast"], 0)
        o.initialise(dh, ah)
        t_factor = time.perf_counter() - t0
        o.apg_reset()
        th = o.apg_continue(2, [1.0, 1.0])
        per_it = []
        for _ in range(timed_iterations):
            t1 = time.perf_counter()
            th = o.apg_continue(1, th)
            per_it.append(time.perf_counter() - t1)
    finally:
        os.sched_setaffinity(0, set(cores))
    med = float(np.median(per_it))
    rate = 1.0 / med * (o.nodes / float(nodes_full))
    return {
        "value": rate, "unit": "iterations/s", "cores": 1, "kind": "port",
        "sample": "%d timed iterations (after 2 warm-ups) of oracle/apg_oracle.c (gcc -O3 -march=native, 1 thread pinned to core %d) on %s, "
                  "%s, %d nodes, step size %.6g: median %.1f ms, min %.1f, max %.1f; factor step + affine terms %.1f s (not timed)%s" % (
                      timed_iterations, core, "the bench workload itself (same network, tree, forecasts)" if not scaled else "a sub-tree of the bench workload",
                      precision, o.nodes, float(p["config"]["stepSize"][0]), 1e3 * med, 1e3 * min(per_it), 1e3 * max(per_it), t_factor, note),
        "ms_per_iteration_median": 1e3 * med, "ms_per_controlStep_500it": 500.0 * 1e3 * med * (float(nodes_full) / o.nodes),
        "scaled_by_nodes": bool(scaled), "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(),
    }


def shard_ceiling(problem, device, steps, unsharded_ms, profile_steps=40):
    """1 GPU: what ONE RANK of a W-rank run executes, for W = 2, 4, 8 -- rank 0's shard (the largest: the subtrees are dealt
    round-robin) of the bench workload through the whole sharded path of the library: rn_partition_create, a real one-rank RCCL
    communicator (every ncclAllReduce of the path is issued and runs), the cut stage, device-resident batches; then the same
    context with the one-shot exchange (the rank writes to and reads from its own inbox).  Everything a rank does except the
    wire and the wait for its peers, so `speedup_before_wire` = unsharded time / this time is the CEILING of the W-GPU
    speed-up, and `exchange_budget_us_for_3p5x` (W = 8) is what the exchange over xGMI may cost per iteration with north_star's
    >= 3.5x still met.  The iterates are not the solution (the other ranks' sums are missing); timing only."""
    from rapidnet_amd import capi, synth

    cut = capi.default_cut_stage(problem["tree"])
    dh, ah = synth.forecast_at(problem["forecast"], 0)
    n_it = max(int(steps), 100)
    rows = []
    for W in (2, 4, 8):
        row, s = {"world": W}, None
        try:
            part = capi.partition_tree(problem["tree"], 0, W, cut)
            s = capi.Solver(problem["network"], part["tree"], problem["config"], precision="f64", device=device)
            s.commInit(0, 1, capi.comm_unique_id())
            s.setCutStage(cut, (part["momE"], part["momP"]))
            s.initialiseSmpcController(dh, ah)
            row["local_nodes"] = int(s.nodes)

            def timed(tag):
                s.apgReset()
                for _ in range(4):
                    s.apgIterate(20, history=False)
                s.synchronize()
                reg = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    s.apgIterate(n_it, history=False)
                    s.synchronize()
                    reg.append(1e3 * (time.perf_counter() - t0) / n_it)
                s.apgReset()
                s.apgIterate(5, history=False)
                s.profileEnable(1); s.profileReset()
                s.apgIterate(profile_steps, history=False)
                ms, n = s.profileRead()
                cms, cn = s.profileReadCollective()
                s.profileEnable(0)
                med = float(np.median(reg))
                return {"ms_per_step": med, "ms_per_step_min": min(reg), "ms_per_step_max": max(reg), "regions": len(reg), "steps_per_region": n_it,
                        "kernel_classes_us": {"stream_gemv": 1e3 * ms[0] / max(n[0], 1), "recursion+shared_gemms(incl. exchange)": 1e3 * ms[1] / max(n[1], 1),
                                              "dual_update": 1e3 * ms[2] / max(n[2], 1), "collective_per_step": 1e3 * cms / max(profile_steps, 1)},
                        "speedup_before_wire": unsharded_ms / med}

            s.setExchangeTransport(capi.EXCHANGE_COLLECTIVE)      # both transports are timed here: nothing for the context to choose
            row["rccl_one_rank"] = timed("rccl")
            s.setExchangeTransport(capi.EXCHANGE_ONESHOT)         # (the library wires the rank's own inbox over its communicator)
            row["one_shot"] = timed("oneshot")
            best = min(row["rccl_one_rank"]["ms_per_step"], row["one_shot"]["ms_per_step"])
            row["exchange_budget_us_for_3p5x"] = 1e3 * (unsharded_ms / 3.5 - best) if W == 8 else None
        except Exception as e:   # noqa: BLE001 -- reported, never fatal for the headline
            row["error"] = "%s: %s" % (type(e).__name__, e)
        finally:
            if s is not None:
                s.close()
        rows.append(row)
    return {"what": "rank 0's shard of a W-rank partition of the bench workload through the whole sharded path on ONE GPU (one-rank RCCL communicator; "
                    "one-shot: the rank's own inbox): per-rank time without the wire and the wait for peers => ceiling of the W-GPU speed-up",
            "cut_stage": int(cut), "unsharded_ms_per_step": unsharded_ms, "shards": rows}


def quasi_newton(problem, device, iterations=40):
    """1 GPU: the global-FBE and NAMA outer loops (SmpcController::algorithmGlobalFbe / algorithmNama, SmpcController.cu:1529-1586) on
    the bench workload, fp64, dense per-node blocks and the structured operator mode: ms per iteration over `iterations`
    iterations after 3 untimed ones (one rn_algorithm_fbe_nama call: line searches, L-BFGS, read-backs included), the step sizes the
    line searches took and the library's counters."""
    from rapidnet_amd import capi, synth

    dh, ah = synth.forecast_at(problem["forecast"], 0)
    out = {"iterations": iterations, "dtype": "f64", "unit": "ms per iteration"}
    for structured in (False, True):
        for alg, key in (("globalFbeAlgorithm", "global_fbe"), ("namaAlgorithm", "nama")):
            name = "%s_%s" % (key, "structured" if structured else "dense")
            s = None
            try:
                s = capi.Solver(problem["network"], problem["tree"], problem["config"], precision="f64", device=device, structured=structured)
                s.initialiseSmpcController(dh, ah)
                s.setAlgorithm(alg, 5)
                run = s.algorithmGlobalFbe if alg == "globalFbeAlgorithm" else s.algorithmNama
                run(3)
                s.synchronize()
                t0 = time.perf_counter()
                h, v, tau = run(iterations)
                dt = time.perf_counter() - t0
                c = s.fbeCounters()
                out[name] = {"ms_per_iteration": 1e3 * dt / iterations, "tau_first": [float(x) for x in tau[:8]], "tau_mean": float(np.mean(tau)),
                             "value_first_last": [float(v[0]), float(v[-1])], "primal_inf_first_last": [float(h[0]), float(h[-1])],
                             "counters": c if isinstance(c, dict) else [int(x) for x in c]}
            except Exception as e:   # noqa: BLE001
                out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            finally:
                if s is not None:
                    s.close()
    return out


_PHASES = []       # [name, seconds since the start of this process] at every phase boundary; "phases_s" of the JSON line = the time each took


def phase(what):
    _PHASES.append([what, time.time()])


def phases_taken():
    ts = _PHASES + [["end", time.time()]]
    return [[ts[i][0], round(ts[i + 1][1] - ts[i][1], 2)] for i in range(len(ts) - 1) if ts[i + 1][1] - ts[i][1] >= 0.05]


def main():
    args = parse()
    phase("start")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))     # one child process per rank, started before anything touches the GPU
    if args.worker and os.environ.get("RAPIDNET_BENCH_FAKE_WORKER"):
        _fake_worker(args, rank)        # test double (tests/test_bench_supervisor.py); never returns
    if args.traffic_probe:              # the run the PMC counters are collected on: iterations only
        args.no_cpu_baseline, args.dense_only, args.profile_steps, args.repeats, args.other_configs, args.no_traffic = True, True, 0, 0, "", True
    # PMC figures of THIS run, keyed by what they were measured on: (workload, precision) -> (traffic dict, source dict).  A figure is only
    # ever attached to the config it was collected on (round 5 attached the fp64 traffic to the fp32 leg of the same workload).
    head_precision = args.precision or ("f32" if args.workload == "wide4096" else "f64")
    measured, mfma_structured = {}, None
    if world == 1 and args.gpus == 1 and not args.no_traffic and not args.structured and not args.force_shard and args.emulate_world == 0:
        phase("PMC passes: HBM traffic (%s %s)" % (args.workload, head_precision))
        measured[(args.workload, head_precision)] = measure_traffic(args)   # child processes; nothing in this process has touched the GPU yet
        for spec in (args.other_configs.split(",") if not args.traffic_probe else []):      # the same workload in the other precision rides along: its own passes
            w_, _, pr_ = spec.partition(":")
            if w_ == args.workload and pr_ in ("f32", "f64") and pr_ != head_precision:
                phase("PMC passes: HBM traffic (%s %s)" % (w_, pr_))
                measured[(w_, pr_)] = measure_traffic(args, precision=pr_)
        if not (args.dense_only or args.traffic_probe):
            phase("PMC pass: MFMA busy (structured)")
            mfma_structured = measure_mfma(args)
    if world > 1 and os.environ.get("RAPIDNET_BENCH_TRAFFIC_JSON"):      # rank 0's worker: what its supervisor measured before starting it
        try:
            handed = json.loads(os.environ["RAPIDNET_BENCH_TRAFFIC_JSON"])
            measured[(args.workload, head_precision)] = (handed.get("traffic") or {}, handed.get("source"))
        except ValueError:
            pass
    knobs = {}
    for kv in args.knob:
        k_, _, v_ = kv.partition("=")
        knobs[k_.strip()] = int(v_)
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was launched with WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    sharded = world > 1 or args.force_shard or args.emulate_world > 0
    # stdout carries exactly ONE line, the JSON: libraries that write to fd 1 on their own (RCCL prints a version banner
    # when a communicator is created) go to stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit_partial(obj, what):
        """rank 0's worker under a supervisor: a result line that is not complete yet, on the real stdout (the supervisor's pipe)"""
        os.write(saved_stdout, (json.dumps(dict(obj, partial=what)) + "\n").encode())
    if sharded:
        import torch
        import torch.distributed as dist

        # torch.distributed is the CONTROL plane only (rendezvous, the 128-byte ncclUniqueId, barriers, max over ranks) and
        # runs over gloo on the host: the one RCCL communicator on each device is the solver library's own
        # (rn_comm_init), whose ncclAllReduce sits on the solver's stream -- torch never creates a second one
        ndev = torch.cuda.device_count()       # counting devices does not initialise the GPU
        oversubscribed = ndev < int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if ndev < 1:
            sys.exit("bench.py: no GPU visible")
        device = local_rank % ndev               # fewer GPUs than ranks: RCCL will refuse the duplicates (see below)
        torch.cuda.set_device(device)
        if "MASTER_ADDR" not in os.environ:   # --force-shard without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: loopback (the box's hostname may not resolve)
        import datetime

        # (a peer that dies closes its sockets: gloo raises in the survivors at once; the long timeout only covers ranks that wait
        #  at the final barrier while rank 0 times the CPU baseline)
        prefix = os.environ.get("RAPIDNET_BENCH_STORE_PREFIX")
        if prefix and world > 1:
            # a rank's second worker (supervise, step 2): the launcher's store still holds the first worker's rendezvous keys
            agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
            store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, rank == 0 and not agent, datetime.timedelta(seconds=300))
            dist.init_process_group("gloo", store=dist.PrefixStore(prefix, store), rank=rank, world_size=world, timeout=datetime.timedelta(seconds=1800))
        else:
            dist.init_process_group("gloo", init_method="env://", timeout=datetime.timedelta(seconds=1800))
    else:
        device, oversubscribed = local_rank, False
    wd = Watchdog(rank, 600.0) if world > 1 else None

    def beat(what, limit=None):
        phase(what)
        if wd is not None:
            wd.beat(what, limit)

    def die(msg):
        """N > 1: leave at once with a non-zero code and without finalisers (a context whose peers are gone must not be torn down
        collectively) -- every rank reaches this through the same agreed verdict, or the launcher ends the others"""
        print("bench.py rank %d: %s" % (rank, msg), file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(1)

    from rapidnet_amd import capi, synth

    precision = args.precision or ("f32" if args.workload == "wide4096" else "f64")
    problem = synth.make_problem(args.workload)
    data_sha256 = synth.fingerprint(problem)      # of the data as generated: taken before anything (a control step advances currentX / prevU in place) touches them
    nodes_full = int(problem["tree"]["nodes"][0])
    dh, ah = synth.forecast_at(problem["forecast"], 0)
    cut_stage = -1
    tree = problem["tree"]
    debug_part = None
    if sharded:
        # the partition itself happens behind the C-ABI (rn_create_sharded); the debug modes (one rank driving the sharded code
        # path: --force-shard, --emulate-world) take the rank-local tree from the same C partitioner and set the pieces by hand
        cut_stage = capi.default_cut_stage(problem["tree"])
        if world == 1:
            debug_part = capi.partition_tree(problem["tree"], 0, max(args.emulate_world, 1), cut_stage)
            tree = debug_part["tree"]
    def fresh_uid():
        """(ncclUniqueId, error): rank 0's, over gloo.  Every context creates a communicator of its own, so every run_mode asks again."""
        box = [None, ""]
        if rank == 0 and args.exchange == "rccl":
            try:
                box[0] = capi.comm_unique_id()
            except Exception as e:   # every rank must learn about it, or the others wait in the communicator set-up
                box[1] = "rn_comm_unique_id: %s" % e
        dist.broadcast_object_list(box, src=0)
        return box[0], box[1]

    def agree(ok, err, what):
        """(everybody succeeded, first error): a MIN all-reduce over gloo -- every rank takes the same branch afterwards"""
        import torch

        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return True, None
        errs = [None] * world
        dist.all_gather_object(errs, err)
        return False, next(("rank %d: %s" % (i, e) for i, e in enumerate(errs) if e), "a peer rank failed in " + what)

    fallback_reason = [None]
    comm_ranks = [None]
    rccl_library = None
    if sharded:
        try:
            rccl_library = capi.comm_library()
        except Exception as e:
            rccl_library = "unavailable: %s" % e

    def run_mode(structured, steps, warmup, profile_steps, repeats=0, problem=problem, tree=tree, precision=precision, workload=args.workload,
                 control_step=True, cut_stage=cut_stage, fatal=True, alt=False, fused_ab=False):
        if sharded:
            import torch
        def make_local():
            if not sharded:
                return capi.Solver(problem["network"], tree, problem["config"], precision=precision, device=device, structured=structured, knobs=knobs)
            if world == 1:    # debug modes: rank 0's shard (or the whole tree) through the sharded code path with a one-rank communicator
                s_ = capi.Solver(problem["network"], tree, problem["config"], precision=precision, device=device, structured=structured, knobs=knobs)
                s_.commInit(0, 1, None)
                s_.setCutStage(cut_stage, (debug_part["momE"], debug_part["momP"]))
                s_.setExchangeTransport(capi.EXCHANGE_COLLECTIVE)      # (these timing modes say which transport they time: --one-shot switches below)
                return s_
            # the real thing: partition + cut stage + children moments in ONE call of the C-ABI -- WITHOUT a communicator yet
            s_ = capi.Solver(problem["network"], problem["tree"], problem["config"], precision=precision, device=device, structured=structured,
                             rank=rank, nranks=world, cut_stage=cut_stage, unique_id=None, knobs=knobs)
            # the headline worker runs the communicator's all-reduce and NOTHING else: fixed before rn_comm_init, so the library never
            # creates an inbox or maps a peer in this process.  The second worker (alt) leaves the default, RN_EXCHANGE_AUTO.
            if not alt:
                s_.setExchangeTransport(capi.EXCHANGE_COLLECTIVE)
            return s_

        dh, ah = synth.forecast_at(problem["forecast"], 0)
        s = None
        if sharded:
            # Two agreed steps, so that nobody ever waits inside ncclCommInitRank (blocking, no timeout) for a peer that has
            # already failed: (1) every rank creates its shard context WITHOUT a communicator and the ranks agree over gloo that
            # all of them succeeded -- otherwise everybody leaves with a non-zero code; (2) only then rn_comm_init, under the
            # watchdog, and a second agreement on its outcome.
            beat("rn_create_sharded (%s)" % workload)
            ok, err = True, ""
            try:
                s = make_local()
            except Exception as e:   # noqa: BLE001 -- reported by every rank, then the job ends
                ok, err = False, "%s: %s" % (type(e).__name__, e)
            all_ok, why = agree(ok, err, "rn_create_sharded")
            if not all_ok:
                if s is not None:
                    s.close()
                if not fatal:
                    raise AgreedFailure("a rank could not create its shard context (%s)" % why)
                die("a rank could not create its shard context, nobody starts the communicator set-up (%s)" % why)
            _fault("before_comm_init", rank)
            new_uid, uid_error = fresh_uid()
            ok, err = True, ""
            if args.exchange == "torch":
                ok, err = False, "--exchange torch"
            elif new_uid is None:
                ok, err = False, uid_error or "no unique id"
            else:
                beat("rn_comm_init / ncclCommInitRank (%s)" % workload, 180.0)
                try:
                    s.commInit(rank if world > 1 else 0, world, new_uid)
                except capi.RapidNetError as e:
                    ok, err = False, str(e)
                beat("communicator ready", 600.0)
            all_ok, why = agree(ok, err, "rn_comm_init")
            if not all_ok:
                fallback_reason[0] = why
                if not (args.exchange == "torch" or (oversubscribed and args.allow_oversubscribe)):
                    if not fatal:
                        s.close()
                        raise AgreedFailure("RCCL could not create the communicator (%s)" % why)
                    die("RCCL could not create the communicator (%s); one rank per GPU is required (--allow-oversubscribe rehearses the "
                        "launcher path over gloo on a box with fewer GPUs)" % why)
                if ok and args.exchange != "torch":   # a communicator its peers do not have is of no use: start over without one
                    s.close()
                    s = make_local()
            comm_ranks[0] = s.shardInfo()["comm_ranks"]
            if world == 1 and args.one_shot:
                try:
                    s.setExchangeTransport(capi.EXCHANGE_ONESHOT)     # a one-rank communicator: the library has wired the rank's own inbox
                except capi.RapidNetError:
                    s.peerInboxConnect([s.peerInboxCreate()])         # no communicator: by hand
                    s.setExchangeTransport(capi.EXCHANGE_ONESHOT)
        else:
            s = make_local()
        beat("factor step + affine terms (%s)" % workload)
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        tune_info = None
        if alt and sharded and world > 1 and not fallback_reason[0]:
            # every rank the same sequence of gloo collectives whatever happens to it (a one-shot reader's time-out raises on every rank of the batch)
            beat("rn_exchange_autotune (%s)" % workload, 300.0)
            ok, err = True, ""
            try:
                tune_info = s.exchangeAutotune(args.tune_iterations)
            except capi.RapidNetError as e:
                ok, err = False, str(e)
            all_ok, why = agree(ok, err, "rn_exchange_autotune")
            if not all_ok:
                s.close()
                raise AgreedFailure("rn_exchange_autotune failed (%s)" % why)
            s.apgReset()
        theta = [1.0, 1.0]
        n_cut = 0
        if sharded and fallback_reason[0]:
            nps = problem["tree"]["nodesPerStage"]
            n_cut = int(nps[cut_stage - 1]) * (s.nv + 2 * s.nx)

        def iterate(n):
            """n APG iterations.  Normal path: ONE library call, no host sync inside.  Fallback: the protected step methods
            with the cut payload all-reduced through torch.distributed (exact unless the soft-constraint branch trips)."""
            if not (sharded and fallback_reason[0]):
                s.apgIterate(n, history=False)
                return
            for _ in range(n):
                lam = theta[1] * (1.0 / theta[0] - 1.0)
                theta[0], theta[1] = theta[1], 0.5 * (np.sqrt(theta[1] ** 4 + 4 * theta[1] ** 2) - theta[1] ** 2)
                s.dualExtrapolationStep(lam)
                s.debugSweepPhase(1)
                payload = torch.from_numpy(s.debugCutBuffer(n_cut))
                dist.all_reduce(payload)
                s.debugCutBuffer(n_cut, payload.numpy())
                s.debugSweepPhase(2)
                s.proximalFunG(); s.computeFixedPointResidual(); s.dualUpdate()

        def barrier():
            s.synchronize()
            if dist is not None:
                import torch

                dist.barrier()
                torch.cuda.synchronize()

        # clock ramp (untimed, before the contract's W warm-up steps): a GPU that has idled through the host-side set-up (or the
        # CPU baseline of the previous config) needs tens of milliseconds of work to reach its clocks -- the first timed region of a
        # 0.1 ms-per-step config otherwise measures the ramp (seen: 0.28 ms per step in region 1, 0.098 in regions 2-5)
        # (batches of 20: the kernels of the timed region -- shorter batches take the exact bookkeeping path)
        if sharded:   # every rank must issue the same number of collectives: a fixed count, never a time-based loop
            for _ in range(4):
                iterate(20)
            s.synchronize()
        elif not args.traffic_probe:
            t_ramp = time.perf_counter()
            while time.perf_counter() - t_ramp < 0.08:
                iterate(20)
                s.synchronize()
        beat("warm-up (%s)" % workload)
        iterate(warmup)
        barrier()
        beat("timed region (%s)" % workload)
        t0 = time.perf_counter()
        iterate(steps)
        s.synchronize()
        dt_own = time.perf_counter() - t0      # this rank's own time: enqueue + its GPU work incl. the waits inside the collectives
        barrier()
        dt = time.perf_counter() - t0
        per_rank = None
        if dist is not None and world > 1:     # who was slow, and how uneven the shards are
            rows = [None] * world
            dist.all_gather_object(rows, (int(s.nodes), 1e3 * dt_own / steps))
            per_rank = {"local_nodes": [r[0] for r in rows], "local_nodes_min": min(r[0] for r in rows), "local_nodes_max": max(r[0] for r in rows),
                        "ms_per_step_own": [r[1] for r in rows], "ms_per_step_own_min": min(r[1] for r in rows), "ms_per_step_own_max": max(r[1] for r in rows),
                        "note": "own = K steps enqueued and synchronised on the rank's stream, before the closing barrier; the collectives inside make the ranks wait for each other, "
                                "so the spread shows launch / clock skew, not the shards' work -- see kernel_classes for that"}

        def max_over_ranks(v):
            if dist is None:
                return v
            import torch

            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        dt = max_over_ranks(dt)
        # the contract's timed region is the one above; `repeats` further regions of the same K steps give its spread
        rep = []
        for _ in range(max(0, repeats)):
            beat("repeat regions (%s)" % workload)
            barrier()
            t0 = time.perf_counter()
            iterate(steps)
            barrier()
            rep.append(max_over_ranks(time.perf_counter() - t0))
        spread = None
        if rep:
            allr = sorted([dt] + rep)
            spread = {"regions": len(allr), "steps_per_region": steps, "ms_per_step_median": 1e3 * float(np.median(allr)) / steps,
                      "ms_per_step_min": 1e3 * allr[0] / steps, "ms_per_step_max": 1e3 * allr[-1] / steps,
                      "value_median": steps / float(np.median(allr)), "note": "region 1 is the contract's timed region (`value`); all regions max over ranks"}
        # the other form of the forward walk + dual update in the SAME context (same buffers: two contexts of one process differ by up to 5 %
        # through the placement of their buffers alone).  Since round 6 the one launch (k_down_chain_dual) is the library's default (rn_set_fused_walk_dual,
        # -1), shaped by the tree: one workgroup per chain where the chains fill the chip -- this workload --, up to four otherwise.  Timed here:
        # the two-launch form forced, `repeats` regions, then back to the default.
        fused = None
        if fused_ab and not sharded and rep:
            try:
                s.setFusedWalkDual(0)
                iterate(40)
                frep = []
                for _ in range(len(rep)):
                    barrier()
                    t0 = time.perf_counter()
                    iterate(steps)
                    barrier()
                    frep.append(time.perf_counter() - t0)
                s.setFusedWalkDual(-1)
                iterate(40)
                s.synchronize()
                tm, um = float(np.median(frep)), float(np.median(rep))
                fused = {"default": "one launch (k_down_chain_dual), workgroups per chain by shape", "value": steps / um, "ms_per_step": 1e3 * um / steps, "regions": len(rep),
                         "two_launches_same_context": {"value": steps / tm, "ms_per_step": 1e3 * tm / steps, "ms_per_step_min": 1e3 * min(frep) / steps,
                                                       "ms_per_step_max": 1e3 * max(frep) / steps, "regions": len(frep)},
                         "speedup": tm / um,
                         "what": "rn_set_fused_walk_dual: k_down_chain + k_dual_stage as one launch (k_down_chain_dual, Hx kept in LDS) against the two launches, identical "
                                 "iterates, same context, interleaved with the headline's regions; the per-launch profiling pass (roofline, kernel_classes) always runs "
                                 "the two-launch form, so the dual update's figures there are those of the kernel north_star names"}
            except Exception as e:   # noqa: BLE001 -- reported, never fatal for the headline
                fused = {"error": "%s: %s" % (type(e).__name__, e)}
        # one whole control step (SmpcController::controlAction: state upload, affine terms, 500 iterations, u0 back)
        ctrl_ms = None
        if not sharded and control_step:
            s.controlAction(dh, ah, maxIterations=5)
            t1 = time.perf_counter()
            s.controlAction(dh, ah, maxIterations=500)
            ctrl_ms = 1e3 * (time.perf_counter() - t1)
        # per-launch hipEvent pass on the solver's own stream
        roofline, classes = None, {}
        if profile_steps > 0:
            beat("per-launch hipEvent pass (%s)" % workload)
            s.apgReset()
            s.apgIterate(5, history=False)
            s.profileEnable(1)
            s.profileReset()
            s.apgIterate(profile_steps, history=False)
            ms, n = s.profileRead()
            s.profileEnable(0)
            bwd_bytes, dual_bytes = s.algorithmicBytes()
            # what a do-nothing streaming kernel reaches on this very device (practical denominator beside the 8 TB/s spec)
            try:
                read_ceiling, copy_ceiling = s.measureHbm(2 << 30, 3)
            except capi.RapidNetError:
                read_ceiling, copy_ceiling = None, None
            names = ("stream_gemv" if not structured else "struct_prep+gemm_m2", "recursion+shared_gemms", "dual_update", "bookkeeping")
            for i, nm in enumerate(names):
                classes[nm] = {"ms_total": float(ms[i]), "launches": int(n[i]), "avg_us": float(1e3 * ms[i] / max(n[i], 1))}
            if sharded:   # the all-reduces, bracketed by hipEvents of their own on the solver's stream (they lie INSIDE recursion+shared_gemms)
                cms, cn = s.profileReadCollective()
                classes["collective"] = {"ms_total": float(cms), "launches": int(cn), "avg_us": float(1e3 * cms / max(cn, 1)),
                                         "per_step_us": float(1e3 * cms / max(profile_steps, 1)),
                                         "note": "hipEvents around every all-reduce (cut payload once per iteration; dist tail and verdict + history once per batch): "
                                                 "wire latency plus the wait for the slowest peer; contained in recursion+shared_gemms"}
            kinfo = s.kernelInfo()
            dual_kernel = "k_dual_stage" if kinfo["dual_stage"] else "k_dual_fused"
            dual_s = 1e-3 * ms[2] / max(n[2], 1)
            dual = {"kernel": dual_kernel, "workgroups": kinfo["dual_blocks"], "vectors_per_thread": kinfo["dual_trips"], "achieved": dual_bytes / dual_s / 1e9 if dual_s > 0 else 0.0, "algorithmic_bytes_per_launch": dual_bytes,
                    "avg_launch_us": 1e6 * dual_s}
            dual["frac"] = dual["achieved"] / 8000.0
            dual["traffic"] = None   # filled below once the source of the counters is known
            # HBM traffic from the PMC counters: measured by this run's own rocprofv3 child passes (measure_traffic, started before
            # the GPU was touched); if that was not possible, carried over from profiles/traffic.json, and only if that file was
            # collected on this workload with exactly the kernel sources this run executes; traffic_source says which
            traffic, traffic_source = {}, {"measured_in_this_run": False, "file": None}
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            measured_traffic, measured_source = measured.get((workload, precision), ({}, None))
            if measured_traffic and (not sharded or (world > 1 and rank == 0)):   # N > 1: rank 0's shard, measured by its supervisor
                traffic, traffic_source = measured_traffic, measured_source
            elif os.path.exists(tpath) and workload == "barcelona493" and precision == "f64" and not sharded:
                try:
                    from rapidnet_amd import build as _b

                    t = json.load(open(tpath))
                    cur = _b.kernel_sources_sha256()
                    traffic_source = {"measured_in_this_run": False, "file": "profiles/traffic.json", "collected_at_commit": t.get("collected_at_commit"),
                                      "kernels_sha256": t.get("kernels_sha256"), "matches_current_kernels": t.get("kernels_sha256") == cur,
                                      "how": t.get("source")}
                    if t.get("kernels_sha256") == cur:
                        traffic = t
                    if measured_source and measured_source.get("why_not"):
                        traffic_source["not_measured_in_this_run_because"] = measured_source["why_not"]
                except Exception:
                    traffic = {}
            if not traffic and measured_source and measured_source.get("why_not"):
                traffic_source.setdefault("why_not", measured_source["why_not"])
            dual["traffic"] = traffic.get("k_dual_stage_bytes_per_launch" if kinfo["dual_stage"] else "k_dual_fused_bytes_per_launch")
            if structured:   # no streaming kernel: the fused dual update is the dominant (HBM-bound) kernel
                roofline = {"kernel": dual_kernel, "bound": "hbm", "achieved": dual["achieved"], "peak": 8000.0, "unit": "GB/s",
                            "frac": dual["frac"], "traffic": traffic.get("k_dual_stage_bytes_per_launch" if kinfo["dual_stage"] else "k_dual_fused_bytes_per_launch"), "traffic_source": traffic_source,
                            "algorithmic_bytes_per_launch": dual_bytes, "avg_launch_us": dual["avg_launch_us"], "launches_per_step": 1}
                if copy_ceiling:
                    roofline.update({"measured_copy_ceiling": copy_ceiling, "frac_of_measured_ceiling": dual["achieved"] / copy_ceiling})
            else:
                avg_s = 1e-3 * ms[0] / max(n[0], 1)
                achieved = bwd_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
                roofline = {"kernel": "k_stream_gemv", "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                            "frac": achieved / 8000.0, "traffic": traffic.get("k_stream_gemv_bytes_per_launch"), "traffic_source": traffic_source,
                            "algorithmic_bytes_per_launch": bwd_bytes, "avg_launch_us": 1e6 * avg_s, "launches_per_step": 1,
                            "dual_update": dual}
                # the kernel north_star's 60 % names, as FLAT keys (a record that keeps scalars only keeps these)
                roofline.update({"dual_update_frac": dual["frac"], "dual_update_us": dual["avg_launch_us"], "dual_update_traffic": dual["traffic"],
                                 "dual_update_bytes": dual_bytes, "dual_update_kernel": dual_kernel})
                if read_ceiling:   # read-only stream vs a read-only probe; the dual update (5 read + 2 write streams) vs the copy probe
                    roofline.update({"measured_read_ceiling": read_ceiling, "frac_of_measured_ceiling": achieved / read_ceiling})
                    dual.update({"measured_copy_ceiling": copy_ceiling, "frac_of_measured_ceiling": dual["achieved"] / copy_ceiling})
        alt_res = None
        if alt and sharded and world > 1:
            # the second worker of a rank: this context was left on RN_EXCHANGE_AUTO and tuned before its warm-up (tune_info); everything above --
            # warm-up, the contract's timed region, the repeats -- ran on the transport the ranks agreed on
            if fallback_reason[0]:
                alt_res = {"error": "not run: no RCCL communicator in this run (%s)" % fallback_reason[0]}
            elif tune_info is None or "error" in tune_info:
                alt_res = {"error": (tune_info or {}).get("error", "rn_exchange_autotune did not run")}
            else:
                alt_res = {"kind": "RN_EXCHANGE_AUTO: rn_exchange_autotune timed the communicator's all-reduce and the one-shot peer-write exchange on this context's own "
                                   "iterations (%d per candidate, max over ranks) and kept the faster; the timed region ran on it" % tune_info["iterations"],
                           "chosen": "one-shot" if tune_info["transport"] == 1 else "collective", "tune": tune_info,
                           "rccl_hints": {k: os.environ[k] for k in RCCL_HINTS if k in os.environ},
                           "value": steps / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / steps, "timing_spread": spread, "per_rank": per_rank}
        batch_counters = s.counters()   # optimistic / exact batches of rn_apg_iterate, replays (0 unless a soft constraint tripped)
        res = {"value": steps / dt, "ms_per_step": 1e3 * dt / steps, "spread": spread, "nodes": s.nodes, "per_rank": per_rank, "ms_per_controlStep_500it_derived": 500 * 1e3 * dt / steps, "batch_counters": batch_counters,
               "ms_per_controlStep_500it_measured": ctrl_ms, "roofline": roofline, "kernel_classes": classes,
               "dims": (s.nx, s.nu, s.nv, s.nd, s.N), "alt_exchange": alt_res, "fused_walk_dual": fused}
        s.close()
        return res

    if args.alt_exchange_only:      # the second worker of a rank (supervise): the context chooses its exchange itself, then the contract's timed region on it
        res = {"error": "not run"}
        try:
            res = run_mode(False, args.steps, args.warmup, 0, repeats=min(args.repeats, 3), control_step=False, fatal=False, alt=True)["alt_exchange"]
        except AgreedFailure as e:
            res = {"error": "AgreedFailure: %s" % e}
        if rank == 0:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            print(json.dumps({"alt_exchange": res}), flush=True)
            os.dup2(2, 1)
        dist.barrier()
        return
    # headline: the reference's storage model (dense per-node blocks); the structured mode is reported beside it on 1 GPU
    dense = None if args.structured else run_mode(False, args.steps, args.warmup, args.profile_steps, repeats=args.repeats,
                                                  fused_ab=not (args.dense_only or args.traffic_probe) and args.workload == "barcelona493")
    struct, struct_error = None, None
    if args.structured:
        struct = run_mode(True, args.steps, args.warmup, args.profile_steps, repeats=args.repeats)
    elif not sharded and not args.dense_only:
        try:      # the opt-in mode rides along: a failure there is reported in the line, it does not take the headline with it
            # (the headline workload: with the repeat regions and the same-context A/B of the fused walk + dual update, as in the dense run)
            sab = not args.traffic_probe and args.workload == "barcelona493"
            struct = run_mode(True, args.steps, args.warmup, args.profile_steps, repeats=args.repeats if sab else 0, fused_ab=sab)
        except Exception as e:   # noqa: BLE001
            struct_error = "%s: %s" % (type(e).__name__, e)
    head = struct if args.structured else dense
    dt = args.steps / head["value"]
    roofline, classes = head["roofline"], head["kernel_classes"]
    nx, nu, nv, nd, N = head["dims"]

    class _S:   # dims for the config string below
        pass
    s = _S()
    s.nx, s.nu, s.nv, s.nd, s.N = nx, nu, nv, nd, N

    # The other configurations of BASELINE.json, timed in the same run with the same protocol (W warm-up steps, K timed steps, the
    # per-launch hipEvent pass), each with its own roofline object and CPU leg.  One GPU: barcelona31 (fp64) and wide4096 (fp32).
    # N > 1: configs[4] -- the wide network, fp32, sharded over the N GPUs like the headline tree (every rank runs it: collectives).
    others = []
    if not args.structured and args.workload == "barcelona493" and not args.traffic_probe:
        others = [w for w in args.other_configs.split(",") if w and w != args.workload and w != "%s:%s" % (args.workload, precision)]
        if sharded:
            others = [w for w in others if w == "wide4096"] if world > 1 else []
    out = None
    if rank == 0:
        out = {
            "metric": "apg_iterations_per_sec", "value": args.steps / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if precision == "f64" else "f32", "data": "synthetic",
            "config": {"workload": "%s: nx=%d nu=%d nv=%d nd=%d N=%d K=%d nodes=%d" % (
                args.workload, s.nx, s.nu, s.nv, s.nd, s.N, int(problem["tree"]["K"][0]), nodes_full),
                "operator_storage": "structured (shared operators, no per-node blocks)" if args.structured else "dense per-node blocks (reference storage model)",
                # which data were solved: a version tag of the generator and a fingerprint of the numbers themselves.  Version f1
                # (round 3 on) re-centres the control bounds of the BASELINE workloads for feasibility: same
                # dimensions, operators, tree and step size -- the kernels do the same work -- but not the iterates of rounds 1-2
                "data_version": synth.data_tag(args.workload), "data_sha256": data_sha256,
                "ms_per_controlStep_500it": head["ms_per_controlStep_500it_derived"],
                "ms_per_controlStep_500it_measured": head["ms_per_controlStep_500it_measured"],
                "parallelism": "1 GPU" if not sharded else ("subtree sharding below stage %d, 1 RCCL all-reduce/iteration" % cut_stage if not fallback_reason[0]
                                                            else "subtree sharding below stage %d, FALLBACK exchange through torch.distributed, step-wise (%s)" % (cut_stage, fallback_reason[0]))},
            "local_nodes": int(head["nodes"]),
            "timing_spread": head["spread"],
            "rccl": None if not sharded else {"ranks": world, "ranks_seen_by_rccl": comm_ranks[0], "library": rccl_library, "communicator": "one per device, owned by librapidnet_hip (rn_comm_init); "
                                              "ncclUniqueId and barriers travel over torch.distributed/gloo",
                                              "exchange": "torch.distributed fallback: " + fallback_reason[0] if fallback_reason[0] else "ncclAllReduce on the solver's stream"},
            "roofline": roofline, "kernel_classes": classes, "batch_counters": head["batch_counters"],
        }
        if knobs:
            out["knobs"] = knobs
        if sharded and isinstance(classes.get("collective"), dict):
            # N > 1: the per-iteration exchange between hipEvents on the solver's stream (the collective of one iteration; wire + wait for the slowest peer)
            out["exchange_us"] = classes["collective"].get("per_step_us")
        if not args.structured and struct is not None:   # the exact shared-operator reformulation (RN_OPS_STRUCTURED), same workload, same iterates
            out["structured_mode"] = {k: struct[k] for k in ("value", "ms_per_step", "ms_per_controlStep_500it_derived",
                                                               "ms_per_controlStep_500it_measured", "roofline", "kernel_classes")}
            out["structured_mode"]["operator_storage"] = ("none: shared-operator MFMA GEMMs -- what RN_OPS_AUTO, the default of the C-ABI and of the C++ class surface, runs "
                                                          "(the headline `value` stays on the dense blocks: the reference's storage model and the roofline this record tracks)")
            if struct.get("spread"):
                out["structured_mode"]["timing_spread"] = struct["spread"]
            if mfma_structured is not None:      # matrix-unit use of the shared-operator products from this run's own SQ counters
                out["structured_mode"]["mfma"] = mfma_structured
                if isinstance(mfma_structured.get("busy_frac"), float):
                    out["structured_mode"]["mfma_busy_frac"] = mfma_structured["busy_frac"]
            if struct.get("fused_walk_dual"):      # rn_set_fused_walk_dual in the structured context (same-context A/B, as for the dense headline)
                out["structured_mode"]["fused_walk_dual"] = {k: v for k, v in struct["fused_walk_dual"].items() if k != "what"}
        if struct_error:
            out["structured_mode"] = {"error": struct_error}
        if dense is not None and dense.get("fused_walk_dual"):
            out["fused_walk_dual"] = dense["fused_walk_dual"]
        if head.get("per_rank"):
            out["per_rank"] = head["per_rank"]
        if args.worker:   # under a supervisor (N > 1): the headline is on record from here on -- a first, partial line; the supervisor keeps
            # the most complete one it has seen and prints ONE (supervise): nothing that follows can take the headline with it
            emit_partial(out, "headline measured; secondary configs (wide4096), the one-shot exchange pass and the CPU baseline pending")
    entries = []
    for spec in others:
        w, _, prec_req = spec.partition(":")
        prec_w = prec_req if prec_req in ("f32", "f64") else ("f32" if w.startswith("wide") else "f64")
        entry = {"workload": w, "dtype": prec_w}
        try:
            beat("problem data (%s)" % w)
            pw = synth.make_problem(w)      # (always fresh: the headline's control steps have advanced currentX / prevU of `problem` in place)
            cut_w = capi.default_cut_stage(pw["tree"]) if sharded else -1
            r = run_mode(False, args.steps, args.warmup, args.profile_steps, repeats=min(args.repeats, 4), problem=pw, tree=pw["tree"],
                         precision=prec_w, workload=w, control_step=False, cut_stage=cut_w, fatal=False)
            nxw, nuw, nvw, ndw, Nw = r["dims"]
            entry.update({"config": "%s: nx=%d nu=%d nv=%d nd=%d N=%d K=%d nodes=%d" % (w, nxw, nuw, nvw, ndw, Nw, int(pw["tree"]["K"][0]), int(pw["tree"]["nodes"][0])),
                          "data_version": synth.data_tag(w),
                          "value": r["value"], "unit": "iterations/s", "ms_per_step": r["ms_per_step"], "steps": args.steps, "warmup": args.warmup,
                          "timing_spread": r["spread"], "roofline": r["roofline"], "kernel_classes": r["kernel_classes"]})
            if sharded:
                entry.update({"n_gpus": world, "local_nodes": int(r["nodes"]), "per_rank": r["per_rank"],
                              "parallelism": "subtree sharding below stage %d, 1 RCCL all-reduce/iteration" % cut_w if not fallback_reason[0]
                              else "subtree sharding below stage %d, FALLBACK exchange through torch.distributed (%s)" % (cut_w, fallback_reason[0])})
            if not args.no_cpu_baseline:
                beat("CPU baseline (%s)%s" % (w, "" if rank == 0 else ": waiting for rank 0"), 1300.0)   # every rank: the others wait in the next collective
            if rank == 0 and not args.no_cpu_baseline:
                entry["cpu_baseline"] = cpu_baseline(w, pw, prec_w, min(args.cpu_iterations, 8), sample_levels=1)
            del pw
        except AgreedFailure as e:   # every rank raised it together: reported, the run goes on
            entry["error"] = "AgreedFailure: %s" % e
        except Exception as e:   # a config that does not fit this device / host is reported, not fatal for the headline
            if world > 1:        # ... unless other ranks are inside collectives this rank will never join
                die("secondary config %s failed on this rank only (%s: %s)" % (w, type(e).__name__, e))
            entry["error"] = "%s: %s" % (type(e).__name__, e)
        entries.append(entry)
    # The replay path, timed: the feasible-by-construction workloads never trip the soft-constraint thresholds, so their batches
    # always take the optimistic path once.  Here the ORIGINAL data of the 31-scenario tree (random bounds: infeasible) with small
    # penalties -- the tree-global distances exceed gamma / lambda -- run one optimistic batch (checkpoint, 20 iterations with the
    # prox as a pure projection, verdict, restore, 20 exact iterations) and then one batch of the back-off (exact path only).
    replay = None
    if not sharded and not args.structured and args.workload == "barcelona493" and not args.traffic_probe and not args.dense_only:
        try:
            beat("replay path")
            pr = synth.make_problem("barcelona31_infeasible", penalty_x=20.0, penalty_xs=5.0)
            sr = capi.Solver(pr["network"], pr["tree"], pr["config"], precision="f64", device=device)
            sr.initialiseSmpcController(*synth.forecast_at(pr["forecast"], 0))
            # batches of 20 after a reset; each is labelled by what the counters say it was: a plain optimistic batch (the first
            # iterations from zero duals stay inside the thresholds), the batch that tripped and was replayed, a batch of the back-off
            kinds = {}
            for rep in range(3):          # the first round includes first-launch costs; the last is reported
                sr.apgReset(); sr.setExchangeMode(1); sr.synchronize()
                for _ in range(4):
                    c0 = sr.counters()
                    t0 = time.perf_counter(); sr.apgIterate(20, history=False); sr.synchronize(); dtb = time.perf_counter() - t0
                    c1 = sr.counters()
                    kind = "replayed" if c1["replayed"] > c0["replayed"] else ("optimistic" if c1["optimistic"] > c0["optimistic"] else "exact_back_off")
                    kinds[kind] = 1e3 * dtb / 20
            replay = {"workload": "barcelona31_infeasible, penaltyStateX 20, penaltySafetyX 5 (the soft-constraint thresholds trip)", "data_version": synth.data_tag("barcelona31_infeasible"),
                      "batch": 20, "ms_per_step_by_batch_kind": kinds, "batch_counters": sr.counters(),
                      "note": "optimistic = checkpoint + 20 iterations with the prox as a pure projection + verdict; replayed = the same, then restore + 20 exact iterations; "
                              "exact_back_off = the next 8 batches go straight through the exact path"}
            sr.close()
        except Exception as e:   # noqa: BLE001 -- reported, never fatal for the headline
            replay = {"error": "%s: %s" % (type(e).__name__, e)}
    # 1 GPU, default run: the one-GPU half of the multi-GPU story and the quasi-Newton loops, timed by THIS run (round 5; they were
    # builder-run files under profiles/ before)
    ceiling, qn = None, None
    if not sharded and not args.structured and args.workload == "barcelona493" and precision == "f64" and not args.traffic_probe and not args.dense_only:
        unsharded_ms = float(head["spread"]["ms_per_step_median"]) if head.get("spread") else 1e3 * dt / args.steps
        if not args.no_shard_ceiling:
            try:
                ceiling = shard_ceiling(problem, device, args.steps, unsharded_ms, profile_steps=max(args.profile_steps, 20))
            except Exception as e:   # noqa: BLE001 -- reported, never fatal for the headline
                ceiling = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_quasi_newton:
            try:
                qn = quasi_newton(problem, device)
            except Exception as e:   # noqa: BLE001
                qn = {"error": "%s: %s" % (type(e).__name__, e)}
    same_box = None
    if sharded and world > 1 and not args.structured and not args.traffic_probe:
        # the ceiling of THIS box: rank 0 runs its own shard once more with a one-rank communicator (every launch and collective of the sharded path,
        # no wire, no peer to wait for) while the other ranks wait at the barrier below; speedup_vs_shard_ceiling = that time / the measured time
        beat("shard ceiling on this box (rank 0)", 300.0)
        if rank == 0:
            sc_s = None
            try:
                part0 = capi.partition_tree(problem["tree"], 0, world, cut_stage)
                sc_s = capi.Solver(problem["network"], part0["tree"], problem["config"], precision=precision, device=device, knobs=knobs)
                sc_s.setExchangeTransport(capi.EXCHANGE_COLLECTIVE)
                sc_s.commInit(0, 1, capi.comm_unique_id())
                sc_s.setCutStage(cut_stage, (part0["momE"], part0["momP"]))
                sc_s.initialiseSmpcController(dh, ah)
                sc_s.apgReset()
                for _ in range(4):
                    sc_s.apgIterate(20, history=False)
                sc_s.synchronize()
                regs = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    sc_s.apgIterate(args.steps, history=False)
                    sc_s.synchronize()
                    regs.append(1e3 * (time.perf_counter() - t0) / args.steps)
                same_box = {"world": world, "local_nodes": int(sc_s.nodes), "ms_per_step": float(np.median(regs)), "ms_per_step_min": min(regs), "ms_per_step_max": max(regs),
                            "what": "rank 0's shard through the whole sharded path with a one-rank RCCL communicator on rank 0's own GPU, right after the headline: "
                                    "a rank's time without the wire and without peers to wait for"}
            except Exception as e:   # noqa: BLE001 -- reported, never fatal
                same_box = {"error": "%s: %s" % (type(e).__name__, e)}
            finally:
                if sc_s is not None:
                    sc_s.close()
        dist.barrier()
    beat("closing: rank 0's CPU baseline, then the last barrier", 1300.0)      # every rank the same limit (rank 0 times the CPU legs meanwhile)
    if rank == 0:
        if replay is not None:
            out["replay_path"] = replay
        if entries:
            out["configs"] = entries
        if ceiling is not None:
            out["shard_ceiling"] = ceiling
        if same_box is not None:
            out["shard_ceiling_same_box"] = same_box
            if same_box.get("ms_per_step"):
                out["speedup_vs_shard_ceiling"] = same_box["ms_per_step"] / out["ms_per_step"]      # 1.0 = the exchange and the peers cost nothing
        if qn is not None:
            out["quasi_newton"] = qn
        if args.worker:
            emit_partial(out, "headline and secondary configs measured; CPU baseline pending")
        if not args.no_cpu_baseline and not args.traffic_probe and (not sharded or world > 1):
            # rank 0 only, after every timed region (the other ranks wait at the closing barrier)
            beat("CPU baseline (%s)" % args.workload, 1300.0)
            out["cpu_baseline"] = cpu_baseline(args.workload, problem, precision, args.cpu_iterations)
        out["phases_s"] = phases_taken()      # where this run's wall-clock went (>= 0.05 s each), for whoever budgets it
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        if "--worker" not in sys.argv[1:]:
            supervise(parse())     # never returns
        # a rank that fails leaves AT ONCE with a non-zero code and without finalisers (its peers may be inside a collective it
        # will never join; tearing down a communicator collectively would hang as well) -- the launcher then ends the others
        try:
            main()
        except SystemExit as e:
            code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
            if isinstance(e.code, str):
                print(e.code, file=sys.stderr)
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(code)
        except BaseException:   # noqa: BLE001
            import traceback

            traceback.print_exc()
            sys.stderr.flush()
            os._exit(1)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    main()
