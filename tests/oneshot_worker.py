"""Worker of tests/test_gpu_oneshot.py -- the one-shot exchange at the cut (rn_set_exchange_transport(ctx, 1)).

  inprocess <problem> <world> <cut> [structured] [f32]
      `world` shard contexts of ONE process, one host thread each, inboxes wired through rn_debug_peer_inbox_connect_local;
      run in a process of its own because the ranks' streams must sit on different hardware queues (a crown kernel that
      waits for a peer's packets must not block that peer's kernels behind it in the same queue): GPU_MAX_HW_QUEUES is set
      by the parent.  Compared with the stand-in transport (bitwise) and with the CPU oracle.
  ipc <problem> <cut>
      one rank of a 2-process run under torch.distributed.run on one GPU: the inboxes travel as hipIpcMemHandle_t over
      gloo (the path `bench.py --gpus N` takes on a multi-GPU node); the per-batch collectives go through an all-reduce
      callback that uses gloo.  Rank 0 compares the reassembled iterates with the oracle.
"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from oracle.oracle import Oracle  # noqa: E402
from rapidnet_amd import capi, partition, synth  # noqa: E402

VECS = ((capi.BUF_X, "x", "nx"), (capi.BUF_U, "u", "nu"), (capi.BUF_V, "v", "nv"), (capi.BUF_UPD_XI, "updXi", "2nx"), (capi.BUF_UPD_PSI, "updPsi", "nu"),
        (capi.BUF_DUAL_XI, "dualXi", "2nx"), (capi.BUF_RES_PSI, "resPsi", "nu"))


def relmax(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    assert a.shape == b.shape and np.isfinite(a).all()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def inprocess(name, world, cut, structured, precision, kw):
    p = synth.make_problem(name, **kw)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"], precision=precision)
    o.initialise(dh, ah)
    ohist = o.apg(44)
    tol = 1e-9 if precision == "f64" else 2e-4
    results = {}
    for transport in (0, 1):
        group = capi.local_group_create(world)
        shards = []
        for r in range(world):
            s = capi.Solver(p["network"], p["tree"], p["config"], rank=r, nranks=world, cut_stage=cut, structured=structured, precision=precision)
            s.joinLocalGroup(group, r)
            shards.append(s)
        if transport == 1:
            for s in shards:
                s.peerInboxCreate()
            capi.peer_inbox_connect_local(shards)
            for s in shards:
                s.setExchangeTransport(1)
                if "wrap" in sys.argv[5:]:
                    s.debugPeerSeq(0xFFFFFFF0)      # the 44 exchanges of this run walk the 32-bit sequence tag across its wrap (... ffffffff, 2, 3 ...)
        out, errs = [None] * world, []

        def work(i):
            try:
                s = shards[i]
                s.initialiseSmpcController(dh, ah)
                s.apgReset()
                h = [s.apgIterate(20), s.apgIterate(4), s.apgIterate(20)]     # optimistic, exact (short batch: no -- sharded batches are always optimistic), optimistic
                out[i] = (np.concatenate(h), s.counters())
            except Exception as e:   # noqa: BLE001
                errs.append((i, e))

        ts = [threading.Thread(target=work, args=(i,)) for i in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        d = {"nx": shards[0].nx, "nu": shards[0].nu, "nv": shards[0].nv, "2nx": 2 * shards[0].nx}
        nodes = shards[0].full_nodes
        got = {nm: partition.scatter_to_global([s.get(bid) for s in shards], [s.global_nodes for s in shards], nodes, d[dm]) for bid, nm, dm in VECS}
        results[transport] = (got, [x[0] for x in out], [x[1] for x in out])
        for nm in got:
            assert relmax(got[nm], o.get(nm)) < tol, (transport, nm, relmax(got[nm], o.get(nm)))
        for h, _ in out:
            assert np.abs(h - ohist).max() <= tol * np.abs(ohist).max(), transport
        for s in shards:
            s.close()
        capi.local_group_destroy(group)
    # the same sums in the same (rank) order: the one-shot transport reproduces the stand-in's bits
    for nm in results[0][0]:
        assert np.array_equal(results[0][0][nm], results[1][0][nm]), nm
    assert all(np.array_equal(a, b) for a, b in zip(results[0][1], results[1][1]))
    assert results[0][2] == results[1][2], (results[0][2], results[1][2])
    print("oneshot inprocess ok: %s world %d cut %d structured %d %s, batches %s" % (name, world, cut, structured, precision, results[1][2][0]), flush=True)


def auto_case(name, world, cut):
    """RN_EXCHANGE_AUTO (the default of a sharded context): `world` shard contexts of one process with wired inboxes.
    (A) nobody says anything: the first device-resident batch times both transports on the context's own iterations, the ranks agree on
        one, and neither the iterates nor the batch counters show that it happened -- bitwise the run with the transport fixed;
    (B) the selection: ONE rank's one-shot time is biased by +/- 10 s per iteration (rn_debug_set_knob): every rank must take the same
        decision, and it must be the one the MAX over the ranks' times says -- collective with the slow rank, one-shot with the fast one."""
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(dh, ah)
    ohist = o.apg(44)

    def ranks(fixed=None, bias=None):
        group = capi.local_group_create(world)
        shards = []
        for r in range(world):
            kn = {"tune_bias_us": bias} if (bias is not None and r == 1) else None
            s = capi.Solver(p["network"], p["tree"], p["config"], rank=r, nranks=world, cut_stage=cut, knobs=kn)
            s.joinLocalGroup(group, r)
            s.peerInboxCreate()
            shards.append(s)
        capi.peer_inbox_connect_local(shards)
        if fixed is not None:
            for s in shards:
                s.setExchangeTransport(fixed)
        return group, shards

    def on_threads(shards, fn):
        out, errs = [None] * world, []

        def work(i):
            try:
                out[i] = fn(shards[i])
            except Exception as e:   # noqa: BLE001
                errs.append((i, e))

        ts = [threading.Thread(target=work, args=(i,)) for i in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        return out

    def solve(s):
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        h = np.concatenate([s.apgIterate(20), s.apgIterate(4), s.apgIterate(20)])
        return h, s.counters(), {nm: s.get(bid) for bid, nm, _ in VECS}, s.exchangeAutotune(0)

    # (A)
    runs = {}
    for fixed in (None, 0):
        group, shards = ranks(fixed)
        runs[fixed] = on_threads(shards, solve)
        for s in shards:
            s.close()
        capi.local_group_destroy(group)
    infos = [r[3] for r in runs[None]]
    assert all(i["tunes"] == 1 and i["candidates"] == 3 and i["iterations"] == 20 for i in infos), infos
    assert len({i["transport"] for i in infos}) == 1 and len({(i["collective_us"], i["oneshot_us"]) for i in infos}) == 1, infos     # every rank: the same figures, the same choice
    assert infos[0]["transport"] == (1 if infos[0]["oneshot_us"] < infos[0]["collective_us"] else 0), infos
    assert all(i["tunes"] == 0 for i in (r[3] for r in runs[0]))
    for a, b in zip(runs[None], runs[0]):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1], (a[1], b[1])       # history and batch counters: as if the timing runs had not happened
        for nm in a[2]:
            assert np.array_equal(a[2][nm], b[2][nm]), nm
        assert np.abs(a[0] - ohist).max() <= 1e-9 * np.abs(ohist).max()
    # (B)
    chosen = {}
    for bias in (10_000_000, -10_000_000):
        group, shards = ranks(None, bias)

        def tune(s):
            s.initialiseSmpcController(dh, ah)
            s.apgReset()
            s.apgIterate(3)                      # (a short batch first -- which tunes by itself: the explicit call below starts from a state that is not all zeros)
            state = (capi.BUF_XI, capi.BUF_PSI, capi.BUF_UPD_XI, capi.BUF_UPD_PSI, capi.BUF_ACC_XI, capi.BUF_ACC_PSI)      # the iterate state: y, y+, w
            before = {b: s.get(b) for b in state}
            info = s.exchangeAutotune(16)
            after = {b: s.get(b) for b in state}
            h = s.apgIterate(20)
            return info, all(np.array_equal(before[k], after[k]) for k in before), h, s.counters()

        res = on_threads(shards, tune)
        assert all(r[1] for r in res)                                          # the iterate state (y, y+, w) is untouched by the tuner
        assert len({r[0]["transport"] for r in res}) == 1, [r[0] for r in res]
        assert len({(r[0]["collective_us"], r[0]["oneshot_us"]) for r in res}) == 1
        assert res[0][0]["own_oneshot_us"] != res[1][0]["own_oneshot_us"]      # rank 1's own figure carries the bias, the agreed one is the MAX
        chosen[bias] = res[0][0]["transport"]
        assert np.abs(res[0][2] - ohist[3:23]).max() <= 1e-9 * np.abs(ohist).max()
        assert res[0][3]["optimistic"] == 2 and res[0][3]["replayed"] == 0 and res[0][0]["tunes"] == 2, (res[0][3], res[0][0])
        for s in shards:
            s.close()
        capi.local_group_destroy(group)
    assert chosen == {10_000_000: 0, -10_000_000: 1}, chosen
    print("oneshot auto ok: %s world %d cut %d: unbiased choice %d (collective %.1f us, one-shot %.1f us per iteration), biased %s" % (
        name, world, cut, infos[0]["transport"], infos[0]["collective_us"], infos[0]["oneshot_us"], chosen), flush=True)


def timeout_case():
    """a rank that never pushes: the waiting rank's batch returns RN_E_COMM after the time-out instead of hanging"""
    import time

    p = synth.make_problem("medium")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    group = capi.local_group_create(2)
    shards = []
    for r in range(2):
        s = capi.Solver(p["network"], p["tree"], p["config"], rank=r, nranks=2)
        s.joinLocalGroup(group, r)
        s.peerInboxCreate()
        shards.append(s)
    capi.peer_inbox_connect_local(shards)
    for s in shards:
        s.setExchangeTransport(1)
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
    res = {}

    def lonely():
        t0 = time.time()
        try:
            shards[0].apgIterate(20, history=False)
            res["err"] = None
        except capi.RapidNetError as e:
            res["err"] = str(e)
        res["t"] = time.time() - t0

    t = threading.Thread(target=lonely)
    t.start()
    t.join()
    # rank 1 never iterates: rank 0's crown kernels time out (every one of them: bounded), and the closing collective of the batch
    # fails in the stand-in group as well -- either way an error, not a hang
    assert res["err"] is not None and res["t"] < 60, res
    print("oneshot timeout ok: %.1f s, %s" % (res["t"], res["err"][:120]), flush=True)
    os._exit(0)      # the half-finished group is not torn down


def latecomer_case():
    """Three ranks; rank 2 starts its batch 0.6 s late, the readers' time-out is 0.4 s.  Ranks 0 and 1 give up on rank 2's packets of
    the FIRST exchange (and carry on with incomplete sums); rank 2 then finds everybody's packets where they belong, catches up and
    never times out itself.  The batch is nevertheless invalid everywhere -- ranks 0 and 1 pushed sums derived from incomplete ones --
    so every rank must return RN_E_COMM: the flag rides in the per-batch MAX all-reduce (round 5; before, rank 2 returned RN_OK)."""
    import time

    p = synth.make_problem("medium")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    world = 3
    group = capi.local_group_create(world)
    shards = []
    for r in range(world):
        s = capi.Solver(p["network"], p["tree"], p["config"], rank=r, nranks=world)
        s.joinLocalGroup(group, r)
        s.peerInboxCreate()
        shards.append(s)
    capi.peer_inbox_connect_local(shards)
    for s in shards:
        s.setExchangeTransport(1)
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        s.synchronize()
    res = [None] * world

    def work(i):
        if i == 2:
            time.sleep(0.6)
        try:
            shards[i].apgIterate(20, history=False)
            res[i] = "ok"
        except capi.RapidNetError as e:
            res[i] = str(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert all(r is not None and r != "ok" for r in res), res                     # EVERY rank failed the batch, the one that saw all packets too
    assert all("did not arrive" in r for r in res), res                           # ... with the one-shot time-out's message (RN_E_COMM)
    for s in shards:                                                              # and refuses to go on until it is reset
        try:
            s.apgIterate(1, history=False)
            raise AssertionError("a poisoned context iterated")
        except capi.RapidNetError:
            pass
    print("oneshot latecomer ok: %s" % [r[:60] for r in res], flush=True)
    os._exit(0)


def ipc(name, cut):
    import ctypes as C

    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", init_method="env://")
    hip = C.CDLL("libamdhip64.so")
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    s = capi.Solver(p["network"], p["tree"], p["config"], rank=rank, nranks=world, cut_stage=cut)

    def all_reduce(buf, count, f64, op, stream):       # stands where ncclAllReduce would: device buffer -> host -> gloo -> device
        hip.hipStreamSynchronize(C.c_void_p(stream))
        host = np.empty(count, np.float64 if f64 else np.float32)
        if hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(buf), C.c_size_t(host.nbytes), 2) != 0:
            return 1
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 2 else dist.ReduceOp.SUM)
        return 0 if hip.hipMemcpy(C.c_void_p(buf), C.c_void_p(host.ctypes.data), C.c_size_t(host.nbytes), 1) == 0 else 2

    s.debugSetAllreduce(all_reduce)
    handles = [None] * world
    dist.all_gather_object(handles, s.peerInboxCreate())
    s.peerInboxConnect(handles)
    dist.barrier()
    s.initialiseSmpcController(dh, ah)
    out = {}
    for transport in (0, 1):
        s.setExchangeTransport(transport)
        s.apgReset()
        h = np.concatenate([s.apgIterate(20), s.apgIterate(20)])
        out[transport] = (h, {nm: s.get(bid) for bid, nm, _ in VECS})
    for nm in out[0][1]:
        assert np.array_equal(out[0][1][nm], out[1][1][nm]), nm         # two ranks: a + b in either order -- the same bits
    assert np.array_equal(out[0][0], out[1][0])
    rows = [None] * world
    dist.all_gather_object(rows, (s.global_nodes, out[1][1], out[1][0], s.counters()))
    if rank == 0:
        o = Oracle(p["network"], p["tree"], p["config"])
        o.initialise(dh, ah)
        ohist = o.apg(40)
        d = {"nx": s.nx, "nu": s.nu, "nv": s.nv, "2nx": 2 * s.nx}
        for bid, nm, dm in VECS:
            full = partition.scatter_to_global([r[1][nm] for r in rows], [r[0] for r in rows], s.full_nodes, d[dm])
            assert relmax(full, o.get(nm)) < 1e-9, nm
        for r in rows:
            assert np.abs(r[2] - ohist).max() <= 1e-9 * np.abs(ohist).max()
        print("oneshot ipc ok: %s, %d ranks, cut %d, batches %s" % (name, world, cut, rows[0][3]), flush=True)
    dist.barrier()
    sys.stdout.flush()
    os._exit(0)      # torch + the solver library in one process: no interpreter tear-down (see tests/test_gpu_bench_contract.py)


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "inprocess":
        kw = {"penalty_x": 20.0, "penalty_xs": 5.0} if "trip" in sys.argv[5:] else {}
        inprocess(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), "structured" in sys.argv[5:], "f32" if "f32" in sys.argv[5:] else "f64", kw)
    elif mode == "auto":
        auto_case(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    elif mode == "timeout":
        timeout_case()
    elif mode == "latecomer":
        latecomer_case()
    elif mode == "ipc":
        ipc(sys.argv[2], int(sys.argv[3]))
    else:
        sys.exit("unknown mode " + mode)
