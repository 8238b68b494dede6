"""Multi-rank (world_size 2, gloo, CPU) test of the subtree sharding: partitioner + one all-reduce of the cut parents'
children sums per APG iteration.  Every rank runs the CPU oracle on its LOCAL tree, the exchange goes through
torch.distributed (gloo), and the reassembled iterates must equal the single-process solve of the whole tree.
This is the logic bench.py --gpus N runs with RCCL in place of gloo (rapidnet_amd/csrc: k_cut_partial_sums)."""
import os
import socket

import numpy as np
import pytest

from rapidnet_amd import partition, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _theta_lambdas(n):
    th0, th1, out = 1.0, 1.0, []
    for _ in range(n):
        out.append(th1 * (1 / th0 - 1))
        th0, th1 = th1, 0.5 * (np.sqrt(th1 ** 4 + 4 * th1 ** 2) - th1 ** 2)
    return out


def _worker(rank, world, port, name, cut, iters, outdir):
    import torch
    import torch.distributed as dist

    from oracle.oracle import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    # the rank-local tree comes from the partitioner behind the C-ABI (rn_partition_create), the one the product uses
    from rapidnet_amd import capi

    part = capi.partition_tree(p["tree"], rank, world, cut)
    ltree, gids = part["tree"], part["globalNode"]
    o = Oracle(p["network"], ltree, p["config"], alias_operators=False)
    o.initialise(dh, ah)
    # beta of the cut parents needs ALL their children (calculateZeta); the product gets that from static tree
    # moments (partition.cut_children_moments -> rn_set_cut_children_moments), here it is taken from the full tree
    full = Oracle(p["network"], p["tree"], p["config"])
    full.initialise(dh, ah)
    crown = p["tree"]["nodesPerStageCumul"][cut]
    beta = o.get("beta")
    beta[: crown * o.nv] = full.get("beta")[: crown * o.nv]
    o.set("beta", beta)
    o.apg_reset()
    pn = ltree["nodesPerStage"][cut - 1]
    for lam in _theta_lambdas(iters):
        o.extrapolate(lam)
        o.solve_step_phase(0, cut)
        for name_, dim in (("q", o.nx), ("r", o.nv)):   # the ONE exchange step of an iteration
            buf = torch.from_numpy(o.buf(name_)[: pn * dim].copy())
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            o.buf(name_)[: pn * dim] = buf.numpy()
        o.solve_step_phase(1, cut)
        o.prox(); o.residual(); o.dual_update()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), gids=gids, x=o.get("x"), u=o.get("u"), upd=o.get("updXi"),
             psi=o.get("updPsi"))
    dist.barrier()
    dist.destroy_process_group()


# ragged: non-uniform tree, subtrees of unequal size per rank.  Only cuts whose rank-local trees are trees the REFERENCE can
# represent run here, because every rank runs the reference-faithful oracle on its local tree: the reference decides
# "branching stage" by node counts (SmpcController.cu:661, nk > pn) and indexes nChildren by node id, both of which break
# when a cut parent has no local child (ragged cut at 2 or 3).  The HIP path reads explicit child ranges; those cuts are
# covered on the GPU against the oracle of the FULL tree (tests/test_gpu_sharding.py, tests/test_gpu_sharded_batched.py).
@pytest.mark.parametrize("name,cut", [("medium", 2), ("medium", 1), ("ragged", 1)])
def test_two_rank_sharded_solve_matches_single_process(tmp_path, name, cut):
    import torch.multiprocessing as mp

    from oracle.oracle import Oracle

    iters, world = 6, 2
    mp.spawn(_worker, args=(world, _free_port(), name, cut, iters, str(tmp_path)), nprocs=world, join=True)
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(dh, ah)
    o.apg(iters)
    nodes = o.nodes
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    ids = [q["gids"] for q in parts]
    for key, ref, dim in (("x", "x", o.nx), ("u", "u", o.nu), ("upd", "updXi", 2 * o.nx), ("psi", "updPsi", o.nu)):
        full = partition.scatter_to_global([q[key] for q in parts], ids, nodes, dim)
        r = o.get(ref)
        assert np.abs(full - r).max() <= 1e-10 * np.abs(r).max(), key
    # the crown is replicated: both ranks must hold identical values there
    crown = p["tree"]["nodesPerStageCumul"][cut]
    assert np.array_equal(ids[0][:crown], ids[1][:crown])
    assert np.abs(parts[0]["x"][: crown * o.nx] - parts[1]["x"][: crown * o.nx]).max() <= 1e-9 * np.abs(parts[0]["x"]).max()


def test_partition_properties():
    for name, world in (("medium", 2), ("medium", 3), ("small", 2), ("barcelona31", 8)):
        p = synth.make_problem(name, step_size=1e-6)
        tree = p["tree"]
        cut = partition.default_cut_stage(tree)
        nodes = tree["nodes"][0]
        seen = np.zeros(nodes, int)
        crown = tree["nodesPerStageCumul"][cut]
        leaves = 0
        for r in range(world):
            lt, gids = partition.local_tree(tree, r, world, cut)
            assert list(gids[:crown]) == list(range(crown))          # crown replicated, same numbering
            seen[gids[crown:]] += 1
            assert lt["nodesPerStageCumul"][-2] == lt["nodes"][0] == len(gids)
            anc = np.array(lt["ancestor"])
            assert anc[0] == 0 and (anc[1:] >= 1).all() and (anc[1:] <= np.arange(1, len(gids))).all()
            # children of every node are contiguous and in stage order
            assert (np.diff(anc[1:]) >= 0).all()
            assert np.allclose(np.array(lt["probNode"]), np.array(tree["probNode"])[gids])
            leaves += lt["K"][0]
        assert (seen[crown:] == 1).all() and (seen[:crown] == 0).all()   # every subtree node owned exactly once
        assert leaves == tree["K"][0]
