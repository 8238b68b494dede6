"""The subtree partitioner behind the C-ABI (rn_partition_create, rapidnet_amd/csrc/partition.hpp) against the Python
cross-check rapidnet_amd/partition.py: local trees, global node maps and the children moments of the cut parents must be
identical, value for value.  Host-only code: runs without a GPU."""
import numpy as np
import pytest

from rapidnet_amd import capi, partition, synth


@pytest.mark.parametrize("name,world,cut", [("medium", 2, 2), ("medium", 3, 1), ("small", 2, 3), ("small", 5, 2), ("deep", 4, 3),
                                            ("late", 2, 5), ("fan", 7, 1), ("barcelona493", 8, 0), ("ragged", 3, 1), ("ragged", 2, 2)])
def test_c_partition_equals_python_partition(name, world, cut):
    p = synth.make_problem(name)
    c = cut or partition.default_cut_stage(p["tree"])
    assert capi.default_cut_stage(p["tree"]) == partition.default_cut_stage(p["tree"])
    E, P = partition.cut_children_moments(p["tree"], c)
    owned = np.zeros(int(p["tree"]["nodes"][0]), int)
    for r in range(world):
        lt, ids = partition.local_tree(p["tree"], r, world, c)
        q = capi.partition_tree(p["tree"], r, world, cut)
        for k in lt:
            assert np.array_equal(np.asarray(lt[k]), np.asarray(q["tree"][k])), (name, r, k)
        assert np.array_equal(ids, q["globalNode"]) and q["cutStage"] == c
        assert np.array_equal(E, q["momE"]) and np.array_equal(P, q["momP"])
        owned[q["globalNode"]] += 1
    crown = p["tree"]["nodesPerStageCumul"][c]
    assert (owned[:crown] == world).all() and (owned[crown:] == 1).all()


def test_partition_errors_are_reported():
    p = synth.make_problem("tiny")            # 2 x 2 tree: 2 subtrees below stage 1
    with pytest.raises(capi.RapidNetError, match="more ranks than subtrees"):
        capi.partition_tree(p["tree"], 2, 3, 1)
    with pytest.raises(capi.RapidNetError, match="bad rank"):
        capi.partition_tree(p["tree"], 2, 2, 1)
    with pytest.raises(capi.RapidNetError, match="cut stage"):
        capi.partition_tree(p["tree"], 0, 2, int(p["tree"]["N"][0]))
    one = synth.make_problem("horizon1")
    with pytest.raises(capi.RapidNetError, match="cut stage"):
        capi.partition_tree(one["tree"], 0, 2, 0)


def test_c_partition_equals_python_partition_on_random_nonuniform_trees():
    """30 random trees with per-node child counts (1 .. 4 children, 1 .. 3 branching stages), random rank counts and cuts: the C
    partitioner and the Python cross-check must agree on every array, and every rank's local tree must be a consistent
    scenario tree (children contiguous, ancestors in the previous stage, probabilities of the owned leaves summing to 1 over
    the ranks)."""
    rng = np.random.default_rng(20260303)
    done = 0
    while done < 30:
        N = int(rng.integers(3, 9))
        depth = int(rng.integers(1, min(3, N - 1) + 1))
        branching, width = [], 1
        for _ in range(depth):
            counts = [int(c) for c in rng.integers(1, 5, width)]
            branching.append(counts)
            width = sum(counts)
        tree = synth.make_tree(N, branching, rng, nd=3, nu=4)
        world = int(rng.integers(2, 6))
        cut = int(rng.integers(1, N))
        if tree["nodesPerStage"][cut] < world:
            continue                                   # more ranks than subtrees: rejected by both (covered above)
        done += 1
        E, P = partition.cut_children_moments(tree, cut)
        leaf_prob = 0.0
        for r in range(world):
            lt, ids = partition.local_tree(tree, r, world, cut)
            q = capi.partition_tree(tree, r, world, cut)
            for k in lt:
                assert np.array_equal(np.asarray(lt[k]), np.asarray(q["tree"][k])), (branching, world, cut, r, k)
            assert np.array_equal(ids, q["globalNode"]) and np.array_equal(E, q["momE"]) and np.array_equal(P, q["momP"])
            t = q["tree"]
            anc, st = np.asarray(t["ancestor"]), np.asarray(t["stages"])
            assert anc[0] == 0 and (st[anc[1:] - 1] == st[1:] - 1).all() and (np.diff(anc[1:]) >= 0).all()
            crown = tree["nodesPerStageCumul"][cut]
            owned_leaves = [i for i in range(len(st)) if st[i] == N - 1 and q["globalNode"][i] >= crown]
            leaf_prob += float(np.asarray(t["probNode"])[owned_leaves].sum())
        assert abs(leaf_prob - 1.0) < 1e-12
