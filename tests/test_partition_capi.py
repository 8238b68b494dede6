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
