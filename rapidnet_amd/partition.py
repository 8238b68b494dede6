"""Subtree partitioner for multi-GPU sharding (new capability; the reference is single-GPU).

The scenario tree is cut below stage `c`: nodes of stages < c (the "crown") are replicated on every rank, the
subtrees rooted at stage c are dealt round-robin to the ranks.  Each rank then holds an ordinary scenario tree (same
JSON schema, nodes renumbered stage by stage, children contiguous) whose stage c-1 nodes have only their LOCAL
children; the solver all-reduces the children sums of those cut parents once per APG iteration
(rn_set_cut_stage / k_cut_partial_sums) and everything else runs without communication.
"""
import numpy as np


def default_cut_stage(tree):
    """First stage at which the tree has stopped branching (all K scenario chains exist): the most balanced cut."""
    nps = np.asarray(tree["nodesPerStage"], int)
    K = int(tree["K"][0])
    N = int(tree["N"][0])
    for c in range(1, N):
        if nps[c] == K:
            return c
    return max(1, N - 1)


def local_tree(tree, rank, world, cut_stage):
    """Returns (local tree dict, global node id of every local node)."""
    N = int(tree["N"][0])
    nodes = int(tree["nodes"][0])
    nd, nu = int(tree["dimDemand"][0]), int(tree["dimPrice"][0])
    assert 1 <= cut_stage < N
    stages = np.asarray(tree["stages"], int)
    anc = np.asarray(tree["ancestor"], int) - 1
    cum = np.asarray(tree["nodesPerStageCumul"], int)
    owner = np.full(nodes, -1, int)  # -1: replicated crown node
    roots = np.arange(cum[cut_stage], cum[cut_stage + 1])
    owner[roots] = np.arange(roots.size) % world
    for i in range(cum[cut_stage + 1], nodes):  # BFS order: the parent is already labelled
        owner[i] = owner[anc[i]]
    keep = np.flatnonzero((owner == -1) | (owner == rank))  # increasing global id == stage by stage
    new_id = np.full(nodes, -1, int)
    new_id[keep] = np.arange(keep.size)
    l_stages = stages[keep]
    l_nodes = keep.size
    l_anc = np.where(anc[keep] >= 0, new_id[np.maximum(anc[keep], 0)] + 1, 0)
    per_stage = np.bincount(l_stages, minlength=N)
    assert (per_stage > 0).all(), "rank %d owns no node in some stage" % rank
    l_cum = np.concatenate([[0], np.cumsum(per_stage)])
    child_count = np.bincount(l_anc[l_anc > 0] - 1, minlength=l_nodes)
    K_local = int(per_stage[N - 1])
    leaves = np.flatnonzero(child_count == 0)
    nonleaf = np.flatnonzero(child_count > 0)
    # reference convention: nChildren lists the non-leaf nodes; a cut parent without local children is a local leaf
    n_children_cumul = np.cumsum(child_count)
    children = [i + 1 for i in range(1, l_nodes)]
    ed = np.asarray(tree["errorDemandNode"], float).reshape(nodes, nd)[keep]
    ep = np.asarray(tree["errorPriceNode"], float).reshape(nodes, nu)[keep]
    out = {
        "N": [N], "K": [K_local], "dimDemand": [nd], "dimPrice": [nu], "nodes": [l_nodes],
        "nChildrenTot": [l_nodes - 1], "nNonLeafNodes": [int(nonleaf.size)],
        "stages": l_stages.tolist(), "nodesPerStage": per_stage.tolist() + [0],
        "nodesPerStageCumul": l_cum.tolist() + [l_nodes],
        "leaves": (leaves + 1).tolist(), "children": children, "ancestor": l_anc.tolist(),
        "nChildren": child_count[nonleaf].tolist(), "nChildrenCumul": n_children_cumul.tolist(),
        "probNode": np.asarray(tree["probNode"], float)[keep].tolist(),
        "errorDemandNode": ed.ravel().tolist(), "errorPriceNode": ep.ravel().tolist(),
    }
    return out, keep


def cut_children_moments(tree, cut_stage):
    """For every cut parent i (stage cut_stage-1 of the FULL tree): E_i = sum_c p_c errorDemand_c over ALL its
    children and P_i = sum_c p_c.  A rank only holds its local children, but beta_i needs all of them
    (calculateZeta, Utilities.cu:100-131: zeta_i = p_i dUhat_i - sum_c p_c (uhat_c - uhat_i) with
    uhat_c = Lhat (errorDemand_c + dhat[stage+1])), and sum_c p_c uhat_c = Lhat (E_i + P_i dhat[stage+1])."""
    nodes = int(tree["nodes"][0])
    nd = int(tree["dimDemand"][0])
    cum = np.asarray(tree["nodesPerStageCumul"], int)
    anc = np.asarray(tree["ancestor"], int) - 1
    p = np.asarray(tree["probNode"], float)
    ed = np.asarray(tree["errorDemandNode"], float).reshape(nodes, nd)
    first, n_par = cum[cut_stage - 1], cum[cut_stage] - cum[cut_stage - 1]
    E = np.zeros((n_par, nd))
    P = np.zeros(n_par)
    for c in range(cum[cut_stage], cum[cut_stage + 1]):
        E[anc[c] - first] += p[c] * ed[c]
        P[anc[c] - first] += p[c]
    return E, P


def scatter_to_global(parts, global_ids, nodes, dim):
    """Reassemble a node-major vector of the full tree from per-rank local vectors (crown taken from rank 0)."""
    full = np.zeros((nodes, dim))
    for vec, ids in zip(parts, global_ids):
        full[ids] = np.asarray(vec, float).reshape(len(ids), dim)
    return full.ravel()
