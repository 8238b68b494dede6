"""tests/numpy_engine.py (the node-by-node numpy evaluation of the factor step and the affine terms, used on the HIP path where
the oracle's dense blocks do not fit the host) pinned to the CPU oracle -- which is itself pinned to the reference's fixtures --
on small trees: every operator block of every node, the scaled bounds, uhat / e / alpha / beta."""
import numpy as np
import pytest

from numpy_engine import NumpyEngine
from oracle.oracle import Oracle
from rapidnet_amd import synth


@pytest.mark.parametrize("name,alias", [("small", True), ("odd", True), ("ragged", True), ("late", False)])
def test_numpy_engine_matches_the_oracle(name, alias):
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"], alias_operators=alias)
    o.initialise(dh, ah)
    e = NumpyEngine(p["network"], p["tree"], p["config"])
    nv, nx, nu, nodes = e.nv, e.nx, e.nu, e.nodes
    blocks = {k: o.get(k).reshape(nodes, -1) for k in ("Phi", "D", "Psi", "Ftil")}
    for node in range(nodes):
        ops = e.operators(node)
        for k, ref in blocks.items():
            got = ops[k].ravel(order="F")
            assert np.abs(got - ref[node]).max() <= 1e-12 * max(np.abs(ref[node]).max(), 1e-300), (k, node)
    b = e.bounds_of(np.arange(nodes))
    for k in ("xmin", "xmax", "xs", "umin", "umax"):
        ref = o.get(k).reshape(nodes, -1)
        assert np.abs(b[k] - ref).max() <= 1e-13 * np.abs(ref).max(), k
    a = e.affine(dh, ah)
    for k in ("uhat", "e", "alpha", "beta"):
        ref = o.get(k).reshape(nodes, -1)
        assert np.abs(a[k] - ref).max() <= 1e-11 * np.abs(ref).max(), k
