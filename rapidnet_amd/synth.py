"""Deterministic synthetic DWN / scenario-tree / controller-config generator.

The reference ships only the 3-tank fixture; its Barcelona network.json and large trees are missing
(/root/reference/.MISSING_LARGE_BLOBS).  This module synthesises inputs of the same *shape* and in the
same JSON schema the reference's loaders read (DwnNetwork.cuh:23-37, ScenarioTree.cuh:23-40,
SmpcConfiguration.cuh:24-47, Forecaster.cu:94,108): scalars are 1-element lists, matrices are
column-major flat lists, tree indices are 1-based (`ancestor`, `children`, `leaves`), `nodesPerStage`
has N+1 entries and `nodesPerStageCumul` N+2.

Seed convention: 20260101 + config index.
"""
import json
import os

import numpy as np

# name -> (config index, nx, nu, nd, ne, N, branching)
CONFIGS = {
    "toy": (0, 3, 6, 4, 2, 5, []),                       # BASELINE.json configs[0]: N=5, 1 scenario
    "barcelona31": (1, 63, 114, 88, 17, 24, [31]),       # configs[1]: K=31, nodes=714
    "barcelona493": (2, 63, 114, 88, 17, 24, [17, 29]),  # configs[2]/[3]: K=493, nodes=10864 (headline)
    "wide4096": (4, 200, 360, 280, 54, 24, [16, 16, 16]),  # configs[4]: K=4096, nodes=86289
    "wide256": (4, 200, 360, 280, 54, 24, [16, 16]),     # the wide network on a 256-scenario tree (10 GB of blocks in fp32): tuning probe
    # small shapes for parity tests
    "tiny": (10, 3, 6, 4, 2, 6, [2, 2]),
    "small": (11, 5, 9, 6, 3, 8, [3, 2, 2]),
    "odd": (12, 7, 13, 5, 4, 7, [2, 3]),
    "medium": (13, 21, 38, 30, 6, 12, [4, 5]),
    # edge shapes: branching that starts late (single-child crown nodes), a deep binary crown (127 crown nodes: the
    # stage-by-stage crown kernels instead of the fused one), the shortest horizons, one wide fan-out
    "late": (14, 4, 7, 5, 2, 9, [1, 1, 3, 1, 2]),
    "deep": (15, 3, 6, 4, 2, 10, [2, 2, 2, 2, 2, 2, 2]),
    "horizon1": (16, 3, 6, 4, 2, 1, []),
    "horizon2": (17, 3, 6, 4, 2, 2, [3]),
    "fan": (18, 4, 8, 6, 3, 4, [70]),
    # many inputs, few tanks: 2 nv = 580 rows per operator column (several 16-byte slots per thread in k_stream_gemv) and
    # shared-operator products too large for the LDS slab kernels (tile-kernel fallback)
    "tall": (19, 8, 300, 40, 10, 3, [2]),
    # nv + 2 nx = 1200 components per crown node: more than the 1024 threads of a k_up_crown workgroup (and shared-operator
    # products far too large for the LDS slab kernels)
    "widecrown": (20, 400, 420, 30, 20, 4, [2, 2]),
    # NON-UNIFORM branching: per-node child counts (the reference's solveSumChildren / solveChildNodesUpdate walk the
    # nChildrenCumul offsets, Utilities.cu:142-201): root 3 -> (2, 4, 1) -> (1, 2, 1, 3, 1, 2, 1) -> 11 chains; subtrees of
    # unequal size below stage 1 (4, 7 and 2 ... nodes per stage) for the sharded path
    "ragged": (21, 6, 11, 5, 3, 9, [3, [2, 4, 1], [1, 2, 1, 3, 1, 2, 1]]),
    # non-uniform two-stage crown (root 3 -> 2, 4, 1 chains) with an even ny: the chain-local sweep's crown kernel on unequal children counts
    "small2": (23, 5, 10, 6, 3, 8, [3, [2, 4, 1]]),
    "ragged2": (22, 4, 7, 5, 2, 7, [[2], [3, 1], [1, 1, 2, 4], [2, 1, 1, 1, 1, 3, 1, 1]]),
    # crown nodes with many children: 40 and 3 under the root's two (k_up_crown_lin: several thread groups per node, several children per group),
    # and 20 children per node on a network whose crown rows are wider than half a workgroup (one group, two batches of loads)
    "bigfan": (25, 4, 8, 6, 3, 5, [2, [40, 3]]),
    "widefan": (26, 150, 240, 60, 20, 4, [2, 20]),
}


# 0.95 / lipschitz_estimate(...) of the configs whose 40 power iterations take minutes on a host CPU (the estimate walks the
# whole tree in numpy: ~90 GFLOP per application on wide4096); value computed once with exactly that call
STEP_SIZE_CACHE = {"wide4096": 1.5525146548810805e-06}


def make_tree(N, branching, rng, nd, nu, err_scale=0.05, dhat=None, ahat=None):
    """Stage-contiguous BFS tree that branches in the leading stages, then chains to N.

    branching[k] = children per node of stage k: an int (uniform per stage) or a list with one count per node of the stage
    (non-uniform trees: the reference supports per-node child counts through nChildren / nChildrenCumul,
    Utilities.cu:142-201).  Returns the tree dict in the reference schema.  Branching only in the leading stages is the
    only shape the reference's operator aliasing supports (Engine.cu:210-221).
    """
    per_stage = [1]
    counts = []          # per stage: child count of every node
    for b in branching:
        cb = [int(b)] * per_stage[-1] if np.isscalar(b) else [int(v) for v in b]
        assert len(cb) == per_stage[-1] and min(cb) >= 1, "one child count (>= 1) per node of the stage"
        counts.append(cb)
        per_stage.append(int(sum(cb)))
    while len(per_stage) < N:
        per_stage.append(per_stage[-1])
    per_stage = per_stage[:N]
    assert len(branching) < N
    cumul = np.concatenate([[0], np.cumsum(per_stage)]).astype(int)
    nodes = int(cumul[-1])
    K = per_stage[-1]
    stages = np.concatenate([np.full(n, k) for k, n in enumerate(per_stage)]).astype(int)
    ancestor = np.zeros(nodes, int)  # 1-based, root = 0
    prob = np.ones(nodes)
    n_children = []
    children = []
    for k in range(N - 1):
        first = cumul[k + 1]
        for j in range(per_stage[k]):
            i = cumul[k] + j
            b = counts[k][j] if k < len(counts) else 1
            if j > 0:
                first += counts[k][j - 1] if k < len(counts) else 1
            w = rng.dirichlet(np.ones(b) * 4.0) if b > 1 else np.ones(1)
            for c in range(b):
                ancestor[first + c] = i + 1
                prob[first + c] = prob[i] * w[c]
                children.append(int(first + c + 1))
            n_children.append(int(b))
    n_nonleaf = nodes - K
    n_children_cumul = np.zeros(nodes, int)
    n_children_cumul[:n_nonleaf] = np.cumsum(n_children)
    n_children_cumul[n_nonleaf:] = n_children_cumul[n_nonleaf - 1] if n_nonleaf > 0 else 0
    leaves = np.arange(cumul[N - 1], nodes) + 1
    if dhat is None:
        dhat = np.ones((N, nd))
    if ahat is None:
        ahat = np.ones((N, nu))
    err_d = err_scale * rng.standard_normal((nodes, nd)) * dhat[stages]
    err_a = err_scale * rng.standard_normal((nodes, nu)) * ahat[stages]
    err_d[0] = 0.0
    err_a[0] = 0.0
    return {
        "N": [N], "K": [K], "dimDemand": [nd], "dimPrice": [nu], "nodes": [nodes],
        "nChildrenTot": [nodes - 1], "nNonLeafNodes": [n_nonleaf],
        "stages": stages.tolist(),
        "nodesPerStage": [int(v) for v in per_stage] + [0],
        "nodesPerStageCumul": cumul.tolist() + [nodes],
        "leaves": leaves.tolist(), "children": children, "ancestor": ancestor.tolist(),
        "nChildren": n_children, "nChildrenCumul": n_children_cumul.tolist(),
        "probNode": prob.tolist(),
        "errorDemandNode": err_d.ravel().tolist(), "errorPriceNode": err_a.ravel().tolist(),
    }


def make_network(nx, nu, nd, ne, rng):
    """Sparse incidence-style DWN: x+ = x + B u + Gd d, 0 = E u + Ed d (DwnNetwork.cuh:41-57)."""
    B = np.zeros((nx, nu))
    for j in range(nu):
        B[rng.integers(nx), j] = 1.0           # every actuator fills one tank ...
        if rng.random() < 0.35:
            t = rng.integers(nx)
            if B[t, j] == 0:
                B[t, j] = -1.0                 # ... and some drain another
    for i in range(nx):                        # no isolated tank
        if not B[i].any():
            B[i, rng.integers(nu)] = 1.0
    Gd = np.zeros((nx, nd))
    for j in range(nd):
        Gd[rng.integers(nx), j] = -1.0
    E = np.zeros((ne, nu))
    cols = rng.permutation(nu)[:ne]
    for i in range(ne):                        # dedicated pivot column per mixing node => full row rank
        E[i, cols[i]] = 1.0
        for j in rng.permutation(nu)[:3]:
            if j not in cols:
                E[i, j] = rng.choice([-1.0, 1.0])
    Ed = np.zeros((ne, nd))
    for i in range(ne):
        Ed[i, rng.integers(nd)] = -1.0
    xmax = rng.uniform(500, 5000, nx).round(0)
    xsafe = (rng.uniform(0.05, 0.15, nx) * xmax).round(1)
    umax = rng.uniform(100, 2000, nu).round(0)
    alpha1 = np.where(rng.random(nu) < 0.3, rng.uniform(0.02, 0.15, nu), 0.0).round(4)
    col = lambda M: M.ravel(order="F").tolist()
    return {
        "nx": [nx], "nu": [nu], "ne": [ne], "nd": [nd],
        "matA": col(np.eye(nx)), "matB": col(B), "matGd": col(Gd), "matE": col(E), "matEd": col(Ed),
        "vecXmin": [0.0] * nx, "vecXmax": xmax.tolist(), "vecXsafe": xsafe.tolist(),
        "vecUmin": [0.0] * nu, "vecUmax": umax.tolist(), "costAlpha1": alpha1.tolist(),
    }


def null_space_and_particular(E, Ed):
    """L = null(E) taken as the trailing left singular vectors of E' and Lhat = -pinv(E) Ed, exactly the
    construction of Engine::calculateMatLandMatLhat (Engine.cu:466-669) but on the host in fp64."""
    ne, nu = E.shape
    U, S, Vt = np.linalg.svd(E.T, full_matrices=True)  # E' = U S V'
    L = U[:, ne:]
    Sinv = np.where(np.abs(S) > 0, 1.0 / S, 0.0)
    pinvE = U[:, :ne] @ (Sinv[:, None] * Vt)
    return L, -pinvE @ Ed


class _Structured:
    """numpy evaluation of the *linear part* of the dual-gradient sweep (beta = uhat = e = 0, x0 = 0), used
    only to estimate the Lipschitz constant for the generated stepSize.  Uses the algebraic structure of the
    reference's operators: every per-node block is a shared matrix times a per-stage
    diagonal times a power of p_i."""

    def __init__(self, network, tree, config):
        g = lambda d, k: int(d[k][0])
        self.nx, self.nu, self.nd = g(network, "nx"), g(network, "nu"), g(network, "nd")
        self.nv, self.N = g(config, "nv"), g(tree, "N")
        nx, nu, nv, N = self.nx, self.nu, self.nv, self.N
        self.B = np.array(network["matB"], float).reshape(nx, nu, order="F")
        self.L = np.array(config["matL"], float).reshape(nu, nv, order="F")
        W = np.array(config["costW"], float).reshape(nu, nu, order="F")
        self.Rinv = np.linalg.inv(self.L.T @ W @ self.L)
        self.Bbar_t = self.L.T @ self.B.T
        self.diag = np.array(config["matDiagPrecnd"], float).reshape(N, 2 * nx + nu)
        self.cum = np.array(tree["nodesPerStageCumul"], int)
        self.nps = np.array(tree["nodesPerStage"], int)
        self.anc = np.array(tree["ancestor"], int) - 1
        self.p = np.array(tree["probNode"], float)
        self.stage = np.array(tree["stages"], int)
        self.nodes = g(tree, "nodes")

    def apply(self, xi, psi):
        nx, nu, nv, N = self.nx, self.nu, self.nv, self.N
        sp = np.sqrt(self.p)[:, None]
        du, dx, dxs = self.diag[:, :nu], self.diag[:, nu:nu + nx], self.diag[:, nu + nx:]
        a = sp * (dx[self.stage] * xi[:, :nx] + dxs[self.stage] * xi[:, nx:])
        b = sp * du[self.stage] * psi
        c = a @ self.Bbar_t.T + b @ self.L
        r = np.zeros((self.nodes, nv))
        q = np.zeros((self.nodes, nx))
        v = np.zeros((self.nodes, nv))
        for k in range(N - 1, -1, -1):
            sl = slice(self.cum[k], self.cum[k + 1])
            r[sl] += c[sl]
            q[sl] += a[sl]
            v[sl] = (-0.5 / self.p[sl, None]) * (r[sl] @ self.Rinv.T)
            if k > 0:
                np.add.at(r, self.anc[sl], r[sl] + q[sl] @ self.Bbar_t.T)
                np.add.at(q, self.anc[sl], q[sl])
        w = np.zeros((self.nodes, nu))
        x = np.zeros((self.nodes, nx))
        for k in range(N):
            sl = slice(self.cum[k], self.cum[k + 1])
            w[sl] = v[sl] @ self.L.T
            x[sl] = w[sl] @ self.B.T
            if k > 0:
                w[sl] += w[self.anc[sl]]
                x[sl] = x[self.anc[sl]] + w[sl] @ self.B.T
        return np.hstack([sp * dx[self.stage] * x, sp * dxs[self.stage] * x]), sp * du[self.stage] * w


def lipschitz_estimate(network, tree, config, iters=40, seed=0):
    op = _Structured(network, tree, config)
    rng = np.random.default_rng(seed)
    xi = rng.standard_normal((op.nodes, 2 * op.nx))
    psi = rng.standard_normal((op.nodes, op.nu))
    lam = 1.0
    for _ in range(iters):
        nrm = np.sqrt((xi ** 2).sum() + (psi ** 2).sum())
        xi, psi = xi / nrm, psi / nrm
        xi, psi = op.apply(xi, psi)
        lam = np.sqrt((xi ** 2).sum() + (psi ** 2).sum())
    return float(lam)


# Configs whose constraints are made FEASIBLE by construction (make_feasible below): the BASELINE.json workloads.  With the
# random bounds of make_network the hard constraints of these problems cannot all be met (the primal infeasibility of the
# APG iterates never falls below a few hundred), the dual iteration then wanders without converging and amplifies rounding
# differences ~10x per 50 iterations -- no second implementation can track a first one to 1e-8 for 500 iterations on such
# data.  On a feasible problem the same iteration is stable (two oracle runs whose beta differs by 1e-13 stay 1e-15 apart for
# 500 iterations), so north_star's tolerance can be asserted directly at the reference's maxIterations.  The small parity
# shapes keep their original data ("<name>_infeasible" gives the original data of a feasible config).
FEASIBLE = {"barcelona31", "barcelona493", "wide4096", "wide256"}


def steady_state_map(network, config):
    """M (nu x nd) with u = M d the control that meets the mass balance at the mixing nodes, E u + Ed d = 0 (u = L v + Lhat d,
    DwnNetwork.cuh:41-57), and keeps every tank level constant, B u + Gd d = 0: v = -(B L)^+ (B Lhat + Gd) d.
    Returns (M, largest |B M + Gd| entry): the drift per unit demand that (B L) cannot cancel (0 if it has full row rank)."""
    nx, nu, nd = (int(network[k][0]) for k in ("nx", "nu", "nd"))
    nv = int(config["nv"][0])
    B = np.array(network["matB"], float).reshape(nx, nu, order="F")
    Gd = np.array(network["matGd"], float).reshape(nx, nd, order="F")
    L = np.array(config["matL"], float).reshape(nu, nv, order="F")
    Lhat = np.array(config["matLhat"], float).reshape(nu, nd, order="F")
    V = -np.linalg.lstsq(B @ L, B @ Lhat + Gd, rcond=None)[0]
    M = Lhat + L @ V
    return M, float(np.abs(B @ M + Gd).max())


def make_feasible(problem, margin=0.25):
    """Re-centres the control bounds (and the previous control) so that the problem has a strictly feasible point for every
    forecast of the simulation horizon: the policy u_i = M d_i (steady_state_map) keeps x_i = currentX at every node of the
    tree, and umin / umax are placed `margin` of their spread (plus 2 % of the level) outside the range that policy needs.
    The state bounds stay (xsafe < currentX < xmax by construction of the generator).  The optimum is elsewhere -- the
    economic cost pushes the controls down until umin or the safety level of a tank binds -- so constraints are active, but
    the dual problem has a solution and APG converges."""
    network, tree, config, forecast = (problem[k] for k in ("network", "tree", "config", "forecast"))
    nu, nd = int(network["nu"][0]), int(network["nd"][0])
    nodes = int(tree["nodes"][0])
    M, drift = steady_state_map(network, config)
    if drift > 1e-12 * max(1.0, float(np.abs(M).max())):
        # (B L) has no full row rank: some combination of tank levels cannot be controlled (e.g. two tanks joined by one
        # pipe and nothing else), and the demands would drive it out of any bounds.  The demands' effect on exactly those
        # combinations, R = (I - (B L)(B L)^+)(B Lhat + Gd), is taken out of Gd; B -- hence the Lipschitz constant and the
        # step size -- and M are unchanged.
        nx, nd_ = int(network["nx"][0]), nd
        B = np.array(network["matB"], float).reshape(nx, nu, order="F")
        Gd = np.array(network["matGd"], float).reshape(nx, nd_, order="F")
        Gd = Gd - (B @ M + Gd)
        network["matGd"] = Gd.ravel(order="F").tolist()
        M2, drift = steady_state_map(network, config)
        assert drift < 1e-9 * max(1.0, float(np.abs(M).max())) and np.abs(M2 - M).max() < 1e-9 * max(1.0, float(np.abs(M).max()))
    st = np.asarray(tree["stages"], int)
    err = np.asarray(tree["errorDemandNode"], float).reshape(nodes, nd)
    lo, hi = np.full(nu, np.inf), np.full(nu, -np.inf)
    for t in range(int(forecast["simHorizon"][0])):
        dh, _ = forecast_at(forecast, t)
        U = (err + dh.reshape(-1, nd)[st]) @ M.T
        lo, hi = np.minimum(lo, U.min(0)), np.maximum(hi, U.max(0))
    uprev = M @ np.asarray(config["prevDemand"], float)
    lo, hi = np.minimum(lo, uprev), np.maximum(hi, uprev)
    pad = margin * (hi - lo) + 0.02 * np.maximum(np.abs(lo), np.abs(hi)) + 1e-3
    network["vecUmin"] = (lo - pad).tolist()
    network["vecUmax"] = (hi + pad).tolist()
    config["prevU"] = uprev.tolist()
    xs, xmax, x0 = (np.asarray(v, float) for v in (network["vecXsafe"], network["vecXmax"], config["currentX"]))
    assert (xs < x0).all() and (x0 < xmax).all(), "currentX must lie strictly between the safety level and the capacity of every tank"
    return problem


def make_problem(name, max_iterations=500, sim_horizon=2, penalty_x=1e6, penalty_xs=1e4, step_size=None, feasible=None):
    """Returns {"network","tree","config","forecast"} dicts (reference JSON schema) for a named config.

    The generator's linear algebra (an SVD, a least-squares solve, matrix products) runs with ONE BLAS thread: a threaded BLAS
    blocks its reductions by the thread count, so the same seed gave data that differed in the last bits between a 256-thread host
    and an 8-thread one (`fingerprint` told them apart: bench lines of one round carried two different `data_sha256`)."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:      # not installed: the data are then only reproducible per host
        return _make_problem(name, max_iterations, sim_horizon, penalty_x, penalty_xs, step_size, feasible)
    with threadpool_limits(limits=1):
        return _make_problem(name, max_iterations, sim_horizon, penalty_x, penalty_xs, step_size, feasible)


def _make_problem(name, max_iterations, sim_horizon, penalty_x, penalty_xs, step_size, feasible):
    if name.endswith("_infeasible") and name[:-11] in CONFIGS:
        name, feasible = name[:-11], False
    if feasible is None:
        feasible = name in FEASIBLE
    idx, nx, nu, nd, ne, N, branching = CONFIGS[name]
    rng = np.random.default_rng(20260101 + idx)
    network = make_network(nx, nu, nd, ne, rng)
    nv = nu - ne
    hours = np.arange(N + sim_horizon)
    base_d = rng.uniform(5, 80, nd)
    base_a = rng.uniform(0.05, 0.15, nu)
    dhat_all = base_d[None, :] * (1 + 0.3 * np.sin(2 * np.pi * (hours[:, None] + rng.uniform(0, 24, nd)[None, :]) / 24))
    ahat_all = base_a[None, :] * (1 + 0.5 * np.sin(2 * np.pi * (hours[:, None] + 6) / 24))
    tree = make_tree(N, branching, rng, nd, nu, dhat=dhat_all[:N], ahat=ahat_all[:N])
    E = np.array(network["matE"], float).reshape(ne, nu, order="F")
    Ed = np.array(network["matEd"], float).reshape(ne, nd, order="F")
    L, Lhat = null_space_and_particular(E, Ed)
    W = np.diag(rng.uniform(0.5, 2.0, nu))
    diag = rng.uniform(0.8, 2.2, (N, 2 * nx + nu))
    xmax = np.array(network["vecXmax"])
    umax = np.array(network["vecUmax"])
    col = lambda M: np.asarray(M).ravel(order="F").tolist()
    config = {
        "nx": [nx], "nu": [nu], "ne": [ne], "nv": [nv], "nd": [nd], "N": [N],
        "matL": col(L), "matLhat": col(Lhat), "matDiagPrecnd": diag.ravel().tolist(), "costW": col(W),
        "currentX": (0.5 * xmax).tolist(), "prevDemand": dhat_all[0].tolist(),
        "prevU": (rng.uniform(0.1, 0.4, nu) * umax).tolist(),
        "stepSize": [1e-4], "maxIterations": [max_iterations],
        "penaltyStateX": [penalty_x], "penaltySafetyX": [penalty_xs],
        "pathToNetwork": "network.json", "pathToScenarioTree": "scenarioTree.json",
        "pathToForecaster": "forecastor.json", "algorithmName": "proximalAlgorithm", "lbfgsBufferSize": [5],
    }
    if step_size is None:
        step_size = STEP_SIZE_CACHE.get(name) or 0.95 / lipschitz_estimate(network, tree, config)
    config["stepSize"] = [float(step_size)]
    forecast = {"N": [N], "simHorizon": [sim_horizon], "dimDemand": [nd], "dimPrices": [nu]}
    for t in range(sim_horizon):  # Forecaster reads members 4+2t / 5+2t in file order (Forecaster.cu:94,108)
        forecast["timeIdDemand%d" % t] = dhat_all[t:t + N].ravel().tolist()
        forecast["timeIdPrice%d" % t] = ahat_all[t:t + N].ravel().tolist()
    problem = {"network": network, "tree": tree, "config": config, "forecast": forecast}
    return make_feasible(problem) if feasible else problem


# Version tag of the generated data.  "f1" (round 3 on): the BASELINE.json shapes (names in FEASIBLE) are made feasible by
# construction (make_feasible: control bounds, prevU and -- for rank-deficient B L -- a few columns of Gd are re-centred); "g0": the
# generator's original random bounds (what rounds 1-2 benchmarked under the plain names, still available as "<name>_infeasible").
# Dimensions, B, L, W, the preconditioner, the tree and the step size are the same in both, so the kernels do the same work; the
# iterates, the goldens and the active sets differ -- numbers of different versions are not comparable digit for digit.
DATA_VERSION_FEASIBLE, DATA_VERSION_ORIGINAL = "f1", "g0"


def data_tag(name):
    """'<workload>@<data version>' of a named config, e.g. 'barcelona493@f1', 'barcelona31_infeasible@g0'."""
    base = name[:-11] if name.endswith("_infeasible") else name
    feasible = (base in FEASIBLE) and not name.endswith("_infeasible")
    return "%s@%s" % (name, DATA_VERSION_FEASIBLE if feasible else DATA_VERSION_ORIGINAL)


def fingerprint(problem):
    """sha256 over the numbers a solve depends on (network, tree, configuration, both forecasts), as float64 bytes in a fixed
    key order: two runs that print the same fingerprint solved the same problem."""
    import hashlib

    h = hashlib.sha256()
    for part, keys in (("network", ("matB", "matGd", "matE", "matEd", "vecXmin", "vecXmax", "vecXsafe", "vecUmin", "vecUmax", "costAlpha1")),
                       ("tree", ("ancestor", "stages", "probNode", "errorDemandNode", "errorPriceNode")),
                       ("config", ("matL", "matLhat", "costW", "matDiagPrecnd", "currentX", "prevU", "prevDemand", "stepSize", "penaltyStateX", "penaltySafetyX"))):
        for k in keys:
            h.update(k.encode() + b"\0")
            h.update(np.ascontiguousarray(np.asarray(problem[part][k], dtype=np.float64)).tobytes())
    for k in sorted(problem["forecast"]):
        h.update(k.encode() + b"\0")
        h.update(np.ascontiguousarray(np.asarray(problem["forecast"][k], dtype=np.float64)).tobytes())
    return h.hexdigest()


def forecast_at(forecast, sim_time):
    """Forecaster::predictDemand/predictPrices member-order rule (Forecaster.cu:93-119)."""
    keys = list(forecast.keys())
    return (np.asarray(forecast[keys[4 + 2 * sim_time]], float), np.asarray(forecast[keys[5 + 2 * sim_time]], float))


def write_problem(problem, directory):
    """Write the four JSON files the reference's loaders read; config paths are made absolute."""
    os.makedirs(directory, exist_ok=True)
    cfg = dict(problem["config"])
    cfg["pathToNetwork"] = os.path.join(directory, "network.json")
    cfg["pathToScenarioTree"] = os.path.join(directory, "scenarioTree.json")
    cfg["pathToForecaster"] = os.path.join(directory, "forecastor.json")
    for fname, d in (("network.json", problem["network"]), ("scenarioTree.json", problem["tree"]),
                     ("forecastor.json", problem["forecast"]), ("controllerConfig.json", cfg)):
        with open(os.path.join(directory, fname), "w") as f:
            json.dump(d, f)
    return os.path.join(directory, "controllerConfig.json")
