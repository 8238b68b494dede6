"""Operator storage on the boundary (round 6): RN_OPS_AUTO -- the default of the C-ABI and of the C++ class surface -- runs the exact
structured form (no per-node block is ever stored) for as long as every block is the factor step's own, and switches to dense storage
the moment a caller hands in a block of its own (rn_set_operator: the counterpart of writing through the reference's getMatPhi() /
getPtrMatPhi()[node] device pointers, Engine.cuh:170-230).  Checked against the CPU oracle, whose per-node blocks (Phi, D, Psi, Ftil:
the arrays solveStep multiplies with at SmpcController.cu:617-638) can be overwritten the same way."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from rapidnet_amd import capi, synth

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9
VECS = ((capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_V, "v"), (capi.BUF_UPD_XI, "updXi"), (capi.BUF_UPD_PSI, "updPsi"), (capi.BUF_DUAL_XI, "dualXi"),
        (capi.BUF_RES_PSI, "resPsi"))
OPS = ((capi.OP_PHI, "Phi"), (capi.OP_D, "D"), (capi.OP_PSI, "Psi"), (capi.OP_F, "Ftil"))


def relmax(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    assert a.shape == b.shape and np.isfinite(a).all()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def solver(p, mode, precision="f64"):
    s = capi.Solver(p["network"], p["tree"], p["config"], operator_mode=mode, precision=precision)
    s.initialiseSmpcController(*synth.forecast_at(p["forecast"], 0))
    return s


@pytest.mark.parametrize("name", ["medium", "ragged", "barcelona31"])
def test_auto_is_the_structured_form_until_a_block_is_handed_in(name):
    p = synth.make_problem(name)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(*synth.forecast_at(p["forecast"], 0))
    a, d = solver(p, "auto"), solver(p, "dense")
    assert a.operatorMode() == ("auto", "structured") and d.operatorMode() == ("dense", "dense")
    ha, hd, ho = a.algorithmApg(30), d.algorithmApg(30), o.apg(30)
    for bid, nm in VECS:
        assert relmax(a.get(bid), o.get(nm)) < REL_TOL and relmax(d.get(bid), o.get(nm)) < REL_TOL, nm
    assert np.abs(ha - ho).max() <= REL_TOL * np.abs(ho).max() and np.abs(hd - ho).max() <= REL_TOL * np.abs(ho).max()
    # the factor step's own block handed back in: dense storage from here on, and the very bits of a context that was dense from the start
    node = a.nodes // 2
    a.setOperator(capi.OP_PSI, node, a.getOperator(capi.OP_PSI, node))
    assert a.operatorMode() == ("auto", "dense")
    h2a, h2d = a.algorithmApg(30), d.algorithmApg(30)
    assert np.array_equal(h2a, h2d)
    for bid, _ in VECS:
        assert np.array_equal(a.get(bid), d.get(bid))
    a.close(); d.close()


@pytest.mark.parametrize("name,precision,tol", [("medium", "f64", REL_TOL), ("small", "f64", REL_TOL), ("medium", "f32", 2e-4)])
def test_blocks_of_the_callers_own_against_the_oracle(name, precision, tol):
    """every kind of per-node block (Phi, D, Psi, Ftil) overwritten on a few nodes -- crown, chain, leaf -- in the oracle and, through
    rn_set_operator, in an auto context and a dense one: same iterates as the oracle, and not those of the unmodified problem"""
    p = synth.make_problem(name)
    o = Oracle(p["network"], p["tree"], p["config"], precision=precision)
    o.initialise(*synth.forecast_at(p["forecast"], 0))
    base = solver(p, "auto", precision)
    hbase = base.algorithmApg(25)
    a, d = solver(p, "auto", precision), solver(p, "dense", precision)
    nx, nu, nv, n = o.nx, o.nu, o.nv, o.nodes
    rng = np.random.default_rng(5)
    dims = {"Phi": nv * 2 * nx, "D": nv * 2 * nx, "Psi": nv * nu, "Ftil": nv * nu}
    for k, (op, oname) in enumerate(OPS):
        for node in sorted({0, 1 + k, n // 2 + k, n - 1 - k}):
            blocks = o.buf(oname).reshape(n, dims[oname])          # a view: written through
            mine = np.array(blocks[node], dtype=np.float64) * (1.0 + 0.3 * rng.standard_normal(dims[oname]))
            blocks[node] = mine
            mine = np.array(blocks[node], dtype=np.float64)        # (fp32 oracle: what it really holds)
            for s in (a, d):
                s.setOperator(op, node, mine)
                assert relmax(s.getOperator(op, node), mine) < (1e-15 if precision == "f64" else 1e-7)
    assert a.operatorMode() == ("auto", "dense")
    ha, hd, ho = a.algorithmApg(25), d.algorithmApg(25), o.apg(25)
    for bid, nm in VECS:
        assert relmax(a.get(bid), o.get(nm)) < tol and relmax(d.get(bid), o.get(nm)) < tol, nm
    assert np.abs(ha - ho).max() <= tol * np.abs(ho).max() and np.array_equal(ha, hd)
    assert np.abs(ha - hbase).max() > 1e-6 * np.abs(hbase).max()      # the blocks matter
    # a new factor step recomputes every block: back to the unmodified problem's iterates (and an auto context stays dense: it was told once)
    a.factorStep()
    assert np.abs(a.algorithmApg(25) - hbase).max() <= (1e-9 if precision == "f64" else 2e-4) * np.abs(hbase).max()
    for s in (base, a, d):
        s.close()


def test_what_cannot_be_handed_in():
    p = synth.make_problem("small")
    st = solver(p, "structured")
    blk = st.getOperator(capi.OP_PHI, 1)
    with pytest.raises(capi.RapidNetError, match="RN_OPS_STRUCTURED"):
        st.setOperator(capi.OP_PHI, 1, blk)
    a = solver(p, "auto")
    with pytest.raises(capi.RapidNetError, match="shared matrices"):
        a.setOperator(capi.OP_OMEGA, 1, a.getOperator(capi.OP_OMEGA, 1))
    with pytest.raises(capi.RapidNetError, match="size"):
        a.setOperator(capi.OP_PHI, 1, blk[:-1])
    assert a.operatorMode() == ("auto", "structured")          # a refused call changes nothing
    fresh = capi.Solver(p["network"], p["tree"], p["config"], operator_mode="auto")
    with pytest.raises(capi.RapidNetError, match="before rn_factor_step"):
        fresh.setOperator(capi.OP_PHI, 1, blk)
    for s in (st, a, fresh):
        s.close()


@pytest.mark.parametrize("alg", ["globalFbeAlgorithm", "namaAlgorithm"])
def test_quasi_newton_loops_after_the_switch_to_dense(alg):
    """an auto context running global FBE / NAMA is handed a block: the loops carry on in dense storage (NAMA with its paired Hessian sweep,
    whose buffers only exist in dense mode) and give the iterates of a context that was dense from the start"""
    p = synth.make_problem("medium")
    runs = []
    for mode in ("auto", "dense"):
        s = capi.Solver(p["network"], p["tree"], p["config"], operator_mode=mode)
        s.setAlgorithm(alg, 5)
        s.initialiseSmpcController(*synth.forecast_at(p["forecast"], 0))
        node = s.nodes // 3
        s.setOperator(capi.OP_D, node, 1.2 * s.getOperator(capi.OP_D, node))
        run = s.algorithmGlobalFbe if alg == "globalFbeAlgorithm" else s.algorithmNama
        h, v, t = run(8)
        runs.append((h, v, t, s.get(capi.BUF_X), s.fbeCounters()))
        s.close()
    assert np.array_equal(runs[0][2], runs[1][2]) and np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][3], runs[1][3])
    if alg == "namaAlgorithm":
        assert runs[0][4]["sweep_pairs"] == runs[1][4]["sweep_pairs"] > 0


@pytest.mark.parametrize("name,precision,tol", [("medium", "f64", 1e-9), ("ragged", "f64", 1e-9), ("small2", "f64", 1e-9), ("late", "f64", 1e-9), ("deep", "f64", 1e-9),
                                                ("horizon1", "f64", 1e-9), ("horizon2", "f64", 1e-9), ("toy", "f64", 1e-9), ("fan", "f64", 1e-9), ("odd", "f64", 1e-9),
                                                ("barcelona31", "f64", 1e-9), ("medium", "f32", 2e-4), ("barcelona31", "f32", 2e-4),
                                                ("bigfan", "f64", 1e-9), ("widefan", "f32", 2e-4)])
def test_linear_form_of_the_structured_sweep(name, precision, tol):
    """Structured mode, round 6: the running sums are taken of the INPUTS of the shared-operator products (k_up_chain_lin, k_up_crown_lin, the
    root's step in the v / Lv launch) and the first product k_gemm_prep_m2 disappears -- against the oracle, and against the form with that
    product (rn_debug_set_knob struct_linear = 0): the same sums in another association, so to rounding, not bitwise.  Chains from the root,
    late branching, a deep crown (stage-by-stage crown launches), ragged child counts, odd ny (flat dual update), the shortest horizons, crown nodes
    with 40 children (several thread groups per node in k_up_crown_lin) and crown rows wider than half a workgroup (one group, two batches)."""
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    # (late branching and the shortest horizons: the reference's aliasing of Omega / Theta by scenario position does not apply, tests/test_gpu_parity.py EDGE_SHAPES)
    o = Oracle(p["network"], p["tree"], p["config"], precision=precision, alias_operators=name not in ("late", "horizon1", "horizon2"))
    o.initialise(dh, ah)
    runs = []
    # the linear form as it ships (the subtree sums of beta a constant of the control step, one product with the composite operator), the form with the
    # first product, the linear form with beta walked every iteration, the constant with v / [Lv; BLv] as two products, and the composite operator
    # without the forward walk's affine terms in its constant
    for lin in (1, 0, 3, 4, 5):
        s = capi.Solver(p["network"], p["tree"], p["config"], operator_mode="structured", precision=precision, knobs={"struct_linear": lin})
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        h = np.concatenate([s.apgIterate(20), s.apgIterate(3), s.apgIterate(17)])      # optimistic batch, short exact batch, optimistic batch
        # step-wise entry points on top of the iterated state
        s.dualExtrapolationStep(0.5); s.solveStep()
        runs.append((h, {nm: s.get(bid) for bid, nm in VECS}, s.get(capi.BUF_PRIMAL_XI), s.profileRead))
        cls = None
        s.close()
    ho = o.apg(40)
    o.extrapolate(0.5); o.solve_step()
    for h, vecs, hx, _ in runs:
        assert np.abs(h - ho).max() <= tol * np.abs(ho).max()
        for nm in ("x", "u", "v"):
            assert relmax(vecs[nm], o.get(nm)) < tol, nm
        assert relmax(hx, o.get("primalXi")) < tol
    assert relmax(runs[0][1]["x"], runs[1][1]["x"]) < (1e-10 if precision == "f64" else 2e-4)
    assert relmax(runs[0][1]["x"], runs[2][1]["x"]) < (1e-10 if precision == "f64" else 2e-4)
    assert relmax(runs[0][1]["x"], runs[3][1]["x"]) < (1e-10 if precision == "f64" else 2e-4)
    assert relmax(runs[0][1]["v"], runs[3][1]["v"]) < (1e-10 if precision == "f64" else 2e-4)      # v_i by its own launch in the iterations that store it
    assert relmax(runs[0][1]["x"], runs[4][1]["x"]) < (1e-10 if precision == "f64" else 2e-4)
    assert relmax(runs[0][1]["u"], runs[4][1]["u"]) < (1e-10 if precision == "f64" else 2e-4)


def test_linear_form_launches_no_first_product():
    """the per-launch profile of a structured context: class 0 (the product in front of the chain walks) holds no launch in the linear form"""
    p = synth.make_problem("medium")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    for lin, launches in ((1, 0), (0, 20)):
        s = capi.Solver(p["network"], p["tree"], p["config"], operator_mode="structured", knobs={"struct_linear": lin})
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        s.apgIterate(5, history=False)
        s.profileEnable(1); s.profileReset()
        s.apgIterate(20, history=False)
        ms, n = s.profileRead()
        s.profileEnable(0)
        assert n[0] == launches and n[1] == 20 and (ms[0] > 0) == (launches > 0), (lin, ms, n)
        s.close()


@pytest.mark.parametrize("name,precision", [("medium", "f64"), ("ragged", "f64"), ("small2", "f64"), ("deep", "f64"), ("medium", "f32"), ("barcelona31", "f64")])
def test_chain_walk_riding_in_the_fused_walk_and_dual_update(name, precision):
    """Structured mode, linear form, fused walk + dual update (what the 493-scenario tree runs by default; forced here): the NEXT iteration's
    leaf-to-top running sums are formed by the same workgroup from the accelerated dual it has just computed (k_down_chain_dual UPLIN, phase C),
    the next sweep starts at its crown launch and that launch hosts the bookkeeping workgroup.  The same sums in the same order as the
    stand-alone chain walk: bitwise its iterates, histories and batch counters (struct_linear = 2: the linear form without the ride) -- and the
    oracle's.  barcelona31: the crown is the root alone (its step rides in the v / Lv launch), no crown launch to host the bookkeeping: no ride."""
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    tol = 1e-9 if precision == "f64" else 2e-4
    o = Oracle(p["network"], p["tree"], p["config"], precision=precision)
    o.initialise(dh, ah)
    ho = o.apg(45)
    runs = []
    for lin in (1, 2):
        s = capi.Solver(p["network"], p["tree"], p["config"], operator_mode="structured", precision=precision, knobs={"struct_linear": lin})
        s.setFusedWalkDual(1)
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        h = np.concatenate([s.apgIterate(20), s.apgIterate(5), s.apgIterate(20)])
        runs.append((h, {nm: s.get(bid) for bid, nm in VECS}, s.counters()))
        s.close()
    assert runs[0][2] == runs[1][2]
    assert np.array_equal(runs[0][0], runs[1][0])
    for nm in runs[0][1]:
        assert np.array_equal(runs[0][1][nm], runs[1][1][nm]), nm
        assert relmax(runs[0][1][nm], o.get(nm)) < tol, nm
    assert np.abs(runs[0][0] - ho).max() <= tol * np.abs(ho).max()


@pytest.mark.parametrize("name", ["medium", "ragged", "barcelona31"])
def test_constants_of_the_control_step_follow_the_affine_terms(name):
    """Structured mode keeps the subtree sums of beta and their share of v_i / [L v_i; B L v_i] as constants of the control step (Ctx::lin_const_refresh).
    They must follow every way the affine terms can change: a write to beta through rn_set between two batches, a new control step
    (eliminateInputDistubanceCoupling with another forecast), a new state (updateStateControl) -- against a dense context that is given exactly the
    same calls, iterate by iterate; and a context in the form that walks beta every iteration (rn_debug_set_knob struct_linear = 3)."""
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    dh1, ah1 = synth.forecast_at(p["forecast"], 1)
    ctx = [capi.Solver(p["network"], p["tree"], p["config"], operator_mode="structured"),
           capi.Solver(p["network"], p["tree"], p["config"], operator_mode="dense"),
           capi.Solver(p["network"], p["tree"], p["config"], operator_mode="structured", knobs={"struct_linear": 3})]
    outs = []
    for s in ctx:
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        h = [s.apgIterate(20)]
        s.set(capi.BUF_BETA, 1.3 * s.get(capi.BUF_BETA))            # the affine term itself, by hand
        h.append(s.apgIterate(20))
        s.eliminateInputDistubanceCoupling(dh1, ah1)                 # the next control step's forecast
        s.apgReset()
        h.append(s.apgIterate(20))
        s.updateStateControl(0.9 * np.asarray(p["config"]["currentX"], dtype=float).ravel(), np.asarray(p["config"]["prevU"], dtype=float).ravel(),
                             np.asarray(p["config"]["prevDemand"], dtype=float).ravel())
        h.append(s.apgIterate(20))
        outs.append((np.concatenate(h), s.get(capi.BUF_X), s.get(capi.BUF_U), s.get(capi.BUF_V)))
        s.close()
    for k in (1, 2):
        assert np.abs(outs[0][0] - outs[k][0]).max() <= 1e-9 * np.abs(outs[k][0]).max(), k
        for a, b in zip(outs[0][1:], outs[k][1:]):
            assert relmax(a, b) < 1e-9, k
