"""The shared-operator products v = m1 - RT [s; kappa] / (2p) and [L v; B L v] have three forms: k_gemm_vlv with the
software-pipelined MFMA loop (at most one slab workgroup per CU, and fp32), k_gemm_vlv with the lean loop, and k_gemm_vlv_wide
(2 or 3 slabs per workgroup, every operator fragment used for all of them: trees with more than four slabs per CU).
Which one a context takes is decided by the tree's size; every output element is the same chain of MFMAs over k in all of
them, so forcing the wide kernel or the other loop on a small tree (rn_debug_set_knob) must reproduce the default bit for bit.
(The lean loop and the wide kernel at their own sizes are covered by the full-size tests: tests/test_gpu_fullsize.py,
tests/test_gpu_baseline_configs.py.  The LDS-staged and register-resident forms of round 5 measured slower and were removed in
round 6: profiles/r06_pruned_forms.md.)"""
import numpy as np
import pytest

from rapidnet_amd import capi, synth

pytestmark = pytest.mark.gpu
synth.CONFIGS.setdefault("wide16", (4, 200, 360, 280, 54, 24, [4, 2, 2]))   # the wide network (nv = 306: 20 row tiles), 16 scenarios
BUFS = (capi.BUF_X, capi.BUF_U, capi.BUF_V, capi.BUF_UPD_XI, capi.BUF_UPD_PSI, capi.BUF_PRIMAL_XI, capi.BUF_PRIMAL_PSI)


def run(p, structured, precision, n=40, knobs=None):
    s = capi.Solver(p["network"], p["tree"], p["config"], structured=structured, precision=precision, knobs=knobs)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    s.initialiseSmpcController(dh, ah)
    s.apgReset()
    hist = s.apgIterate(n)
    out = {b: s.get(b) for b in BUFS}
    kernel = s.kernelInfo()
    s.close()
    return hist, out, kernel


def same(a, b):
    assert np.array_equal(a[0], b[0])
    for k in BUFS:
        assert np.array_equal(a[1][k], b[1][k]), k


@pytest.mark.parametrize("name,structured,precision", [("medium", False, "f64"), ("medium", True, "f64"), ("ragged", False, "f64"),
                                                       ("barcelona31", False, "f64"), ("medium", False, "f32"), ("wide16", False, "f32")])
@pytest.mark.parametrize("ct", [2, 3])
def test_wide_slab_kernel_is_bitwise_the_default(name, structured, precision, ct):
    p = synth.make_problem(name)
    same(run(p, structured, precision), run(p, structured, precision, knobs={"vlv_wide": ct}))


@pytest.mark.parametrize("name,structured,precision", [("medium", False, "f64"), ("barcelona31", True, "f64"), ("medium", False, "f32")])
def test_lean_and_pipelined_mfma_loops_are_bitwise_equal(name, structured, precision):
    """slab_mfma: the software-pipelined loop (few slabs, fp32) and the lean one (more slabs than CUs) feed the same operands into the same
    chain of MFMAs."""
    p = synth.make_problem(name)
    same(run(p, structured, precision, knobs={"slab_pipe": 0}), run(p, structured, precision, knobs={"slab_pipe": 1}))


@pytest.mark.parametrize("name,structured,precision,wide", [("medium", False, "f64", None), ("medium", True, "f64", None), ("ragged", False, "f64", None),
                                                            ("barcelona31", True, "f64", None), ("medium", False, "f32", None), ("wide16", False, "f32", 3),
                                                            ("wide16", True, "f32", None), ("barcelona493", True, "f64", None)])
def test_fragment_ordered_operands_are_bitwise_the_column_major_ones(name, structured, precision, wide):
    """RN_KNOB_SLAB_FRAG (default on): the slab products take their A operands from the operators' fragment-ordered copies (GemmArgs::Mf: one
    contiguous 16-byte request per lane and pair of k-steps) instead of the column-major ones (two strided 8-byte requests) -- the same
    operands into the same chain of MFMAs, so the same bits: the software-pipelined loop (small trees), the wide kernel (forced on wide16)
    and, on the whole 493-scenario tree in structured mode (no per-node blocks: cheap), the lean loop of k_gemm_vlv and k_gemm_prep_m2."""
    p = synth.make_problem(name)
    n = 12 if name == "barcelona493" else 40
    extra = {"vlv_wide": wide} if wide else {}
    same(run(p, structured, precision, n, knobs=dict(extra, slab_frag=0)), run(p, structured, precision, n, knobs=dict(extra, slab_frag=1)))
