"""The shared-operator products v = m1 - RT [s; kappa] / (2p) and [L v; B L v] have four forms: k_gemm_vlv with the
software-pipelined MFMA loop (at most one slab workgroup per CU, and fp32), k_gemm_vlv with the lean loop, k_gemm_vlv_wide
(2 or 3 slabs per workgroup, every operator fragment used for all of them) and -- round 5, the default for launches with more slabs
than CUs -- k_gemm_vlv_lds / k_gemm_prep_m2_lds (operator chunks copied into an LDS ring by loader waves, A and B fragments from LDS).
Which one a context takes is decided by the tree's size; every output element is the same chain of MFMAs over k in all of
them, so forcing the wide kernel on a small tree must reproduce the default bit for bit.  (The lean loop and the wide kernel
at their own sizes are covered by the full-size tests: tests/test_gpu_fullsize.py, tests/test_gpu_baseline_configs.py.)"""
import numpy as np
import pytest

from rapidnet_amd import capi, synth

pytestmark = pytest.mark.gpu
synth.CONFIGS.setdefault("wide16", (4, 200, 360, 280, 54, 24, [4, 2, 2]))   # the wide network (nv = 306: 20 row tiles), 16 scenarios
BUFS = (capi.BUF_X, capi.BUF_U, capi.BUF_V, capi.BUF_UPD_XI, capi.BUF_UPD_PSI, capi.BUF_PRIMAL_XI, capi.BUF_PRIMAL_PSI)


def run(p, structured, precision, n=40):
    s = capi.Solver(p["network"], p["tree"], p["config"], structured=structured, precision=precision)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    s.initialiseSmpcController(dh, ah)
    s.apgReset()
    hist = s.apgIterate(n)
    out = {b: s.get(b) for b in BUFS}
    kernel = s.kernelInfo()
    s.close()
    return hist, out, kernel


@pytest.mark.parametrize("name,structured,precision", [("medium", False, "f64"), ("medium", True, "f64"), ("ragged", False, "f64"),
                                                       ("barcelona31", False, "f64"), ("medium", False, "f32"), ("wide16", False, "f32")])
@pytest.mark.parametrize("ct", [2, 3])
def test_wide_slab_kernel_is_bitwise_the_default(monkeypatch, name, structured, precision, ct):
    p = synth.make_problem(name)
    monkeypatch.delenv("RAPIDNET_VLV_WIDE", raising=False)
    h0, o0, _ = run(p, structured, precision)
    monkeypatch.setenv("RAPIDNET_VLV_WIDE", str(ct))     # read when a context launches its first sweep
    h1, o1, _ = run(p, structured, precision)
    assert np.array_equal(h0, h1)
    for b in BUFS:
        assert np.array_equal(o0[b], o1[b]), b


@pytest.mark.parametrize("name,structured,precision", [("medium", False, "f64"), ("medium", True, "f64"), ("ragged", False, "f64"),
                                                       ("barcelona31", False, "f64"), ("barcelona31", True, "f64"), ("medium", False, "f32"),
                                                       ("wide16", False, "f32"), ("wide16", True, "f32")])
def test_lds_staged_slab_kernels_are_bitwise_the_default(monkeypatch, name, structured, precision):
    """RAPIDNET_SLAB_LDS=1 forces the LDS-staged products on trees that would not take them by themselves (fewer slabs than CUs):
    same chain of MFMAs over k for every output element => the same bits.  wide16: operators of 20 and 35 row tiles -- several
    passes of 16 tiles, MFMA waves with two tiles each, the second tile's epilogue operands fetched late."""
    p = synth.make_problem(name)
    monkeypatch.setenv("RAPIDNET_SLAB_LDS", "0")
    h0, o0, _ = run(p, structured, precision)
    monkeypatch.setenv("RAPIDNET_SLAB_LDS", "1")
    h1, o1, _ = run(p, structured, precision)
    assert np.array_equal(h0, h1)
    for b in BUFS:
        assert np.array_equal(o0[b], o1[b]), b


synth.CONFIGS.setdefault("barcelona64", (1, 63, 114, 88, 17, 24, [8, 8]))   # the Barcelona network on 64 chains: 1 417 nodes = 89 slabs


@pytest.mark.parametrize("variant", ["1", "2"])
@pytest.mark.parametrize("name,structured,precision,grid", [("barcelona31", False, "f64", 0), ("barcelona31", True, "f64", 7), ("barcelona64", False, "f64", 7),
                                                            ("barcelona64", False, "f64", 32), ("barcelona64", False, "f32", 5)])
def test_register_resident_slab_kernel_is_bitwise_the_default(monkeypatch, name, structured, precision, grid, variant):
    """k_gemm_vlv_reg (round 5: the operator tiles live in the waves' registers for the whole launch, persistent workgroups walk the
    slabs; the default on the 493-scenario tree) forced on small trees of the same network: RAPIDNET_SLAB_REG_GRID caps the grid so
    that a workgroup walks several slabs (89 slabs on 7 workgroups: 12 or 13 each, both halves of the double-buffered slab in use,
    the last slab partial).  Same chain of MFMAs over k for every output element => the same bits as k_gemm_vlv."""
    p = synth.make_problem(name)
    monkeypatch.setenv("RAPIDNET_SLAB_REG", "0")
    h0, o0, _ = run(p, structured, precision)
    monkeypatch.setenv("RAPIDNET_SLAB_REG", variant)      # 1: four waves x the whole register file; 2: eight waves, LDS-DMA staging
    if grid:
        monkeypatch.setenv("RAPIDNET_SLAB_REG_GRID", str(grid))
    h1, o1, _ = run(p, structured, precision)
    assert np.array_equal(h0, h1)
    for b in BUFS:
        assert np.array_equal(o0[b], o1[b]), b


@pytest.mark.parametrize("name,structured,precision,wide", [("medium", False, "f64", None), ("medium", True, "f64", None), ("ragged", False, "f64", None),
                                                            ("barcelona31", True, "f64", None), ("medium", False, "f32", None), ("wide16", False, "f32", "3"),
                                                            ("wide16", True, "f32", None), ("barcelona493", True, "f64", None)])
def test_fragment_ordered_operands_are_bitwise_the_column_major_ones(monkeypatch, name, structured, precision, wide):
    """RAPIDNET_SLAB_FRAG (default 1): the slab products take their A operands from the operators' fragment-ordered copies (GemmArgs::Mf: one
    contiguous 16-byte request per lane and pair of k-steps) instead of the column-major ones (two strided 8-byte requests) -- the same
    operands into the same chain of MFMAs, so the same bits: the software-pipelined loop (small trees), the wide kernel (forced on wide16)
    and, on the whole 493-scenario tree in structured mode (no per-node blocks: cheap), the lean loop of k_gemm_vlv and k_gemm_prep_m2."""
    p = synth.make_problem(name)
    if wide:
        monkeypatch.setenv("RAPIDNET_VLV_WIDE", wide)
    n = 12 if name == "barcelona493" else 40
    monkeypatch.setenv("RAPIDNET_SLAB_FRAG", "0")
    h0, o0, _ = run(p, structured, precision, n)
    monkeypatch.setenv("RAPIDNET_SLAB_FRAG", "1")
    h1, o1, _ = run(p, structured, precision, n)
    assert np.array_equal(h0, h1)
    for b in BUFS:
        assert np.array_equal(o0[b], o1[b]), b
