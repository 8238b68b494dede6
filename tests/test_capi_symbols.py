"""The C-ABI library loads without a GPU and exports exactly the symbols include/*.h declare: the drop-in boundary
(rapidnet.h) and the test hooks (rapidnet_debug.h, every one of them named rn_debug_*)."""
import os
import re

from conftest import ROOT
from rapidnet_amd import capi


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "rapidnet.h")).read()
    dbg = open(os.path.join(ROOT, "include", "rapidnet_debug.h")).read()
    product = set(re.findall(r"\b(rn_[a-z_0-9]+)\s*\(", hdr))
    hooks = set(re.findall(r"\b(rn_[a-z_0-9]+)\s*\(", dbg))
    assert not [s for s in product if s.startswith("rn_debug_")], "test hooks belong in rapidnet_debug.h"
    assert hooks and all(s.startswith("rn_debug_") for s in hooks), hooks
    declared = product | hooks
    lib = capi.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(capi.SYMBOLS), (declared ^ set(capi.SYMBOLS))


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under rapidnet_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rapidnet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".inc")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                # ("Hessian oracle" / rn_compute_hessian_oracle is the reference's name of an FBE step, not the checker)
                for needle in ("import oracle", "from oracle", "oracle/", "oracle.py", "liboracle", "apg_oracle", "oracle_"):
                    assert needle not in txt, (needle, os.path.join(dirpath, f))


def test_create_fails_loudly_without_gpu_or_with_bad_input():
    import pytest
    import torch

    from rapidnet_amd import synth

    p = synth.make_problem("tiny")
    if not torch.cuda.is_available():
        with pytest.raises(capi.RapidNetError):
            capi.Solver(p["network"], p["tree"], p["config"])


def test_build_failed_marker_of_other_sources_is_ignored(monkeypatch):
    """Ranks other than local rank 0 wait for its build and abort at once when it leaves a failure marker -- but only a marker
    written for the sources they are waiting for: one an earlier run left behind (other sources) must not fail the launch."""
    import pytest

    from rapidnet_amd import build

    capi.load()
    marker = build.LIB_HIP + ".buildfailed"
    calls = [0]

    def stale_twice():
        calls[0] += 1
        return calls[0] < 3

    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setattr(build, "hip_is_stale", stale_twice)
    try:
        open(marker, "w").write("0" * 64 + "\nan error of an earlier run")
        monkeypatch.setattr(capi, "_LIB", None)
        assert capi.load() is not None and calls[0] == 3          # waited through the stale marker
        calls[0] = 0
        open(marker, "w").write(build.hip_fingerprint() + "\nhipcc: error: boom")
        monkeypatch.setattr(capi, "_LIB", None)
        with pytest.raises(RuntimeError, match="boom"):
            capi.load()
    finally:
        if os.path.exists(marker):
            os.remove(marker)
        monkeypatch.setattr(capi, "_LIB", None)
    monkeypatch.undo()
    capi.load()
