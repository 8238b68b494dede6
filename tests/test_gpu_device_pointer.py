"""rn_device_pointer: the raw device pointers of the arrays the library keeps in the reference's own layout (the counterpart of the
reference's raw getters, Engine.cuh:108-318, and protected device vectors, SmpcController.cuh:336-462): what a device-to-host copy
from the pointer returns must be what rn_get returns, in both precisions; the scaled bounds come as node-major copies made on request; the
dual iterates refuse."""
import ctypes as C

import numpy as np
import pytest

from rapidnet_amd import capi, synth

pytestmark = pytest.mark.gpu


def _d2h(ptr, n, prec):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.empty(n, dtype=np.float64 if prec == "f64" else np.float32)
    assert hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0      # hipMemcpyDeviceToHost
    return out.astype(np.float64)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_device_pointers_hold_what_the_getters_return(precision):
    p = synth.make_problem("medium")
    s = capi.Solver(p["network"], p["tree"], p["config"], precision=precision)
    s.initialiseSmpcController(*synth.forecast_at(p["forecast"], 0))
    s.algorithmApg(30)
    s.synchronize()
    dims = {capi.BUF_X: s.nx, capi.BUF_U: s.nu, capi.BUF_V: s.nv, capi.BUF_UHAT: s.nu, capi.BUF_E: s.nx, capi.BUF_BETA: s.nv}
    for bid, dim in dims.items():
        ptr, n, prec = s.devicePointer(bid)
        assert ptr and n == s.nodes * dim and prec == precision
        assert np.array_equal(_d2h(ptr, n, prec), s.get(bid)), bid
    # the scaled bounds (Engine.cuh:294-314 getSysXmin ... getSysUmax): node-major copies made by the first request, refreshed by a factor step
    bounds = {capi.BUF_XMIN: s.nx, capi.BUF_XMAX: s.nx, capi.BUF_XS: s.nx, capi.BUF_UMIN: s.nu, capi.BUF_UMAX: s.nu}
    mem0 = s.deviceMemoryInfo()["context_bytes"]
    ptrs = {}
    for bid, dim in bounds.items():
        ptr, n, prec = s.devicePointer(bid)
        assert ptr and n == s.nodes * dim and prec == precision
        assert np.array_equal(_d2h(ptr, n, prec), s.get(bid)), bid
        ptrs[bid] = ptr
    assert s.deviceMemoryInfo()["context_bytes"] == mem0 + s.nodes * (3 * s.nx + 2 * s.nu) * (8 if precision == "f64" else 4)      # made once, all five
    s.network["vecUmax"] = [1.5 * v for v in np.asarray(s.network["vecUmax"], float).ravel()]
    s.factorStep()
    s.synchronize()
    for bid, dim in bounds.items():
        ptr, n, prec = s.devicePointer(bid)
        assert ptr == ptrs[bid] and np.array_equal(_d2h(ptr, n, prec), s.get(bid)), bid     # same arrays, new contents
    for bid in (capi.BUF_XI, capi.BUF_UPD_PSI, capi.BUF_PRIMAL_XI):      # kept interleaved [node][2nx + nu]: not the reference's layout
        with pytest.raises(capi.RapidNetError, match="not kept in the reference's layout"):
            s.devicePointer(bid)
    s.close()
