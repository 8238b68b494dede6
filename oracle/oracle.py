"""ctypes wrapper around oracle/apg_oracle.c -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's APG path (see the header of apg_oracle.c for the
reference file:line map).  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this module; nothing under rapidnet_amd/ does.

Inputs are plain dicts in the reference's JSON schema (scalars are 1-element lists):
  network   : nx nu nd ne matA matB matGd matE matEd vecXmin vecXmax vecXsafe vecUmin vecUmax costAlpha1
  tree      : N K nodes nNonLeafNodes nChildrenTot stages nodesPerStage nodesPerStageCumul leaves children
              ancestor nChildren nChildrenCumul probNode dimDemand dimPrice errorDemandNode errorPriceNode
  config    : nx nu nd nv N matL matLhat costW penaltyStateX penaltySafetyX matDiagPrecnd currentX prevU
              prevDemand stepSize maxIterations ...
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False, march=None, variant=""):
    """Compile apg_oracle.c into liboracle_f64<variant>.so / liboracle_f32<variant>.so next to this file.

    Default -march=x86-64-v3 so that a library built in one container runs on another host.  bench.py's cpu_baseline
    leg builds a SEPARATE pair (variant="_native", -march=native) on the machine it times, so the portable test oracle
    is never overwritten by a host-specific binary."""
    march = march or os.environ.get("ORACLE_MARCH", "x86-64-v3")
    src = os.path.join(_HERE, "apg_oracle.c")
    for tag, define in (("f64", []), ("f32", ["-DORACLE_REAL=float"])):
        out = os.path.join(_HERE, "liboracle_%s%s.so" % (tag, variant))
        if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            tmp = out + ".tmp.%d" % os.getpid()
            cmd = ["gcc", "-O3", "-march=" + march, "-fPIC", "-shared", "-std=c99"] + define + ["-o", tmp, src, "-lm"]
            subprocess.check_call(cmd)
            os.replace(tmp, out)
    return _HERE


def _lib(precision, variant=""):
    tag = {"f64": "f64", "f32": "f32"}[precision] + variant
    if tag not in _LIBS:
        path = os.path.join(_HERE, "liboracle_%s.so" % tag)
        src = os.path.join(_HERE, "apg_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            try:
                build(variant=variant, march="native" if variant == "_native" else None)
            except Exception:  # a stale binary still loads; rebuild is best effort
                if not os.path.exists(path):
                    raise
        lib = C.CDLL(path)
        lib.oracle_create.restype = C.c_void_p
        lib.oracle_create.argtypes = [C.c_int] * 8 + [C.c_void_p] * 7
        lib.oracle_destroy.argtypes = [C.c_void_p]
        lib.oracle_set_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        lib.oracle_factor_step.restype = C.c_int
        lib.oracle_factor_step.argtypes = [C.c_void_p] * 13
        lib.oracle_update_state_control.argtypes = [C.c_void_p] * 4
        lib.oracle_eliminate.argtypes = [C.c_void_p] * 5 + [C.c_double, C.c_int, C.c_int]
        for fn in ("oracle_apg_reset", "oracle_solve_step", "oracle_prox", "oracle_residual", "oracle_dual_update"):
            getattr(lib, fn).argtypes = [C.c_void_p]
        lib.oracle_extrapolate.argtypes = [C.c_void_p, C.c_double]
        lib.oracle_solve_step_phase.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.oracle_primal_infeasibility.restype = C.c_double
        lib.oracle_primal_infeasibility.argtypes = [C.c_void_p]
        lib.oracle_apg.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.oracle_apg_continue.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.oracle_buffer.restype = C.c_void_p
        lib.oracle_buffer.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_long)]
        lib.oracle_final_branch_node.argtypes = [C.c_void_p]
        lib.oracle_dist.restype = C.c_double
        lib.oracle_dist.argtypes = [C.c_void_p, C.c_int]
        lib.oracle_set_algorithm.argtypes = [C.c_void_p, C.c_int, C.c_int]
        for fn in ("oracle_fbe_reset", "oracle_hessian_oracle", "oracle_gradient_fbe", "oracle_nama_residual",
                   "oracle_lbfgs_direction"):
            getattr(lib, fn).argtypes = [C.c_void_p]
        lib.oracle_value_fbe.restype = C.c_double
        lib.oracle_value_fbe.argtypes = [C.c_void_p]
        for fn in ("oracle_line_search_fbe", "oracle_line_search_ame"):
            getattr(lib, fn).restype = C.c_double
            getattr(lib, fn).argtypes = [C.c_void_p, C.c_double]
        lib.oracle_fbe_nama.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_lbfgs_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                           C.POINTER(C.c_double)]
        lib.oracle_control_action.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.oracle_move_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_kpi.restype = C.c_double
        lib.oracle_kpi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _LIBS[tag] = lib
    return _LIBS[tag]


def _scalar(d, key):
    v = d[key]
    return v[0] if isinstance(v, (list, tuple, np.ndarray)) else v


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel().astype(np.int32))


def load_json(path):
    with open(path) as f:
        return json.load(f)


def forecast_at(forecast, sim_time):
    """Forecaster::predictDemand / predictPrices (Forecaster.cu:93-119): members 4+2t and 5+2t in file order."""
    keys = list(forecast.keys())
    return _f64(forecast[keys[4 + 2 * sim_time]]), _f64(forecast[keys[5 + 2 * sim_time]])


class Oracle:
    """CPU oracle for one (network, tree, config) triple."""

    def __init__(self, network, tree, config, precision="f64", alias_operators=True, variant=""):
        self.lib = _lib(precision, variant)
        self.lib.oracle_config_aliasing(1 if alias_operators else 0)
        self.dtype = np.float64 if precision == "f64" else np.float32
        self.network, self.tree, self.config = network, tree, config
        self.nx, self.nu, self.nd = (int(_scalar(network, k)) for k in ("nx", "nu", "nd"))
        self.nv = int(_scalar(config, "nv"))
        self.N, self.K, self.nodes = (int(_scalar(tree, k)) for k in ("N", "K", "nodes"))
        self.n_nonleaf = int(_scalar(tree, "nNonLeafNodes"))
        self._keep = [_i32(tree[k]) for k in ("stages", "nodesPerStage", "nodesPerStageCumul", "ancestor",
                                              "nChildren", "nChildrenCumul")]
        nps, npsc = self._keep[1], self._keep[2]
        # the loaders need N+1 / N+2 entries (ScenarioTree.cu:66-75 reads whatever the JSON carries)
        assert len(nps) >= self.N + 1 and len(npsc) >= self.N + 2, "nodesPerStage needs N+1, Cumul N+2 entries"
        prob = _f64(tree["probNode"])
        self._keep.append(prob)
        self.h = self.lib.oracle_create(self.nx, self.nu, self.nv, self.nd, self.N, self.K, self.nodes, self.n_nonleaf,
                                        *[a.ctypes.data for a in self._keep])
        self.lib.oracle_set_params(self.h, float(_scalar(config, "stepSize")), float(_scalar(config, "penaltyStateX")),
                                   float(_scalar(config, "penaltySafetyX")))
        self.max_iterations = int(_scalar(config, "maxIterations"))
        self.err_demand = _f64(tree["errorDemandNode"])
        self.err_price = _f64(tree["errorPriceNode"])

    def __del__(self):
        try:
            self.lib.oracle_destroy(self.h)
        except Exception:
            pass

    # --- Engine ---------------------------------------------------------------------------------------
    def factor_step(self):
        n, c = self.network, self.config
        args = [_f64(n["matB"]), _f64(c["matL"]), _f64(c["matLhat"]), _f64(n["matGd"]), _f64(c["costW"]),
                _f64(c["matDiagPrecnd"]), _f64(n["vecXmin"]), _f64(n["vecXmax"]), _f64(n["vecXsafe"]),
                _f64(n["vecUmin"]), _f64(n["vecUmax"]), _f64(n["costAlpha1"])]
        rc = self.lib.oracle_factor_step(self.h, *[a.ctypes.data for a in args])
        if rc:
            raise RuntimeError("factor step: p*Rbar singular at node %d" % (rc - 1))

    def update_state_control(self, x0=None, u_prev=None, d_prev=None):
        c = self.config
        a = [_f64(c["currentX"] if x0 is None else x0), _f64(c["prevU"] if u_prev is None else u_prev),
             _f64(c["prevDemand"] if d_prev is None else d_prev)]
        self.lib.oracle_update_state_control(self.h, *[v.ctypes.data for v in a])

    def eliminate(self, nominal_demand, nominal_prices, weight_economical=1.0, demand_uncertainty=True,
                  price_uncertainty=True):
        dh, ah = _f64(nominal_demand), _f64(nominal_prices)
        self.lib.oracle_eliminate(self.h, dh.ctypes.data, ah.ctypes.data, self.err_demand.ctypes.data,
                                  self.err_price.ctypes.data, float(weight_economical), int(demand_uncertainty),
                                  int(price_uncertainty))

    def initialise(self, nominal_demand, nominal_prices):
        """SmpcController::initialiseSmpcController, SmpcController.cu:476-487."""
        self.factor_step()
        self.update_state_control()
        self.eliminate(nominal_demand, nominal_prices)

    # --- SmpcController -------------------------------------------------------------------------------
    def apg_reset(self):
        self.lib.oracle_apg_reset(self.h)

    def extrapolate(self, lam):
        self.lib.oracle_extrapolate(self.h, float(lam))

    def solve_step(self):
        self.lib.oracle_solve_step(self.h)

    def solve_step_phase(self, phase, cut_stage):
        """tests only: backward sweep down to the cut (phase 0) / the rest (phase 1); see oracle_solve_step_phase."""
        self.lib.oracle_solve_step_phase(self.h, int(phase), int(cut_stage))

    def prox(self):
        self.lib.oracle_prox(self.h)

    def residual(self):
        self.lib.oracle_residual(self.h)

    def dual_update(self):
        self.lib.oracle_dual_update(self.h)

    def primal_infeasibility(self):
        return self.lib.oracle_primal_infeasibility(self.h)

    def apg(self, iters=None):
        iters = self.max_iterations if iters is None else int(iters)
        hist = np.zeros(max(iters, 1), dtype=np.float64)
        self.lib.oracle_apg(self.h, iters, hist.ctypes.data)
        return hist[:iters]

    def apg_continue(self, iters, theta):
        th = np.asarray(theta, dtype=np.float64).copy()
        self.lib.oracle_apg_continue(self.h, int(iters), th.ctypes.data)
        return th

    def dist(self):
        return self.lib.oracle_dist(self.h, 0), self.lib.oracle_dist(self.h, 1)

    # --- global FBE / NAMA (SmpcController.cu:884-1476, 1529-1586) ------------------------------------
    ALGORITHMS = {"proximalAlgorithm": 0, "globalFbeAlgorithm": 1, "namaAlgorithm": 2}  # Engine.cu:151-163

    def set_algorithm(self, name, lbfgs_buffer_size=None):
        if lbfgs_buffer_size is None:
            lbfgs_buffer_size = int(_scalar(self.config, "lbfgsBufferSize")) if "lbfgsBufferSize" in self.config else 5
        self.algorithm = self.ALGORITHMS[name]
        self.lbfgs_size = int(lbfgs_buffer_size)
        self.lib.oracle_set_algorithm(self.h, self.algorithm, self.lbfgs_size)

    def fbe_reset(self):
        self.lib.oracle_fbe_reset(self.h)

    def hessian_oracle(self):
        self.lib.oracle_hessian_oracle(self.h)

    def gradient_fbe(self):
        self.lib.oracle_gradient_fbe(self.h)

    def nama_residual(self):
        self.lib.oracle_nama_residual(self.h)

    def lbfgs_direction(self):
        self.lib.oracle_lbfgs_direction(self.h)

    def value_fbe(self):
        return self.lib.oracle_value_fbe(self.h)

    def line_search_fbe(self, value_y):
        return self.lib.oracle_line_search_fbe(self.h, float(value_y))

    def line_search_ame(self, value_y):
        return self.lib.oracle_line_search_ame(self.h, float(value_y))

    def lbfgs_state(self, col=None, mem=None, H=None):
        """get (no arguments) or set (col, mem, H) lbfgsBufferCol / lbfgsBufferMemory / lbfgsBufferHessian."""
        c, m, h = C.c_int(0 if col is None else col), C.c_int(0 if mem is None else mem), C.c_double(0 if H is None else H)
        self.lib.oracle_lbfgs_state(self.h, 0 if col is None else 1, C.byref(c), C.byref(m), C.byref(h))
        return c.value, m.value, h.value

    def fbe_nama(self, iters=None):
        """algorithmGlobalFbe / algorithmNama: returns (vecPrimalInfs, vecValueFbe, vecTau)."""
        iters = self.max_iterations if iters is None else int(iters)
        hist, val, tau = (np.zeros(max(iters, 1)) for _ in range(3))
        self.lib.oracle_fbe_nama(self.h, iters, hist.ctypes.data, val.ctypes.data, tau.ctypes.data)
        return hist[:iters], val[: max(iters - 1, 0)], tau[: max(iters - 1, 0)]

    # --- closed loop (SmpcController.cu:1607-1716, 1778-1859) -----------------------------------------
    def control_action(self, nominal_demand, nominal_prices, max_iterations=None, project=True, **elim):
        """controlAction(fstream&) (project=True) / controlAction(real_t*) (project=False) at the oracle's current
        state / previous control / previous demand."""
        a = [np.array(self.buf(k), dtype=np.float64) for k in ("curX", "prevU", "prevD")]
        self.update_state_control(*a)
        self.eliminate(nominal_demand, nominal_prices, **elim)
        n = self.max_iterations if max_iterations is None else int(max_iterations)
        u = np.zeros(self.nu)
        self.lib.oracle_control_action(self.h, n, int(project), u.ctypes.data)
        return u

    def move_forward(self, nominal_demand, nominal_prices, weight_economical=1.0, plant_mode=0):
        """moveForewardInTime with the in-built simulator; returns the shifted (currentX, prevU, prevDemand)."""
        dh, ah, xs = _f64(nominal_demand), _f64(nominal_prices), _f64(self.network["vecXsafe"])
        x, u, d = np.zeros(self.nx), np.zeros(self.nu), np.zeros(self.nd)
        self.lib.oracle_move_forward(self.h, dh.ctypes.data, ah.ctypes.data, xs.ctypes.data, float(weight_economical),
                                     int(plant_mode), x.ctypes.data, u.ctypes.data, d.ctypes.data)
        return x, u, d

    def kpis(self, simulation_time):
        """(economic, smooth, network, safety) KPI read-outs."""
        xs = _f64(self.network["vecXsafe"])
        return tuple(self.lib.oracle_kpi(self.h, w, int(simulation_time), xs.ctypes.data) for w in range(4))

    @property
    def final_branch_node(self):
        return self.lib.oracle_final_branch_node(self.h)

    def buf(self, name):
        """numpy VIEW (read/write) of one of the oracle's arrays."""
        cnt = C.c_long(0)
        p = self.lib.oracle_buffer(self.h, name.encode(), C.byref(cnt))
        if not p:
            raise KeyError(name)
        ct = C.c_double if self.dtype == np.float64 else C.c_float
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(cnt.value,))

    def get(self, name):
        return np.array(self.buf(name), dtype=np.float64)

    def set(self, name, values):
        b = self.buf(name)
        v = np.asarray(values, dtype=np.float64).ravel()
        assert v.size == b.size, (name, v.size, b.size)
        b[:] = v.astype(self.dtype)
