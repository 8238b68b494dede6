"""ctypes binding of include/rapidnet.h (librapidnet_hip.so) and a thin Python controller on top of it.

There is no CPU fallback: if the HIP library is missing or fails to load this module raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

RN_F32, RN_F64 = 0, 1
(BUF_X, BUF_U, BUF_V, BUF_XI, BUF_PSI, BUF_ACC_XI, BUF_ACC_PSI, BUF_UPD_XI, BUF_UPD_PSI, BUF_PRIMAL_XI,
 BUF_PRIMAL_PSI, BUF_DUAL_XI, BUF_DUAL_PSI, BUF_RES_XI, BUF_RES_PSI, BUF_UHAT, BUF_E, BUF_BETA, BUF_ALPHA,
 BUF_XMIN, BUF_XMAX, BUF_XS, BUF_UMIN, BUF_UMAX,
 BUF_PREV_XI, BUF_PREV_PSI, BUF_LBFGS_CUR_YVEC_XI, BUF_LBFGS_CUR_YVEC_PSI, BUF_LBFGS_PREV_YVEC_XI,
 BUF_LBFGS_PREV_YVEC_PSI, BUF_LBFGS_DIR_XI, BUF_LBFGS_DIR_PSI, BUF_PRIMAL_XI_DIR, BUF_PRIMAL_PSI_DIR,
 BUF_XDIR, BUF_UDIR) = range(36)
ALG_APG, ALG_GLOBAL_FBE, ALG_NAMA = 0, 1, 2
ALGORITHMS = {"proximalAlgorithm": ALG_APG, "globalFbeAlgorithm": ALG_GLOBAL_FBE, "namaAlgorithm": ALG_NAMA}  # Engine.cu:151-163
OP_PHI, OP_PSI, OP_D, OP_F, OP_OMEGA, OP_THETA, OP_G = range(7)
OPS_DENSE, OPS_STRUCTURED, OPS_AUTO = 0, 1, 2
OPS_MODES = {"dense": OPS_DENSE, "structured": OPS_STRUCTURED, "auto": OPS_AUTO}
# include/rapidnet_debug.h, RN_KNOB_*
KNOBS = {k: i for i, k in enumerate(("dual_trips", "dual_pipe", "vlv_wide", "slab_pipe", "slab_frag", "unscaled_walk", "stream_two_per_cu",
                                     "stream_split", "nama_pair", "ls_sequential", "value_mfma", "tune_bias_us", "struct_linear", "fuse_split"))}
EXCHANGE_COLLECTIVE, EXCHANGE_ONESHOT, EXCHANGE_AUTO = 0, 1, 2

# every symbol include/rapidnet.h (the boundary) and include/rapidnet_debug.h (test hooks, rn_debug_*) declare
SYMBOLS = [
    "rn_create", "rn_destroy", "rn_last_error", "rn_synchronize", "rn_factor_step", "rn_set_tree_errors",
    "rn_set_uncertainty", "rn_update_state_control", "rn_eliminate_input_disturbance_coupling", "rn_set_parameters",
    "rn_apg_reset", "rn_apg_iterate", "rn_algorithm_apg", "rn_control_action", "rn_dual_extrapolation_step",
    "rn_solve_step", "rn_proximal_fun_g", "rn_compute_fixed_point_residual", "rn_dual_update",
    "rn_update_primal_infeasibility", "rn_get_prox_distances", "rn_buffer_size", "rn_get", "rn_set", "rn_get_operator",
    "rn_device_pointer", "rn_profile_enable", "rn_profile_reset", "rn_profile_read", "rn_algorithmic_bytes", "rn_stream",
    "rn_comm_unique_id", "rn_comm_init", "rn_comm_init_timeout", "rn_comm_check", "rn_comm_library", "rn_set_cut_stage", "rn_get_history_parts", "rn_get_counters", "rn_debug_sweep_phase",
    "rn_debug_cut_buffer", "rn_set_cut_children_moments", "rn_set_operator_mode", "rn_get_operator_mode", "rn_set_operator", "rn_set_warm_start", "rn_set_exchange_mode",
    "rn_measure_hbm", "rn_set_algorithm", "rn_fbe_reset", "rn_algorithm_fbe_nama", "rn_compute_hessian_oracle", "rn_compute_gradient_fbe",
    "rn_update_fixed_point_residual_nama", "rn_compute_lbfgs_direction", "rn_update_lbfgs_buffer", "rn_two_loop_recursion_lbfgs", "rn_compute_value_fbe",
    "rn_line_search_lbfgs_update", "rn_line_search_ame_lbfgs_update", "rn_lbfgs_state", "rn_lbfgs_column",
    "rn_get_range", "rn_set_range", "rn_get_kernel_info", "rn_default_cut_stage", "rn_partition_create", "rn_partition_destroy", "rn_create_sharded", "rn_shard_info", "rn_shard_global_nodes",
    "rn_debug_set_allreduce", "rn_debug_local_group_create", "rn_debug_local_group_join", "rn_debug_local_group_destroy",
    "rn_guard_check", "rn_device_memory_info", "rn_reserve_iterations", "rn_profile_read_collective", "rn_debug_inject_allocation", "rn_guard_report", "rn_debug_guard_poke",
    "rn_fbe_counters", "rn_peer_inbox_create", "rn_peer_inbox_connect", "rn_debug_peer_inbox_connect_local", "rn_debug_peer_seq", "rn_set_exchange_transport", "rn_exchange_autotune", "rn_set_fused_walk_dual", "rn_debug_set_knob",
]


class RnDims(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("nx", "nu", "nv", "nd", "N", "K", "nodes", "nNonLeafNodes")]


class RnTree(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("stages", "nodesPerStage", "nodesPerStageCumul", "ancestor", "nChildren",
                                         "nChildrenCumul", "probNode")]


class RnSystem(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("matB", "matGd", "matL", "matLhat", "costW", "matDiagPrecnd", "vecXmin",
                                         "vecXmax", "vecXsafe", "vecUmin", "vecUmax", "costAlpha1")]


class RnPartition(C.Structure):
    _fields_ = [("dims", RnDims), ("tree", RnTree), ("globalNode", C.POINTER(C.c_int)), ("errorDemandNode", C.POINTER(C.c_double)),
                ("errorPriceNode", C.POINTER(C.c_double)), ("rank", C.c_int), ("nranks", C.c_int), ("cutStage", C.c_int),
                ("nCutParents", C.c_int), ("momE", C.POINTER(C.c_double)), ("momP", C.POINTER(C.c_double)), ("owner", C.c_void_p)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)

_LIB = None


def lib_path():
    return os.environ.get("RAPIDNET_LIB") or _build.LIB_HIP


def load():
    """Load librapidnet_hip.so (must have been built: `python -m rapidnet_amd.build`)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if os.environ.get("RAPIDNET_LIB"):
        if not os.path.exists(path):
            raise RuntimeError("$RAPIDNET_LIB names a missing library: %s" % path)
    else:
        # (re)build in-tree when the library is missing or older than its sources (a no-op otherwise; hipcc is part of the
        # image).  Under a launcher only local rank 0 compiles, before any GPU call; the others wait for the finished file
        # (build_hip renames it into place).  There is no CPU fallback.
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # left by local rank 0 when its build fails, so that the waiting ranks abort at once.  Its first line is the fingerprint of
        # the sources that failed to build: a marker an EARLIER run left behind for other sources is ignored (the waiting ranks may
        # look before local rank 0 has removed it)
        failed = path + ".buildfailed"
        try:
            if local_rank == 0:
                if os.path.exists(failed):
                    os.remove(failed)
                try:
                    _build.build_hip()
                except Exception as e:
                    with open(failed, "w") as f:
                        f.write("%s\n%s" % (_build.hip_fingerprint(), e))
                    raise
            else:
                import time

                for _ in range(1800):
                    if not _build.hip_is_stale():
                        break
                    if os.path.exists(failed):
                        try:
                            stamp, _, why = open(failed).read().partition("\n")
                        except OSError:      # removed between the two calls
                            stamp, why = "", ""
                        if stamp.strip() == _build.hip_fingerprint():
                            raise RuntimeError("local rank 0 could not build librapidnet_hip.so: %s" % why.strip())
                    time.sleep(0.5)
                else:   # every rank must run the same binary: a library that is still stale after 15 minutes is an error, not a fallback
                    raise RuntimeError("librapidnet_hip.so is still older than its sources after waiting 900 s for local rank 0 to rebuild it")
        except Exception as e:
            if local_rank != 0 or not os.path.exists(path) or int(os.environ.get("WORLD_SIZE", "1")) > 1:   # ranks never run different binaries
                raise RuntimeError("librapidnet_hip.so (%s) is missing or stale and could not be built (%s); run __graft_entry__.build() -- "
                                   "there is no CPU fallback" % (path, e))
            import warnings

            warnings.warn("librapidnet_hip.so could not be rebuilt (%s); loading the existing (possibly stale) binary" % e)
    lib = C.CDLL(path)
    vp, dp, ip = C.c_void_p, C.c_void_p, C.c_int
    lib.rn_create.argtypes = [C.POINTER(RnDims), C.POINTER(RnTree), ip, ip, C.POINTER(vp)]
    lib.rn_destroy.argtypes = [vp]
    lib.rn_last_error.argtypes = [vp]
    lib.rn_last_error.restype = C.c_char_p
    lib.rn_synchronize.argtypes = [vp]
    lib.rn_factor_step.argtypes = [vp, C.POINTER(RnSystem)]
    lib.rn_set_tree_errors.argtypes = [vp, dp, dp]
    lib.rn_set_uncertainty.argtypes = [vp, ip, ip, C.c_double]
    lib.rn_update_state_control.argtypes = [vp, dp, dp, dp]
    lib.rn_eliminate_input_disturbance_coupling.argtypes = [vp, dp, dp]
    lib.rn_set_parameters.argtypes = [vp, C.c_double, C.c_double, C.c_double]
    lib.rn_apg_reset.argtypes = [vp]
    lib.rn_apg_iterate.argtypes = [vp, ip, dp]
    lib.rn_algorithm_apg.argtypes = [vp, ip, dp]
    lib.rn_control_action.argtypes = [vp, dp, dp, dp, dp, dp, ip, ip, dp]
    lib.rn_dual_extrapolation_step.argtypes = [vp, C.c_double]
    for f in ("rn_solve_step", "rn_proximal_fun_g", "rn_compute_fixed_point_residual", "rn_dual_update"):
        getattr(lib, f).argtypes = [vp]
    lib.rn_update_primal_infeasibility.argtypes = [vp, dp]
    lib.rn_get_prox_distances.argtypes = [vp, dp, dp]
    lib.rn_buffer_size.argtypes = [vp, ip]
    lib.rn_buffer_size.restype = C.c_size_t
    lib.rn_get.argtypes = [vp, ip, dp, C.c_size_t]
    lib.rn_set.argtypes = [vp, ip, dp, C.c_size_t]
    lib.rn_get_operator.argtypes = [vp, ip, ip, dp, C.c_size_t]
    lib.rn_get_range.argtypes = [vp, ip, C.c_size_t, C.c_size_t, dp]
    lib.rn_set_range.argtypes = [vp, ip, C.c_size_t, C.c_size_t, dp]
    lib.rn_device_pointer.argtypes = [vp, ip, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    lib.rn_profile_enable.argtypes = [vp, ip]
    lib.rn_profile_reset.argtypes = [vp]
    lib.rn_profile_read.argtypes = [vp, dp, dp]
    lib.rn_algorithmic_bytes.argtypes = [vp, dp, dp]
    lib.rn_stream.argtypes = [vp]
    lib.rn_stream.restype = vp
    lib.rn_comm_unique_id.argtypes = [dp]
    lib.rn_comm_init.argtypes = [vp, ip, ip, dp]
    lib.rn_comm_init_timeout.argtypes = [vp, ip, ip, dp, C.c_double]
    lib.rn_comm_check.argtypes = [vp]
    lib.rn_comm_library.argtypes = [C.c_char_p, C.c_size_t]
    lib.rn_set_cut_stage.argtypes = [vp, ip]
    lib.rn_get_history_parts.argtypes = [vp, ip, ip, dp]
    lib.rn_get_counters.argtypes = [vp, dp]
    lib.rn_get_kernel_info.argtypes = [vp, dp]
    lib.rn_set_cut_children_moments.argtypes = [vp, dp, dp, C.c_size_t]
    lib.rn_set_operator_mode.argtypes = [vp, ip]
    lib.rn_get_operator_mode.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.rn_set_operator.argtypes = [vp, ip, ip, dp, C.c_size_t]
    lib.rn_set_warm_start.argtypes = [vp, ip]
    lib.rn_set_exchange_mode.argtypes = [vp, ip]
    lib.rn_debug_sweep_phase.argtypes = [vp, ip]
    lib.rn_debug_cut_buffer.argtypes = [vp, ip, dp, C.c_size_t]
    lib.rn_measure_hbm.argtypes = [vp, C.c_size_t, ip, dp, dp]
    lib.rn_set_algorithm.argtypes = [vp, ip, ip]
    for f in ("rn_fbe_reset", "rn_compute_hessian_oracle", "rn_compute_gradient_fbe", "rn_update_fixed_point_residual_nama",
              "rn_compute_lbfgs_direction", "rn_update_lbfgs_buffer", "rn_two_loop_recursion_lbfgs"):
        getattr(lib, f).argtypes = [vp]
    lib.rn_algorithm_fbe_nama.argtypes = [vp, ip, dp, dp, dp]
    lib.rn_compute_value_fbe.argtypes = [vp, dp]
    lib.rn_line_search_lbfgs_update.argtypes = [vp, C.c_double, dp]
    lib.rn_line_search_ame_lbfgs_update.argtypes = [vp, C.c_double, dp]
    lib.rn_lbfgs_state.argtypes = [vp, ip, dp, dp, dp, dp]
    lib.rn_lbfgs_column.argtypes = [vp, ip, ip, ip, dp, C.c_size_t]
    lib.rn_default_cut_stage.argtypes = [C.POINTER(RnDims), C.POINTER(RnTree)]
    lib.rn_partition_create.argtypes = [C.POINTER(RnDims), C.POINTER(RnTree), dp, dp, ip, ip, ip, C.POINTER(RnPartition)]
    lib.rn_partition_destroy.argtypes = [C.POINTER(RnPartition)]
    lib.rn_partition_destroy.restype = None
    lib.rn_create_sharded.argtypes = [C.POINTER(RnDims), C.POINTER(RnTree), dp, dp, ip, ip, ip, ip, ip, dp, C.POINTER(vp)]
    lib.rn_shard_info.argtypes = [vp, dp]
    lib.rn_shard_global_nodes.argtypes = [vp, dp, C.c_size_t]
    lib.rn_debug_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    lib.rn_debug_local_group_create.argtypes = [ip, C.POINTER(vp)]
    lib.rn_debug_local_group_join.argtypes = [vp, vp, ip]
    lib.rn_debug_local_group_destroy.argtypes = [vp]
    lib.rn_guard_check.argtypes = [vp, dp]
    lib.rn_device_memory_info.argtypes = [vp, dp]
    lib.rn_reserve_iterations.argtypes = [vp, ip]
    lib.rn_debug_inject_allocation.argtypes = [vp, C.c_size_t]
    lib.rn_guard_report.argtypes = [dp]
    lib.rn_debug_guard_poke.argtypes = [vp, ip]
    lib.rn_fbe_counters.argtypes = [vp, dp]
    lib.rn_peer_inbox_create.argtypes = [vp, dp]
    lib.rn_peer_inbox_connect.argtypes = [vp, dp, ip]
    lib.rn_debug_peer_inbox_connect_local.argtypes = [C.POINTER(vp), ip]
    lib.rn_debug_peer_seq.argtypes = [vp, C.c_uint]
    lib.rn_set_exchange_transport.argtypes = [vp, ip]
    lib.rn_exchange_autotune.argtypes = [vp, ip, dp]
    lib.rn_set_fused_walk_dual.argtypes = [vp, ip]
    lib.rn_debug_set_knob.argtypes = [vp, ip, ip]
    lib.rn_profile_read_collective.argtypes = [vp, dp, dp]
    _LIB = lib
    return lib


class RapidNetError(RuntimeError):
    pass


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel().astype(np.int32))


def _s(d, k):
    v = d[k]
    return v[0] if isinstance(v, (list, tuple, np.ndarray)) else v


class Solver:
    """Python mirror of the reference's Engine + SmpcController pair for one (network, tree, config) triple.

    Method names follow the reference (Engine.cuh:74-93, SmpcController.cuh:57-203); every method is one call
    through the C-ABI.  `problem` dicts use the reference's JSON schema.
    """

    def __init__(self, network, tree, config, precision="f64", device=0, structured=False, rank=0, nranks=1, cut_stage=0,
                 unique_id=None, knobs=None, operator_mode=None):
        """nranks > 1: `tree` is the FULL scenario tree and the context is rank `rank`'s shard of it (rn_create_sharded:
        partition, communicator from `unique_id` -- None = none, the exchange is a test's job --, cut stage, children
        moments); self.nodes is then the LOCAL node count and self.global_nodes maps local -> full-tree node ids.
        operator_mode: "dense" | "structured" | "auto" (rn_set_operator_mode); None: `structured` decides between the first two --
        the tests and bench.py always say which storage they measure; the C-ABI's own default is auto."""
        self.lib = load()
        self.structured = bool(structured)
        self.network, self.tree, self.config = network, tree, config
        self.nx, self.nu, self.nd = (int(_s(network, k)) for k in ("nx", "nu", "nd"))
        self.nv = int(_s(config, "nv"))
        self.N, self.K, self.nodes = (int(_s(tree, k)) for k in ("N", "K", "nodes"))
        self.ny = 2 * self.nx + self.nu
        self.max_iterations = int(_s(config, "maxIterations"))
        dims, t, keep = tree_structs(tree, self.nx, self.nu, self.nv, self.nd)
        ed, ep = _f64(tree["errorDemandNode"]), _f64(tree["errorPriceNode"])
        h = C.c_void_p()
        prec = RN_F64 if precision == "f64" else RN_F32
        self.rank, self.nranks = int(rank), int(nranks)
        if self.nranks > 1:
            idbuf = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
            rc = self.lib.rn_create_sharded(C.byref(dims), C.byref(t), ed.ctypes.data, ep.ctypes.data, prec, int(device), self.rank, self.nranks,
                                            int(cut_stage), C.cast(idbuf, C.c_void_p) if idbuf is not None else None, C.byref(h))
            if rc != 0:
                raise RapidNetError("rn_create_sharded failed (%d): %s" % (rc, self.lib.rn_last_error(None).decode()))
            self.h = h
            info = self.shardInfo()
            self.full_nodes, self.nodes = self.nodes, info["local_nodes"]
            self.global_nodes = self.shardGlobalNodes()
        else:
            rc = self.lib.rn_create(C.byref(dims), C.byref(t), prec, int(device), C.byref(h))
            if rc != 0:
                raise RapidNetError("rn_create failed (%d): %s" % (rc, self.lib.rn_last_error(None).decode()))
            self.h = h
            self._check(self.lib.rn_set_tree_errors(self.h, ed.ctypes.data, ep.ctypes.data))
        self._check(self.lib.rn_set_parameters(self.h, float(_s(config, "stepSize")), float(_s(config, "penaltyStateX")),
                                               float(_s(config, "penaltySafetyX"))))
        mode = OPS_MODES[operator_mode] if operator_mode is not None else (OPS_STRUCTURED if self.structured else OPS_DENSE)
        self._check(self.lib.rn_set_operator_mode(self.h, mode))
        self.structured = mode != OPS_DENSE
        for k, v in (knobs or {}).items():
            self.debugSetKnob(k, v)

    def close(self):
        if getattr(self, "h", None):
            self.lib.rn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RapidNetError("rapidnet error %d: %s" % (rc, self.lib.rn_last_error(self.h).decode()))

    # ---- Engine -------------------------------------------------------------------------------------------
    def factorStep(self):
        n, c = self.network, self.config
        arrs = [_f64(n["matB"]), _f64(n["matGd"]), _f64(c["matL"]), _f64(c["matLhat"]), _f64(c["costW"]),
                _f64(c["matDiagPrecnd"]), _f64(n["vecXmin"]), _f64(n["vecXmax"]), _f64(n["vecXsafe"]), _f64(n["vecUmin"]),
                _f64(n["vecUmax"]), _f64(n["costAlpha1"])]
        assert arrs[0].size == self.nx * self.nu and arrs[2].size == self.nu * self.nv and arrs[3].size == self.nu * self.nd
        assert arrs[4].size == self.nu * self.nu and arrs[5].size == self.N * (2 * self.nx + self.nu)
        s = RnSystem(*[a.ctypes.data for a in arrs])
        self._check(self.lib.rn_factor_step(self.h, C.byref(s)))

    def updateStateControl(self, currentX=None, prevU=None, prevDemand=None):
        c = self.config
        a = [_f64(c["currentX"] if currentX is None else currentX), _f64(c["prevU"] if prevU is None else prevU),
             _f64(c["prevDemand"] if prevDemand is None else prevDemand)]
        assert a[0].size == self.nx and a[1].size == self.nu and a[2].size == self.nd
        self._check(self.lib.rn_update_state_control(self.h, *[v.ctypes.data for v in a]))

    def eliminateInputDistubanceCoupling(self, nominalDemand, nominalPrices):
        dh, ah = _f64(nominalDemand), _f64(nominalPrices)
        assert dh.size == self.N * self.nd and ah.size == self.N * self.nu
        self._check(self.lib.rn_eliminate_input_disturbance_coupling(self.h, dh.ctypes.data, ah.ctypes.data))

    def setUncertainty(self, demand=True, price=True, weightEconomical=1.0):
        self._check(self.lib.rn_set_uncertainty(self.h, int(demand), int(price), float(weightEconomical)))

    def initialiseSmpcController(self, nominalDemand, nominalPrices):
        """SmpcController::initialiseSmpcController, SmpcController.cu:476-487."""
        self.factorStep()
        self.updateStateControl()
        self.eliminateInputDistubanceCoupling(nominalDemand, nominalPrices)

    # ---- SmpcController -------------------------------------------------------------------------------------
    def apgReset(self):
        self._check(self.lib.rn_apg_reset(self.h))

    def apgIterate(self, n, history=True):
        hist = np.zeros(max(n, 1)) if history else None
        self._check(self.lib.rn_apg_iterate(self.h, int(n), hist.ctypes.data if history else None))
        return hist[:n] if history else None

    def algorithmApg(self, maxIterations=None):
        n = self.max_iterations if maxIterations is None else int(maxIterations)
        hist = np.zeros(max(n, 1))
        self._check(self.lib.rn_algorithm_apg(self.h, n, hist.ctypes.data))
        return hist[:n]

    def controlAction(self, nominalDemand, nominalPrices, currentX=None, prevU=None, prevDemand=None, maxIterations=None,
                      project=False):
        c = self.config
        a = [_f64(c["currentX"] if currentX is None else currentX), _f64(c["prevU"] if prevU is None else prevU),
             _f64(c["prevDemand"] if prevDemand is None else prevDemand), _f64(nominalDemand), _f64(nominalPrices)]
        u0 = np.zeros(self.nu)
        n = self.max_iterations if maxIterations is None else int(maxIterations)
        self._check(self.lib.rn_control_action(self.h, *[v.ctypes.data for v in a], n, int(project), u0.ctypes.data))
        return u0

    def setExchangeMode(self, optimistic=True):
        """False / 0: exact, True / 1: optimistic batches (default) (include/rapidnet.h, rn_set_exchange_mode)."""
        self._check(self.lib.rn_set_exchange_mode(self.h, int(optimistic)))

    def setWarmStart(self, on=True):
        self._check(self.lib.rn_set_warm_start(self.h, int(on)))

    def dualExtrapolationStep(self, lam):
        self._check(self.lib.rn_dual_extrapolation_step(self.h, float(lam)))

    def solveStep(self):
        self._check(self.lib.rn_solve_step(self.h))

    def proximalFunG(self):
        self._check(self.lib.rn_proximal_fun_g(self.h))

    def computeFixedPointResidual(self):
        self._check(self.lib.rn_compute_fixed_point_residual(self.h))

    def dualUpdate(self):
        self._check(self.lib.rn_dual_update(self.h))

    def updatePrimalInfeasibity(self):
        v = C.c_double(0)
        self._check(self.lib.rn_update_primal_infeasibility(self.h, C.byref(v)))
        return v.value

    def proxDistances(self):
        a, b = C.c_double(0), C.c_double(0)
        self._check(self.lib.rn_get_prox_distances(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- global FBE / NAMA (SmpcController.cu:884-1476, 1529-1586) ---------------------------------------------
    def setAlgorithm(self, name, lbfgsBufferSize=None):
        if lbfgsBufferSize is None:
            lbfgsBufferSize = int(_s(self.config, "lbfgsBufferSize")) if "lbfgsBufferSize" in self.config else 5
        self.algorithm = ALGORITHMS[name] if isinstance(name, str) else int(name)
        self.lbfgsBufferSize = int(lbfgsBufferSize)
        self._check(self.lib.rn_set_algorithm(self.h, self.algorithm, self.lbfgsBufferSize))

    def fbeReset(self):
        self._check(self.lib.rn_fbe_reset(self.h))

    def _algorithmFbeNama(self, maxIterations):
        n = self.max_iterations if maxIterations is None else int(maxIterations)
        hist, val, tau = (np.zeros(max(n, 1)) for _ in range(3))
        self._check(self.lib.rn_algorithm_fbe_nama(self.h, n, hist.ctypes.data, val.ctypes.data, tau.ctypes.data))
        return hist[:n], val[: max(n - 1, 0)], tau[: max(n - 1, 0)]

    def algorithmGlobalFbe(self, maxIterations=None):
        """returns (vecPrimalInfs, vecValueFbe, vecTau)"""
        assert getattr(self, "algorithm", ALG_APG) == ALG_GLOBAL_FBE
        return self._algorithmFbeNama(maxIterations)

    def algorithmNama(self, maxIterations=None):
        assert getattr(self, "algorithm", ALG_APG) == ALG_NAMA
        return self._algorithmFbeNama(maxIterations)

    def computeHessianOracalGlobalFbe(self):
        self._check(self.lib.rn_compute_hessian_oracle(self.h))

    def computeGradientFbe(self):
        self._check(self.lib.rn_compute_gradient_fbe(self.h))

    def updateFixedPointResidualNamaAlgorithm(self):
        self._check(self.lib.rn_update_fixed_point_residual_nama(self.h))

    def computeLbfgsDirection(self):
        self._check(self.lib.rn_compute_lbfgs_direction(self.h))

    def computeValueFbe(self):
        v = C.c_double(0)
        self._check(self.lib.rn_compute_value_fbe(self.h, C.byref(v)))
        return v.value

    def computeLineSearchLbfgsUpdate(self, valueFbeY):
        t = C.c_double(0)
        self._check(self.lib.rn_line_search_lbfgs_update(self.h, float(valueFbeY), C.byref(t)))
        return t.value

    def computeLineSearchAmeLbfgsUpdate(self, valueAmeY):
        t = C.c_double(0)
        self._check(self.lib.rn_line_search_ame_lbfgs_update(self.h, float(valueAmeY), C.byref(t)))
        return t.value

    def fbeCounters(self):
        """dict(searches, batches, sequential, sweep_pairs): how the line searches of the FBE / NAMA loops ran, and how many NAMA
        iterations ran their two Hessian sweeps in one pass over the operator blocks"""
        out = np.zeros(4, dtype=np.int64)
        self._check(self.lib.rn_fbe_counters(self.h, out.ctypes.data))
        return dict(zip(("searches", "batches", "sequential", "sweep_pairs"), (int(v) for v in out)))

    def lbfgsState(self, col=None, mem=None, H=None, rho=None):
        """get (no arguments) or set lbfgsBufferCol / Memory / Hessian / Rho; returns (col, mem, H, rho[size+1])."""
        c, m, h = C.c_int(0 if col is None else col), C.c_int(0 if mem is None else mem), C.c_double(0 if H is None else H)
        r = np.zeros(self.lbfgsBufferSize + 1)
        if rho is not None:
            r[:] = rho
        self._check(self.lib.rn_lbfgs_state(self.h, 0 if col is None else 1, C.addressof(c), C.addressof(m), C.addressof(h),
                                            r.ctypes.data))
        return c.value, m.value, h.value, r

    def lbfgsColumn(self, which, col, values=None):
        """column `col` of matS (which=0) / matY (which=1) in the reference's (all xi | all psi) order."""
        n = self.nodes * (2 * self.nx + self.nu)
        a = np.zeros(n) if values is None else _f64(values).copy()
        assert a.size == n
        self._check(self.lib.rn_lbfgs_column(self.h, 0 if values is None else 1, int(which), int(col), a.ctypes.data, n))
        return a

    # ---- raw access -----------------------------------------------------------------------------------------
    def get(self, buf):
        n = self.lib.rn_buffer_size(self.h, buf)
        out = np.zeros(n)
        self._check(self.lib.rn_get(self.h, buf, out.ctypes.data, n))
        return out

    def set(self, buf, values):
        v = _f64(values)
        self._check(self.lib.rn_set(self.h, buf, v.ctypes.data, v.size))

    def getRange(self, buf, first, n):
        out = np.zeros(n)
        self._check(self.lib.rn_get_range(self.h, buf, int(first), int(n), out.ctypes.data))
        return out

    def setRange(self, buf, first, values):
        v = _f64(values)
        self._check(self.lib.rn_set_range(self.h, buf, int(first), v.size, v.ctypes.data))

    def getOperator(self, op, node):
        nx, nu, nv = self.nx, self.nu, self.nv
        n = {OP_PHI: nv * 2 * nx, OP_D: nv * 2 * nx, OP_PSI: nv * nu, OP_F: nv * nu, OP_OMEGA: nv * nv, OP_THETA: nv * nx,
             OP_G: nv * nx}[op]
        out = np.zeros(n)
        self._check(self.lib.rn_get_operator(self.h, op, int(node), out.ctypes.data, n))
        return out

    def setOperator(self, op, node, values):
        """hand in one node's block (OP_PHI, OP_PSI, OP_D, OP_F; col-major nv x (2nx | nu)); an auto context becomes dense"""
        v = _f64(values)
        self._check(self.lib.rn_set_operator(self.h, int(op), int(node), v.ctypes.data, v.size))

    def operatorMode(self):
        """(requested, active) as "dense" | "structured" | "auto" """
        r, a = C.c_int(0), C.c_int(0)
        self._check(self.lib.rn_get_operator_mode(self.h, C.byref(r), C.byref(a)))
        names = {v: k for k, v in OPS_MODES.items()}
        return names[r.value], names[a.value]

    def synchronize(self):
        self._check(self.lib.rn_synchronize(self.h))

    # ---- measurement ----------------------------------------------------------------------------------------
    def profileEnable(self, on=1):
        self._check(self.lib.rn_profile_enable(self.h, int(on)))

    def profileReset(self):
        self._check(self.lib.rn_profile_reset(self.h))

    def profileRead(self):
        ms = np.zeros(4)
        n = np.zeros(4, dtype=np.int64)
        self._check(self.lib.rn_profile_read(self.h, ms.ctypes.data, n.ctypes.data))
        return ms, n

    def profileReadCollective(self):
        """(ms, count) of the all-reduces timed since the last profileReset (sharded contexts; inside class 1 of profileRead)."""
        ms, n = C.c_double(0), C.c_long(0)
        self._check(self.lib.rn_profile_read_collective(self.h, C.addressof(ms), C.addressof(n)))
        return ms.value, n.value

    def guardCheck(self):
        """red-zone bytes overwritten so far (0 unless the context was created under RAPIDNET_GUARD=1)."""
        bad = C.c_long(0)
        self._check(self.lib.rn_guard_check(self.h, C.addressof(bad)))
        return bad.value

    def peerInboxCreate(self):
        """this rank's inbox for the one-shot exchange: returns its 64-byte IPC handle (bytes)"""
        buf = C.create_string_buffer(64)
        self._check(self.lib.rn_peer_inbox_create(self.h, C.cast(buf, C.c_void_p)))
        return bytes(buf.raw)

    def peerInboxConnect(self, handles):
        """handles: the 64-byte IPC handles of all ranks, in rank order"""
        blob = b"".join(bytes(h) for h in handles)
        buf = C.create_string_buffer(blob, len(blob))
        self._check(self.lib.rn_peer_inbox_connect(self.h, C.cast(buf, C.c_void_p), len(handles)))

    def debugPeerSeq(self, seq):
        self._check(self.lib.rn_debug_peer_seq(self.h, int(seq) & 0xFFFFFFFF))

    def setExchangeTransport(self, transport):
        """EXCHANGE_COLLECTIVE (0): the cut payload is all-reduced by the communicator; EXCHANGE_ONESHOT (1): one-shot peer writes (needs
        connected inboxes); EXCHANGE_AUTO (2, the default): the context times both in its first batch and keeps the faster"""
        self._check(self.lib.rn_set_exchange_transport(self.h, int(transport)))

    def exchangeAutotune(self, iterations=0):
        """rn_exchange_autotune: iterations > 0 tunes now (collective: every rank calls it), 0 reports the last result"""
        out = np.zeros(8)
        self._check(self.lib.rn_exchange_autotune(self.h, int(iterations), out.ctypes.data))
        return {"transport": int(out[0]), "candidates": int(out[1]), "collective_us": float(out[2]), "oneshot_us": float(out[3]),
                "own_collective_us": float(out[4]), "own_oneshot_us": float(out[5]), "iterations": int(out[6]), "tunes": int(out[7])}

    def setFusedWalkDual(self, on):
        """1 / 0: forward walk + dual update in one launch inside batches of >= 16 iterations (identical iterates) on / off; -1: by shape (default)"""
        self._check(self.lib.rn_set_fused_walk_dual(self.h, int(on)))

    def debugSetKnob(self, knob, value):
        """rn_debug_set_knob (include/rapidnet_debug.h): force a launch-shape choice the library otherwise makes by problem size; before factorStep.
        knob: a KNOB_* id or its name without the prefix ("dual_trips", "vlv_wide", ...); value -1: the library's own choice"""
        k = KNOBS[knob.lower()] if isinstance(knob, str) else int(knob)
        self._check(self.lib.rn_debug_set_knob(self.h, k, int(value)))

    def debugGuardPoke(self, nbytes):
        self._check(self.lib.rn_debug_guard_poke(self.h, int(nbytes)))

    def deviceMemoryInfo(self):
        out = (C.c_size_t * 4)()
        self._check(self.lib.rn_device_memory_info(self.h, C.addressof(out)))
        return dict(zip(("free", "total", "context_bytes", "live_contexts"), (int(v) for v in out)))

    def reserveIterations(self, n):
        self._check(self.lib.rn_reserve_iterations(self.h, int(n)))

    def measureHbm(self, nbytes=1 << 30, reps=3):
        """(read-only GB/s, copy GB/s) of do-nothing flat streaming kernels on this device."""
        r, c = C.c_double(0), C.c_double(0)
        self._check(self.lib.rn_measure_hbm(self.h, int(nbytes), int(reps), C.addressof(r), C.addressof(c)))
        return r.value, c.value

    def devicePointer(self, buffer_id):
        """(raw device address, element count, 'f64' | 'f32') of a buffer kept in the reference's node-major layout (X, U, V, UHAT, E,
        BETA, ALPHA, XDIR, UDIR): the counterpart of the reference's raw device getters; RapidNetError for the others (use get)."""
        p, n, prec = C.c_void_p(), C.c_size_t(0), C.c_int(0)
        self._check(self.lib.rn_device_pointer(self.h, int(buffer_id), C.byref(p), C.byref(n), C.byref(prec)))
        return p.value, n.value, "f64" if prec.value == RN_F64 else "f32"

    def algorithmicBytes(self):
        a, b = C.c_double(0), C.c_double(0)
        self._check(self.lib.rn_algorithmic_bytes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- multi-GPU ------------------------------------------------------------------------------------------
    def commInit(self, rank, nranks, unique_id_bytes, timeout=None):
        """unique_id_bytes None: record rank / nranks only (tests emulate the exchange with debugCutBuffer).
        timeout (seconds; None: $RAPIDNET_COMM_TIMEOUT_S or 120): a rank whose peers do not arrive gets RapidNetError (RN_E_COMM)
        back instead of waiting for ever; the context stays usable without a communicator."""
        if unique_id_bytes is None:
            self._check(self.lib.rn_comm_init(self.h, int(rank), int(nranks), None))
            return
        buf = C.create_string_buffer(bytes(unique_id_bytes), 128)
        if timeout is None:
            self._check(self.lib.rn_comm_init(self.h, int(rank), int(nranks), C.cast(buf, C.c_void_p)))
        else:
            self._check(self.lib.rn_comm_init_timeout(self.h, int(rank), int(nranks), C.cast(buf, C.c_void_p), float(timeout)))

    def commCheck(self):
        self._check(self.lib.rn_comm_check(self.h))

    def debugSweepPhase(self, phase):
        self._check(self.lib.rn_debug_sweep_phase(self.h, int(phase)))

    def debugCutBuffer(self, n, values=None):
        buf = np.zeros(n) if values is None else _f64(values)
        self._check(self.lib.rn_debug_cut_buffer(self.h, 0 if values is None else 1, buf.ctypes.data, buf.size))
        return buf

    def setCutStage(self, stage, moments=None):
        """moments = partition.cut_children_moments(full_tree, stage), required when nranks > 1."""
        self._check(self.lib.rn_set_cut_stage(self.h, int(stage)))
        if moments is not None:
            E, P = _f64(moments[0]), _f64(moments[1])
            self._check(self.lib.rn_set_cut_children_moments(self.h, E.ctypes.data, P.ctypes.data, P.size))

    def shardInfo(self):
        out = np.zeros(7, dtype=np.int32)
        self._check(self.lib.rn_shard_info(self.h, out.ctypes.data))
        return dict(zip(("rank", "nranks", "cut_stage", "cut_parents", "comm_ranks", "local_nodes", "full_nodes"), (int(v) for v in out)))

    def shardGlobalNodes(self):
        n = self.shardInfo()["local_nodes"]
        out = np.zeros(n, dtype=np.int32)
        self._check(self.lib.rn_shard_global_nodes(self.h, out.ctypes.data, n))
        return out

    def debugSetAllreduce(self, fn):
        """fn(dev_ptr, count, is_f64, op, stream) -> 0; called in place of ncclAllReduce (test facility)."""
        self._ar_cb = ALLREDUCE_FN(lambda user, buf, count, f64, op, stream: int(fn(buf, count, f64, op, stream))) if fn is not None else C.cast(None, ALLREDUCE_FN)
        self._check(self.lib.rn_debug_set_allreduce(self.h, self._ar_cb, None))

    def joinLocalGroup(self, group, rank):
        self._check(self.lib.rn_debug_local_group_join(self.h, group, int(rank)))

    def kernelInfo(self):
        out = np.zeros(8, dtype=np.int32)
        self._check(self.lib.rn_get_kernel_info(self.h, out.ctypes.data))
        keys = ("dual_stage", "dual_blocks", "dual_trips", "dual_pipe", "stream_G", "stream_NL", "chain_stage", "vlv_slab")
        return dict(zip(keys, (int(v) for v in out)))

    def counters(self):
        """rn_apg_iterate batch bookkeeping: dict(optimistic, exact, replayed, hold)."""
        out = np.zeros(4, dtype=np.int64)
        self._check(self.lib.rn_get_counters(self.h, out.ctypes.data))
        return dict(zip(("optimistic", "exact", "replayed", "hold"), (int(v) for v in out)))

    def historyParts(self, first, n):
        out = np.zeros(4 * n)
        self._check(self.lib.rn_get_history_parts(self.h, int(first), int(n), out.ctypes.data))
        return out.reshape(n, 4)


def tree_structs(tree, nx, nu, nv, nd):
    """(RnDims, RnTree, arrays to keep alive) of a tree dict in the reference's JSON schema."""
    N, K, nodes = (int(_s(tree, k)) for k in ("N", "K", "nodes"))
    dims = RnDims(nx, nu, nv, nd, N, K, nodes, int(_s(tree, "nNonLeafNodes")))
    keep = [_i32(tree[k]) for k in ("stages", "nodesPerStage", "nodesPerStageCumul", "ancestor", "nChildren", "nChildrenCumul")]
    keep.append(_f64(tree["probNode"]))
    if len(keep[1]) < N + 1 or len(keep[2]) < N + 2:
        raise RapidNetError("nodesPerStage needs N+1 and nodesPerStageCumul N+2 entries (ScenarioTree.cu:66-75)")
    return dims, RnTree(*[a.ctypes.data for a in keep]), keep


def default_cut_stage(tree, nx=1, nu=1, nv=1, nd=1):
    dims, t, _keep = tree_structs(tree, nx, nu, nv, nd)
    c = load().rn_default_cut_stage(C.byref(dims), C.byref(t))
    if c < 0:
        raise RapidNetError("rn_default_cut_stage failed (%d)" % c)
    return c


def partition_tree(tree, rank, nranks, cut_stage=0):
    """rn_partition_create on a tree dict: returns a dict with the local tree (reference JSON schema, like
    rapidnet_amd.partition.local_tree), 'globalNode', 'cutStage', 'momE' [parents, nd], 'momP' [parents]."""
    lib = load()
    nd, nu = int(_s(tree, "dimDemand")), int(_s(tree, "dimPrice"))
    dims, t, _keep = tree_structs(tree, 1, nu, 1, nd)
    ed, ep = _f64(tree["errorDemandNode"]), _f64(tree["errorPriceNode"])
    part = RnPartition()
    rc = lib.rn_partition_create(C.byref(dims), C.byref(t), ed.ctypes.data, ep.ctypes.data, int(rank), int(nranks), int(cut_stage), C.byref(part))
    if rc != 0:
        raise RapidNetError("rn_partition_create failed (%d): %s" % (rc, lib.rn_last_error(None).decode()))
    try:
        n, N = part.dims.nodes, part.dims.N
        iarr = lambda ptr, cnt: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int)), (cnt,)).copy() if cnt else np.zeros(0, np.int32)
        darr = lambda ptr, cnt: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), (cnt,)).copy() if cnt else np.zeros(0)
        local = {
            "N": [N], "K": [part.dims.K], "dimDemand": [nd], "dimPrice": [nu], "nodes": [n], "nChildrenTot": [n - 1],
            "nNonLeafNodes": [part.dims.nNonLeafNodes],
            "stages": iarr(part.tree.stages, n).tolist(), "nodesPerStage": iarr(part.tree.nodesPerStage, N + 1).tolist(),
            "nodesPerStageCumul": iarr(part.tree.nodesPerStageCumul, N + 2).tolist(), "ancestor": iarr(part.tree.ancestor, n).tolist(),
            "nChildren": iarr(part.tree.nChildren, part.dims.nNonLeafNodes).tolist(), "nChildrenCumul": iarr(part.tree.nChildrenCumul, n).tolist(),
            "probNode": darr(part.tree.probNode, n).tolist(),
            "errorDemandNode": darr(part.errorDemandNode, n * nd).tolist(), "errorPriceNode": darr(part.errorPriceNode, n * nu).tolist(),
        }
        cc = np.bincount(np.asarray(local["ancestor"])[1:] - 1, minlength=n)
        local["leaves"] = (np.flatnonzero(cc == 0) + 1).tolist()
        local["children"] = [i + 1 for i in range(1, n)]
        return {"tree": local, "globalNode": iarr(part.globalNode, n), "cutStage": part.cutStage,
                "momE": darr(part.momE, part.nCutParents * nd).reshape(part.nCutParents, nd), "momP": darr(part.momP, part.nCutParents)}
    finally:
        lib.rn_partition_destroy(C.byref(part))


def guard_report():
    """(contexts checked at rn_destroy in guard mode, red-zone bytes found overwritten) of this process."""
    out = (C.c_long * 2)()
    load().rn_guard_report(C.addressof(out))
    return int(out[0]), int(out[1])


def peer_inbox_connect_local(solvers):
    """contexts of ONE process, in rank order, each with an inbox (peerInboxCreate): wire them to each other (tests)"""
    arr = (C.c_void_p * len(solvers))(*[s.h for s in solvers])
    rc = load().rn_debug_peer_inbox_connect_local(arr, len(solvers))
    if rc != 0:
        raise RapidNetError("rn_debug_peer_inbox_connect_local failed (%d): %s" % (rc, "; ".join(load().rn_last_error(s.h).decode() for s in solvers)))


def local_group_create(nranks):
    g = C.c_void_p()
    rc = load().rn_debug_local_group_create(int(nranks), C.byref(g))
    if rc != 0:
        raise RapidNetError("rn_debug_local_group_create failed (%d)" % rc)
    return g


def local_group_destroy(group):
    load().rn_debug_local_group_destroy(group)


def comm_library():
    """Path of the RCCL image librapidnet_hip bound (an already-loaded image is reused)."""
    buf = C.create_string_buffer(4096)
    rc = load().rn_comm_library(buf, 4096)
    if rc != 0:
        raise RapidNetError("rn_comm_library failed (%d)" % rc)
    return buf.value.decode()


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = load().rn_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != 0:
        raise RapidNetError("rn_comm_unique_id failed (%d)" % rc)
    return bytes(buf.raw)
