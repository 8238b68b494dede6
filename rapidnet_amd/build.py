"""Builds the in-tree native libraries (hipcc cross-compiles for gfx950 without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_HIP = os.path.join(HERE, "librapidnet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _fingerprint(srcs, extra=()):
    import hashlib

    h = hashlib.sha256()
    for f in srcs:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def _stale(out, srcs, extra=()):
    """A built file is current when the stamp beside it (<out>.srchash) equals the fingerprint of its sources: content,
    not mtimes -- a snapshot copied to another box keeps its binaries valid whatever the copy did to the timestamps."""
    stamp = out + ".srchash"
    if not os.path.exists(out) or not os.path.exists(stamp):
        return True
    return open(stamp).read().strip() != _fingerprint(srcs, extra)


def _stamp(out, srcs, extra=()):
    with open(out + ".srchash", "w") as f:
        f.write(_fingerprint(srcs, extra) + "\n")


# Translation units of librapidnet_hip.so and what each is compiled from (besides itself and the instantiation list of its name): one kernel
# family per unit, so that an edit of one kernel header recompiles that unit and the host driver (which only parses the headers: its
# kernels are declared `extern template`, csrc/instantiations/*.inc, tools/gen_instantiations.py).
UNITS = {
    "k_stream": ["common.hpp", "k_stream.hpp"],
    "k_dual": ["common.hpp", "k_dual.hpp"],
    "k_walks": ["common.hpp", "k_dual.hpp", "k_walks.hpp"],
    "k_slab": ["common.hpp", "k_dual.hpp", "k_walks.hpp", "k_slab.hpp"],
    "k_misc": ["common.hpp", "k_misc.hpp"],
    "k_fbe": ["common.hpp", "k_dual.hpp", "k_walks.hpp", "k_slab.hpp", "fbe_kernels.hpp"],
    "rapidnet_capi": ["common.hpp", "k_dual.hpp", "k_walks.hpp", "k_slab.hpp", "k_stream.hpp", "k_misc.hpp", "kernels.hpp", "fbe_kernels.hpp", "fbe_methods.inc",
                      "partition.hpp"],
}
OBJ_DIR = os.path.join(HERE, "build")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _unit_deps(unit):
    inst = [os.path.join(CSRC, "instantiations", f) for f in sorted(os.listdir(os.path.join(CSRC, "instantiations")))] if unit == "rapidnet_capi" \
        else [os.path.join(CSRC, "instantiations", unit[2:] + ".inc")]
    hdr = [os.path.join(ROOT, "include", "rapidnet.h"), os.path.join(ROOT, "include", "rapidnet_debug.h")] if unit == "rapidnet_capi" else []
    return [os.path.join(CSRC, unit + ".hip")] + [os.path.join(CSRC, f) for f in UNITS[unit]] + inst + hdr


def kernel_sources():
    """every file device code is compiled from"""
    seen, out = set(), []
    for u in UNITS:
        for f in _unit_deps(u):
            if f not in seen and not f.endswith((".h",)):
                seen.add(f); out.append(f)
    return out


def kernel_sources_sha256():
    """Fingerprint of the HIP sources: measurement artefacts (profiles/traffic.json) carry it, and bench.py only reports
    PMC traffic collected with the kernels it is running."""
    import hashlib

    h = hashlib.sha256()
    for f in kernel_sources():
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _hip_deps():
    """Everything librapidnet_hip.so is compiled from."""
    return kernel_sources() + [os.path.join(ROOT, "include", "rapidnet.h"), os.path.join(ROOT, "include", "rapidnet_debug.h")]


def _compile_unit(unit, flags, verbose):
    obj = os.path.join(OBJ_DIR, unit + ".o")
    deps = _unit_deps(unit)
    if not _stale(obj, deps, flags):
        return obj, False, ""
    tmp = "%s.tmp.%d" % (obj, os.getpid())
    cmd = [HIPCC] + flags + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + ["-c", "-o", tmp, deps[0]]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed on %s.hip:\n%s" % (unit, r.stderr[-6000:]))
    os.replace(tmp, obj)
    _stamp(obj, deps, flags)
    return obj, True, r.stderr


def build_hip(force=False, verbose=False, defines=(), out=None, jobs=None):
    """librapidnet_hip.so: the C-ABI (include/rapidnet.h) + every HIP kernel, for gfx950 only.  Seven translation units (UNITS) compiled in
    parallel into rapidnet_amd/build/*.o -- each with a content stamp of ITS sources, so an unchanged unit is not compiled again -- and linked.

    `defines` / `out` build tuning variants (e.g. RN_STREAM_G=6) next to the default library (objects of their own); capi.load() picks the
    library named by $RAPIDNET_LIB when set.  Returns the path; build_hip.remarks holds the compiler's resource-usage remarks of the units that
    were compiled when verbose."""
    from concurrent.futures import ThreadPoolExecutor

    global OBJ_DIR
    out = out or LIB_HIP
    srcs = _hip_deps()
    flags = HIP_FLAGS + ["-D" + d for d in defines]
    build_hip.remarks = ""
    if not (force or _stale(out, srcs, defines)):
        return out
    obj_dir = OBJ_DIR if not defines and out == LIB_HIP else OBJ_DIR + "_" + "_".join(d.replace("=", "-") for d in defines)[:60]
    saved, OBJ_DIR = OBJ_DIR, obj_dir
    try:
        os.makedirs(OBJ_DIR, exist_ok=True)
        if force:
            for u in UNITS:
                for f in (os.path.join(OBJ_DIR, u + ".o"), os.path.join(OBJ_DIR, u + ".o.srchash")):
                    if os.path.exists(f):
                        os.remove(f)
        with ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 4)) as ex:
            res = list(ex.map(lambda u: _compile_unit(u, flags, verbose), UNITS))
        build_hip.remarks = "".join(r[2] for r in res)
        tmp = "%s.tmp.%d" % (out, os.getpid())   # link beside the target, then rename: a concurrent loader never sees a partial file
        try:
            subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [r[0] for r in res] + ["-ldl"])
            os.replace(tmp, out)
            _stamp(out, srcs, defines)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    finally:
        OBJ_DIR = saved
    return out


def hip_is_stale():
    return _stale(LIB_HIP, _hip_deps())


def hip_fingerprint():
    """Content fingerprint of everything librapidnet_hip.so is built from (what its .srchash stamp holds)."""
    return _fingerprint(_hip_deps())


LIB_HOST = os.path.join(HERE, "librapidnet_host.so")
BIN_DIR = os.path.join(HERE, "bin")
TEST_HOST = os.path.join(BIN_DIR, "test_host")


def build_host(force=False):
    """librapidnet_host.so (the reference's C++ class surface over the C-ABI) and the C++ test driver."""
    hdir = os.path.join(CSRC, "host")
    srcs = [os.path.join(hdir, f) for f in ("DataModel.cpp", "Engine.cpp", "SmpcController.cpp", "NullSpace.cpp")]
    deps = srcs + [os.path.join(hdir, f) for f in ("DataModel.hpp", "Engine.hpp", "SmpcController.hpp", "JsonLite.hpp", "Configuration.h", "NullSpace.hpp")]
    build_hip()
    hdr = [os.path.join(ROOT, "include", "rapidnet.h"), os.path.join(ROOT, "include", "rapidnet_debug.h")]
    if force or _stale(LIB_HOST, deps + hdr):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-fPIC", "-shared", "-o", LIB_HOST] + srcs +
                              ["-L" + HERE, "-lrapidnet_hip", "-Wl,-rpath,$ORIGIN"])
        _stamp(LIB_HOST, deps + hdr)
    test_src = os.path.join(ROOT, "tests", "cpp", "test_host.cpp")
    os.makedirs(BIN_DIR, exist_ok=True)
    if force or _stale(TEST_HOST, [test_src] + deps + hdr):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-pthread", "-o", TEST_HOST, test_src, "-L" + HERE, "-lrapidnet_host",
                               "-lrapidnet_hip", "-Wl,-rpath,$ORIGIN/.."])
        _stamp(TEST_HOST, [test_src] + deps + hdr)
    return LIB_HOST


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_host(force="--force" in sys.argv))
