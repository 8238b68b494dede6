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


def kernel_sources():
    return [os.path.join(CSRC, f) for f in ("rapidnet_capi.hip", "kernels.hpp", "fbe_kernels.hpp", "fbe_methods.inc")]


def kernel_sources_sha256():
    """Fingerprint of the HIP sources: measurement artefacts (profiles/traffic.json) carry it, and bench.py only reports
    PMC traffic collected with the kernels it is running."""
    import hashlib

    h = hashlib.sha256()
    for f in kernel_sources():
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _hip_deps():
    """Everything librapidnet_hip.so is compiled from (first entry = the translation unit)."""
    return kernel_sources() + [os.path.join(CSRC, "partition.hpp"), os.path.join(ROOT, "include", "rapidnet.h"), os.path.join(ROOT, "include", "rapidnet_debug.h")]


def build_hip(force=False, verbose=False, defines=(), out=None):
    """librapidnet_hip.so: the C-ABI (include/rapidnet.h) + every HIP kernel, for gfx950 only.

    `defines` / `out` build tuning variants (e.g. RN_STREAM_G=6) next to the default library; capi.load() picks the
    library named by $RAPIDNET_LIB when set."""
    out = out or LIB_HIP
    srcs = _hip_deps()
    if force or _stale(out, srcs, defines):
        tmp = "%s.tmp.%d" % (out, os.getpid())   # compile beside the target, then rename: a concurrent loader never sees a partial file
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
               "-Wno-pass-failed"] + ["-D" + d for d in defines] + ["-o", tmp, srcs[0], "-ldl"]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, out)
            _stamp(out, srcs, defines)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
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
