"""Builds the in-tree native libraries (hipcc cross-compiles for gfx950 without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_HIP = os.path.join(HERE, "librapidnet_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_hip(force=False, verbose=False, defines=(), out=None):
    """librapidnet_hip.so: the C-ABI (include/rapidnet.h) + every HIP kernel, for gfx950 only.

    `defines` / `out` build tuning variants (e.g. RN_STREAM_G=6) next to the default library; capi.load() picks the
    library named by $RAPIDNET_LIB when set."""
    out = out or LIB_HIP
    srcs = [os.path.join(CSRC, "rapidnet_capi.hip"), os.path.join(CSRC, "kernels.hpp"), os.path.join(CSRC, "fbe_kernels.hpp"),
            os.path.join(CSRC, "fbe_methods.inc"), os.path.join(ROOT, "include", "rapidnet.h")]
    if force or _stale(out, srcs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
               "-Wno-pass-failed"] + ["-D" + d for d in defines] + ["-o", out, srcs[0], "-ldl"]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
    return out


LIB_HOST = os.path.join(HERE, "librapidnet_host.so")
BIN_DIR = os.path.join(HERE, "bin")
TEST_HOST = os.path.join(BIN_DIR, "test_host")


def build_host(force=False):
    """librapidnet_host.so (the reference's C++ class surface over the C-ABI) and the C++ test driver."""
    hdir = os.path.join(CSRC, "host")
    srcs = [os.path.join(hdir, f) for f in ("DataModel.cpp", "Engine.cpp", "SmpcController.cpp", "NullSpace.cpp")]
    deps = srcs + [os.path.join(hdir, f) for f in ("DataModel.hpp", "Engine.hpp", "SmpcController.hpp", "JsonLite.hpp", "Configuration.h", "NullSpace.hpp")]
    build_hip()
    if force or _stale(LIB_HOST, deps + [LIB_HIP]):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-fPIC", "-shared", "-o", LIB_HOST] + srcs +
                              ["-L" + HERE, "-lrapidnet_hip", "-Wl,-rpath,$ORIGIN"])
    test_src = os.path.join(ROOT, "tests", "cpp", "test_host.cpp")
    os.makedirs(BIN_DIR, exist_ok=True)
    if force or _stale(TEST_HOST, [test_src, LIB_HOST]):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-o", TEST_HOST, test_src, "-L" + HERE, "-lrapidnet_host",
                               "-lrapidnet_hip", "-Wl,-rpath,$ORIGIN/.."])
    return LIB_HOST


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_host(force="--force" in sys.argv))
