"""No kernel of librapidnet_hip.so may spill: hipcc's own resource report (-Rpass-analysis=kernel-resource-usage) for the
translation unit the library is built from must show 0 bytes of scratch per lane for every kernel (round 4 shipped two that
did: k_stream_gemv<float, NL, false, 2> and k_ls_eval).  Cross-compiles for gfx950: no GPU needed."""
import os
import re
import subprocess
import tempfile

from rapidnet_amd import build


def test_no_kernel_uses_scratch():
    src = build._hip_deps()[0]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [build.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", "-Wno-unused-function", "-Wno-pass-failed",
               "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "k.o"), src]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    cur, scratch = None, {}
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and cur:
            scratch[cur] = int(m.group(1))
    assert len(scratch) > 150, "the resource report lists %d kernels" % len(scratch)
    spilling = {k: v for k, v in scratch.items() if v > 0}
    assert not spilling, spilling
