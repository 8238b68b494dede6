"""No kernel of librapidnet_hip.so may spill, and every kernel has exactly one home.  hipcc's own resource report
(-Rpass-analysis=kernel-resource-usage) of every translation unit the library is built from must show 0 bytes of scratch per lane for
every kernel (round 4 shipped two that spilled: k_stream_gemv<float, NL, false, 2> and k_ls_eval); the host driver rapidnet_capi.hip
must compile NO kernel but its own k_sum_ranks -- everything it launches is declared `extern template` from the generated lists
(csrc/instantiations/*.inc) and compiled in one of the k_*.hip units: a kernel that the driver instantiates by itself means the lists are
out of date (python tools/gen_instantiations.py).  Cross-compiles for gfx950: no GPU needed."""
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

from rapidnet_amd import build


def _report(unit):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [build.HIPCC] + [f for f in build.HIP_FLAGS if f != "-fPIC"] + ["--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", "-o",
                                                                              os.path.join(tmp, "k.o"), os.path.join(build.CSRC, unit + ".hip")]
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
    return scratch


def test_no_kernel_uses_scratch_and_every_kernel_has_one_home():
    with ThreadPoolExecutor(max_workers=min(len(build.UNITS), os.cpu_count() or 4)) as ex:
        reports = dict(zip(build.UNITS, ex.map(_report, build.UNITS)))
    spilling = {k: v for r in reports.values() for k, v in r.items() if v > 0}
    assert not spilling, spilling
    driver = reports.pop("rapidnet_capi")
    assert all("k_sum_ranks" in k for k in driver) and len(driver) == 2, "rapidnet_capi.hip compiles kernels of its own (stale csrc/instantiations/*.inc?): %s" % sorted(driver)
    homes = {}
    for unit, r in reports.items():
        for k in r:
            assert k not in homes, "%s is compiled in %s and in %s" % (k, homes[k], unit)
            homes[k] = unit
    declared = sum(open(os.path.join(build.CSRC, "instantiations", f)).read().count("RN_LINKAGE template") for f in os.listdir(os.path.join(build.CSRC, "instantiations")))
    assert len(homes) == declared, (len(homes), declared)
    assert 150 < len(homes) < 200, "the resource reports list %d kernels" % len(homes)      # (round 5: 232)
