"""Same-context A/B of the sweep's two forms (rn_set_sweep_form: 0 = six launches, 1 = chain-fused), interleaved regions of 100 iterations:
python tools/ab_sweep_form.py [config ...] [--structured] [--f32] [--rounds R] [--profile]   (config: barcelona493, barcelona31, medium ...)
--profile: hipEvent classes of either form (rn_profile_*); under rocprofv3 --kernel-trace --stats the kernel names tell the forms apart."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from rapidnet_amd import capi, synth  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    structured, f32, prof = "--structured" in sys.argv, "--f32" in sys.argv, "--profile" in sys.argv
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 4
    if "--rounds" in sys.argv:
        args = [a for a in args if a != str(rounds)]
    for name in args or ["barcelona493"]:
        p = synth.make_problem(name)
        dh, ah = synth.forecast_at(p["forecast"], 0)
        s = capi.Solver(p["network"], p["tree"], p["config"], structured=structured, precision="f32" if f32 else "f64")
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            s.apgIterate(20, history=False)
            s.synchronize()
        res = {0: [], 1: []}
        active = {}
        for r in range(rounds):
            for form in (0, 1):
                active[form] = s.setSweepForm(form)
                s.apgIterate(40, history=False)
                s.synchronize()
                t0 = time.perf_counter()
                s.apgIterate(100, history=False)
                s.synchronize()
                res[form].append(1e3 * (time.perf_counter() - t0) / 100)
        out = {"config": name, "structured": structured, "dtype": "f32" if f32 else "f64", "active": active,
               "ms_per_iteration": {str(f): {"median": float(np.median(v)), "min": min(v), "max": max(v)} for f, v in res.items()}}
        if prof:
            for form in (0, 1):
                s.setSweepForm(form)
                s.apgIterate(40, history=False)
                s.profileEnable(True)
                s.profileReset()
                s.apgIterate(100, history=False)
                ms, n = s.profileRead()
                s.profileEnable(False)
                out["classes_us_form%d" % form] = {k: 1e3 * ms[i] / max(1, n[i]) for i, k in enumerate(("stream", "helpers", "dual", "bookkeeping"))}
        print(json.dumps(out), flush=True)
        s.close()


if __name__ == "__main__":
    main()
