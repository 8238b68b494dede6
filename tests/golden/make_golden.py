#!/usr/bin/env python3
"""Generates tests/golden/synthetic/*.npz: iterates of the CPU oracle (oracle/apg_oracle.c, itself pinned to the
reference's golden vectors by tests/test_oracle_reference_fixtures.py) on the seeded synthetic problems of
rapidnet_amd/synth.py after k = 1, 10 and 50 APG iterations.  Small fixtures: x, u and the primal-infeasibility
history of every run plus, for the large configs, a strided sample of the duals.

    python tests/golden/make_golden.py            # rewrites the fixtures (deterministic)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402
from rapidnet_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "synthetic")
CASES = {"toy": (1, 10, 50), "tiny": (1, 10, 50), "small": (1, 10, 50), "odd": (1, 10, 50), "medium": (1, 10, 50),
         "barcelona31": (1, 10), "ragged": (1, 10, 50)}


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, ks in CASES.items():
        p = synth.make_problem(name)
        dh, ah = synth.forecast_at(p["forecast"], 0)
        out = {"stepSize": np.array(p["config"]["stepSize"])}
        for k in ks:
            o = Oracle(p["network"], p["tree"], p["config"])
            o.initialise(dh, ah)
            hist = o.apg(k)
            stride = 1 if o.nodes * (2 * o.nx) < 20000 else 37
            out["x_%d" % k] = o.get("x")[::stride]
            out["u_%d" % k] = o.get("u")[::stride]
            out["updXi_%d" % k] = o.get("updXi")[::stride]
            out["updPsi_%d" % k] = o.get("updPsi")[::stride]
            out["hist_%d" % k] = hist
            out["stride"] = np.array([stride])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items() if k.startswith("x_")})


if __name__ == "__main__":
    main()
