"""EPANET .inp -> network JSON (rapidnet_amd/epanet.py) against the pair the reference holds: src/paser/testEpanet.inp and the
src/paser/network.json its MATLAB tool chain wrote from it (parserEpanet.m, createDwnDataJson.m, generateJsonFile.m), both
committed as data under tests/golden/reference_fixture/epanet/."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import REF_FIXTURE, ROOT
from rapidnet_amd import epanet

PAIR = os.path.join(REF_FIXTURE, "epanet")


def test_reference_pair_byte_for_byte(tmp_path):
    out = str(tmp_path / "network.json")
    text = epanet.convert(os.path.join(PAIR, "testEpanet.inp"), out)
    want = open(os.path.join(PAIR, "network.json")).read()
    assert text == want and open(out).read() == want
    # and through the command line
    out2 = str(tmp_path / "cli.json")
    subprocess.check_call([sys.executable, "-m", "rapidnet_amd.epanet", os.path.join(PAIR, "testEpanet.inp"), out2], cwd=ROOT)
    assert open(out2).read() == want


def test_output_is_a_loadable_network(tmp_path):
    """Standard JSON with every key the DwnNetwork loader reads (DwnNetwork.cuh:23-37), column-major matrices of the right size."""
    out = str(tmp_path / "network.json")
    epanet.convert(os.path.join(PAIR, "testEpanet.inp"), out)
    d = json.load(open(out))
    nx, nu, nd, ne = (d[k][0] for k in ("nx", "nu", "nd", "ne"))
    assert (nx, nu, nd, ne) == (3, 5, 3, 1)
    for key, n in (("matA", nx * nx), ("matB", nx * nu), ("matGd", nx * nd), ("matE", ne * nu), ("matEd", ne * nd), ("vecXmin", nx),
                   ("vecXmax", nx), ("vecXsafe", nx), ("vecUmin", nu), ("vecUmax", nu), ("costAlpha1", nu)):
        assert len(d[key]) == n, key
    B = np.array(d["matB"]).reshape(nx, nu, order="F")
    assert B[:, 0].tolist() == [-1, 1, 0] and B[:, 4].tolist() == [0, 0, -1]     # pump 19: tank 3 -> tank 2; pump 223: junction 12 -> tank 4
    E = np.array(d["matE"]).reshape(ne, nu, order="F")
    assert E.tolist() == [[0, 0, -1, -1, 1]]                                     # junction 12: fed by pumps 221, 222, drained by 223
    lib = os.path.join(ROOT, "oracle", "_ref", "libref_loaders.so")
    if os.path.exists(lib):   # the reference's own DwnNetwork loader reads the file (child process: see tests/test_ref_loaders.py)
        code = ("import ctypes as C, sys\n"
                "lib = C.CDLL(%r)\n"
                "lib.ref_network_new.restype = C.c_void_p; lib.ref_network_new.argtypes = [C.c_char_p]\n"
                "lib.ref_network_dims.argtypes = [C.c_void_p, C.c_void_p]\n"
                "lib.ref_network_array.restype = C.POINTER(C.c_float); lib.ref_network_array.argtypes = [C.c_void_p, C.c_char_p]\n"
                "h = lib.ref_network_new(%r.encode()); dims = (C.c_int * 8)(); lib.ref_network_dims(h, dims)\n"
                "assert list(dims[:4]) == [3, 5, 3, 1], list(dims[:4])\n"
                "b = lib.ref_network_array(h, b'matB'); assert [b[i] for i in range(3)] == [-1.0, 1.0, 0.0]\n"
                "print('OK', flush=True); import os; os._exit(0)\n") % (lib, out)
        env = dict(os.environ)
        pad = os.path.join(ROOT, "oracle", "_ref", "libmalloc_pad.so")
        if os.path.exists(pad):
            env["LD_PRELOAD"] = pad
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0 and "OK" in r.stdout, (r.stdout, r.stderr[-2000:])


def test_valves_and_networks_without_coupling(tmp_path):
    inp = tmp_path / "v.inp"
    inp.write_text("""[JUNCTIONS]
;ID Elev Demand Pattern
 J1 10 5 ;
 J2 10 7 ;
[RESERVOIRS]
 R1 100 ;
[TANKS]
;ID Elevation InitLevel MinLevel MaxLevel Diameter MinVol VolCurve
 T1 50 10 2 20 5 0 ;
 T2 50 10 1 30 5 0 ;
[PIPES]
;ID Node1 Node2 Length Diameter Roughness MinorLoss Status
 P1 T1 J1 10 1 100 0 Open ;
 P2 J2 T2 10 1 100 0 Open ;
[PUMPS]
 PU1 R1 T1 HEAD 1 ;
[VALVES]
;ID Node1 Node2 Diameter Type Setting MinorLoss
 V1 T1 T2 12 PRV 0 0 ;
[TAGS]
""")
    d = epanet.parse_epanet(str(inp))
    assert (d["nx"], d["nu"], d["nd"], d["ne"]) == (2, 2, 2, 1)
    assert d["matB"].tolist() == [[-1.0, 1.0], [0.0, -1.0]]          # pump fills T1; valve T1 -> T2
    assert d["matGd"].tolist() == [[1.0, 0.0], [0.0, 1.0]]
    assert d["matE"].tolist() == [[0.0, 0.0]] and d["matEd"].tolist() == [[0.0, 0.0]]   # no junction on a pump or valve
    assert d["vecXmin"].ravel().tolist() == [2.0, 1.0] and d["vecXmax"].ravel().tolist() == [20.0, 30.0]
    full = epanet.add_horizon_and_safety(d, horizon=12, safety=0.25)
    txt = epanet.to_json_text(full)
    j = json.loads(txt)
    assert j["N"] == [12] and j["vecXsafe"] == [5.0, 7.5]
