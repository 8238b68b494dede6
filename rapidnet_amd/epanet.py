"""EPANET .inp -> DwnNetwork JSON (the reference's offline tool chain src/paser/parserEpanet.m + generateJsonFile.m +
createDwnDataJson.m, MATLAB).

    python -m rapidnet_amd.epanet network.inp network.json [--horizon 24] [--safety 0.35]

What the reference's parser derives from the EPANET sections, and this module with it:
  [TANKS]      -> states: nx tanks, vecXmin / vecXmax = MinLevel / MaxLevel (parserEpanet.m:70-97, 278-279)
  [PUMPS]+[VALVES] -> inputs, pumps first: matB(tank, link) = +1 if the link's Node1 is the tank, -1 if its Node2 is
                  (parserEpanet.m:190-214)
  [JUNCTIONS]  -> demands: matGd(tank, junction) = 1 if a pipe joins them (parserEpanet.m:215-228)
  [PIPES]      -> only used for that adjacency
  input-demand coupling: one row of matE / matEd per junction that is an end node of a pump or valve:
                  matE(row, link) = +1 (junction is Node1) / -1 (Node2), matEd(row, junction) = -1 (parserEpanet.m:231-266);
                  a single zero row if there is none (:268-271)
  fixed choices of the reference: matA = I, vecUmin = 0, vecUmax = 100, costAlpha1 = 10 (:276-284); N = 24 and
                  vecXsafe = 0.35 vecXmax are added by createDwnDataJson.m:9-10.
The writer reproduces generateJsonFile.m byte for byte (field order, scalars as 1-element arrays, matrices column-major
with one column per line, MATLAB's %d rendering of non-integers as %e) so that the reference-held pair
src/paser/testEpanet.inp -> src/paser/network.json is a golden test (tests/test_epanet.py).

Differences from the MATLAB code: sections are read line by line (comments and the trailing ';' stripped) instead of
walking a whitespace token stream with hard-coded header lengths; the valve loops of parserEpanet.m index the pump arrays
(valveNode1(iPump), :204-211, :254-264 -- unreachable for the reference's own file, which has no valves) and are
implemented here as the pump loops are.
"""
import sys

import numpy as np


def read_sections(path):
    """{SECTION: [token list per data line]} of an EPANET input file."""
    sections, cur = {}, None
    with open(path) as f:
        for raw in f:
            line = raw.strip()
            if not line:
                continue
            if line.startswith("["):
                cur = line[1:line.index("]")].upper()
                sections.setdefault(cur, [])
                continue
            if cur is None or line.startswith(";"):
                continue
            body = line.split(";", 1)[0].split()       # drop the row terminator / trailing comment
            if body:
                sections[cur].append(body)
    return sections


def parse_epanet(path):
    """dwnData of parserEpanet.m as a dict of numpy arrays."""
    s = read_sections(path)
    junction_id = [r[0] for r in s.get("JUNCTIONS", [])]
    tanks = s.get("TANKS", [])
    tank_id = [r[0] for r in tanks]
    x_min = np.array([float(r[3]) for r in tanks])
    x_max = np.array([float(r[4]) for r in tanks])
    pipes = [(r[1], r[2]) for r in s.get("PIPES", [])]
    links = [(r[1], r[2]) for r in s.get("PUMPS", [])] + [(r[1], r[2]) for r in s.get("VALVES", [])]
    nx, nu, nd = len(tank_id), len(links), len(junction_id)
    B = np.zeros((nx, nu))
    Gd = np.zeros((nx, nd))
    for i, t in enumerate(tank_id):
        for j, (n1, n2) in enumerate(links):
            B[i, j] = 1.0 if n1 == t else (-1.0 if n2 == t else 0.0)
        neighbours = [b if a == t else a for a, b in pipes if t in (a, b)]
        for k, jn in enumerate(junction_id):
            if jn in neighbours:
                Gd[i, k] = 1.0
    rows_e, rows_ed = [], []
    for k, jn in enumerate(junction_id):
        row = np.zeros(nu)
        hit = False
        for j, (n1, n2) in enumerate(links):
            if jn == n1:
                row[j], hit = 1.0, True
            if jn == n2:
                row[j], hit = -1.0, True
        if hit:
            ed = np.zeros(nd)
            ed[k] = -1.0
            rows_e.append(row); rows_ed.append(ed)
    E = np.array(rows_e) if rows_e else np.zeros((1, nu))
    Ed = np.array(rows_ed) if rows_ed else np.zeros((1, nd))
    return {"nx": nx, "nu": nu, "ne": E.shape[0], "nd": nd, "matA": np.eye(nx), "matB": B, "matGd": Gd, "matE": E, "matEd": Ed,
            "vecXmin": x_min.reshape(-1, 1), "vecXmax": x_max.reshape(-1, 1), "vecUmin": np.zeros((nu, 1)),
            "vecUmax": 100.0 * np.ones((nu, 1)), "costAlpha1": 10.0 * np.ones((nu, 1))}


def add_horizon_and_safety(dwn, horizon=24, safety=0.35):
    """createDwnDataJson.m:9-10."""
    out = dict(dwn)
    out["N"] = horizon
    out["vecXsafe"] = safety * dwn["vecXmax"]
    return out


def _matlab_d(v):
    """fprintf('%d', v): integers as integers, anything else in %e notation."""
    v = float(v)
    return "%d" % int(v) if v == int(v) else "%e" % v


def to_json_text(data):
    """generateJsonFile.m:15-50."""
    parts = ["{ \n"]
    keys = list(data.keys())
    for i, key in enumerate(keys):
        val = data[key]
        if isinstance(val, str):
            parts.append('"%s" : "%s"' % (key, val))
        elif np.isscalar(val) or np.asarray(val).size == 1:
            parts.append('"%s" : [%s]' % (key, _matlab_d(np.asarray(val).ravel()[0])))
        else:
            m = np.atleast_2d(np.asarray(val, float))
            rows, cols = m.shape
            parts.append('"%s" : [' % key)
            for jj in range(1, cols + 1):
                for kk in range(1, rows + 1):
                    parts.append(_matlab_d(m[kk - 1, jj - 1]) + (", " if jj * kk < m.size else "]"))
                if jj * rows < m.size:
                    parts.append("\n")
        if i < len(keys) - 1:
            parts.append(",\n")
    parts.append("\n}")
    return "".join(parts)


def convert(inp_path, json_path, horizon=24, safety=0.35):
    text = to_json_text(add_horizon_and_safety(parse_epanet(inp_path), horizon, safety))
    with open(json_path, "w") as f:
        f.write(text)
    return text


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("inp")
    ap.add_argument("json")
    ap.add_argument("--horizon", type=int, default=24)
    ap.add_argument("--safety", type=float, default=0.35)
    a = ap.parse_args()
    convert(a.inp, a.json, a.horizon, a.safety)
    sys.exit(0)
