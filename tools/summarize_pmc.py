"""Median FETCH_SIZE / WRITE_SIZE per launch and kernel from two rocprofv3 --pmc passes -> the JSON kept in profiles/traffic.json."""
import csv, glob, json, os, statistics, subprocess, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapidnet_amd import build as _build  # noqa: E402


def load(directory, counter):
    out = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            out.setdefault(row["Kernel_Name"], {}).setdefault(row.get("Dispatch_Id", len(out)), 0.0)
            out[row["Kernel_Name"]][row.get("Dispatch_Id")] += float(row["Counter_Value"])
    return {k: list(v.values()) for k, v in out.items()}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
raw = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith(("void rn::", "rn::")):
        continue
    raw[k] = {"FETCH_SIZE_KiB_median": statistics.median(fetch[k]) if k in fetch else None,
              "WRITE_SIZE_KiB_median": statistics.median(write[k]) if k in write else None,
              "launches": len(fetch.get(k, write.get(k, [])))}


def total(prefix):
    for k, v in raw.items():
        if k.startswith(prefix) and v["FETCH_SIZE_KiB_median"] is not None and v["launches"] >= 4:
            return 1024.0 * (2.0 * v["FETCH_SIZE_KiB_median"] + (v["WRITE_SIZE_KiB_median"] or 0.0))
    return None


def _git_head():
    try:
        return subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


print(json.dumps({
    "kernels_sha256": _build.kernel_sources_sha256(), "collected_at_commit": _git_head(),
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_traffic.sh), bench.py --steps 20 --dense-only, "
              "barcelona493 fp64, medians per launch",
    "correction": "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request on wide streaming reads: MI355X_MICROARCH.md, HBM); WRITE_SIZE as is; KiB*1024",
    "k_stream_gemv_bytes_per_launch": total("void rn::k_stream_gemv<double"),
    "k_dual_stage_bytes_per_launch": total("void rn::k_dual_stage<double, false"),
    "raw": raw}, indent=1))
