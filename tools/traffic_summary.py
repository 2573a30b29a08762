#!/usr/bin/env python3
"""Memory-side traffic of one bench step from the rocprofv3 --pmc passes that
tools/collect_profiles.sh leaves in <dir>/pmc_{fetch,write,tcc}_<workload>/ (sqlite output).

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE
and WRITE_SIZE are KiB, and on gfx950 FETCH_SIZE reports half of a wide coalesced read stream.  They
are the L2's memory-side (fabric) request counters: Infinity-Cache hits are included, so this bounds
HBM bytes from above; TCC_EA0_RDREQ x 64 B is reported beside it.  Summed over every kernel of the find
path (tokeniser, find_kernel, wsweep_kernel, finalize, merges) of the ONE timed step the profiled command
makes (bench.py --steps 1 --warmup 0 --static-choice); the kernels of the counted build (namespace
blurrily::counted: the extra, untimed launch that counts requests) are left out.
Prints one JSON object: {workload: bytes per step, "detail": {...}}."""
import glob
import json
import os
import sqlite3
import sys

OURS = ("find_kernel", "find_small_kernel", "find_one_kernel", "wsweep_kernel", "tokenise", "finalize_rows", "merge_", "normalise",
        "apply_tombstones")
CALLS_PER_RUN = 1


def totals(dirpath):
    dbs = glob.glob(os.path.join(dirpath, "**", "*.db"), recursive=True)
    if not dbs:
        return {}
    c = sqlite3.connect(dbs[0])
    rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) "
                     "from counters_collection group by kernel_name, counter_name").fetchall()
    out, per_kernel = {}, {}
    for k, n, v, disp in rows:
        if not any(o in k for o in OURS) or "blurrily::counted::" in k:
            continue
        out[n] = out.get(n, 0.0) + v
        short = k.replace("void ", "").replace("blurrily::(anonymous namespace)::", "").split("(")[0]
        per_kernel.setdefault(short, {})[n] = [v, disp]
    return {"sum": out, "per_kernel": per_kernel}


def kernel_source_hash():
    """bench.py's: sha256 over the code (comments apart) of the kernel and launch sources this profile was taken at."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernel_source_hash()


def main():
    base = sys.argv[1]
    out = {"_how": " ".join(__doc__.split("\n\n")[1].split()),
           # what bench.py checks the figure's freshness against (the GPU box has no .git: the commit comes from
           # the caller, PROFILE_COMMIT=$(git rev-parse --short HEAD) in the gpurun command line)
           "kernel_source_hash": kernel_source_hash(), "commit": os.environ.get("PROFILE_COMMIT")}
    detail = {}
    for wl in ("geonames", "words", "skewed", "geonames_x4", "geonames_miss"):
        d = {}
        for sub in ("fetch", "write", "tcc"):
            t = totals(os.path.join(base, f"pmc_{sub}_{wl}"))
            if t:
                d.update({k: v / CALLS_PER_RUN for k, v in t["sum"].items()})
                for k, ctr in t["per_kernel"].items():
                    d.setdefault("per_kernel", {}).setdefault(k, {}).update(ctr)
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            out[wl] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        if "TCC_EA0_RDREQ_sum" in d:
            d["TCC_EA0_RDREQ_x64B"] = d["TCC_EA0_RDREQ_sum"] * 64
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        if d:
            detail[wl] = d
    out["detail"] = detail
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
