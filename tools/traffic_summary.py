#!/usr/bin/env python3
"""HBM-side traffic of the dominant find launch from the rocprofv3 --pmc passes that
tools/collect_profiles.sh leaves in <dir>/pmc_{fetch,write,tcc}/ (sqlite output).

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE
and WRITE_SIZE are KiB, and on gfx950 FETCH_SIZE reports half of a wide coalesced read stream.
Infinity-Cache hits are included, so this is an upper bound on HBM bytes (the 247 MB posting
array of the Geonames-scale index fits the 256 MiB Infinity Cache); TCC_EA0_RDREQ x 64 B is
reported beside it as the lower reading.  Prints one JSON object."""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise  # noqa: E402


def dominant(dirpath):
    dbs = glob.glob(os.path.join(dirpath, "**", "*.db"), recursive=True)
    if not dbs:
        return None, {}
    per_kernel = summarise(dbs[0], only="find_kernel")
    if not per_kernel:
        return None, {}
    # the launch that owns the time: largest max dispatch duration
    name = max(per_kernel, key=lambda k: max(v[2] for v in per_kernel[k].values()))
    return name, per_kernel[name]


def main():
    base = sys.argv[1]
    out = {"_how": __doc__.split("\n\n")[1].replace("\n", " ")}
    detail = {}
    kname = None
    for sub in ("pmc_fetch", "pmc_write", "pmc_tcc"):
        name, ctr = dominant(os.path.join(base, sub))
        if name is None:
            continue
        kname = kname or name
        for c, (v, disp, dur) in ctr.items():
            detail[c] = v / max(disp, 1)                # per launch
            detail.setdefault("dispatch_ns", dur)
    detail["kernel"] = kname
    if "FETCH_SIZE" in detail and "WRITE_SIZE" in detail:
        out["geonames"] = (2 * detail["FETCH_SIZE"] + detail["WRITE_SIZE"]) * 1024
        detail["fabric_TBps"] = out["geonames"] / (detail["dispatch_ns"] * 1e-9) / 1e12
    if "TCC_EA0_RDREQ_sum" in detail:
        detail["TCC_EA0_RDREQ_x64B"] = detail["TCC_EA0_RDREQ_sum"] * 64
    if "TCC_HIT_sum" in detail and "TCC_MISS_sum" in detail:
        detail["l2_hit_rate"] = detail["TCC_HIT_sum"] / (detail["TCC_HIT_sum"] + detail["TCC_MISS_sum"])
    out["detail"] = detail
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
