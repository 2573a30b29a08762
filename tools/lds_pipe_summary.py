#!/usr/bin/env python3
"""LDS-pipe occupancy of the batch kernels from the rocprofv3 --pmc passes tools/lds_pipe.sh leaves in
<dir>/pmc_{idx,inst,mix}_<workload>/ (sqlite output).

Per workload, for the ONE kernel the timed step spends its time in (the counted build's launch -- namespace
blurrily::counted -- left out): the SQ counters summed over its dispatches, and

  lds_busy_frac      SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES   -- the share of a busy CU's cycles in which its LDS array works
                     (MI355X_MICROARCH.md: SQ_LDS_IDX_ACTIVE = all LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra ones)
  conflict_frac      SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_lds_frac      SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES       -- the share of wave-cycles spent waiting on an LDS instruction
  valu_busy_frac     SQ_ACTIVE_INST_VALU x 4 cycles / (4 SIMDs x SQ_BUSY_CU_CYCLES) -- the share of the VALU issue slots taken
                     (rocprofiler's derived VALUBusy = 100*SQ_ACTIVE_INST_VALU*4/SIMD_NUM/GRBM_GUI_ACTIVE, over busy CU cycles)

Prints one JSON object stamped with bench.py's kernel_source_hash (what the bench line checks freshness against)."""
import glob
import json
import os
import sqlite3
import sys

LEAVE = "find_kernel<unsigned char, 1024, false, true>"
KERNEL = {"geonames": LEAVE, "words": "find_small_kernel", "geonames_x4": LEAVE, "skewed": LEAVE, "geonames_miss": LEAVE}


def counters(dirpath, kernel):
    dbs = glob.glob(os.path.join(dirpath, "**", "*.db"), recursive=True)
    if not dbs:
        return {}
    c = sqlite3.connect(dbs[0])
    rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), sum(duration) "
                     "from counters_collection group by kernel_name, counter_name").fetchall()
    out = {}
    for k, n, v, disp, dur in rows:
        if kernel not in k or "blurrily::counted::" in k:
            continue
        out[n] = out.get(n, 0.0) + v
        out["_dispatches"] = disp
    return out


def main():
    base = sys.argv[1]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = {"_how": " ".join(__doc__.split("\n\n")[1].split()), "kernel_source_hash": bench.kernel_source_hash(),
           "commit": os.environ.get("PROFILE_COMMIT")}
    for wl, kernel in KERNEL.items():
        d = {"kernel": kernel}
        for sub in ("idx", "inst", "mix", "valu", "salu"):
            d.update(counters(os.path.join(base, f"pmc_{sub}_{wl}"), kernel))
        g = d.get
        if g("SQ_LDS_IDX_ACTIVE") and g("SQ_BUSY_CU_CYCLES"):
            d["lds_busy_frac"] = g("SQ_LDS_IDX_ACTIVE") / g("SQ_BUSY_CU_CYCLES")
            d["conflict_frac"] = g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE")
        # a wave64 VALU instruction holds its SIMD for one quad-cycle (SQ_ACTIVE_INST_VALU counts those; it equals
        # SQ_INSTS_VALU to within a percent on these kernels), a CU has four SIMDs: busy share of the VALU issue slots
        valu_q = g("SQ_ACTIVE_INST_VALU") or g("SQ_INSTS_VALU")
        if valu_q and g("SQ_BUSY_CU_CYCLES"):
            d["valu_busy_frac"] = valu_q * 4.0 / (4.0 * g("SQ_BUSY_CU_CYCLES"))
            d["valu_busy_from"] = "SQ_ACTIVE_INST_VALU" if g("SQ_ACTIVE_INST_VALU") else "SQ_INSTS_VALU"
        if g("SQ_INST_CYCLES_SALU") and g("SQ_BUSY_CU_CYCLES"):     # (rocprofiler's SALUBusy, the same way)
            d["salu_busy_frac"] = g("SQ_INST_CYCLES_SALU") * 4.0 / (4.0 * g("SQ_BUSY_CU_CYCLES"))
        if g("SQ_WAIT_INST_LDS") and g("SQ_WAVE_CYCLES"):
            d["wait_lds_frac"] = g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES")
        if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
            d["wait_any_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        try:
            line = json.load(open(os.path.join(base, f"pmc_idx_{wl}", "detail.json")))      # (bench.py --detail: the full record)
            d["kernel_ms_under_pmc"] = line.get("kernel_ms")
            d["sweep"] = line.get("roofline", {}).get("sweep")
        except Exception:
            pass
        out[wl] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
