#!/usr/bin/env python3
"""Phase breakdown of the find kernel: the counted build of the kernels (blurrily_storage_set_stats) also keeps
wave 0's shader clocks per phase of the sweep.

Usage on the GPU box:
    python tools/phase_profile.py [scale] [n_queries]
Prints shader clocks per window per phase as seen by wave 0 of each workgroup.
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np  # noqa: E402
import workloads as W  # noqa: E402
from blurrily_amd import RawMap, _native  # noqa: E402

# needles with <= 64 trigrams run sweep_coop: there slot 0 also holds the selection of the previous step and
# slot 7 is the publishing turn (next window's unit descriptors); in sweep_pipelined slot 7 is the selection
PHASES = ["loop/table (+select)", "count prefetched", "count rest", "barrier(count)", "issue next head", "scan",
          "barrier(scan)", "publish | select"]


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    n = int(8423769 * scale)
    if os.environ.get("PH_WORKLOAD"):                                  # e.g. words, skewed: bench.py's haystacks
        hay, off = W.bench_haystack(os.environ["PH_WORKLOAD"], scale)
        n = len(off) - 1
    else:
        hay, off = W.geonames(n, max(1000, int(500000 * min(1.0, scale * 4))), 3)
    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    for key in ("nm_cmin", "nm_dense", "wsweep", "nm_min_windows", "ws_autotune", "few_max"):                 # e.g. NM_CMIN=0: nothing left out of the needle-major count
        if os.environ.get(key.upper()):
            try:
                m.set_option(key, int(os.environ[key.upper()]))
            except OSError:                                           # (a build from before round 4)
                pass
    m.sync_device()
    info = m.device_info()
    lib = _native.lib()
    lib.blurrily_debug_phase_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    m.set_stats(True)
    for batch in ([int(x) for x in os.environ['PH_BATCHES'].split()] if os.environ.get('PH_BATCHES') else (512, nq)):
        qp, qo = W.queries(hay, off, batch, 3000)
        t = time.perf_counter()
        rows, counts = m.find_batch_packed(qp, qo, 10)
        dt = time.perf_counter() - t
        nwg = min(batch, 512)
        full = np.zeros((4096 + nwg, 16), dtype=np.uint64)           # rows 4096..: sweep_role's manager waves
        assert lib.blurrily_debug_phase_clocks(m.handle, full.ctypes.data, 4096 + nwg) == 0
        buf, mgr = full[:nwg], full[4096:]
        tot_all = buf.sum(axis=0).astype(np.float64)
        tot = tot_all[:8]
        per_window = tot / (batch * info["n_windows"])
        print(f"batch {batch}: {dt * 1e3:.2f} ms wall, {info['n_windows']} windows; clocks per (query, window):")
        for name, v in zip(PHASES, per_window):
            print(f"   {name:22s} {v:9.0f}")
        print(f"   {'TOTAL':22s} {per_window.sum():9.0f}")
        if mgr.sum():
            mg = mgr.sum(axis=0).astype(np.float64)[:8] / (batch * info["n_windows"])
            print("   manager wave (sweep_role): " + "  ".join(f"{nm_} {v:.0f}" for nm_, v in zip(
                ["loop", "-", "choose+fetch", "barrier(count)", "-", "settle", "barrier(scan)", "publish"], mg)) + f"  TOTAL {mg.sum():.0f}")
        st = m.find_stats()
        print(f"   steps/needle {st['steps'] / batch:.1f} postings/needle {st['posting_entries'] / batch:.0f} last_sweep {m.get_option('last_sweep')} "
              f"compactions/needle {st['compactions'] / batch:.2f} resweeps/needle {st['resweeps'] / batch:.2f} tasks/needle {st['tasks'] / batch:.1f}")
        per_needle = tot_all[10:13] / batch
        print(f"   per needle: setup {per_needle[0]:9.0f}  sweeps {per_needle[1]:9.0f}  final compaction + rows {per_needle[2]:9.0f}")
        if tot_all[8]:
            print(f"   wave 0 head units per (query, window): {tot_all[8] / (batch * info['n_windows']):.2f}, "
                  f"live lanes per unit: {tot_all[9] / tot_all[8]:.1f} of 64")


if __name__ == "__main__":
    main()
