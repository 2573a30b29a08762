#!/usr/bin/env python3
"""bench.py -- batched trigram find on MI355X (the metric of BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (device tokeniser + find kernels) over one batch of
synthetic needles that is already resident in HBM.  At N=1 the workload is
BASELINE.json configs[2]: the synthetic Geonames-scale haystack (8 423 769 multi-word strings,
~118 M trigram entries) and one batch of 1 M needles.  For N>1 (configs[3]) the haystack is
replicated on every GPU, every rank gets its own 1 M-needle shard (weak scaling) and the
per-rank result blocks are collected on rank 0 by one RCCL gather inside the timed region.

Rank 0 prints ONE JSON line; `value` is whole-job needles/s.  The same line carries
`roofline` (algorithmic bytes of SURVEY.md section 8(d) over the HIP-event time of the find
kernels) and, at N=1, `cpu_baseline` (the reference's own C -- oracle/_ref -- timed on one
host core on a bounded sample of the same needles).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (imported before the HIP library so both share one HIP runtime)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (haystack generator kwargs, needles per rank, limit, BASELINE.json config)
    "geonames": dict(kind="geonames", n=8423769, vocab=500000, hay_seed=3, queries=1_000_000, limit=10,
                     label="configs[2]: synthetic Geonames-scale haystack, 1M batched needles"),
    "words":    dict(kind="words", n=235886, hay_seed=1, queries=100_000, limit=10,
                     label="configs[1]: 235k-word haystack, 100k batched needles"),
    "skewed":   dict(kind="skewed", n=4_000_000, hay_seed=5, queries=100_000, limit=100,
                     label="configs[4]: adversarial hot-trigram haystack, limit=100"),
}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def build_haystack(spec, scale):
    import workloads as W
    from blurrily_amd import RawMap
    n = max(1000, int(spec["n"] * scale))
    t0 = time.time()
    if spec["kind"] == "geonames":
        hay, off = W.geonames(n, max(1000, int(spec["vocab"] * min(1.0, scale * 4))), spec["hay_seed"])
    elif spec["kind"] == "words":
        hay, off = W.words(n, spec["hay_seed"])
    else:
        hay, off = W.skewed(n, spec["hay_seed"])
    t1 = time.time()
    m = RawMap()
    refs = np.arange(1, n + 1, dtype=np.uint32)
    entries = m.put_many_packed(hay, off, refs)
    t2 = time.time()
    m.sync_device()
    t3 = time.time()
    log(f"haystack: {n} strings, {entries} entries ({entries / n:.2f}/string); "
        f"generate {t1 - t0:.1f}s, put {t2 - t1:.1f}s, device index {t3 - t2:.1f}s")
    return m, hay, off, entries


def cpu_baseline(m, hay, hay_off, qp, qo, limit, budget_s):
    """The reference's own C (oracle/_ref, kind "reference") -- or, if that build is absent, the
    oracle port -- on ONE host core (the reference is single-threaded), on a bounded prefix of
    the step's needles.  The haystack reaches the reference as a .trigrams file."""
    import workloads as W
    from helpers import Oracle, Reference
    raw = W.unpack(qp, qo[:min(len(qo) - 1, 2048) + 1])
    packed = np.frombuffer(b"\0".join(raw) + b"\0", dtype=np.uint8)      # C strings
    starts = np.zeros(len(raw), dtype=np.uint32)
    starts[1:] = np.cumsum([len(r) + 1 for r in raw])[:-1]
    rows = (C.c_uint32 * (3 * max(limit, 1)))()
    if Reference.available():
        kind = "reference"
        path = f"/tmp/blurrily_bench_{os.getpid()}.trigrams"
        m.save(path)
        ref = Reference(path)
        S = Reference.shim()

        def run(lo, hi):
            seg = np.ascontiguousarray(starts[lo:hi])
            t = time.perf_counter()
            S.ref_find_many(ref.h, packed.ctypes.data, seg.ctypes.data, hi - lo, limit, rows)
            return time.perf_counter() - t

        def done():
            ref.close()
            os.unlink(path)
    else:
        kind = "port"
        o = Oracle()
        o.put_many(hay, hay_off)

        def run(lo, hi):
            t = time.perf_counter()
            for k in range(lo, hi):
                o.L.oracle_find(o.h, raw[k], limit, rows)
            return time.perf_counter() - t

        def done():
            pass
    run(0, 1)                                        # page the index in
    k = min(4, len(raw))
    per = run(0, k) / k
    n = int(max(k, min(len(raw), budget_s / max(per, 1e-7))))
    dt = run(0, n)
    out = {"value": n / dt, "unit": "queries/s", "cores": 1, "kind": kind,
           "sample": f"first {n} needles of the step batch, limit {limit}, one thread "
                     f"(flags of ext/blurrily/extconf.rb: -Os)",
           "ms_per_query": 1e3 * dt / n}
    # The reference is single-threaded; for scale, the same read-only map queried by one forked
    # process per host core (each maps the same file), every process timing the same n needles.
    if kind == "reference" and hasattr(os, "fork"):
        try:
            cores = min(os.cpu_count() or 1, 128)
            import multiprocessing as mp
            ctx = mp.get_context("fork")
            q = ctx.Queue()

            n_all = max(2, n // 10)              # memory-bound when every core runs: keep it short

            def worker(i):
                t0 = time.perf_counter()
                run(0, n_all)
                q.put((t0, time.perf_counter()))
            procs = [ctx.Process(target=worker, args=(i,)) for i in range(cores)]
            for p_ in procs:
                p_.start()
            spans = [q.get(timeout=300) for _ in procs]
            for p_ in procs:
                p_.join()
            wall = max(e for _, e in spans) - min(b for b, _ in spans)
            out["all_cores"] = {"value": cores * n_all / wall, "unit": "queries/s", "cores": cores,
                                "note": f"one process per hardware thread on the shared read-only map, "
                                        f"{n_all} needles each"}
        except Exception as e:                       # informational only
            out["all_cores"] = {"error": str(e)}
    done()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="geonames", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink haystack and batch (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline work")
    ap.add_argument("--latency-probes", type=int, default=200)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with python -m torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: blurrily_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import workloads as W
    from blurrily_amd import _native
    from blurrily_amd.sharding import gather_results

    spec = dict(WORKLOADS[args.workload])
    limit = spec["limit"]
    n_q = max(100, int(spec["queries"] * args.scale))
    m, hay, hay_off, entries_resident = build_haystack(spec, args.scale)
    # this rank's shard of the global batch (world x n_q needles, contiguous shards)
    q_seed = (3 if world == 1 else 4) * 1000 + rank
    qp, qo = W.queries(hay, hay_off, n_q, q_seed)
    sum_T = W.count_trigrams(qp, qo)

    dev = torch.device("cuda", local_rank)
    d_packed = torch.from_numpy(qp).to(dev)
    d_off = torch.from_numpy(qo.astype(np.int64)).to(dev)
    d_results = torch.empty((n_q, limit, 3), dtype=torch.int32, device=dev)
    d_counts = torch.empty((n_q,), dtype=torch.int32, device=dev)
    d_nb = torch.empty((n_q,), dtype=torch.int32, device=dev)
    gathered = None
    if world > 1 and rank == 0:
        gathered = (torch.empty((world, n_q, limit, 3), dtype=torch.int32, device=dev),
                    torch.empty((world, n_q), dtype=torch.int32, device=dev))
    lib = _native.lib()
    m.set_timing(True)
    stream = torch.cuda.current_stream().cuda_stream
    kernel_ms = []

    def step():
        res = lib.blurrily_storage_find_batch_device(
            m.handle, d_packed.data_ptr(), int(qo[-1]), d_off.data_ptr(), n_q, limit,
            d_results.data_ptr(), d_counts.data_ptr(), d_nb.data_ptr(), stream)
        if res < 0:
            raise RuntimeError(f"find_batch_device failed: errno {C.get_errno()}")
        kernel_ms.append(m.device_info()["last_find_kernel_ms"])
        if world > 1:
            gather_results(dist, d_results, d_counts, gathered, rank)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- derived figures (outside the timed region) ----------------------------------------
    nb = d_nb.cpu().numpy().astype(np.uint32).astype(np.int64)
    counts = d_counts.cpu().numpy().astype(np.int64)
    sum_nb, sum_rows = int(nb.sum()), int(counts.sum())
    algo_bytes = 8 * sum_nb + 8 * sum_T + 12 * sum_rows + 4 * n_q            # SURVEY.md 8(d), one launch
    k_ms = float(np.mean(kernel_ms))
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9
    totals = torch.tensor([float(sum_nb)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(totals)
    total_entries = float(totals.item())

    out = None
    if rank == 0:
        # p50 single-needle latency through blurrily_storage_find (host buffers, sync per call)
        raw = W.unpack(qp, qo[:args.latency_probes + 1])
        rows = (_native.TrigramMatch * limit)()
        lat = []
        for nd in raw:
            t = time.perf_counter()
            lib.blurrily_storage_find(m.handle, nd, limit, rows)
            lat.append(time.perf_counter() - t)
        p50_us = float(np.median(lat) * 1e6) if lat else None
        # the same batch through the host-buffer entry point: H2D of the needles and D2H of the
        # result rows included (reported beside `value`, never as `value`)
        t = time.perf_counter()
        m.find_batch_packed(qp, qo, limit)
        host_rate = n_q / (time.perf_counter() - t)
        info = m.device_info()
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and args.scale == 1.0:      # measured on the full workload only
            try:
                traffic = json.load(open(tpath)).get(args.workload)
            except Exception:
                traffic = None
        out = {
            "metric": "find() queries/sec (batched), Geonames-scale haystack",
            "value": world * n_q * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": spec["label"], "haystack_strings": int(len(hay_off) - 1),
                       "haystack_entries": int(entries_resident), "needles_per_gpu": n_q, "limit": limit,
                       "index_replicated": world > 1, "parallelism": f"query-shard x{world}",
                       "scale": args.scale},
            "p50_query_us": p50_us,
            "host_buffer_queries_per_sec": host_rate,
            "matched_entries_per_sec": total_entries * args.steps / elapsed,
            "entries_per_query": sum_nb / n_q,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # the physical stream: PMC bytes per launch (a separate rocprofv3 --pmc run of this
                         # workload, profiles/) over this run's kernel time, against the same peak
                         "traffic_gbs": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "kernel": "find_kernel<uint8_t,%s>" % os.environ.get("BLURRILY_FIND_THREADS", "1024"),
                         "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "resident_index_bytes": int(info["device_bytes"])},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(m, hay, hay_off, qp, qo, limit, args.cpu_budget)
            except Exception as e:  # the GPU numbers stand on their own
                out["cpu_baseline"] = {"error": str(e)}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
