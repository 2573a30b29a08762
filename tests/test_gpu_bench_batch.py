"""The call and the batch bench.py times, checked: blurrily_storage_find_batch_device on
BASELINE.json configs[2] -- haystack seed 3 (8 423 769 strings), needle seed 3000, ONE batch of
1 000 000 device-resident needles, limit 10, nb_entries requested -- with the inputs built by the
same tools/workloads.py functions bench.py uses.

  * every one of the 1 M needles: rows in the reference's total order (matches desc, weight asc,
    reference asc -- storage.c:129-138 + the stable-sort tie order pinned by
    spec/integration_spec.rb:37-42), no reference twice, 1 <= matches <= T, counts <= limit,
    weight == strlen of the indexed string (storage.c:409), and d_nb_entries equal to the
    oracle's nb_entries (storage.c:498-503) -- all 1 M of them;
  * the first 100 000 needles row for row, through SHA-256 digests of oracle-produced result blocks
    (tests/golden/digest_*.json, tools/make_digests.py; configs[1] and configs[4]: that is the whole batch);
  * stratified by kernel path: a second launch with the kernels' request counters on yields, per needle, the
    paths its find went through (blurrily_storage_find_path_flags: 4-bit / byte counters, cold start, pool
    overflow and re-sweep, windows stepped over, left-out slices, robust scan ...); its rows must equal the
    timed launch's, every class that occurs in the batch must occur among the digest-covered needles, and a
    further sample drawn class by class from the REST of the batch is compared with the oracle row for row;
  * sampled needles of the whole batch row for row against the oracle;
  * the host-buffer entry point on a slice of the same batch gives the same rows.
"""
import ctypes as C

import numpy as np
import pytest

import workloads as W
from blurrily_amd import RawMap, _native
from helpers import Oracle, block_digests, load_golden

pytestmark = pytest.mark.gpu


def _order_ok(rows, counts, limit):
    """Vectorised: consecutive rows ascend strictly in (-matches, weight, reference)."""
    r = rows.astype(np.int64)
    ok = True
    for k in range(limit - 1):
        valid = counts > k + 1
        a, b = r[valid, k], r[valid, k + 1]
        lt = (a[:, 1] > b[:, 1]) | ((a[:, 1] == b[:, 1]) & ((a[:, 2] < b[:, 2]) |
                                                            ((a[:, 2] == b[:, 2]) & (a[:, 0] < b[:, 0]))))
        ok &= bool(lt.all())
    return ok


@pytest.mark.parametrize("name,n_sample,per_class", [("geonames", 1200, 150), ("words", 3000, 0), ("skewed", 300, 0)])
def test_the_benched_call_on_the_benched_batch(name, n_sample, per_class, geonames_full):
    import torch
    spec = W.BENCH_WORKLOADS[name]
    limit = spec["limit"]
    hay, off = (geonames_full.hay, geonames_full.off) if name == "geonames" else W.bench_haystack(name)
    n = len(off) - 1
    qp, qo = W.bench_needles(hay, off, name)
    n_q = len(qo) - 1
    assert n == spec["n"] and n_q == spec["queries"]

    m = RawMap()
    m.put_many_packed(hay, off, np.arange(1, n + 1, dtype=np.uint32))
    m.sync_device()
    dev = torch.device("cuda", 0)
    d_packed = torch.from_numpy(qp).to(dev)
    d_off = torch.from_numpy(qo.astype(np.int64)).to(dev)
    d_results = torch.full((n_q, limit, 3), -1, dtype=torch.int32, device=dev)
    d_counts = torch.full((n_q,), -1, dtype=torch.int32, device=dev)
    d_nb = torch.full((n_q,), -1, dtype=torch.int32, device=dev)
    lib = _native.lib()
    stream = torch.cuda.current_stream().cuda_stream
    res = lib.blurrily_storage_find_batch_device(m.handle, d_packed.data_ptr(), int(qo[-1]), d_off.data_ptr(), n_q,
                                                 limit, d_results.data_ptr(), d_counts.data_ptr(), d_nb.data_ptr(),
                                                 stream)
    assert res == 0, C.get_errno()
    torch.cuda.synchronize()
    rows = d_results.cpu().numpy().view(np.uint32)
    counts = d_counts.cpu().numpy().view(np.uint32).astype(np.int64)
    nb = d_nb.cpu().numpy().view(np.uint32).astype(np.uint64)

    if name == "geonames":
        o = geonames_full.oracle                                   # (the session's: tests/conftest.py)
    else:
        o = Oracle()
        o.put_many(hay, off)
    every = o.batch(qp, qo, find=False, nb=True, ntri=True)

    # ---- all needles ---------------------------------------------------------------------
    assert np.array_equal(nb, np.minimum(every["nb"], 0xFFFFFFFF)), "d_nb_entries != oracle nb_entries"
    assert counts.min() >= 1 and counts.max() <= limit           # every needle is an edited haystack string
    assert _order_ok(rows, counts, limit)
    live = np.arange(limit)[None, :] < counts[:, None]
    T = every["ntri"].astype(np.int64)
    matches = rows[:, :, 1].astype(np.int64)
    assert ((matches >= 1) | ~live).all() and ((matches <= T[:, None]) | ~live).all()
    refs = rows[:, :, 0].astype(np.int64)
    assert ((refs >= 1) & (refs <= n) | ~live).all()
    srt = np.sort(np.where(live, refs, -np.arange(1, limit + 1)[None, :]), axis=1)   # distinct fillers
    assert (srt[:, 1:] != srt[:, :-1]).all(), "a reference twice in one needle's rows"
    hay_len = (off[1:] - off[:-1]).astype(np.int64)               # weight 0 -> strlen (storage.c:409)
    assert (np.where(live, rows[:, :, 2].astype(np.int64) - hay_len[np.clip(refs - 1, 0, n - 1)], 0) == 0).all()

    # ---- a sample, row for row against the oracle ------------------------------------------
    rng = np.random.default_rng(2024)
    idx = np.sort(rng.choice(n_q, size=n_sample, replace=False)).astype(np.uint32)
    want = o.batch(qp, qo, idx=idx, limit=limit)
    assert np.array_equal(counts[idx], want["counts"])
    for k, q in enumerate(idx):
        c = int(counts[q])
        assert np.array_equal(rows[q, :c], want["rows"][k, :c]), (int(q), rows[q, :c].tolist(),
                                                                  want["rows"][k, :c].tolist())

    # ---- the first 100 000 needles, row for row, through digests of the oracle's result blocks ------
    covered = min(n_q, 100_000)
    gold = load_golden(f"digest_{name}.json")
    assert gold["limit"] == limit and gold["needles"] == covered
    got = block_digests(rows[:covered], counts[:covered], gold["block"])
    bad = [k for k, (a, b) in enumerate(zip(got, gold["digests"])) if a != b]
    assert not bad and len(got) == len(gold["digests"]), f"result blocks {bad[:8]} (of {gold['block']} needles) differ"
    assert int(counts[:covered].sum()) == gold["sum_counts"]

    # ---- by kernel path: the counted build says which paths every needle took ----------------------
    m.set_stats(True)
    d_results2 = torch.full_like(d_results, -1)
    d_counts2 = torch.full_like(d_counts, -1)
    res = lib.blurrily_storage_find_batch_device(m.handle, d_packed.data_ptr(), int(qo[-1]), d_off.data_ptr(), n_q,
                                                 limit, d_results2.data_ptr(), d_counts2.data_ptr(), 0, stream)
    assert res == 0, C.get_errno()
    torch.cuda.synchronize()
    flags = m.find_path_flags(n_q)
    m.set_stats(False)
    counts2 = d_counts2.cpu().numpy().view(np.uint32).astype(np.int64)
    rows2 = d_results2.cpu().numpy().view(np.uint32)
    live = np.arange(limit)[None, :] < counts[:, None]
    assert np.array_equal(counts2, counts) and np.array_equal(np.where(live[:, :, None], rows2, 0),
                                                              np.where(live[:, :, None], rows, 0))
    assert flags.all(), "a needle without any path flag"
    hist = {nm: int(((flags >> b) & 1).sum()) for b, nm in enumerate(m.PATH_FLAGS) if ((flags >> b) & 1).any()}
    print(f"{name}: needles by kernel path {hist}")
    rest_idx = []
    for b, nm in enumerate(m.PATH_FLAGS):
        has = np.nonzero((flags >> b) & 1)[0]
        if len(has) == 0:
            continue
        in_cover = int((has < covered).sum())
        # a class of at least 1 in 10 000 needles is present among the digest-covered needles ...
        assert in_cover > 0 or len(has) * 10000 < n_q, (nm, len(has))
        # ... and sampled from the rest of the batch as well
        outside = has[has >= covered]
        if len(outside):
            rest_idx.append(np.random.default_rng(b).choice(outside, size=min(len(outside), per_class), replace=False))
    if rest_idx:
        idx2 = np.unique(np.concatenate(rest_idx)).astype(np.uint32)
        want2 = o.batch(qp, qo, idx=idx2, limit=limit)
        assert np.array_equal(counts[idx2], want2["counts"])
        for k, q in enumerate(idx2):
            c = int(counts[q])
            assert np.array_equal(rows[q, :c], want2["rows"][k, :c]), (int(q), hex(int(flags[q])))

    # ---- the host-buffer entry point agrees on a slice of the batch --------------------------
    lo, hi = n_q // 2, n_q // 2 + 5000
    sub_off = (qo[lo:hi + 1] - qo[lo]).astype(np.uint64)
    sub = qp[int(qo[lo]):int(qo[hi])]
    h_rows, h_counts = m.find_batch_packed(sub, sub_off, limit)
    assert np.array_equal(h_counts.astype(np.int64), counts[lo:hi])
    live_s = np.arange(limit)[None, :] < counts[lo:hi, None]
    assert np.array_equal(np.where(live_s[:, :, None], h_rows, 0), np.where(live_s[:, :, None], rows[lo:hi], 0))


def test_bench_two_ranks_plumbing(tmp_path, request):
    """bench.py's N > 1 path end to end -- per-rank shards, ONE gather of the result blocks, max-over-ranks
    timing, the JSON line's per-rank fields -- as two ranks sharing this box's GPU, the blocks gathered
    through host memory over gloo (BLURRILY_DIST_BACKEND; RCCL needs a GPU per rank).  A plumbing test,
    not a measurement."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BLURRILY_DIST_BACKEND="gloo")
    started = getattr(request.config, "_two_ranks", None)      # (tests/conftest.py starts it with the session)
    if started is not None:
        out, err = started.communicate(timeout=600)
        request.config._two_ranks = None
        assert started.returncode == 0, err[-2000:]
    else:
        from conftest import two_ranks_command
        res = subprocess.run(two_ranks_command(port), env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        out = res.stdout
    from conftest import TWO_RANKS_DETAIL
    line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
    short = json.loads(line)
    assert len(line) <= 6000                                   # what the driver parses: short, numbers and tokens
    # the line names the collective and the devices it ran on; everything else is in the detail file
    assert short["n_gpus"] == min(2, __import__("torch").cuda.device_count()) and short["replicas"] == 2
    assert short["collective"]["backend"] == "gloo" and short["collective"]["world"] == 2 and "rccl_version" in short["collective"]
    assert len(short["per_rank_kernel_ms"]) == 2 and short["gather_checked"] == 2
    with open(TWO_RANKS_DETAIL) as fh:
        d = json.load(fh)
    assert d["value"] == __import__("pytest").approx(short["value"], rel=1e-3)
    # n_gpus counts PHYSICAL devices (two ranks on this box's one GPU are one GPU); `replicas` the shards
    import torch
    phys = min(2, torch.cuda.device_count())
    assert d["n_gpus"] == d["distinct_devices"] == phys and d["replicas"] == 2 and d["collective"]["distinct_devices"] == phys
    assert d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["needles_per_gpu"] == 50000 and d["config"]["index_replicated"] is True
    assert abs(d["value"] - 2 * 50000 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6      # whole-job needles/s
    assert len(d["per_rank"]["kernel_ms"]) == 2 and len(d["per_rank"]["gather_ms"]) == 2
    assert d["gather_bytes_per_rank"] == 50000 * (10 * 12 + 4) and d["gather_ms"] > 0
    # the gathers overlap the next step's search: both blocks went round and arrived as they were sent
    assert d["gather_overlapped"] is True and d["gather_checked"] == 2
    assert "cpu_baseline" not in d and "extra_configs" not in d
    # which backend and devices the collective ran on (RCCL on the driver's multi-GPU runs; gloo here)
    assert d["collective"]["backend"] == "gloo" and d["collective"]["world"] == 2 and len(d["collective"]["device_ids"]) == 2


def test_bench_two_ranks_strong_scaling_plumbing():
    """`--scaling strong`: configs[3]'s literal batch -- 8 x the workload's needles IN ALL -- cut into contiguous shards
    (shard_bounds), here over two ranks sharing the box's GPU, gloo; the line says strong, the value counts every
    rank's needles once."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BLURRILY_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "1", "--scale", "0.02", "--scaling", "strong", "--verify-shards",
           "--detail", f"/tmp/blurrily_strong_detail_{os.getpid()}.json"]
    res = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    short = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    with open(cmd[-1]) as fh:
        d = json.load(fh)
    os.unlink(cmd[-1])
    total = 8 * 20000
    # one GPU served both ranks: the line says so (n_gpus 1, replicas 2), names the collective, and counts every rank's
    # needles once
    import torch
    phys = min(2, torch.cuda.device_count())
    assert short["n_gpus"] == phys and short["replicas"] == 2 and short["scaling"] == "strong"
    assert short["collective"] == {"backend": "gloo", "world": 2, "rccl_version": None, "distinct_devices": phys}
    assert d["scaling"] == "strong" and d["replicas"] == 2 and d["config"]["needles_per_gpu"] == total // 2
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert abs(short["value"] - d["value"]) / d["value"] < 1e-3
    # what rank 0 gathered for shard 1 equals a direct find of shard 1's needles (--verify-shards)
    assert short["shards_verified"] == {"shard": 1, "needles": total // 2, "equal": True}


def test_bench_n1_under_the_launcher_is_the_plain_n1_line():
    """SCALE's N=1 point is `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the same code
    path as plain `python bench.py` (no process group, no collective), hence the same line -- same config, same
    in-run traffic figures, a value within run-to-run noise."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    detail = f"/tmp/blurrily_n1_detail_{os.getpid()}.json"
    tail = [os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--scale", "0.05",
            "--no-extra", "--no-cpu-baseline", "--latency-probes", "0", "--detail", detail]
    lines = []
    for launcher in ([], ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port)]):
        res = subprocess.run([sys.executable] + launcher + tail, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        short = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
        assert short["n_gpus"] == 1 and "collective" not in short and short["roofline"]["kernel"].startswith("find_")
        with open(detail) as fh:
            lines.append(json.load(fh))
        os.unlink(detail)
    plain, launched = lines
    assert plain["n_gpus"] == launched["n_gpus"] == 1 and "collective" not in launched and "per_rank" not in launched
    assert plain["config"] == launched["config"] and plain["metric"] == launched["metric"]
    # same batch, same kernels: the same bytes up to the few windows swept again after a pool overflow (which
    # candidates a full pool drops depends on the order its atomics land in)
    assert abs(plain["roofline"]["traffic"] - launched["roofline"]["traffic"]) / plain["roofline"]["traffic"] < 1e-3
    assert plain["roofline"]["counters"]["tasks"] == launched["roofline"]["counters"]["tasks"]
    assert abs(plain["value"] - launched["value"]) / plain["value"] < 0.25, (plain["value"], launched["value"])
    assert "unpinned" in plain and plain["roofline"]["hbm_only_frac"] is None
    # the three yardsticks of the one rate: the spec, the guide's copy figure, the measured read ceiling
    r = plain["roofline"]
    assert r["peak"] == 8000.0 and r["achievable_peak"] == 6290.0 and r["read_ceiling"] == 7490.0
    assert abs(r["frac"] * r["peak"] - r["achieved"]) < 1e-6 * r["achieved"]
    assert abs(r["frac_of_read_ceiling"] * r["read_ceiling"] - r["achieved"]) < 1e-6 * r["achieved"]


def _run_bench(args, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    detail = f"/tmp/blurrily_bench_detail_{os.getpid()}.json"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args + ["--detail", detail], cwd=root,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        return res, None
    # the full record (what these tests look into), with the short line the driver parses under "_line"
    assert len(lines[-1]) <= 6000, len(lines[-1])
    with open(detail) as fh:
        d = json.load(fh)
    os.unlink(detail)
    d["_line"] = json.loads(lines[-1])
    return res, d


def test_bench_fails_loudly_when_a_leg_fails():
    """A bench line without its CPU baseline, or with a dead extra config, is not a result: the line is still
    printed, carries the error, and the exit status is 1 (VERDICT r03 "weak" #6)."""
    res, d = _run_bench(["--steps", "2", "--warmup", "1", "--scale", "0.02", "--cpu-budget", "1", "--latency-probes", "0",
                         "--inject-failure", "words"])
    assert res.returncode == 1, (res.returncode, res.stderr[-1500:])
    assert d is not None and "injected failure" in d["extra_configs"]["words"]["cpu_baseline"]["error"]
    assert "injected failure" in d["_line"]["extra_configs"]["words"]["cpu_error"]        # ... and in the short line too
    assert d["_line"]["cpu_baseline"]["parity_mismatches"] == 0 and d["_line"]["cpu_baseline"]["parity_checked"] >= 64
    # the legs that did not fail are whole, parity floor included (32 needles whatever the budget)
    assert d["cpu_baseline"]["parity_mismatches"] == 0 and d["parity_checked"] >= 64
    for name in ("skewed", "geonames_x4", "geonames_miss"):
        x = d["extra_configs"][name]
        assert "error" not in x and x["parity_checked"] >= 64 and x["cpu_baseline"]["parity_mismatches"] == 0, (name, x)
        assert x["roofline"]["counted_launch_rows_equal_timed"] is True and x["roofline"]["sweep"] == x["roofline"]["counted_sweep"]
    res, d = _run_bench(["--steps", "2", "--warmup", "1", "--scale", "0.02", "--cpu-budget", "1", "--latency-probes", "0",
                         "--no-extra", "--inject-failure", "geonames"])
    assert res.returncode == 1 and "injected failure" in d["cpu_baseline"]["error"]
    assert "injected failure" in d["_line"]["cpu_baseline"]["error"]


def test_bench_in_process_over_two_logical_devices():
    """`--gpus 2 --in-process`: one process, option "devices" 2 (with one GPU on the box both replicas live on it), the
    two ranks' needles in one call.  Plumbing, not a measurement: the line's bookkeeping, and rows equal to the
    reference's on the checked prefix."""
    res, d = _run_bench(["--gpus", "2", "--in-process", "--steps", "2", "--warmup", "1", "--scale", "0.05", "--cpu-budget", "1",
                         "--latency-probes", "0"])
    assert res.returncode == 0, res.stderr[-2000:]
    # with one GPU on the box both replicas share it: the line does not call that two GPUs
    import torch
    phys = min(2, torch.cuda.device_count())
    assert d["n_gpus"] == d["distinct_devices"] == d["in_process"]["distinct_devices"] == phys and d["replicas"] == 2
    assert d["in_process"]["n_replicas_made"] == 1 and d["in_process"]["same_device_mask"] == (1 if phys == 1 else 0)
    assert d["config"]["needles_per_gpu"] == 50000 and "in-process" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 50000 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    assert d["cpu_baseline"]["parity_mismatches"] == 0 and d["parity_checked"] >= 64 and "extra_configs" not in d
