"""CPU: the multi-rank path (world_size 2, gloo).  Paths are sharded over ranks with no data-path
collective; the only communication is the all_gather of the per-path result rows."""
import os
import socket
import sys

import numpy as np
import pytest

import helpers


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, names, q):
    sys.path.insert(0, helpers.ROOT)
    sys.path.insert(0, os.path.join(helpers.ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from batotp_amd import capi
    from batotp_amd import dist as bdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = bdist.shard_range(len(names), rank, world)
        cases = [helpers.Case(n) for n in names[lo:hi]]
        for c in cases:
            c.problem = helpers.Case(names[0]).problem
        ctx = capi.Context(helpers.load_oracle(), 0)
        outs = helpers.run_pipeline(ctx, cases, mvc=False, details=False) if cases else []
        local = np.array([o["result"] for o in outs], dtype=capi.RESULT_DTYPE) if outs else np.zeros(0, dtype=capi.RESULT_DTYPE)
        allres = bdist.gather_results(local)
        # the variable-length part: both curves of every path, to rank 0
        b = None                 # (a rank without a share has no batch: it takes part in the collectives and sends nothing)
        if cases:
            b = capi.Batch(ctx, cases[0].problem, [c.n for c in cases], max(c.max_steps() for c in cases))
            for k, c in enumerate(cases):
                b.upload_knots(k, [c.y], [c.sres])
            b.optimize()
        curves = {w: bdist.gather_curves(b, w) for w in (-1, 1)}
        if rank == 0:
            q.put((allres.tobytes(), {w: [(s.tobytes(), sd.tobytes()) for s, sd in curves[w]] for w in curves}))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions():
    from batotp_amd import dist as bdist
    for n in (0, 1, 5, 8, 1024, 4097):
        for w in (1, 2, 3, 8):
            spans = [bdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_gather_matches_single_rank(oracle_ctx):
    import torch.multiprocessing as mp
    from batotp_amd import capi
    names = ["GEN7DOF", "synth_gen7dof_s0", "GEN7DOF", "synth_gen7dof_s0", "GEN7DOF"]  # 5 paths -> 3 + 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, names, q)) for r in range(2)]
    for p in procs:
        p.start()
    raw, curves = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gathered = np.frombuffer(raw, dtype=capi.RESULT_DTYPE)
    cases = [helpers.Case(n) for n in names]
    for c in cases:
        c.problem = cases[0].problem
    single = helpers.run_pipeline(oracle_ctx, cases, mvc=False, details=False)
    assert gathered.shape[0] == len(names)
    for k, o in enumerate(single):
        for f in capi.RESULT_DTYPE.names:
            assert gathered[k][f] == o["result"][f], (k, f)
    # curves gathered across the two ranks == the single-rank curves, bit for bit, in path order
    for w, key in ((-1, "rev"), (1, "fwd")):
        assert len(curves[w]) == len(names)
        for k, o in enumerate(single):
            assert curves[w][k][0] == o[key][0].tobytes() and curves[w][k][1] == o[key][1].tobytes(), (w, k)


def test_eight_ranks_gather_rows_and_curves_with_empty_shards(oracle_ctx):
    """the world size of BASELINE configs 4 / 5 (8 GPUs), on CPU: 11 paths over 8 ranks (blocks of 2, 2, 2, 1, 1, 1, 1, 1) and 5 paths
    over 8 ranks (three ranks own nothing): result rows by all_gather in rank order, BOTH curves of every path to rank 0 by the size
    exchange + grouped send/recv of batotp_amd.dist.gather_curves -- everything equal to the single-rank run, bit for bit"""
    import torch.multiprocessing as mp
    from batotp_amd import capi
    pool = ["GEN7DOF", "synth_gen7dof_s0", "GEN7DOF"]
    ref = {}
    for names in ([pool[k % 3] for k in range(11)], [pool[k % 2] for k in range(5)]):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 8, port, names, q)) for r in range(8)]
        for p in procs:
            p.start()
        raw, curves = q.get(timeout=600)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        gathered = np.frombuffer(raw, dtype=capi.RESULT_DTYPE)
        assert gathered.shape[0] == len(names)
        for n in set(names):
            if n not in ref:
                c = helpers.Case(n)
                c.problem = helpers.Case(names[0]).problem
                ref[n] = helpers.run_pipeline(oracle_ctx, [c], mvc=False, details=False)[0]
        for k, n in enumerate(names):
            o = ref[n]
            for f in capi.RESULT_DTYPE.names:
                assert gathered[k][f] == o["result"][f], (len(names), k, f)
            for w, key in ((-1, "rev"), (1, "fwd")):
                assert len(curves[w]) == len(names)
                assert curves[w][k][0] == o[key][0].tobytes() and curves[w][k][1] == o[key][1].tobytes(), (len(names), w, k)


def _bench_line(args, env_extra=None):
    import json, subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_launches_its_own_ranks_and_shards_as_worded():
    """`python bench.py --gpus 2` starts two ranks itself (torch.distributed.run as a child process) -- here on CPU with
    --launch-check (gloo, no GPU work): strong sharding of cfg 4's 1024 paths, gather in rank order, max over ranks"""
    r, line = _bench_line(["--gpus", "2", "--launch-check", "--config", "cfg4"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["paths_total"] == 1024
    assert line["gathered_rows"] == 1024 and line["rows_in_rank_order"] and line["max_over_ranks"] == 2.0
    # weak configurations keep their per-GPU batch
    r, line = _bench_line(["--gpus", "2", "--launch-check", "--config", "cfg2"])
    assert r.returncode == 0 and line["n_gpus"] == 2 and line["paths_total"] == 2 and line["gathered_rows"] == 2


def test_bench_refuses_a_rank_count_that_is_not_the_one_asked_for():
    """--gpus N with a different WORLD_SIZE (the silent single-rank run of round 1) is an error"""
    r, line = _bench_line(["--gpus", "4", "--launch-check"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and line is None and "WORLD_SIZE is 1" in (r.stderr + r.stdout)


def test_bench_rehearses_config_5_on_four_ranks():
    """BASELINE config 5 as worded on 4 ranks without GPUs (--launch-check, gloo): 4096 cable-robot paths in contiguous blocks of
    1024, every block in ONE chunk of its 288 GB GPU (every channel as pairs: ~68 MB per path of 2e5 knots, the layout measure()
    picks for the cable robot in serial form), result rows gathered in rank order, and the size exchange of the curve gather: rank 0
    would receive ~3/4 of all forward curves"""
    r, line = _bench_line(["--gpus", "4", "--launch-check", "--config", "cfg5"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 4 and line["scaling"] == "strong" and line["paths_total"] == 4096 and line["gathered_rows"] == 4096
    assert line["rows_in_rank_order"] and line["max_over_ranks"] == 4.0
    assert line["paths_per_rank"] == 1024 and line["chunks_per_rank"] == [1024] and 2048 <= line["paths_that_fit_one_gpu"] < 4096
    g = line["curve_gather"]
    assert len(g["points_per_rank"]) == 4 and min(g["points_per_rank"]) > 1024 * 0.4 * 200000 * 0.99
    assert 3.5 < g["GB_to_rank0"] < 6.5
    # one GPU holds the whole batch in two chunks of whole multiples of the distinct paths
    r, line = _bench_line(["--gpus", "1", "--launch-check", "--config", "cfg5"])
    assert r.returncode == 0 and line["paths_per_rank"] == 4096 and line["chunks_per_rank"] == [2048] * 2


@pytest.mark.parametrize("config, per_rank", [("cfg4", 128), ("cfg5", 512), ("cfg5_distinct2048", 512)])
def test_bench_rehearses_the_sharded_configs_on_eight_ranks(config, per_rank):
    """BASELINE configs 4 and 5 as worded on their 8 ranks without GPUs (--launch-check, gloo): contiguous blocks of 128 / 512 paths,
    one chunk per GPU, rows gathered in rank order, the size exchange of the curve gather (rank 0 receives 7/8 of the curves)"""
    r, line = _bench_line(["--gpus", "8", "--launch-check", "--config", config])
    assert r.returncode == 0, r.stderr[-2000:]
    total = 8 * per_rank
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["paths_total"] == total and line["gathered_rows"] == total
    assert line["rows_in_rank_order"] and line["max_over_ranks"] == 8.0
    assert line["paths_per_rank"] == per_rank and line["chunks_per_rank"] == [per_rank]
    g = line["curve_gather"]
    assert len(g["points_per_rank"]) == 8 and min(g["points_per_rank"]) > 0
    assert abs(g["GB_to_rank0"] / (16e-9 * sum(g["points_per_rank"])) - 7.0 / 8.0) < 0.02
