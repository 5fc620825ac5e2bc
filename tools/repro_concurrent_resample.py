#!/usr/bin/env python3
"""Reproduction harness for the red GPU run of round 4.

N one-path `baknots` processes (BA::interpInputData -> batotp_hip_resample with B = 1, one HIP context each) run
concurrently on one GPU, optionally beside a resident batch of the same workload that is sweeping; every knots.bin is
compared bit for bit with the knots of the ORACLE's resampler (oracle/_build/dump_knots: the same host shell linked
against the CPU checker).  A mismatch is characterised (sizes, first differing index per channel, whether the knots
are those of another seed) and both files are kept under gpurun_out/repro/.

usage: repro_concurrent_resample.py [--workload gen7] [--knots 50000] [--seeds 1024] [--jobs 64] [--rounds 3]
                                    [--resident 1024] [--out gpurun_out/repro]
"""
import argparse
import concurrent.futures as cf
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from batotp_amd import capi, pathgen  # noqa: E402

BAKNOTS = os.path.join(ROOT, "batotp_amd", "host", "_build", "baknots")
ORACLE_KNOTS = os.path.join(ROOT, "oracle", "_build", "dump_knots")


def prepare(workload, seed, n_target, work):
    """writes path.dat / config.dat; returns the sha256 of path.dat (the generator runs in the calling thread)"""
    w = bench.WORKLOADS[workload]
    n_coarse = max(8, int(round(n_target / w["knots_per_coarse"])))
    theta, cart, tres = w["gen"](seed, n_coarse)
    pathgen.write_traj_bin(os.path.join(work, "path.dat"), tres, theta, cart)
    pathgen.write_config(os.path.join(work, "config.dat"), **w["cfg"])
    return hashlib.sha256(open(os.path.join(work, "path.dat"), "rb").read()).hexdigest()


def run_tool(tool, work, env=None):
    r = subprocess.run([tool, "config.dat"], cwd=work, capture_output=True, text=True, env=env)
    if r.returncode != 0:
        return None, r.stdout[-500:] + r.stderr[-500:]
    # (round 6: BA::interpInputData evaluates the resampler until two consecutive evaluations agree and reports a disagreement)
    report = [l for l in r.stdout.splitlines() if "disagree" in l or "stage checksums" in l]
    return open(os.path.join(work, "knots.bin"), "rb").read(), ("[DISAGREE] " + " || ".join(report) + " || " if report else "") + r.stdout[-300:]


def parse(kb):
    N, nJ, nC = (int(v) for v in np.frombuffer(kb, "<i8", 3, 0))
    sres = float(np.frombuffer(kb, "<f8", 1, 24)[0])
    y = np.frombuffer(kb, "<f8", (nJ + nC) * N, 32).reshape(nJ + nC, N)
    return N, sres, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gen7")
    ap.add_argument("--knots", type=int, default=50000)
    ap.add_argument("--seeds", type=int, default=1024)
    ap.add_argument("--seed0", type=int, default=7000)
    ap.add_argument("--jobs", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--resident", type=int, default=1024, help="paths of a resident batch sweeping meanwhile (0: none)")
    ap.add_argument("--idle", action="store_true", help="the resident batch sweeps once and then sits idle (what the red test had)")
    ap.add_argument("--no-oracle", action="store_true", help="skip the oracle's knots (5 min of host time for 1024 seeds): detection rests on the "
                    "report of the two-evaluations guard alone")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "repro"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    seeds = [a.seed0 + k for k in range(a.seeds)]
    base = tempfile.mkdtemp(prefix="repro_")
    t0 = time.time()

    # the oracle's knots of every seed (CPU), digests only
    def oracle_one(seed):
        work = os.path.join(base, f"o{seed}")
        os.makedirs(work)
        ph = prepare(a.workload, seed, a.knots, work)
        kb, msg = run_tool(ORACLE_KNOTS, work)
        if kb is None:
            raise RuntimeError(f"oracle resampler failed on seed {seed}: {msg}")
        shutil.rmtree(work)
        return hashlib.sha256(kb).hexdigest(), len(kb), ph

    # the synthetic-path generator (numpy / scipy in many threads): serial digests first, so that a wrong INPUT is told
    # apart from a wrong resampling result
    serial = {}
    for seed in seeds:
        work = os.path.join(base, f"g{seed}")
        os.makedirs(work)
        serial[seed] = prepare(a.workload, seed, a.knots, work)
        shutil.rmtree(work)
    print(f"serial generation of {len(seeds)} inputs in {time.time() - t0:.1f} s", flush=True)

    if a.no_oracle:
        want = {s_: (None, 0, serial[s_]) for s_ in seeds}
    else:
        with cf.ThreadPoolExecutor(max_workers=min(a.jobs, os.cpu_count() or 1)) as ex:
            want = dict(zip(seeds, ex.map(oracle_one, seeds)))
    by_digest = {v[0]: s for s, v in want.items()}
    gen_bad = [s for s in seeds if want[s][2] != serial[s]]
    print(f"generator in {a.jobs} threads against serial generation: {len(gen_bad)} different inputs {gen_bad[:8]}", flush=True)
    print(f"oracle knots of {len(seeds)} seeds in {time.time() - t0:.1f} s", flush=True)

    # a resident batch that keeps the GPU busy the way the test had it (its own context, sweeping in a loop)
    stop = threading.Event()
    busy = {"steps": 0}
    thr = None
    if a.resident:
        hip = capi.Context(capi.load_hip(), 0)
        c = dict(workload=a.workload, knots=a.knots)
        inp = bench.Inputs(hip, c["workload"], c["knots"], seeds[: min(len(seeds), a.resident)])
        prob = capi.Problem.from_buffer_copy(bytes(inp.prob))
        if prob.flags & capi.F_NO_SAMPLES:
            prob.flags |= capi.F_COMPACT_SPLINES
        cap = int(int(inp.n_knots.max()) * bench.WORKLOADS[a.workload]["cap"] * 2) + 1024
        b = capi.Batch(hip, prob, [int(inp.n_knots[p % inp.K]) for p in range(a.resident)], cap)
        inp.fill(b, a.resident)

        def spin():
            while not stop.is_set():
                b.precompute(0); b.sweep(-1); b.sweep(+1)
                busy["steps"] += 1
                if a.idle:
                    hip.synchronize()
                    break
        thr = threading.Thread(target=spin)
        thr.start()
        print(f"resident batch of {a.resident} paths sweeping ({b.nbytes() / 2**30:.1f} GiB)", flush=True)

    bad = []

    def device_one(args):
        rnd, seed = args
        work = os.path.join(base, f"d{rnd}_{seed}")
        os.makedirs(work)
        ph = prepare(a.workload, seed, a.knots, work)
        kb, msg = run_tool(BAKNOTS, work)
        rec = None
        if ph != serial[seed]:
            rec = dict(round=rnd, seed=seed, kind="generator produced a different input in this thread")
        elif kb is None:
            rec = dict(round=rnd, seed=seed, kind="failed", msg=msg)
        elif want[seed][0] is not None and hashlib.sha256(kb).hexdigest() != want[seed][0]:
            keep = os.path.join(a.out, f"r{rnd}_s{seed}")
            os.makedirs(keep, exist_ok=True)
            open(os.path.join(keep, "device_knots.bin"), "wb").write(kb)
            okb, _ = run_tool(ORACLE_KNOTS, work)
            open(os.path.join(keep, "oracle_knots.bin"), "wb").write(okb)
            shutil.copy(os.path.join(work, "path.dat"), keep)
            shutil.copy(os.path.join(work, "config.dat"), keep)
            Nd, sd, yd = parse(kb)
            No, so, yo = parse(okb)
            rec = dict(round=rnd, seed=seed, kind="mismatch", N_dev=Nd, N_oracle=No, sres_dev=sd, sres_oracle=so,
                       is_other_seed=by_digest.get(hashlib.sha256(kb).hexdigest()), msg=msg)
            if Nd == No:
                diff = yd != yo
                rec["n_diff"] = int(diff.sum())
                rec["first_diff_per_channel"] = [int(np.argmax(d)) if d.any() else -1 for d in diff]
                rec["last_diff_per_channel"] = [int(len(d) - 1 - np.argmax(d[::-1])) if d.any() else -1 for d in diff]
                with np.errstate(all="ignore"):
                    rec["max_abs_diff"] = float(np.nanmax(np.abs(yd - yo)))
                rec["nan_dev"] = int(np.isnan(yd).sum())
            # a second attempt right away, alone: does the same input come out right?
            kb2, _ = run_tool(BAKNOTS, work)
            rec["retry_equal_oracle"] = bool(kb2 is not None and hashlib.sha256(kb2).hexdigest() == want[seed][0])
        elif msg.startswith("[DISAGREE]"):
            rec = dict(round=rnd, seed=seed, kind="right knots after the guard reported two evaluations that disagree", msg=msg)
            keep = os.path.join(a.out, f"r{rnd}_s{seed}")
            os.makedirs(keep, exist_ok=True)
            for f in os.listdir(work):
                if f.startswith("resampler_disagreement_") or f in ("path.dat", "config.dat"):
                    shutil.copy(os.path.join(work, f), keep)
        shutil.rmtree(work, ignore_errors=True)
        return rec

    for rnd in range(a.rounds):
        t1 = time.time()
        with cf.ThreadPoolExecutor(max_workers=a.jobs) as ex:
            recs = [r for r in ex.map(device_one, [(rnd, s) for s in seeds]) if r]
        bad += recs
        print(f"round {rnd}: {len(seeds)} one-path resamples, {a.jobs} at a time, {time.time() - t1:.1f} s, "
              f"{len(recs)} bad, resident steps so far {busy['steps']}", flush=True)
        for r in recs:
            print(json.dumps(r), flush=True)
    stop.set()
    if thr:
        thr.join()
    json.dump(bad, open(os.path.join(a.out, "summary.json"), "w"), indent=1)
    shutil.rmtree(base, ignore_errors=True)
    print(f"total bad: {len(bad)} of {a.rounds * len(seeds)}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
