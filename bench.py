#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X:  Goldilocks NTT field-elements/s at 2^24.

One step = one pass of the hot path over one batch: a natural-order forward NTT (zp_ntt through the
C-ABI) of the rank's column shard u64[cols][2^logn], resident in HBM before the timed region.
Columns are independent, so ranks shard columns with no data-path collective ("scaling": "weak":
every GPU always owns `--cols` columns).  Rank 0 prints ONE JSON line.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--logn 24] [--cols 64]
  N > 1 is launched by torch.distributed.run (one process per GPU, RCCL for the barrier/MAX only).

Extra objects on the line: "roofline" (dominant kernel, HIP-event timing of every pass launch taken
live inside this run), "cpu_baseline" (oracle/ CPU restatement timed on this host, rank 0, N=1) and
"pipeline" (LDE + Poseidon Merkle commit of the same shard, timed outside the K steps).
The oracle is only used as the CPU baseline here, never in the measured path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def random_field_tensor(torch, shape, device, seed):
    """uniform 64-bit patterns reduced into [0,p): int64 storage of canonical u64 values"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lo = torch.randint(0, 1 << 32, shape, dtype=torch.int64, device=device, generator=g)
    hi = torch.randint(0, 1 << 32, shape, dtype=torch.int64, device=device, generator=g)
    x = (hi << 32) | lo
    del lo, hi
    # unsigned x >= p  <=>  signed x in [-(2^32-1), -1]  -> subtract p == add 2^32-1 (mod 2^64)
    bad = (x < 0) & (x >= -((1 << 32) - 1))
    x = torch.where(bad, x + ((1 << 32) - 1), x)
    return x


def cpu_baseline(logn, budget_cols):
    """oracle NTT (OpenMP over columns) on a bounded sample of the same workload"""
    import numpy as np
    from oracle import oracle as O
    cores = O.num_threads()
    cols = max(cores, min(budget_cols, 2 * cores))
    x = O.random_field((cols, 1 << logn), 0xE16E2E70 + 2)
    t0 = time.perf_counter()
    y = O.ntt(x)
    dt = time.perf_counter() - t0
    del y
    # subtract nothing: copy + transform is what the CPU path does per call
    return {"value": cols * (1 << logn) / dt, "unit": "field-elems/s", "cores": cores, "kind": "port",
            "sample": "%d columns x 2^%d rows, oracle/gl_oracle.c radix-2 NTT, OpenMP over columns, %.2f s"
                      % (cols, logn, dt),
            "note": "CPU restatement, not the eigen-zkvm prover (parity unpinned, SURVEY.md 8c)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--logn", type=int, default=24)
    ap.add_argument("--cols", type=int, default=64, help="columns per GPU")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from eigen_zeth_amd.native import Prover

    logn, cols = args.logn, args.cols
    N = 1 << logn
    prover = Prover(local, stream=torch.cuda.current_stream().cuda_stream)
    x = random_field_tensor(torch, (cols, N), dev, 0xE16E2E70 + 4 + rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        prover.ntt(x, x, logn, cols)
    barrier()
    prover.set_profiling(True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        prover.ntt(x, x, logn, cols)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    passes = prover.pass_timings()
    prover.set_profiling(False)

    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())

    if rank == 0:
        plan = prover.ntt_plan(logn)
        npass = max(1, len(plan["passes"]))
        elems_total = world * cols * N * args.steps
        # dominant kernel = the non-transposing radix pass (2 of the 3 launches per chunk at 2^24)
        by_kind = {}
        for rl, ms in passes:
            by_kind.setdefault(rl, []).append(ms)
        dom = max(by_kind, key=lambda k: sum(by_kind[k])) if by_kind else 0
        launches = len(by_kind.get(dom, []))
        avg_ms = sum(by_kind[dom]) / launches if launches else float("nan")
        chunk_cols = min(cols, max(1, (1 << 28) >> logn))
        # algorithmic bytes of one launch: SURVEY 8d gives 16*N bytes per column transform (ideal single
        # pass); a launch does one of the `npass` passes over chunk_cols columns -> 16*N*chunk/npass
        alg_bytes = 16.0 * N * chunk_cols / npass
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if launches else float("nan")
        traffic = None
        tp = os.path.join(ROOT, "profiles", "ntt_traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch_dominant")
            except Exception:
                traffic = None
        out = {
            "metric": "goldilocks_ntt_field_elems_per_s",
            "value": elems_total / wall_max,
            "unit": "field-elems/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "forward NTT, natural order in/out, 2^%d rows x %d columns per GPU (column-major u64), "
                                   "BASELINE configs[3] shape on one GPU" % (logn, cols),
                       "logn": logn, "cols_per_gpu": cols, "sharding": "columns, no data-path collective",
                       "plan": plan},
            "device_ms_per_step": dev_ms / args.steps,
            "roofline": {
                "bound": "hbm",
                "kernel": "ntt_pass_kernel radix 2^%d (%s)" % (abs(dom), "transposing first pass" if dom < 0 else "strided pass"),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if launches else None,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": avg_ms, "launches_timed": launches,
                "whole_transform_GBs": 16.0 * N * cols * args.steps / (dev_ms * 1e-3) / 1e9,
                "whole_transform_frac": 16.0 * N * cols * args.steps / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "per_pass_avg_ms": {str(k): sum(v) / len(v) for k, v in sorted(by_kind.items())},
            },
        }
        if not args.no_pipeline:
            try:
                out["pipeline"] = pipeline_probe(torch, prover, dev, logn, min(cols, 32))
            except Exception as e:  # reported, not fatal for the headline
                out["pipeline"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(logn, 32)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pipeline_probe(torch, prover, dev, logn, cols):
    """LDE (blow-up 2) + Poseidon Merkle commit of a column shard, timed once after one warm-up"""
    N = 1 << logn
    x = random_field_tensor(torch, (cols, N), dev, 99)
    y = torch.empty((cols, 2 * N), dtype=torch.int64, device=dev)
    tree = torch.empty(((4 * N - 1) * 4,), dtype=torch.int64, device=dev)
    res = {}
    for it in range(2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        prover.lde(x, y, logn, 1, cols)
        e[1].record()
        prover.merkle_commit(y, 2 * N, cols, tree)
        e[2].record()
        torch.cuda.synchronize()
        res = {"cols": cols, "blowup": 2, "lde_ms": e[0].elapsed_time(e[1]), "merkle_ms": e[1].elapsed_time(e[2])}
    perms = ((cols + 7) // 8) * 2 * N + (2 * N - 1)
    res["lde_GBs_algorithmic"] = 8.0 * N * 3 * cols / (res["lde_ms"] * 1e-3) / 1e9
    res["poseidon_perms_per_s"] = perms / (res["merkle_ms"] * 1e-3)
    return res


if __name__ == "__main__":
    main()
