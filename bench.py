#!/usr/bin/env python3
"""bench.py -- ALS ratings/sec per iteration (U + I solve) on synthetic ratings of the
shapes BASELINE.json names, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload mal|c3|c5|c5shard|ml1m|ml100k] [--double]

A "step" is one full ALS iteration = EmfLord.alsTrainIter(): the byUser half-step, the
exchange of the solved user shard, the byItem half-step and its exchange.  Inputs (CSR by
user and by item, both factor matrices) are resident in HBM before the timed region.  With
N > 1 the total problem is fixed and rows are sharded over the ranks ("strong").

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      -- the dominant kernel (per half-step) against the roof that binds it, from HIP-event
                   durations measured inside libycnr_als.so on the launch stream
  cpu_baseline  -- the CPU oracle (a port of the reference algorithm) timed on a bounded row
                   sample of the same workload on this box's host cores (N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "you-can-not-recommend_amd", "python"))
sys.path.insert(0, ROOT)


def self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N per-GPU processes ourselves, the way
    the reference's master forks its workers (EmfMaster.createWorkers, lib/emf/EmfMaster.js:44-98; EmfLord.gatherClusterNodes,
    lib/emf/EmfLord.js:752-828).  The parent never touches the GPU (nothing of torch or libycnr_als is imported before this
    runs) and never replaces itself: the ranks are a CHILD process -- `python -m torch.distributed.run` with the same
    arguments --, its stdout (rank 0's JSON line) is this process's stdout, and its exit code is ours."""
    if os.environ.get("WORLD_SIZE") or os.environ.get("RANK"):
        return  # already one rank of a launched job
    gpus = 1
    argv = sys.argv[1:]
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            gpus = argv[i + 1]
        elif a.startswith("--gpus="):
            gpus = a.split("=", 1)[1]
    try:
        gpus = int(gpus)
    except ValueError:
        return  # argparse reports it
    if gpus <= 1 or "--emulate-world" in argv or any(a.startswith("--emulate-world=") for a in argv) or "-h" in argv or "--help" in argv:
        return
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL and the ipc transport need it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    print("bench.py: launching %d ranks: %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


if __name__ == "__main__":
    self_launch()

# (read by the HIP runtime when it starts, i.e. at torch's first device call: see python/ycnr_als/_lib.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: users, items, nnz, k, max_rating, zipf_a, degree_sigma, description
    "mal": (1_750_000, 12_700, 121_000_000, 100, 10, 0.6, 1.2, "MAL-scale synthetic 1.75Mx12.7K, 121M nnz, k=100"),
    "mal96": (1_750_000, 12_700, 121_000_000, 96, 10, 0.6, 1.2, "MAL-scale synthetic 1.75Mx12.7K, 121M nnz, k=96 (kernel experiments)"),
    "c3": (200_000, 20_000, 20_000_000, 64, 10, 0.8, 1.0, "synthetic 200Kx20K, 20M nnz, k=64"),
    # the any-k path (als_gen_kernels.hip.h): float32 beyond 256 factors; with --double, float64 beyond 128
    "c3k512": (200_000, 20_000, 20_000_000, 512, 10, 0.8, 1.0, "synthetic 200Kx20K, 20M nnz, k=512 (any-k path)"),
    "c3k256": (200_000, 20_000, 20_000_000, 256, 10, 0.8, 1.0, "synthetic 200Kx20K, 20M nnz, k=256 (with --double: any-k path)"),
    "c5": (10_000_000, 100_000, 1_000_000_000, 256, 10, 0.7, 1.0, "synthetic 10Mx100K, 1B nnz, k=256"),
    "c5shard": (1_250_000, 100_000, 125_000_000, 256, 10, 0.7, 1.0, "one GPU's eighth of the 10Mx100K, 1B nnz, k=256 config (1.25Mx100K, 125M nnz)"),
    "ml1m": (6040, 3883, 1_000_209, 100, 5, 0.9, 0.9, "MovieLens-1M-shaped synthetic 6040x3883, 1M nnz, k=100"),
    "ml100k": (943, 1682, 100_000, 20, 5, 0.8, 0.9, "MovieLens-100k-shaped synthetic 943x1682, 100K nnz, k=20"),
}

PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA; a float32 product costs 6 bf16 products (exact 3-way split)
PEAK_FP64_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0     # spec; ~6300 achievable


def algorithmic_flops(nnz, rows_solved, k):
    """Minimum-flop model of one half-step: symmetric Gramian + rhs per rating,
    Cholesky + two triangular solves per row."""
    return nnz * (k * (k + 1) + 2 * k) + rows_solved * (k ** 3 / 3.0 + 2 * k * k)


def algorithmic_bytes(nnz, rows_solved, k, s):
    """Gather-model bytes of one half-step: index + value + gathered factor row
    per rating, row write + row pointer per solved row."""
    return nnz * (4 + s + k * s) + rows_solved * (k * s + 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mal", choices=sorted(WORKLOADS))
    ap.add_argument("--double", action="store_true", help="useDoublePrecision")
    ap.add_argument("--chunk", type=int, default=0, help="ratings per split work unit (0 = library default)")
    ap.add_argument("--factors", type=int, default=0, help="experiments: factorsCount instead of the workload's (the line is then NOT the workload's metric; config.factors says so)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for functional tests)")
    ap.add_argument("--same-device", action="store_true",
                    help="functional test only: every rank uses cuda:0 (several ranks on one GPU, gloo)")
    ap.add_argument("--transport", default="", choices=["", "rccl", "ipc", "shm"],
                    help="exchange transport inside libycnr_als.so for N > 1 (default: rccl; shm with --same-device / gloo; "
                         "ipc = mapped peer replicas + copy engines, also with several ranks on one GPU)")
    ap.add_argument("--transport-ab", default="", choices=["", "rccl", "ipc", "shm"],
                    help="N > 1: after the timed run, the same iterations once more over THIS transport; reported beside it as exchange.ab")
    ap.add_argument("--strict-transport", action="store_true",
                    help="N > 1: exit non-zero when the requested transport cannot be set up on every rank (default: try the other "
                         "device-to-device transport, then torch.distributed's all-gather; exchange.path names the path that RAN, "
                         "exchange.requested the one asked for, exchange.fallback why)")
    ap.add_argument("--allow-fallback", action="store_true", help="(the default since round 5; kept for old command lines)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="one GPU solves the shard of every rank of a world of this size in turn (exchange left out) and "
                         "reports the compute time per rank, before and after the feedback re-cut of the shards")
    ap.add_argument("--rebalance-after", type=int, default=2, help="N > 1: re-cut the shards from measured times after each of the first this-many iterations (0 = never)")
    ap.add_argument("--exchange-chunks", type=int, default=4, help="pieces a large side's shard is solved in (exchange overlaps solve)")
    ap.add_argument("--item-sharding", default="rows", choices=["rows", "bands"],
                    help="N > 1: how the item half-step is sharded -- 'rows': items over the ranks, the user matrix all-gathered after every "
                         "user half-step; 'bands': the users in 8 fixed bands, Gramians of all items per band, reduce-scatter of the band sums "
                         "(EmfLord option itemStepSharding; also at N = 1, where it only fixes the order of the sums)")
    ap.add_argument("--dump-factors", default="", help="write the final factor matrices of rank 0 to this .npz")
    ap.add_argument("--dump-step-info", action="store_true", help="add the library's step info of the last half-step of each side to the line (step_info)")
    ap.add_argument("--debug-mod-idx", type=int, default=0,
                    help="timing experiment only: fold all column ids into [0, N) so every gather hits L1/L2")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the baseline sample")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # (run by hand `python bench.py --gpus N` starts its own ranks: self_launch above; here a launcher's world wins)
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (no CPU fallback)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from ycnr_als.data import synth_ratings
    from ycnr_als.emf import Dataset, EmfLord
    if args.emulate_world > 1:
        return emulate_world(args, local_rank)
    users, items, nnz_target, k, max_rating, zipf_a, sigma, desc = WORKLOADS[args.workload]
    if args.factors:
        k, desc = args.factors, desc + f" [factorsCount overridden: {args.factors}]"
    dev = torch.device("cuda", local_rank)
    t0 = time.time()
    tdt = torch.float64 if args.double else torch.float32
    by_user, by_item = synth_ratings(users, items, nnz_target, max_rating=max_rating, seed=20260004, device=dev,
                                     dtype=tdt, degree_sigma=sigma, zipf_a=zipf_a)
    nnz = by_user.nnz
    if args.debug_mod_idx > 0:
        by_user.indx %= args.debug_mod_idx
        by_item.indx %= args.debug_mod_idx
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    # in-sample RMSE set: every 10th rating of each... keep it simple: a 10 % Bernoulli sample
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    from ycnr_als.data import select_csr
    val = select_csr(by_user, torch.rand(nnz, generator=g, device=dev) < 0.10)
    ds = Dataset(by_user, by_item, validate=val, test=None, total_ratings_avg=float(by_user.vals.double().mean()))
    requested_transport = args.transport or ("shm" if args.same_device or args.backend == "gloo" else "rccl")
    lord = EmfLord(options={"factorsCount": k, "trainIters": args.steps, "useDoublePrecision": args.double,
                            "dbType": "mal" if max_rating == 10 else "ml", "chunkRatings": args.chunk,
                            "dataSetDistr": [90, 10, 0], "exchangeChunks": args.exchange_chunks, "rebalanceAfterIters": args.rebalance_after,
                            "commTransport": requested_transport, "strictTransport": args.strict_transport, "itemStepSharding": args.item_sharding},
                    dist=dist)
    lord.prepareToTrain(ds, seed=20260004, device=local_rank)
    if world > 1 and args.strict_transport and lord.exchangePath != "libycnr_als:" + requested_transport:
        sys.exit(f"bench.py: the exchange runs over '{lord.exchangePath}', not over the requested 'libycnr_als:{requested_transport}'")
    t_prep = time.time() - t0 - t_gen

    def barrier():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        lord.alsTrainIter()
    lord.stepTimes.clear()
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        lord.alsTrainIter()
    barrier()
    elapsed = time.perf_counter() - t1
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = nnz * args.steps / elapsed

    # per-kernel accounting over the timed region, from the HIP-event durations libycnr_als.so
    # records on its launch stream (ycnr_als_last_step_info)
    s = 8 if args.double else 4
    peak = PEAK_FP64_TFLOPS if args.double else PEAK_FP32_TFLOPS
    gram_rating = k * (k + 1) + 2 * k           # symmetric Gramian + rhs, per rating
    solve_row = k ** 3 / 3.0 + 2 * k * k        # Cholesky + two triangular solves, per row
    bytes_rating = 4 + s + k * s                # index + value + gathered factor row
    # Every entry carries the flops its kernels EXECUTE, split by the pipe they run on: Gramian
    # products of float32 runs go to the bf16 matrix pipe as six bf16 products each (exact 3-way
    # split; peak 2500 / 6 TFLOP/s float32-equivalent) when the library's conditions hold, everything
    # else (solves, float64, k > 112 Gramians) to the fp32 / fp64 MFMA peak.  Rows in dual form execute
    # n x n work (info.dualFlops: n(n+1)k + n^3/3 + 2n^2 + 2nk per row), not the k x k model.
    # kernels are accounted per half-step: the same kernel is a different workload on the two sides
    # (user side: the 5 MB item matrix is cache-resident; item side: 700 MB of user factors are not)
    kern = {}
    step_ms = {"byUser": 0.0, "byItem": 0.0}
    comm = {sd: {"compute_ms": 0.0, "exchange_ms": 0.0, "exposed_exchange_ms": 0.0, "wall_ms": 0.0, "bytes": 0, "pieces": 0, "n": 0}
            for sd in ("byUser", "byItem")}

    def x6_of(side):
        # bf16x6 Gramian: the LDS-DMA kernels (k % 4 == 0, k <= 128, fixed matrix < 2 GB) and the
        # workgroup-per-row kernels of k > 128 (als_wg_*, any fixed matrix size)
        fixed_rows = items if side == "byUser" else users
        # (float32 of other sizes runs padded to a multiple of 4 columns), and the any-k path's Gramian where a panel of 32 ratings
        # fits a workgroup's LDS and loader slots (k % 4 == 0, k <= 512: als_gen_gram_kernel<float, 4, true>)
        kp = (k + 3) // 4 * 4
        # (the any-k Gramian: launch_step_gen enables X6 when a 32-rating panel fits the loader's slots, 32 (k / 4) <= 8 * 512 threads: k <= 512)
        return (not args.double) and (128 < k <= 256 or (kp <= 128 and fixed_rows * kp * 4 < 2 ** 31) or (k > 256 and k % 4 == 0 and k <= 512))

    gen_path = k > (128 if args.double else 256)  # als_gen_kernels.hip.h: float32 / float64 MFMA, matrix in global memory
    for st in lord.stepTimes:
        i = st["info"]
        side = st["stepType"]
        step_ms[side] += i.totalMs
        c = comm[side]
        c["compute_ms"] += i.totalMs
        c["exchange_ms"] += i.exchangeMs
        c["exposed_exchange_ms"] += i.exposedExchangeMs
        c["wall_ms"] += st["wall"] * 1e3
        c["bytes"] += i.exchangeBytes
        c["pieces"] = i.parts
        c["n"] += 1
        x6 = x6_of(side)
        chunk_ratings = i.ratings - i.fusedRatings
        prim_ratings, prim_rows = i.fusedRatings - i.dualRatings, i.fusedRows - i.dualRows
        # (name, ms, flops on the Gramian pipe, flops on the fp32/fp64 pipe, gather-model bytes)
        row_k = ("als_gram_solve_kernel", i.gramSolveMs, prim_ratings * gram_rating, prim_rows * solve_row,
                 prim_ratings * bytes_rating + prim_rows * (k * s + 8))
        dual_by = i.dualRatings * bytes_rating + i.dualRows * (k * s + 8)
        # (priced on the fp32 pipe as a whole: the library reports one flop count per row class)
        dual_k = ("als_dual_solve_kernel", i.dualSolveMs, 0.0, i.dualFlops, dual_by)
        slab_in_group = False
        if i.dualOverlapped:
            # the dual kernels ran on side streams next to the row kernel: one group, one wall time.  They are forked BEFORE
            # the chunk kernel of the step's stream, so that kernel shares its interval with them too (with a side stream per
            # dual class its own event interval is mostly their time): the group spans chunk kernel + row kernel + dual classes
            slab_in_group = not gen_path and i.parts == 1 and i.gramSlabMs > 0
            whole_rows = (("als_gram_solve_kernel+als_dual_solve_kernel" + ("+als_gram_slab_kernel" if slab_in_group else ""),
                           row_k[1] + dual_k[1] + (i.gramSlabMs if slab_in_group else 0.0),
                           row_k[2] + (chunk_ratings * gram_rating if slab_in_group else 0.0), row_k[3] + dual_k[3],
                           row_k[4] + dual_by + (chunk_ratings * bytes_rating if slab_in_group else 0.0)),)
        else:
            whole_rows = (row_k, dual_k)
        if gen_path:
            # the any-k path reports its Gramian -> slab and slab -> solve kernels (batches of both) as one interval
            # (Gramian flops against the bf16 pipe's float32-equivalent peak when it runs there, else with the solve's on the fp32 / fp64 pipe)
            gx = x6_of(side)
            split_k = (("als_gen_gram_kernel+als_gen_solve_kernel", i.gramSlabMs + i.reduceSolveMs, chunk_ratings * gram_rating if gx else 0.0,
                        (0.0 if gx else chunk_ratings * gram_rating) + i.splitRows * solve_row, chunk_ratings * bytes_rating + i.splitRows * (k * s + 8)),)
        elif slab_in_group:
            split_k = (("als_reduce_solve_kernel", i.reduceSolveMs, 0.0, i.splitRows * solve_row, i.splitRows * (k * s + 8)),)
        else:
            split_k = (("als_gram_slab_kernel", i.gramSlabMs, chunk_ratings * gram_rating, 0.0, chunk_ratings * bytes_rating),
                       ("als_reduce_solve_kernel", i.reduceSolveMs, 0.0, i.splitRows * solve_row, i.splitRows * (k * s + 8)))
        for name, ms, fg, fs, by in whole_rows + split_k:
            if fg + fs > 0:
                d = kern.setdefault(f"{name}[{side}]", {"ms": 0.0, "fg": 0.0, "fs": 0.0, "bytes": 0.0, "launches": 0, "x6": x6})
                d["ms"] += ms
                d["fg"] += fg
                d["fs"] += fs
                d["bytes"] += by
                d["launches"] += 1
    # the dominant entry: the longest one (a kernel, or the group of kernels that share one interval)
    dom = max(kern, key=lambda n: kern[n]["ms"])

    def describe(n):
        d = kern[n]
        t = d["ms"] * 1e-3
        pg = PEAK_BF16_TFLOPS / 6.0 if d["x6"] else peak
        fl = d["fg"] + d["fs"]
        tf = fl / t / 1e12 if t > 0 else 0.0
        gb = d["bytes"] / t / 1e9 if t > 0 else 0.0
        t_min = d["fg"] / (pg * 1e12) + d["fs"] / (peak * 1e12)   # every flop at the peak of its pipe
        pk = fl / t_min / 1e12 if t_min > 0 else peak
        return {"kernel": n, "launches": d["launches"], "avg_launch_ms": round(d["ms"] / max(d["launches"], 1), 4),
                "achieved_TFLOPs": round(tf, 3), "mfma_peak_TFLOPs": round(pk, 1),
                "mfma_peak": ("blend of bf16 MFMA peak / 6 (Gramian products) and " if d["x6"] and d["fg"] > 0 else "")
                + ("fp64 MFMA peak" if args.double else "fp32 MFMA peak") + (" (solves)" if d["x6"] and d["fg"] > 0 else ""),
                "gramian_flops": d["fg"], "other_flops": d["fs"],
                "mfma_frac": round(t_min / t, 4) if t > 0 else 0.0,
                "algorithmic_GBs": round(gb, 1), "hbm_frac": round(gb / PEAK_HBM_GBS, 4)}

    dd = describe(dom)
    # an entry whose flops could not have run in its time (fraction of the pipe's peak above 1) is an accounting error --
    # flops attributed to a kernel that did not execute them -- and is not printed as a measurement
    bad_entries = [n for n in kern if kern[n]["launches"] and describe(n)["mfma_frac"] > 1.0]
    if bad_entries:
        sys.exit("bench.py: accounting error, mfma_frac > 1 for " + ", ".join(bad_entries) + " -- refusing to print the line")
    tot_ms = sum(d["ms"] for d in kern.values())
    tot_fl = sum(d["fg"] + d["fs"] for d in kern.values())
    tot_by = sum(d["bytes"] for d in kern.values())
    # measured L2-miss traffic of this workload's kernels, when a PMC pass of this command has been
    # aggregated (rocprofv3 cannot run inside the timed bench): profiles/traffic.json
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and world == 1:
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("workloads", {}).get(args.workload + ("_f64" if args.double else ""), {})
        except (OSError, ValueError):
            traffic = {}
    # the roof that binds the dominant entry: the one it is closer to.  The memory side is judged
    # by the measured traffic when there is one (the gather model counts every gathered row as a
    # DRAM read; on the user side the fixed matrix lives in L2)
    mem_frac = dd["hbm_frac"]
    if dom in traffic and kern[dom]["ms"] > 0:
        mem_frac = traffic[dom]["hbm_bytes_per_launch"] * kern[dom]["launches"] / (kern[dom]["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS
    hbm_bound = mem_frac >= dd["mfma_frac"]
    roofline = {
        "bound": "hbm" if hbm_bound else "mfma", "kernel": dom,
        "achieved": dd["algorithmic_GBs"] if hbm_bound else dd["achieved_TFLOPs"],
        "peak": PEAK_HBM_GBS if hbm_bound else dd["mfma_peak_TFLOPs"], "unit": "GB/s" if hbm_bound else "TFLOP/s",
        "frac": dd["hbm_frac"] if hbm_bound else dd["mfma_frac"],
        "traffic": None, "launches": dd["launches"], "avg_launch_ms": dd["avg_launch_ms"],
        "bytes_model": "per rating: 4 (column id) + s (rating) + k s (gathered factor row); per solved row: k s + 8; s = sizeof(T)",
        "flops_model": "executed flops: k(k+1)+2k per rating + k^3/3+2k^2 per row solved in primal form, n(n+1)k + n^3/3 + 2n^2 + 2nk per row "
                       "solved in dual form (n ratings < k); each part against the peak of the pipe it runs on (mfma_peak)",
        "kernels": [describe(n) for n in kern if kern[n]["launches"]],
        # all kernels of the iteration together: executed flops against the fp32 (fp64) MFMA peak, gather-model bytes against HBM
        "iteration": {"kernel_ms_per_step": round(tot_ms / args.steps, 3),
                      "mfma_frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4) if tot_ms else 0.0,
                      "hbm_frac": round(tot_by / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if tot_ms else 0.0,
                      # SURVEY.md 8d's secondary figure: minimum-flop model (symmetric Gramian + Cholesky) of the whole
                      # iteration / wall time / fp32 (fp64) peak -- above ~0.5 means the iteration exists only because
                      # the Gramian left the fp32 pipe
                      "flops_min_frac_of_fp32_peak": round(sum(algorithmic_flops(st["info"].ratings, st["info"].rows, k) for st in lord.stepTimes)
                                                           / args.steps / (ms_per_step * 1e-3) / 1e12 / peak / world, 4),
                      "hbm_roof_ratings_per_s": round(PEAK_HBM_GBS * 1e9 / (2 * bytes_rating)),
                      "frac_of_hbm_roof_ratings_per_s": round(value / world / (PEAK_HBM_GBS * 1e9 / (2 * bytes_rating)), 4),
                      "byUser_ms": round(step_ms["byUser"] / args.steps, 3),
                      "byItem_ms": round(step_ms["byItem"] / args.steps, 3)},
    }
    if dom in traffic:
        roofline["traffic"] = traffic[dom]["hbm_bytes_per_launch"]
        roofline["traffic_source"] = tj.get("sources", {}).get(args.workload + ("_f64" if args.double else ""), "profiles/traffic.json")
        roofline["traffic_measured_at_commit"] = tj.get("commits", {}).get(args.workload + ("_f64" if args.double else ""), tj.get("commit", "unknown"))
        roofline["traffic_over_algorithmic"] = round(traffic[dom]["hbm_bytes_per_launch"] / (kern[dom]["bytes"] / kern[dom]["launches"]), 3)
        roofline["traffic_frac_of_hbm_peak"] = round(mem_frac, 4)

    # where a multi-GPU iteration goes, per half-step and averaged over the timed steps (this rank;
    # compute_ms of every rank is gathered so that a straggler shows)
    exchange = {"path": getattr(lord, "exchangePath", "none")}
    if world > 1 and getattr(lord, "commFallback", None):
        exchange["fallback"] = lord.commFallback  # (why exchange.path is not libycnr_als:<requested_transport>)
    if world > 1 and hasattr(lord.backend, "dev"):
        # what the library's communicator says it is: transport, rank, world, and the ranks RCCL itself counts (ncclCommCount)
        exchange["comm"] = lord.backend.dev.comm_info()
    for sd in ("byUser", "byItem"):
        c = comm[sd]
        n = max(c["n"], 1)
        exchange[sd] = {"compute_ms": round(c["compute_ms"] / n, 4), "exchange_ms": round(c["exchange_ms"] / n, 4),
                        "exposed_exchange_ms": round(c["exposed_exchange_ms"] / n, 4), "wall_ms": round(c["wall_ms"] / n, 4),
                        "bytes": int(c["bytes"] // n), "pieces": c["pieces"]}
    lord.finishExchange()  # (itemStepSharding = 'bands': the user matrix is brought up to date everywhere only now)
    if dist:
        # every replica must hold the same bytes after the last exchange: a checksum of both matrices per rank
        sums = []
        for side in (0, 1):
            f = lord.backend.factors(side)
            sums.append(int(f.view(torch.int32 if f.dtype == torch.float32 else torch.int64).to(torch.int64).sum().item()))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"sums": sums, **{sd: exchange[sd]["compute_ms"] for sd in ("byUser", "byItem")}})
        for sd in ("byUser", "byItem"):
            exchange[sd]["compute_ms_by_rank"] = [r[sd] for r in per_rank]
        exchange["replicas_consistent"] = all(r["sums"] == per_rank[0]["sums"] for r in per_rank)
        exchange["requested_transport"] = requested_transport
        if lord.rebalanced:
            exchange["rebalanced_after_iter"] = args.rebalance_after
            exchange["before_rebalance_ms_by_rank"] = {("byUser", "byItem")[sd]: [round(x, 4) for x in v["ms_by_rank"]]
                                                       for sd, v in lord.rebalanced.items()}
        if args.transport_ab and args.transport_ab != requested_transport:
            exchange["ab"] = transport_ab(args, dist, ds, k, max_rating, local_rank, dev, barrier, lord.shards)

    rmse = lord.calcRmse("rmseValidate", False)
    # the RMSE pass BASELINE's metric names beside the solve: device time of its kernel over this rank's users (HIP events inside
    # libycnr_als.so), max over the ranks
    rmse_ms = lord.backend.dev.last_rmse_ms() if hasattr(lord.backend, "dev") else None
    if dist and rmse_ms is not None:
        t = torch.tensor([rmse_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rmse_ms = float(t.item())
    if args.dump_factors and rank == 0:
        np.savez(args.dump_factors, U=lord.backend.get_factors(0), V=lord.backend.get_factors(1), rmse=rmse)

    cpu = cpu_blas = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(lord, by_user, by_item, k, args)
        cpu_blas = cpu_baseline_blas(lord, by_user, by_item, k, args)

    step_info = None
    if args.dump_step_info:
        step_info = {}
        for st in lord.stepTimes:
            step_info[st["stepType"]] = {f: getattr(st["info"], f) for f, _ in type(st["info"])._fields_}
    if rank == 0:
        out = {
            "metric": "ALS ratings/sec per iteration (U+I solve)", "value": value, "unit": "ratings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if args.double else "f32", "data": "synthetic",
            "config": {"workload": desc, "users": users, "items": items, "nnz": nnz, "factorsCount": k,
                       "lambda": 0.05, "parallelism": (f"row-shard x{world} + direct all-gather ({lord.exchangePath})" if args.item_sharding == "rows" else
                                                       f"users in 8 bands over {world} rank(s): items' Gramians reduce-scattered, item rows all-gathered ({lord.exchangePath})")
                       if world > 1 else "1 GPU", "itemStepSharding": args.item_sharding},
            "rmse_in_sample_after_iters": rmse, "iters_run": args.steps + args.warmup,
            "rmse_ms": rmse_ms, "rmse_ratings": val.nnz, "rmse_ratings_per_s": (val.nnz / (rmse_ms * 1e-3)) if rmse_ms else None,
            "roofline": roofline, "exchange": exchange, "cpu_baseline": cpu, "cpu_baseline_blas": cpu_blas,
            "setup_s": {"generate": round(t_gen, 2), "prepare": round(t_prep, 2)},
        }
        if step_info is not None:
            out["step_info"] = step_info
        print(json.dumps(out), flush=True)
    lord.destroy()
    consistent = (not dist) or exchange.get("replicas_consistent", True)
    if dist:
        dist.destroy_process_group()
    if not consistent:
        sys.exit("bench.py: the replicas of the factor matrices differ between the ranks after the last exchange")


def transport_ab(args, dist, ds, k, max_rating, local_rank, dev, barrier, shards):
    """--transport-ab T: the same warm-up + timed iterations once more with the exchange over transport T, on the shards the
    main run ended with (no re-cut), for a line that shows both transports from one process group.  A transport that cannot
    be set up is reported as an error string; it never takes the main line down."""
    from ycnr_als.emf import EmfLord
    world = dist.get_world_size()
    out = {"transport": args.transport_ab}
    lord = None
    try:
        lord = EmfLord(options={"factorsCount": k, "trainIters": args.steps, "useDoublePrecision": args.double,
                                "dbType": "mal" if max_rating == 10 else "ml", "chunkRatings": args.chunk, "dataSetDistr": [90, 10, 0],
                                "exchangeChunks": args.exchange_chunks, "rebalanceAfterIters": 0, "commTransport": args.transport_ab,
                                "strictTransport": True}, dist=dist)
        lord.prepareToTrain(ds, seed=20260004, device=local_rank, shards={s: np.asarray(shards[s]) for s in (0, 1)})
        for _ in range(args.warmup):
            lord.alsTrainIter()
        lord.stepTimes.clear()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            lord.alsTrainIter()
        barrier()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["path"] = lord.exchangePath
        out["ms_per_step"] = 1e3 * float(t.item()) / args.steps
        for sd in ("byUser", "byItem"):
            st = [x for x in lord.stepTimes if x["stepType"] == sd]
            n = max(len(st), 1)
            out[sd] = {"compute_ms": round(sum(x["info"].totalMs for x in st) / n, 4),
                       "exchange_ms": round(sum(x["info"].exchangeMs for x in st) / n, 4),
                       "exposed_exchange_ms": round(sum(x["info"].exposedExchangeMs for x in st) / n, 4),
                       "wall_ms": round(sum(x["wall"] for x in st) * 1e3 / n, 4)}
        sums = []
        for side in (0, 1):
            f = lord.backend.factors(side)
            sums.append(int(f.view(torch.int32 if f.dtype == torch.float32 else torch.int64).to(torch.int64).sum().item()))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, sums)
        out["replicas_consistent"] = all(r == per_rank[0] for r in per_rank)
    except Exception as e:  # noqa: BLE001 -- reported, the main line stands
        out["error"] = str(e)[:300]
    finally:
        if lord is not None and lord.backend is not None:
            try:
                lord.destroy()
            except Exception:  # noqa: BLE001
                pass
    return out


class EmulatedDist:
    """What EmfLord asks of torch.distributed, for ONE process that plays rank `rank` of `world` (--emulate-world)."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def broadcast_object_list(self, box, src=0):
        if box[0] is None:
            box[0] = bytes(128)  # the stub transport's id carries nothing

    def all_gather_object(self, out, obj):
        for i in range(len(out)):
            out[i] = obj

    def all_reduce(self, t, op=None):
        return t

    def barrier(self):
        pass


def emulate_world(args, local_rank):
    """--emulate-world W: the shard of every rank of a W-GPU run, solved on ONE GPU one rank after the other with
    the full fixed matrices and the real pieces (exchange left out: transport 'stub'), first with the cost-model
    cut of the shards, then with the cut the feedback (EmfLord.rebalance -> rebalanced_ranges) derives from the
    measured times.  Prints one JSON line: compute ms per rank and half-step, the imbalance max / mean, the
    launch overhead of the pieces, and what the slowest rank implies for the W-GPU iteration."""
    from ycnr_als.data import synth_ratings
    from ycnr_als.emf import Dataset, EmfLord, rebalanced_ranges
    W = args.emulate_world
    users, items, nnz_target, k, max_rating, zipf_a, sigma, desc = WORKLOADS[args.workload]
    if args.factors:
        k, desc = args.factors, desc + f" [factorsCount overridden: {args.factors}]"
    dev = torch.device("cuda", local_rank)
    tdt = torch.float64 if args.double else torch.float32
    by_user, by_item = synth_ratings(users, items, nnz_target, max_rating=max_rating, seed=20260004, device=dev, dtype=tdt,
                                     degree_sigma=sigma, zipf_a=zipf_a)
    ds = Dataset(by_user, by_item, validate=None, test=None, total_ratings_avg=float(by_user.vals[: 1 << 24].double().mean()))
    cnt = {0: by_user.counts().cpu().numpy(), 1: by_item.counts().cpu().numpy()}
    names = ("byUser", "byItem")

    def run(shards):
        per = {n: {"compute_ms": [], "wall_ms": [], "pieces": 0, "exchange_bytes": [], "kernel_ms": []} for n in names}
        used = None
        for r in range(W):
            lord = EmfLord(options={"factorsCount": k, "trainIters": args.steps, "useDoublePrecision": args.double,
                                    "dbType": "mal" if max_rating == 10 else "ml", "chunkRatings": args.chunk, "dataSetDistr": [100, 0, 0],
                                    "exchangeChunks": args.exchange_chunks, "commTransport": "stub", "rebalanceAfterIters": 0,
                                    "itemStepSharding": args.item_sharding},
                            dist=EmulatedDist(r, W))
            t_rank = time.time()
            lord.prepareToTrain(ds, seed=20260004, device=local_rank, shards=shards)
            used = {s: np.asarray(lord.shards[s]).tolist() for s in (0, 1)}
            for _ in range(max(args.warmup, 1)):
                lord.alsTrainIter()
            lord.stepTimes.clear()
            for _ in range(args.steps):
                lord.alsTrainIter()
            for n in names:
                st = [s for s in lord.stepTimes if s["stepType"] == n]
                per[n]["compute_ms"].append(round(float(np.mean([s["info"].totalMs for s in st])), 4))
                per[n]["wall_ms"].append(round(float(np.mean([s["wall"] for s in st])) * 1e3, 4))
                per[n]["pieces"] = int(st[0]["info"].parts)
                per[n]["exchange_bytes"].append(int(st[0]["info"].exchangeBytes))
                # sums over the pieces: chunk Gramians, row kernel (+ dual classes when they share its interval), dual, reduce
                per[n]["kernel_ms"].append([round(float(np.mean([getattr(s["info"], f) for s in st])), 3)
                                            for f in ("gramSlabMs", "gramSolveMs", "dualSolveMs", "reduceSolveMs")])
            lord.destroy()
            del lord
            torch.cuda.empty_cache()
            # (progress on stderr: at C5 scale a rank takes a minute of host-side preparation, and a silent run looks hung)
            print(f"emulate-world: rank {r + 1}/{W} of the {'cost-model' if shards is None else 'feedback'} cut done in "
                  f"{time.time() - t_rank:.1f} s: byUser {per['byUser']['compute_ms'][-1]} ms, byItem {per['byItem']['compute_ms'][-1]} ms",
                  file=sys.stderr, flush=True)
        for n in names:
            c = np.asarray(per[n]["compute_ms"])
            per[n]["imbalance_max_over_mean"] = round(float(c.max() / c.mean()), 4)
            per[n]["launch_overhead_ms"] = round(float(np.mean(np.asarray(per[n]["wall_ms"]) - c)), 4)
        per["iteration_ms_slowest_rank"] = round(max(per["byUser"]["wall_ms"]) + max(per["byItem"]["wall_ms"]), 4)
        return per, used

    first, shards0 = run(None)
    if args.item_sharding == "bands":  # (the bands are the shards: nothing to re-cut)
        out = {"emulated_world": W, "workload": desc, "factorsCount": k, "dtype": "f64" if args.double else "f32", "steps": args.steps,
               "itemStepSharding": "bands",
               "note": "one GPU solves every rank's share in turn (transport 'stub', exchange left out): byUser = the rank's user band, "
                       "no exchange; byItem = Gramians of ALL items over the band's ratings + reduce and solve of the rank's items (the other "
                       "ranks' band sums are not there: an eighth of the reduce's reads); exchange_bytes = band sums sent + received + "
                       "item rows, per rank and iteration",
               "bands_cut": first, "shards": {"bands": shards0}}
        print(json.dumps(out), flush=True)
        return
    new = {s: rebalanced_ranges(cnt[s], shards0[s], first[names[s]]["compute_ms"], k, args.double) for s in (0, 1)}
    second, shards1 = run(new)
    out = {"emulated_world": W, "workload": desc, "factorsCount": k, "dtype": "f64" if args.double else "f32", "steps": args.steps,
           "note": "one GPU solves every rank's shard in turn: full fixed matrices, real pieces, exchange left out (transport 'stub'); "
                   "iteration_ms_slowest_rank = what the W-GPU iteration costs before any exposed exchange",
           "cost_model_cut": first, "after_feedback_recut": second,
           "shards": {"cost_model": shards0, "feedback": shards1}}
    print(json.dumps(out), flush=True)


def host_cpus():
    """CPUs this process can really use: the affinity mask capped by the cgroup CPU quota.  (On
    the GPU boxes os.cpu_count() is 256 but the quota is 16 CPUs; 256 OpenMP threads under that
    quota ran the oracle 3x slower than 16.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def cpu_baseline(lord, by_user, by_item, k, args):
    """Time the CPU oracle (kind 'port': a structure-faithful restatement of
    EmfWorker.mw_calcTrainAlsPortion) on a bounded, contiguous row sample of both half-steps,
    with the same factor matrices the GPU run ended with.  ratings/s of a full iteration is
    extrapolated as 1 / (1/rate_user + 1/rate_item)."""
    from oracle import oracle as orc
    orc.build()
    cores = host_cpus()
    dt = np.float64 if args.double else np.float32
    U = lord.backend.get_factors(0)
    V = lord.backend.get_factors(1)
    def cost_of(cnt):
        return cnt * (2.0 * k * k + 2 * k) + (cnt > 0) * (2.0 / 3 * k ** 3 + 2 * k * k)

    def run(csr, rp, rows, fixed, solved, lam):
        e = int(rp[rows])
        indx = csr.indx[:e].cpu().numpy()
        vals = csr.vals[:e].cpu().numpy().astype(dt)
        out = solved.copy()
        t = time.perf_counter()
        n = orc.als_step_csr(lam, k, np.ascontiguousarray(rp[:rows + 1]), indx, vals, fixed, out, 0, rows, threads=cores)
        return n, time.perf_counter() - t

    def sample(csr, fixed, solved, lam):
        """Probe a small prefix to measure this box's flop rate, then time a prefix sized for
        about cpu_seconds / 2 of CPU work (bounded: the default run must finish in minutes)."""
        rp = csr.rowPtr.cpu().numpy()
        cost = np.cumsum(cost_of(rp[1:] - rp[:-1]))
        probe_rows = max(1, min(csr.rows, int(np.searchsorted(cost, 2e9 * min(cores, 16))) + 1))
        n, t = run(csr, rp, probe_rows, fixed, solved, lam)
        rate = cost[probe_rows - 1] / max(t, 1e-3)
        rows = max(1, min(csr.rows, int(np.searchsorted(cost, rate * args.cpu_seconds * 0.5)) + 1))
        if rows <= probe_rows:
            return n, t, probe_rows
        n, t = run(csr, rp, rows, fixed, solved, lam)
        return n, t, rows

    nu, tu, ru = sample(by_user, V, U, 0.05)
    ni, ti, ri = sample(by_item, U, V, 0.05)
    rate_u, rate_i = nu / tu, ni / ti
    return {"value": 1.0 / (1.0 / rate_u + 1.0 / rate_i), "unit": "ratings/s", "cores": cores, "kind": "port",
            "sample": f"first {ru} user rows ({nu} ratings, {tu:.1f} s) + first {ri} item rows ({ni} ratings, {ti:.1f} s) "
                      f"of the same workload, OpenMP over rows with {cores} threads (affinity mask capped by the cgroup CPU quota; "
                      f"os.cpu_count() = {os.cpu_count()}); extrapolated to a full iteration",
            "rate_byUser": rate_u, "rate_byItem": rate_i}


def cpu_baseline_blas(lord, by_user, by_item, k, args):
    """The same bounded sample idea through BLAS / LAPACK (kind 'blas'): the reference's per-row gemm +
    gesv structure in PyTorch's CPU BLAS (MKL), one single-threaded worker process per host core like
    the reference's worker processes (oracle/blas_baseline.py, run as a child process so that no
    process forks after it has touched the GPU).  Stands in for the reference's nblas / LAPACK path,
    which cannot run here.  About half the CPU budget of the port sample."""
    import subprocess
    import tempfile
    cores = host_cpus()
    dt = np.float64 if args.double else np.float32
    U = lord.backend.get_factors(0)
    V = lord.backend.get_factors(1)
    helper = os.path.join(ROOT, "oracle", "blas_baseline.py")

    def run(csr, rp, rows, fixed):
        e = int(rp[rows])
        indx = csr.indx[:e].cpu().numpy()
        uniq, inv = np.unique(indx, return_inverse=True)   # only the fixed rows the sample touches travel to the workers
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "sample.npz")
            np.savez(path, k=k, lam=0.05, rowPtr=np.ascontiguousarray(rp[:rows + 1]), indx=inv.astype(np.int32),
                     vals=csr.vals[:e].cpu().numpy().astype(dt), fixed=np.ascontiguousarray(fixed[uniq]))
            out = subprocess.run([sys.executable, helper, path, str(cores)], capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            raise RuntimeError("blas baseline failed: " + out.stderr[-400:])
        j = json.loads(out.stdout.strip().splitlines()[-1])
        return j["ratings"], j["busy_seconds_max"]

    def sample(csr, fixed):
        rp = csr.rowPtr.cpu().numpy()
        # probe ~200 K ratings, then a prefix sized for about cpu_seconds / 4 per side
        probe_rows = max(1, min(csr.rows, int(np.searchsorted(rp, 200_000)) + 1))
        n, t = run(csr, rp, probe_rows, fixed)
        rate = n / max(t, 1e-3)
        rows = max(1, min(csr.rows, int(np.searchsorted(rp, rate * args.cpu_seconds * 0.25)) + 1))
        if rows <= probe_rows:
            return n, t, probe_rows
        n, t = run(csr, rp, rows, fixed)
        return n, t, rows

    try:
        nu, tu, ru = sample(by_user, V)
        ni, ti, ri = sample(by_item, U)
    except Exception as e:  # noqa: BLE001 -- a reported baseline must not take the bench line down
        return {"kind": "blas", "error": str(e)[:300]}
    rate_u, rate_i = nu / tu, ni / ti
    import torch as _t
    return {"value": 1.0 / (1.0 / rate_u + 1.0 / rate_i), "unit": "ratings/s", "cores": cores, "kind": "blas",
            "blas": "PyTorch CPU (" + ("MKL" if _t.backends.mkl.is_available() else "default BLAS") + "): gemm + LAPACK gesv per row",
            "sample": f"first {ru} user rows ({nu} ratings, {tu:.1f} s) + first {ri} item rows ({ni} ratings, {ti:.1f} s) of the same "
                      f"workload, {cores} single-threaded worker processes; extrapolated to a full iteration",
            "rate_byUser": rate_u, "rate_byItem": rate_i}


if __name__ == "__main__":
    main()
