"""The exchange step of the multi-GPU path (include/ycnr_als.h "multi-GPU") on one MI355X.

One GPU cannot show scaling, but it can show that the code the 8-GPU run depends on works:
  * the RCCL transport: communicator creation from a unique id, a self-addressed ncclSend / ncclRecv
    pair and an all-reduce run on hardware (world size 1), and train() end to end under
    torch.distributed's nccl backend;
  * the sharded, pipelined half-step: pieces + exchange ranges give bit-identical factors;
  * several ranks on one GPU, between real processes, against the single-process result bit for bit: the
    host-staged stand-in ('shm') and the device-to-device transport ('ipc': mapped peer replicas, rows pushed
    by hipMemcpyAsync on the communicator's stream, piece by piece behind the kernels);
  * two ranks on two devices over 'rccl' and 'ipc' when the box has them (the scaling job's node does).
"""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from helpers import EPS32, make_problem, numpy_step, row_rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def als():
    import ycnr_als
    L = ycnr_als._lib.load()
    assert L.ycnr_device_count() >= 1, L.ycnr_last_error()
    return ycnr_als


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def problem(k=36, users=900, items=400, seed=3, density=0.08):
    bu, bi, U, V = make_problem(users, items, k, density=density, seed=seed, empty_rows=(5, 17))
    return k, users, items, bu, bi, U, V


# (k, users, items, seed, density): the default small problem; k = 100 (bf16x6 row kernel, packed last block, dual classes);
# k = 256 with rows of ~190 ratings either side of the primal / dual crossover (Gramian -> slab -> four-wave solve in
# batches, dual classes on the side streams) and items of ~330 ratings
# k = 50 (float32, factorsCount % 4 != 0: the kernels work on copies padded to 52 columns, every piece unpads its rows before its exchange)
SHAPES = {"k36": (36, 900, 400, 3, 0.08), "k100": (100, 1200, 500, 4, 0.15), "k256": (256, 700, 400, 5, 0.47), "k50": (50, 1000, 450, 6, 0.12),
          "k64f": (64, 700, 300, 9, 0.3)}  # k % 16 == 0: the Gramian kernels without a padded right-hand-side column


def reference_iteration(als, k, users, items, bu, bi, U, V):
    dev = als.AlsDevice(k, users, items)
    dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
    dev.set_ratings("byItem", bi.rowPtr, bi.indx, bi.vals)
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    dev.step("byUser")
    dev.step("byItem")
    out = dev.get_factors("byUser"), dev.get_factors("byItem")
    dev.destroy()
    return out


def test_rccl_communicator_runs_on_hardware(als):
    """World size 1 over RCCL: unique id, ncclCommInitRank, grouped ncclSend + ncclRecv (to self),
    ncclAllReduce; then a half-step cut into 3 pieces with exchange ranges = the unsharded result."""
    k, users, items, bu, bi, U, V = problem()
    U1, V1 = reference_iteration(als, k, users, items, bu, bi, U, V)
    dev = als.AlsDevice(k, users, items)
    uid = als.AlsDevice.comm_unique_id("rccl")
    assert len(uid) == 128 and any(uid)
    dev.comm_init(uid, 0, 1, "rccl")
    dev.comm_selftest(1 << 18)
    with pytest.raises(als.YcnrError):  # one communicator per handle
        dev.comm_init(uid, 0, 1, "rccl")
    ub = np.array([[0, 300, 610, users]], np.int64)
    ib = np.array([[0, items]], np.int64)
    dev.set_ratings_sharded("byUser", bu.rowPtr, bu.indx, bu.vals, ub)
    dev.set_ratings_sharded("byItem", bi.rowPtr, bi.indx, bi.vals, ib)
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    iu = dev.step("byUser")
    ii = dev.step("byItem")
    assert iu.parts == 3 and ii.parts == 1 and iu.ratings == bu.nnz and iu.rows == users - 2
    assert np.array_equal(dev.get_factors("byUser"), U1) and np.array_equal(dev.get_factors("byItem"), V1)
    s = dev.allreduce_sum(np.array([1.5, -2.0]))
    assert np.array_equal(s, [1.5, -2.0])
    dev.exchange("byUser")
    dev.broadcast_factors("byItem", 0)
    with pytest.raises(als.YcnrError):  # shards must tile the rows in rank order
        dev.set_ratings_sharded("byUser", bu.rowPtr, bu.indx, bu.vals, np.array([[0, 500, 400, users]], np.int64))
    dev.comm_destroy()
    dev.destroy()


def _nccl_train(port, out):
    import torch
    import torch.distributed as dist
    from ycnr_als.data import select_csr, split_to_sets, synth_ratings, transpose_csr
    from ycnr_als.emf import Dataset, EmfLord
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    by_user, _ = synth_ratings(600, 400, 30_000, max_rating=5, seed=11, degree_sigma=0.9, zipf_a=0.8)
    t = split_to_sets(by_user, (85, 10, 5), seed=3)
    tr = select_csr(by_user, t <= 2)
    ds = Dataset(tr, transpose_csr(tr), select_csr(by_user, t == 2), select_csr(by_user, t == 3))
    res = []
    for d in (dist, None):
        lord = EmfLord(options={"factorsCount": 20, "trainIters": 3, "dataDir": "/tmp/ycnr_test_nccl1"}, dist=d)
        lord.prepareToTrain(ds, seed=7)
        hist = lord.train()
        res.append((hist, lord.backend.get_factors(0), lord.backend.get_factors(1), lord.getCalcInfo()["globalAvgShift"]))
        lord.destroy()
    dist.destroy_process_group()
    (h1, U1, V1, s1), (h2, U2, V2, s2) = res
    ok = np.array_equal(U1, U2) and np.array_equal(V1, V2) and h1 == h2 and s1 == s2
    out.put(bool(ok))


def test_train_under_the_nccl_backend(als):
    """train() -- half-steps, the three RMSE passes and globalAvgShift -- with torch.distributed's nccl
    backend initialised (world size 1) gives the results of the plain single-process run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_train, args=(free_port(), q))
    p.start()
    p.join(600)
    assert p.exitcode == 0, "the nccl process failed"
    assert q.get(timeout=5) is True


def _rank_main(rank, world, uid, pieces, out, transport="shm", device=0, shape="k36", back_to_back=False):
    import ycnr_als as als
    k, users, items, bu, bi, U, V = problem(*SHAPES[shape])
    dev = als.AlsDevice(k, users, items, device=device)
    dev.comm_init(uid, rank, world, transport)
    if transport != "ipc":
        dev.comm_selftest(1024)
    cut = lambda n, parts: np.linspace(0, n, parts + 1).astype(np.int64)
    us, its = cut(users, world), cut(items, world)
    ub = np.stack([us[r] + cut(us[r + 1] - us[r], pieces) for r in range(world)])
    ib = np.stack([its[r] + cut(its[r + 1] - its[r], 1) for r in range(world)])
    dev.set_ratings_sharded("byUser", bu.rowPtr, bu.indx, bu.vals, ub)
    dev.set_ratings_sharded("byItem", bi.rowPtr, bi.indx, bi.vals, ib)
    dev.set_factors("byUser", U)
    # only rank 1 holds the item factors at first: the join-time copy (EmfChief.js:55-71)
    dev.set_factors("byItem", V if rank == 1 else np.zeros_like(V))
    dev.broadcast_factors("byItem", 1)
    if back_to_back:
        # both half-steps enqueued before one sync: the item half-step must not read user rows the peers are still
        # pushing (ycnr_als_step_async completes a pending IPC half-step first)
        dev.step_async("byUser")
        dev.step_async("byItem")
        dev.sync()
        ii = dev.last_step_info()
        iu = ii
        ratings = float(bu.nnz)
    else:
        iu = dev.step("byUser")
        ii = dev.step("byItem")
        ratings = float(iu.ratings)
    s = dev.allreduce_sum(np.array([float(rank + 1), ratings]))
    got = dev.get_factors("byUser"), dev.get_factors("byItem")
    dev.destroy()
    out.put((rank, got, s.tolist(), int(iu.parts), int(iu.exchangeBytes), int(ii.exchangeBytes)))


@pytest.mark.parametrize("transport,world,pieces,shape", [("shm", 2, 1, "k36"), ("shm", 3, 4, "k36"), ("ipc", 2, 1, "k36"), ("ipc", 3, 4, "k36"),
                                                          ("ipc", 2, 3, "k100"), ("ipc", 2, 3, "k256"), ("shm", 3, 2, "k256"), ("ipc", 2, 3, "k50")])
def test_ranks_of_one_node_on_one_gpu(als, transport, world, pieces, shape):
    """Several ranks sharing cuda:0: 'shm' stages rows through the host, 'ipc' is the device-to-device path -- every
    rank maps its peers' replicas (hipIpcOpenMemHandle) and pushes its solved rows into them piece by piece on
    the communicator's stream, behind the kernels that produced them.  Exchange, join-time broadcast and
    all-reduce between real processes, against the single-process result bit for bit."""
    k, users, items, bu, bi, U, V = problem(*SHAPES[shape])
    U1, V1 = reference_iteration(als, k, users, items, bu, bi, U, V)
    uid = als.AlsDevice.comm_unique_id(transport)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, uid, pieces, q, transport, 0, shape)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, (Ug, Vg), s, parts, xu, xi in res:
        assert np.array_equal(Ug, U1) and np.array_equal(Vg, V1), f"rank {rank}: replicas differ from the single-process run"
        assert s == [world * (world + 1) / 2, float(bu.nnz)]
        assert parts == pieces and xu > 0 and xi > 0


@pytest.mark.parametrize("transport,world,shape", [("ipc", 2, "k100"), ("ipc", 3, "k100"), ("shm", 2, "k100"), ("ipc", 3, "k50")])
def test_back_to_back_async_half_steps(als, transport, world, shape):
    """step_async(byUser); step_async(byItem); sync between real processes sharing cuda:0: on the push transport the
    second half-step would read user rows that its peers are still writing unless the first is completed before it
    starts (round-3 review; ycnr_als.h documents the rule).  Bit for bit against the single-process result.
    k50: a padded upload (float32, factorsCount % 4 != 0) -- the pad of the fixed matrix is a READ of the matrix the peers push
    into and must sit behind that completion too (round-4 review: it was enqueued in front of it)."""
    k, users, items, bu, bi, U, V = problem(*SHAPES[shape])
    U1, V1 = reference_iteration(als, k, users, items, bu, bi, U, V)
    uid = als.AlsDevice.comm_unique_id(transport)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, uid, 3, q, transport, 0, shape, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, (Ug, Vg), s, parts, xu, xi in res:
        assert np.array_equal(Ug, U1) and np.array_equal(Vg, V1), f"rank {rank}: replicas differ from the single-process run"


def _band_setup(k, users, items, bu, bi, world):
    """8 cost-balanced user bands (whatever the world size), rank r holding 8 / world of them; items cut evenly"""
    from ycnr_als.emf import shard_ranges
    bands = shard_ranges(np.diff(bu.rowPtr), 8, k)
    rank_bands = np.arange(world + 1, dtype=np.int64) * (8 // world)
    owners = np.linspace(0, items, world + 1).astype(np.int64)
    return bands, rank_bands, owners


def _banded_iteration(dev, rank, world, k, users, items, bu, bi, U, V, chunk_check=False):
    from ycnr_als.emf import _columns_between
    bands, rank_bands, owners = _band_setup(k, users, items, bu, bi, world)
    ub = bands[rank_bands]
    if world > 1:
        dev.set_ratings_sharded("byUser", bu.rowPtr, bu.indx, bu.vals, np.stack([ub[r:r + 2] for r in range(world)]))
        dev.defer_exchange("byUser", True)
    else:
        dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
    sub = _columns_between(bi, int(ub[rank]), int(ub[rank + 1]))
    dev.set_ratings_banded("byItem", sub.rowPtr, sub.indx, sub.vals, bands, rank_bands, owners)
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    iu = dev.step("byUser")
    ii = dev.step("byItem")
    # a second iteration: the item half-step above must have left every replica of V current, the user half-step reads it
    dev.step("byUser")
    dev.step("byItem")
    if world > 1:
        dev.exchange("byUser")
    return dev.get_factors("byUser"), dev.get_factors("byItem"), iu, ii


def _band_rank_main(rank, world, uid, out, shape, chunk, transport="ipc", device=0):
    import ycnr_als as als
    k, users, items, bu, bi, U, V = problem(*SHAPES[shape])
    dev = als.AlsDevice(k, users, items, chunkRatings=chunk, device=device)
    dev.comm_init(uid, rank, world, transport)
    Ug, Vg, iu, ii = _banded_iteration(dev, rank, world, k, users, items, bu, bi, U, V)
    dev.destroy()
    out.put((rank, Ug, Vg, int(ii.exchangeBytes), int(iu.exchangeBytes)))


@pytest.mark.parametrize("shape,chunk", [("k36", 0), ("k100", 32), ("k64f", 0)])
def test_item_half_step_sharded_by_user_bands(als, shape, chunk):
    """ycnr_als_set_ratings_banded: the item half-step as Gramians of ALL items over every rank's own users' ratings, one slab per
    (item, band of users), the owner adds an item's bands in band order and solves.  One rank with all 8 bands must solve every
    item within the conditioning bound of its float64 solve (another order of the sums than the row-sharded half-step, not
    another result); 2 and 4 ranks sharing cuda:0 over `ipc` -- band sums pushed into the owners' buffers, the solved item rows
    pushed into every replica, the user matrix exchanged only at the very end -- must reproduce the one-rank run bit for bit
    after two iterations.  chunk = 32: (item, band) segments cut into several chunks (the sum kernel)."""
    k, users, items, bu, bi, U, V = problem(*SHAPES[shape])
    dev = als.AlsDevice(k, users, items, chunkRatings=chunk)
    U1, V1, iu, ii = _banded_iteration(dev, 0, 1, k, users, items, bu, bi, U, V)
    dev.destroy()
    # against float64, one iteration at a time: redo the two iterations on the host
    Uh, Vh = U.astype(np.float64), V.astype(np.float64)
    for _ in range(2):
        Uh, _c = numpy_step(0.05, k, bu, Vh, Uh)
        Vh, ci = numpy_step(0.05, k, bi, Uh, Vh)
    e = row_rel_err(V1, Vh)
    assert (e <= np.maximum(64 * ci * EPS32, 1e-5)).all(), float((e / np.maximum(64 * ci * EPS32, 1e-5)).max())
    for world in (2, 4):
        uid = als.AlsDevice.comm_unique_id("ipc")
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_band_rank_main, args=(r, world, uid, q, shape, chunk)) for r in range(world)]
        for p in procs:
            p.start()
        res = []
        while len(res) < world:
            try:
                res.append(q.get(timeout=5))
            except Exception:  # noqa: BLE001 -- queue.Empty: is everybody still alive?
                dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
                assert not dead, f"a rank process died (exit codes {dead})"
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        for rank, Ug, Vg, xi, xu in res:
            assert np.array_equal(Vg, V1), f"world {world}, rank {rank}: item factors differ from the one-rank run"
            assert np.array_equal(Ug, U1), f"world {world}, rank {rank}: user factors differ from the one-rank run"
            assert xi > 0 and xu == 0   # band sums + item rows travelled; the user half-step exchanged nothing


@pytest.mark.parametrize("transport", ["rccl", "ipc"])
def test_two_ranks_on_two_gpus(als, transport):
    """The product transports between two DEVICES (skipped on the one-GPU boxes; the scaling job's node has
    eight): grouped ncclSend / ncclRecv into the live matrix, and the mapped-replica pushes, pipelined per piece,
    against the single-process result bit for bit."""
    L = als._lib.load()
    if L.ycnr_device_count() < 2:
        pytest.skip("needs two visible devices")
    k, users, items, bu, bi, U, V = problem()
    U1, V1 = reference_iteration(als, k, users, items, bu, bi, U, V)
    uid = als.AlsDevice.comm_unique_id(transport)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, uid, 3, q, transport, r)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, (Ug, Vg), s, parts, xu, xi in res:
        assert np.array_equal(Ug, U1) and np.array_equal(Vg, V1), f"rank {rank}: replicas differ from the single-process run"
        assert parts == 3 and xu > 0 and xi > 0


@pytest.mark.parametrize("transport", ["rccl", "ipc"])
def test_user_bands_on_two_gpus(als, transport):
    """The banded item half-step between two DEVICES (skipped on the one-GPU boxes): the band sums travel by grouped ncclSend /
    ncclRecv (rccl, stream-ordered) or by pushes into the owner's mapped buffer (ipc), bit for bit the one-rank result."""
    L = als._lib.load()
    if L.ycnr_device_count() < 2:
        pytest.skip("needs two visible devices")
    k, users, items, bu, bi, U, V = problem(*SHAPES["k100"])
    dev = als.AlsDevice(k, users, items, chunkRatings=32)
    U1, V1, _, _ = _banded_iteration(dev, 0, 1, k, users, items, bu, bi, U, V)
    dev.destroy()
    uid = als.AlsDevice.comm_unique_id(transport)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_band_rank_main, args=(r, 2, uid, q, "k100", 32, transport, r)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, Ug, Vg, xi, xu in res:
        assert np.array_equal(Vg, V1) and np.array_equal(Ug, U1), f"rank {rank}: factors differ from the one-rank run"


def test_emulated_world_reports_every_rank(tmp_path):
    """bench.py --emulate-world: one GPU solves every rank's shard of a 3-GPU run in turn over the stub transport and
    reports compute ms per rank before and after the feedback re-cut (the tool behind DESIGN.md's 8-rank table)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "ml1m", "--emulate-world", "3", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["emulated_world"] == 3
    for cut in ("cost_model_cut", "after_feedback_recut"):
        for side in ("byUser", "byItem"):
            c = d[cut][side]
            assert len(c["compute_ms"]) == 3 and all(x > 0 for x in c["compute_ms"]) and c["imbalance_max_over_mean"] >= 1.0
            assert len(c["kernel_ms"]) == 3 and sum(c["exchange_bytes"]) > 0
    for side in (0, 1):
        b0, b1 = d["shards"]["cost_model"][str(side)], d["shards"]["feedback"][str(side)]
        assert b0[0] == b1[0] == 0 and b0[-1] == b1[-1] and len(b0) == len(b1) == 4
