"""Host-side mirror of the reference's trainer classes for the ALS path, in Python.

Same method names, argument meaning and result files as lib/emf/EmfLord.js /
EmfMaster.js / EmfManager.js, minus everything outside the hot path (PostgreSQL, worker
processes, TCP cluster).  The NodeJS twin lives in ../../lib/emf/.  One process drives one
GPU; with torch.distributed initialised, rows are sharded over the ranks and the solved
shards are exchanged after every half-step (the role of 'alsSaveCalcedFactors',
lib/emf/EmfMaster.js:711-723).
"""
import copy
import datetime
import json
import math
import os
import shutil
import time

import numpy as np

from .data import Csr
from .trainer import AlsDevice


def default_options():
    """The path's subset of EmfBase.DefaultOptions (lib/emf/EmfBase.js:52-140)."""
    return {
        "dbType": "ml",
        "maxRating": {"mal": 10, "ml": 5},
        "als": {"userFactReg": 0.05, "itemFactReg": 0.05, "initFirstFactorAsAvgRating": False},
        "warmStart": False,                 # prepareToTrain reuses / extends <dbType>_factors_ready (EmfManager.js:405-457)
        "saveCalcResultsEveryIter": False,  # checkpoint after every iteration (reference todo, YcnrController.js:288)
        "factorsCount": 100,
        "trainIters": 10,
        "alg": "als",
        "dataSetDistr": [85, 10, 5],
        "ratingsInPortionForRmse": 10 * 1000,
        "ratingsInPortionForAls": {"byUser": 10 * 1000, "byItem": 10 * 1000},
        "numThreadsForTrain": {"als": os.cpu_count() or 1, "sgd": 1},
        "numThreadsForRmse": os.cpu_count() or 1,
        "useDoublePrecision": False,
        # not in the reference (opt-in): consume the ratings the REFERENCE's portion packer really hands to
        # its workers -- the last rating of every portion dropped (lib/emf/EmfMaster.js:594-603) -- for bug-for-bug replays
        # of a reference run; portions as splitToPortions cuts them for each pass (lib/emf/EmfLord.js:510-612)
        "dropLastRatingPerPortion": False,
        # not in the reference: where the factor directories live (reference: <repo>/data)
        "dataDir": "data",
        # ratings per wave-level work unit on the GPU (0 = library default)
        "chunkRatings": 0,
        # multi-GPU only: a side whose matrix is large is solved in this many pieces per rank, so that
        # the exchange of one piece overlaps with the solve of the next (1 = solve, then exchange)
        "exchangeChunks": 4,
        # multi-GPU only: "rccl" (the product path), "ipc" (peers' replicas mapped through hipIpc, rows pushed by
        # the copy engines; also runs with several ranks on one GPU) or "shm" (functional stand-in: ranks of
        # one node stage rows through POSIX shared memory)
        "commTransport": "rccl",
        # True: fail when commTransport cannot be set up on every rank (default: rccl <-> ipc are tried in turn, then
        # torch.distributed's all-gather; exchangePath / commFallback say what ran and why)
        "strictTransport": False,
        # multi-GPU only: after this many iterations (counted from the first one this Lord runs) the row shards
        # are cut again from the compute time every rank measured (0 = never); the static cut comes from a cost
        # model (row_cost), this is the feedback that corrects it -- the role of the reference's work-stealing
        # portion dispenser (EmfLord.m_incrNextPortion, lib/emf/EmfLord.js:996-1006)
        "rebalanceAfterIters": 2,  # a re-cut after each of the first N iterations
        # multi-GPU only: how the ITEM half-step is sharded.  "rows": every rank solves a range of items against the whole user
        # matrix, which is therefore all-gathered after every user half-step (700 MB at MAL scale).  "bands": the users are cut
        # into 8 cost-balanced bands (the same for every world size), a rank holds its bands' users, accumulates the Gramians of
        # ALL items over its own users' ratings and the per-band sums travel to the items' owners (ycnr_als_set_ratings_banded):
        # the user matrix is never all-gathered, the order of every sum is fixed by the bands, 1 / 2 / 4 / 8 ranks give the same
        # bits.  Needs world in {1, 2, 4, 8} and a backend with set_ratings_banded; the shards are not re-cut in this mode.
        "itemStepSharding": "rows",
    }


def deepmerge(*dicts):
    """deepmerge.all([...]) as used by EmfBase.init (lib/emf/EmfBase.js:284-287)."""
    out = {}
    for d in dicts:
        for k, v in (d or {}).items():
            if isinstance(v, dict) and isinstance(out.get(k), dict):
                out[k] = deepmerge(out[k], v)
            else:
                out[k] = copy.deepcopy(v)
    return out


def split_to_portions(cnt_per_row, rows_cnt, ratings_in_portion, num_threads, pct=0):
    """Row partitioner of EmfLord.splitToPortions (lib/emf/EmfLord.js:510-612) for one step.

    cnt_per_row[id] <= 0 are the holes of the reference's sparse array.  pct: 0 for the ALS
    steps, dataSetDistr[1]+1 / dataSetDistr[2]+1 for rmseValidate / rmseTest.
    Returns (portionsRowIdTo [1-based inclusive upper ids], maxRatingsInPortion, maxRowsInPortion)."""
    cnt = np.asarray(cnt_per_row, np.int64)
    pos = cnt[cnt > 0]
    if len(pos) == 0:
        return np.zeros(0, np.int64), ratings_in_portion, 0
    ratings_count = int(pos.sum())
    max_per_row = int(pos.max())
    if pct > 0:
        ratings_count = math.ceil(ratings_count * (pct / 100))
        max_per_row = math.ceil(max_per_row * (pct / 100))
    avg_portions = math.ceil(ratings_count / ratings_in_portion)
    if avg_portions < num_threads:
        avg_portions = num_threads
        ratings_in_portion = math.ceil(ratings_count / avg_portions)
    if rows_cnt // max(avg_portions, 1) < 1:
        avg_portions = rows_cnt
        ratings_in_portion = math.ceil(ratings_count / avg_portions)
    if ratings_in_portion < max_per_row:
        ratings_in_portion = max_per_row
    ids = np.nonzero(cnt > 0)[0]
    c = cnt[ids]
    if pct > 0:
        c = np.ceil(c * (pct / 100)).astype(np.int64)
    # greedy: open a new portion when the next row would overflow (EmfLord.js:582-591)
    row_id_to, rtgs, rows, max_rows = [], 0, 0, 0
    for j in range(len(ids)):
        if rtgs + c[j] > ratings_in_portion:
            rtgs, rows = 0, 0
            row_id_to.append(0)
        if not row_id_to:
            row_id_to.append(0)
        rtgs += int(c[j])
        rows += 1
        max_rows = max(max_rows, rows)
        row_id_to[-1] = int(ids[j]) + 1
    return np.asarray(row_id_to, np.int64), ratings_in_portion, max_rows


def row_cost(counts, k, double=False):
    """Modelled cost (SIMD-cycles) of re-solving one row with `counts` ratings in a half-step, used
    to cut shards that finish together.  Balancing ratings alone (what the reference's splitToPortions
    does, lib/emf/EmfLord.js:571-592) gives the shard with many short rows more work per rating.

    The model follows what the kernels execute, with constants from measured kernel times:
      * a rating costs the update of the T = nb (nb + 1) / 2 upper 16 x 16 tiles of the Gramian (nb = ceil(k / 16)):
        4.9 cycles per tile on the one-wave bf16x6 kernels of k <= 128 (MAL scale, k = 100: 120 cycles per rating
        with the packed last block of k = 16 m + 4, which saves half a tile column), 5.4 on the workgroup kernels
        of 128 < k <= 256 (2.95 us of a CU per 32 ratings at k = 256);
      * a row in primal form costs its solve whatever its length: 1500 cycles per diagonal tile (the pivot chains)
        + 35 per float32 MFMA of panel and trailing update up to k = 128 (16.5 K at k = 100, where the four edge
        columns are eliminated first), 0.0153 k^3 on the four-wave solve beyond (32 us of a CU at k = 256);
      * rows with fewer ratings than factors take the dual (n x n) form: 2700 m^1.36 k / 100 for m = ceil(n / 16)
        blocks (2.7 K ... 24 K for the classes of 16 ... 80 ratings at k = 100; 132 K measured for 176 ratings at k = 256);
      * float64 and the any-k path (no dual classes): coarse multiples of the above.
    The feedback re-cut (rebalanced_ranges) corrects what the model gets wrong on a given box."""
    n = np.asarray(counts, np.float64)
    if not double:
        k = (k + 3) // 4 * 4  # float32 sizes that are not multiples of 4 run on matrices padded to the next one (kPad): its classes, its edge
    nb = (k + 15) // 16
    tiles = nb * (nb + 1) / 2.0
    dual_max = 0 if double else 16 * min(12 if k > 128 else 5, nb - 1)
    edge4 = (not double) and k <= 128 and nb >= 2 and k % 16 == 4
    nbs = nb - 1 if edge4 else nb
    mfmas = 4.0 * (nbs * (nbs - 1) / 2.0 + (nbs - 1) * nbs * (nbs + 1) / 6.0) + (nbs * (nbs + 1) / 2.0 if edge4 else 0.0)
    if double:
        per_rating, per_row = 16.0 * tiles, 2.0 * (1500.0 * nbs + 35.0 * mfmas) if k <= 128 else 0.1 * float(k) ** 3
        if k > 128:
            per_rating = 40.0 * tiles
    elif k <= 128:
        per_rating, per_row = 4.9 * (tiles - (nb / 2.0 if edge4 else 0.0)), 1500.0 * nbs + 35.0 * mfmas
    elif k <= 256:
        per_rating, per_row = 5.4 * tiles, 0.0153 * float(k) ** 3
    else:
        per_rating, per_row = 40.0 * tiles, 0.1 * float(k) ** 3
    primal = n * per_rating + per_row
    dual = 2700.0 * np.ceil(n / 16.0) ** 1.36 * (k / 100.0)
    c = np.where(n <= dual_max, dual, primal)
    return np.where(n > 0, c, 0.0)


def shard_ranges(counts, world, k=None, double=False):
    """Contiguous row ranges for `world` ranks with equal modelled cost (row_cost; plain rating
    counts when k is None): the greedy cumulative cut of splitToPortions (EmfLord.js:571-592) with
    one portion per GPU.  Returns int64[world+1]."""
    counts = np.asarray(counts, np.int64)
    w = counts.astype(np.float64) if k is None else row_cost(counts, k, double)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    b = [0]
    for r in range(1, world):
        b.append(int(np.searchsorted(cum, total * r / world, side="left")))
    b.append(len(counts))
    b = np.maximum.accumulate(np.asarray(b, np.int64))
    return b


def rebalanced_ranges(counts, bounds, ms_by_rank, k=None, double=False):
    """Row ranges for the iterations that follow, from the time every rank's shard of this side just took.

    The reference hands portions to whichever node asks next (EmfLord.m_incrNextPortion, lib/emf/EmfLord.js:996-1006;
    EmfChief._incrNextPortion, lib/emf/EmfChief.js:308-318), so a slow node simply takes fewer.  Shards here are
    static within a half-step, so the feedback acts between iterations: the modelled cost of the rows of shard r
    is scaled by (measured ms of r) / (modelled cost of r) and the ranges are cut again at equal scaled cost --
    a shard that ran long gives rows away.  Results do not depend on the cuts: every row is solved by one wave /
    workgroup, or summed over chunks whose boundaries follow the row and the ratings of the whole side, not the
    shard (libycnr_als derives the chunk length from the side's total; tests/test_gpu_comm.py re-cuts at a size where
    a per-shard length would differ)."""
    counts = np.asarray(counts, np.int64)
    bounds = np.asarray(bounds, np.int64)
    world = len(bounds) - 1
    ms = np.asarray(ms_by_rank, np.float64)
    w = counts.astype(np.float64) if k is None else row_cost(counts, k, double)
    if len(ms) != world or not np.all(np.isfinite(ms)) or not np.all(ms > 0):
        return bounds.copy()
    w = w.copy()
    for r in range(world):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        c = float(w[lo:hi].sum())
        if c > 0:
            w[lo:hi] *= ms[r] / c
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    b = [0] + [int(np.searchsorted(cum, total * r / world, side="left")) for r in range(1, world)] + [len(counts)]
    return np.maximum.accumulate(np.asarray(b, np.int64))


class Dataset:
    """What prepareToTrain leaves behind in the reference: the split ratings and stats
    (EmfLord.getStats / splitToSets, lib/emf/EmfLord.js:48-250), here given directly.

    train_by_user / train_by_item: Csr of dataset_type IN (1, 2) (EmfMaster.js:502-503);
    validate / test: Csr by user of dataset_type 2 / 3 (may be None)."""

    def __init__(self, train_by_user, train_by_item, validate=None, test=None, total_ratings_avg=None):
        self.train_by_user, self.train_by_item = train_by_user, train_by_item
        self.validate, self.test = validate, test
        self.totalUsersCount = train_by_user.rows
        self.totalItemsCount = train_by_user.cols
        if total_ratings_avg is None:
            v = train_by_user.vals
            total_ratings_avg = float(v.double().mean()) if hasattr(v, "double") else float(np.mean(v, dtype=np.float64))
        self.totalRatingsAvg = total_ratings_avg


class HipBackend:
    """The product compute backend: libycnr_als.so on cuda:<device>.  Factor matrices are torch
    CUDA tensors bound into the handle (torch is the allocator; nothing computes in torch).

    With several ranks the library itself exchanges the solved rows (ycnr_als_comm_init +
    ycnr_als_set_ratings_sharded, RCCL point-to-point over xGMI): step() returns when every replica
    holds every solved row."""

    native_exchange = True

    def __init__(self, opts, users, items, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("ycnr_als needs a HIP device; there is no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.dtype_np = np.float64 if opts["useDoublePrecision"] else np.float32
        tdt = torch.float64 if opts["useDoublePrecision"] else torch.float32
        self.opts, self.users, self.items, self.devno = opts, users, items, device
        k = opts["factorsCount"]
        self.fac = [torch.zeros(users, k, dtype=tdt, device=self.device),
                    torch.zeros(items, k, dtype=tdt, device=self.device)]
        o = self.opts
        self.dev = AlsDevice(o["factorsCount"], self.users, self.items, o["useDoublePrecision"], o["als"]["userFactReg"],
                             o["als"]["itemFactReg"], device=self.devno, chunkRatings=o.get("chunkRatings", 0))
        self.dev.bind_factors(0, self.fac[0])
        self.dev.bind_factors(1, self.fac[1])
        self.world = 1

    # -- communicator ---------------------------------------------------------------------
    def comm_unique_id(self, transport):
        return AlsDevice.comm_unique_id(transport)

    def comm_init(self, unique_id, rank, world, transport):
        self.dev.comm_init(unique_id, rank, world, transport)
        self.world = world

    def comm_destroy(self):
        self.dev.comm_destroy()
        self.world = 1

    def allreduce_sum(self, arr):
        return self.dev.allreduce_sum(arr)

    def broadcast_factors(self, side, root=0):
        self.dev.broadcast_factors(side, root)

    # -- data ------------------------------------------------------------------------------
    def factors(self, side):
        return self.fac[side]

    def set_factors(self, side, arr):
        self.fac[side].copy_(self.torch.from_numpy(np.ascontiguousarray(arr, self.dtype_np)))
        self.torch.cuda.synchronize(self.device)

    def get_factors(self, side):
        return self.fac[side].cpu().numpy()

    def _vals(self, csr):
        return csr.astype(self.dtype_np)

    def set_ratings(self, side, csr, rb, re, bounds=None):
        """bounds: per rank the ascending row ids [begin, cut, ..., end] of its pieces (sharded
        upload for all ranks); None: the plain shard [rb, re) of a single process."""
        c = self._vals(csr)
        if bounds is not None:
            self.dev.set_ratings_sharded(side, c.rowPtr, c.indx, c.vals, bounds)
        else:
            self.dev.set_ratings(side, c.rowPtr, c.indx, c.vals, rb, re)

    def set_ratings_banded(self, side, csr, band_bounds, rank_bands, owner_bounds):
        c = self._vals(csr)
        self.dev.set_ratings_banded(side, c.rowPtr, c.indx, c.vals, band_bounds, rank_bands, owner_bounds)

    def set_ratings_deferred(self, side, csr, bounds):
        """sharded upload whose half-steps leave the solved rows where they are until exchange(side)"""
        c = self._vals(csr)
        self.dev.set_ratings_sharded(side, c.rowPtr, c.indx, c.vals, bounds)
        self.dev.defer_exchange(side, True)

    def exchange(self, side):
        self.torch.cuda.synchronize(self.device)
        self.dev.exchange(side)

    def set_rmse_ratings(self, which, csr, rb, re):
        c = self._vals(csr)
        self.dev.set_rmse_ratings(which, c.rowPtr, c.indx, c.vals, rb, re)

    def step(self, side):
        if os.environ.get("YCNR_TORCH_DEVICE_SYNC"):  # (A/B: the device-wide wait of rounds 1-4)
            self.torch.cuda.synchronize(self.device)
        else:
            # what the host enqueued through torch on its current stream is done (the library solves on a stream of its own).
            # A wait for THAT stream, not torch.cuda.synchronize(): the device-wide wait visits every stream of the process --
            # the library's side streams among them -- and cost 47 us per half-step at the ML-1M shape (0.44 -> 0.35 ms per iteration)
            self.torch.cuda.current_stream(self.device).synchronize()
        return self.dev.step(side)

    def rmse(self, which, shift, portion_row_end):
        self.torch.cuda.current_stream(self.device).synchronize()  # (as step())
        return self.dev.rmse(which, shift, portion_row_end)

    def destroy(self):
        self.dev.destroy()


class EmfLord:
    """train() / alsTrainIter() / alsTrainStep() / calcRmse() of lib/emf/EmfLord.js:864-1081
    and the result files of lib/emf/EmfManager.js:158-176,463-570, driving one GPU per process."""

    STEP_SIDE = {"byUser": 0, "byItem": 1}

    def __init__(self, config=None, options=None, backend_factory=None, dist=None):
        config = config or {}
        self.options = deepmerge(default_options(), config.get("common"), config.get("emf"), options)
        self._status = ""
        self.globalAvgShift = 0
        self.globalBias = 0
        self.calcCnt = 0
        self.calcDate = None
        self.trainIter = 0
        self.rmse = None
        self.predAvg = None
        self.history = []
        self.stepTimes = []
        self._backend_factory = backend_factory or (lambda o, u, i, d: HipBackend(o, u, i, d))
        self._dist = dist  # torch.distributed module when initialised, else None
        self.rank = dist.get_rank() if dist else 0
        self.world = dist.get_world_size() if dist else 1
        self.backend = None
        self.userFactorsFilename = "user_factors"  # EmfBase.js:289-293
        self.itemFactorsFilename = "item_factors"
        self.calcInfoFilename = "calc_info.json"

    # -- EmfBase getters ---------------------------------------------------------------
    @property
    def status(self):
        return self._status

    @property
    def factorsCount(self):
        return self.options["factorsCount"]

    @property
    def factorsReadyPath(self):
        return os.path.join(self.options["dataDir"], self.options["dbType"] + "_factors_ready")

    @property
    def factorsTempPath(self):
        return os.path.join(self.options["dataDir"], self.options["dbType"] + "_factors_tmp")

    # -- prepareToTrain (EmfLord.js:617-653, without the db) ---------------------------
    def prepareToTrain(self, dataset, userFactors=None, itemFactors=None, seed=1, device=0, shards=None):
        """Upload the ratings (sharded by row over the ranks), build the RMSE portions and
        create / load the factor matrices (prepareSharedFactors, EmfMaster.js:347-358)."""
        if self.options["alg"] != "als":
            raise ValueError("only alg='als' is implemented (sgd is obsolete in the reference, README.md:13)")
        self._status = "preparing"
        ds = self.dataset = dataset
        self.totalUsersCount, self.totalItemsCount = ds.totalUsersCount, ds.totalItemsCount
        self.totalRatingsAvg = ds.totalRatingsAvg
        self.calcDate = datetime.datetime.now(datetime.timezone.utc).isoformat()
        cu = _to_np(ds.train_by_user.counts())
        ci = _to_np(ds.train_by_item.counts())
        self.ratingsCntPerUser, self.ratingsCntPerItem = cu, ci
        if self.options.get("dropLastRatingPerPortion", False):
            ds = self.dataset = self._with_reference_portion_quirk(ds, cu, ci)
        k, dbl = self.factorsCount, self.options["useDoublePrecision"]
        # shards: {side: int64[world + 1]} overrides the cost-model cut (bench.py --emulate-world replays measured cuts)
        self.shards = shards or {0: shard_ranges(cu, self.world, k, dbl), 1: shard_ranges(ci, self.world, k, dbl)}
        # the item half-step sharded by user bands: 8 cost-balanced bands of users whatever the world size; rank r holds
        # 8 / world consecutive bands = its user shard
        self.bands = None
        if self.options.get("itemStepSharding", "rows") == "bands":
            if self.world not in (1, 2, 4, 8):
                raise ValueError("itemStepSharding='bands' needs 1, 2, 4 or 8 ranks (8 user bands), got %d" % self.world)
            self.bands = shard_ranges(cu, 8, k, dbl)
            self.rankBands = np.arange(self.world + 1, dtype=np.int64) * (8 // self.world)
            self.shards = {0: self.bands[self.rankBands], 1: self.shards[1]}
        self._itersRun, self.rebalanced = 0, None
        self.backend = self._backend_factory(self.options, self.totalUsersCount, self.totalItemsCount, device)
        ub, ue = self.shards[0][self.rank], self.shards[0][self.rank + 1]
        ib, ie = self.shards[1][self.rank], self.shards[1][self.rank + 1]
        self.native_exchange = self.world > 1 and getattr(self.backend, "native_exchange", False)
        self.commTransport, self.commFallback = None, []
        if self.native_exchange:
            # the communicator's id travels over the host's control plane (here torch.distributed's
            # store; in the NodeJS host the Lord's process.send), the rows over the library's transport.
            # Unless strictTransport is set, a device-to-device transport that cannot be set up on EVERY rank is followed by the
            # other one (rccl -> ipc, ipc -> rccl) before the host falls back to torch.distributed: a run on a node nobody has
            # seen before should produce a number, and exchangePath / commFallback say over which path and why.
            wanted = self.options.get("commTransport", "rccl")
            strict = bool(self.options.get("strictTransport", False))
            candidates = [wanted] + ([t for t in ("rccl", "ipc") if t != wanted] if (not strict and wanted in ("rccl", "ipc")) else [])
            for transport in candidates:
                ok, why = 1, ""
                try:
                    box = [self.backend.comm_unique_id(transport) if self.rank == 0 else None]
                except Exception as e:  # noqa: BLE001 -- reported below, collectively
                    box, ok, why = [None], 0, str(e)
                self._dist.broadcast_object_list(box, src=0)
                inited = False
                if box[0] is not None:
                    try:
                        self.backend.comm_init(box[0], self.rank, self.world, transport)
                        inited = True
                    except Exception as e:  # noqa: BLE001
                        ok, why = 0, str(e)
                else:
                    ok = 0
                # all ranks must agree on the path: one failed rank sends everybody on
                flags = [None] * self.world
                self._dist.all_gather_object(flags, (ok, why))
                if all(f[0] for f in flags):
                    self.commTransport = transport
                    break
                why = "; ".join("rank %d: %s" % (r, f[1]) for r, f in enumerate(flags) if not f[0])
                self.commFallback.append("%s: %s" % (transport, why))
                if inited and hasattr(self.backend, "comm_destroy"):
                    self.backend.comm_destroy()
            if self.commTransport is None:
                why = " | ".join(self.commFallback)
                if strict:
                    # a measurement must not silently run over another path than the one it names (bench.py --strict-transport)
                    raise RuntimeError("exchange transport '%s' unavailable (%s)" % (wanted, why))
                import warnings
                warnings.warn("native exchange unavailable (%s); using torch.distributed all-gather" % why)
                self.native_exchange = False
            elif self.commFallback:
                import warnings
                warnings.warn("exchange over '%s' instead of '%s' (%s)" % (self.commTransport, wanted, " | ".join(self.commFallback)))
        self.exchangePath = ("libycnr_als:" + self.commTransport) if self.native_exchange else \
            ("torch.distributed" if self.world > 1 else "none")
        if self.bands is not None and not hasattr(self.backend, "set_ratings_banded"):
            raise RuntimeError("itemStepSharding='bands' needs a backend with set_ratings_banded (the HIP backend)")
        if self.bands is not None and self.world > 1 and not self.native_exchange:
            raise RuntimeError("itemStepSharding='bands' needs the library's own exchange (commTransport rccl or ipc)")
        self._upload_shards((0, 1))
        self.trainRatingsCount = int(cu.sum())
        # portions of the RMSE passes (EmfLord.js:523-598); kept as exclusive 0-based row ends
        self.portionsRowIdTo = {}
        nthreads = self.options["numThreadsForTrain"]["als"]
        distr = self.options["dataSetDistr"]
        for name, csr, pct in (("rmseValidate", ds.validate, distr[1] + 1), ("rmseTest", ds.test, distr[2] + 1)):
            if csr is None:
                continue
            self.backend.set_rmse_ratings(name, csr, int(ub), int(ue))
            # rows_cnt = trainUsersCount: the users that have ratings (EmfLord.js:526), not the max id
            ends, _, _ = split_to_portions(cu, int((cu > 0).sum()), self.options["ratingsInPortionForRmse"],
                                           nthreads, pct)
            if len(ends):
                ends[-1] = self.totalUsersCount
            self.portionsRowIdTo[name] = ends
        from .data import init_factors
        dt = np.float64 if self.options["useDoublePrecision"] else np.float32
        k = self.factorsCount
        # prepareSharedFactors (EmfMaster.js:347-358) over _loadSharedFactorsForTrain
        # (EmfManager.js:405-457): explicit matrices win; else reuse the ready files when they
        # are compatible, keeping the old rows and drawing random rows only for users / items
        # added since (initSharedFactorsRandom(oldUsersCnt, oldItemsCnt), EmfBase.js:457-513);
        # else draw everything
        self.recreated, self.extended = True, False
        if userFactors is None and itemFactors is None and self.options.get("warmStart", False):
            prev = self.loadCalcResults()
            if prev is not None and prev[1].shape[0] <= self.totalUsersCount and prev[2].shape[0] <= self.totalItemsCount \
                    and prev[1].shape[0] == prev[0]["totalUsersCount"] and prev[2].shape[0] == prev[0]["totalItemsCount"]:
                ci_prev, U0, V0 = prev
                userFactors = init_factors(self.totalUsersCount, k, seed * 2 + 0, dt)
                itemFactors = init_factors(self.totalItemsCount, k, seed * 2 + 1, dt)
                userFactors[:U0.shape[0]] = U0
                itemFactors[:V0.shape[0]] = V0
                self.calcCnt = int(ci_prev.get("calcCnt") or 0)
                self.globalAvgShift = ci_prev.get("globalAvgShift") or 0
                self.recreated = False
                self.extended = U0.shape[0] != self.totalUsersCount or V0.shape[0] != self.totalItemsCount
        drawn = (0 if userFactors is None else None, 0 if itemFactors is None else None)  # first row drawn here, per side
        if not self.recreated:
            drawn = (U0.shape[0], V0.shape[0])
        if userFactors is None:
            userFactors = init_factors(self.totalUsersCount, k, seed * 2 + 0, dt)
        if itemFactors is None:
            itemFactors = init_factors(self.totalItemsCount, k, seed * 2 + 1, dt)
        if self.options["als"].get("initFirstFactorAsAvgRating", False):
            # EmfBase.js:493-511: the first factor of every row drawn here = the row's average rating
            # (ratings_count / avg_rating over the train sets), rows without ratings keep the draw
            for fac, csr, first in ((userFactors, ds.train_by_user, drawn[0]), (itemFactors, ds.train_by_item, drawn[1])):
                if first is None:
                    continue
                rp, vals = _to_np(csr.rowPtr), _to_np(csr.vals).astype(np.float64)
                cnt = np.diff(rp)
                sums = np.add.reduceat(np.concatenate([vals, [0.0]]), np.minimum(rp[:-1], len(vals))) * (cnt > 0)
                rows = np.flatnonzero(cnt > 0)
                rows = rows[rows >= first]
                fac[rows, 0] = (sums[rows] / cnt[rows]).astype(dt)
        self.backend.set_factors(0, userFactors)
        self.backend.set_factors(1, itemFactors)
        self._status = "ready"

    def _with_reference_portion_quirk(self, ds, cu, ci):
        """options.dropLastRatingPerPortion: the data set as the reference's workers see it -- every pass cut into the portions
        of splitToPortions (stats of the full data, as the reference's come from the db), each portion without its last rating."""
        from .data import drop_last_rating_per_portion
        o = self.options
        nthreads, distr = o["numThreadsForTrain"]["als"], o["dataSetDistr"]
        per = o["ratingsInPortionForAls"]
        eu, _, _ = split_to_portions(cu, int((cu > 0).sum()), per["byUser"], nthreads)
        ei, _, _ = split_to_portions(ci, int((ci > 0).sum()), per["byItem"], nthreads)
        out = Dataset(drop_last_rating_per_portion(ds.train_by_user, eu), drop_last_rating_per_portion(ds.train_by_item, ei),
                      ds.validate, ds.test, ds.totalRatingsAvg)
        for name, pct in (("validate", distr[1] + 1), ("test", distr[2] + 1)):
            csr = getattr(ds, name)
            if csr is not None:
                ends, _, _ = split_to_portions(cu, int((cu > 0).sum()), o["ratingsInPortionForRmse"], nthreads, pct)
                if len(ends):
                    ends[-1] = ds.totalUsersCount
                setattr(out, name, drop_last_rating_per_portion(csr, ends))
        return out

    def _upload_shards(self, sides):
        """(Re)upload the train ratings of `sides` for the current self.shards: this rank's rows, in pieces when
        the side's matrix is large (the exchange of one piece overlaps with the solve of the next)."""
        ds, k, dbl = self.dataset, self.factorsCount, self.options["useDoublePrecision"]
        cnts = {0: self.ratingsCntPerUser, 1: self.ratingsCntPerItem}
        csrs = {0: ds.train_by_user, 1: ds.train_by_item}
        rows = {0: self.totalUsersCount, 1: self.totalItemsCount}
        if self.bands is not None:
            b0 = self.shards[0]
            if 0 in sides:
                # the users as plain row shards whose half-steps exchange nothing: nobody reads a user row outside its band
                # until the training ends (finishExchange)
                if self.world > 1:
                    self.backend.set_ratings_deferred(0, csrs[0], np.stack([b0[r:r + 2] for r in range(self.world)]))
                else:
                    self.backend.set_ratings(0, csrs[0], 0, rows[0])
            if 1 in sides:
                lo, hi = int(b0[self.rank]), int(b0[self.rank + 1])
                self.backend.set_ratings_banded(1, _columns_between(csrs[1], lo, hi), self.bands, self.rankBands, self.shards[1])
            return
        if self.native_exchange:
            s = 8 if dbl else 4
            self.pieceBounds = getattr(self, "pieceBounds", {})
            for side in sides:
                nch = int(self.options.get("exchangeChunks", 4)) if rows[side] * k * s / self.world >= (8 << 20) else 1
                b = self.shards[side]
                self.pieceBounds[side] = np.stack([b[r] + shard_ranges(cnts[side][b[r]:b[r + 1]], max(nch, 1), k, dbl)
                                                   for r in range(self.world)])
                self.backend.set_ratings(side, csrs[side], int(b[self.rank]), int(b[self.rank + 1]), bounds=self.pieceBounds[side])
        else:
            for side in sides:
                b = self.shards[side]
                self.backend.set_ratings(side, csrs[side], int(b[self.rank]), int(b[self.rank + 1]))

    def _gather_ms(self, ms):
        """ms of this rank -> float64[world] on every rank, over whatever carries the RMSE partial sums"""
        v = np.zeros(self.world, np.float64)
        v[self.rank] = ms
        if self.native_exchange:
            return self.backend.allreduce_sum(v)
        torch = _torch()
        dev = self.backend.factors(0).device if hasattr(self.backend, "factors") else "cpu"
        t = torch.from_numpy(v).to(dev)
        self._dist.all_reduce(t)
        return t.cpu().numpy()

    def finishExchange(self):
        """itemStepSharding='bands': bring every rank's copy of the USER matrix up to date (the half-steps never all-gather it).
        Collective; a no-op in the other modes."""
        if self.bands is not None and self.world > 1:
            self.backend.exchange(0)

    def rebalance(self):
        """Cut both sides' row shards again from the compute time every rank measured in the last iteration
        (rebalanced_ranges) and upload the ratings for the new cuts.  Collective.  Returns {side: new bounds}."""
        if self.bands is not None:
            return {}  # (the bands are the shards: fixed for every world size)
        last = {}
        for st in self.stepTimes[::-1]:
            if st["stepType"] not in last:
                last[st["stepType"]] = st
            if len(last) == 2:
                break
        k, dbl = self.factorsCount, self.options["useDoublePrecision"]
        cnts = {0: self.ratingsCntPerUser, 1: self.ratingsCntPerItem}
        out, changed = {}, []
        for name, side in self.STEP_SIDE.items():
            if name not in last:
                continue
            info = last[name]["info"]
            ms = self._gather_ms(float(getattr(info, "totalMs", 0.0) or last[name]["wall"] * 1e3))
            nb = rebalanced_ranges(cnts[side], self.shards[side], ms, k, dbl)
            out[side] = {"ms_by_rank": ms.tolist(), "bounds": nb.tolist()}
            # leave a side alone when it is within 3 % of balanced already: a re-upload is not free
            if ms.max() > 1.03 * ms.mean() and not np.array_equal(nb, self.shards[side]):
                self.shards[side] = nb
                changed.append(side)
        if changed:
            self._upload_shards(changed)
        self.rebalanced = out
        return out

    # -- training ----------------------------------------------------------------------
    def getCanTrainError(self):
        if self._status == "training":
            return "Training is already in progress"
        if self._status != "ready":
            return "Not ready to train. Status is " + self._status
        return None

    def train(self):
        """EmfLord.train (lib/emf/EmfLord.js:864-926): trainIters x (alsTrainIter, rmseValidate,
        rmseTest, rmseTest with shift), then saveCalcResults."""
        err = self.getCanTrainError()
        if err is not None:
            raise RuntimeError(err)
        self._status = "training"
        self.trainIter = 0
        while self.trainIter < self.options["trainIters"]:
            self.alsTrainIter()
            rec = {"iter": self.trainIter}
            for name, shift in (("rmseValidate", False), ("rmseTest", False), ("rmseTest", True)):
                r = self.calcRmse(name, shift)
                if r is not None:
                    rec[name + ("Shifted" if shift else "")] = r
            rec["globalAvgShift"] = self.globalAvgShift
            self.history.append(rec)
            self.trainIter += 1
            # the reference's open todo "saveCalcResults every iter!" (lib/YcnrController.js:288): a
            # checkpoint a later prepareToTrain(warmStart) can resume from; calcCnt counts finished trains
            if self.options.get("saveCalcResultsEveryIter", False) and self.trainIter < self.options["trainIters"]:
                self.finishExchange()
                if self.rank == 0:
                    self.saveCalcResults(self.getCalcInfo())
        self.calcCnt += 1
        self.finishExchange()
        if self.rank == 0:
            self.saveCalcResults(self.getCalcInfo())
        self._status = "ready"
        return self.history

    def alsTrainIter(self):
        """2 steps - first fix item vectors and calc user vectors, then vice versa (EmfLord.js:954-958)."""
        # (Both half-steps enqueued before the host waits -- AlsDevice.iteration() -- was measured and is not used here: the
        # 55 us turn-around it saves on the small shapes is lost again when the second half-step's graph is enqueued into
        # the hardware queues the first is still using: ML-100k shape 0.125 -> 0.115 ms, ML-1M 0.43 -> 0.48, MAL unchanged.)
        self.alsTrainStep("byUser")
        self.alsTrainStep("byItem")
        self._itersRun = getattr(self, "_itersRun", 0) + 1
        # feedback for the static shards: a re-cut after each of the first rebalanceAfterIters iterations (the second
        # re-cut corrects what the first one's extrapolation got wrong; a side within 3 % of balanced is left alone)
        if self.world > 1 and self._itersRun <= int(self.options.get("rebalanceAfterIters", 0) or 0):
            self.rebalance()

    def alsTrainStep(self, stepType):
        """EmfLord.alsTrainStep (lib/emf/EmfLord.js:963-984): resolves once every row of the side
        has been re-solved everywhere, i.e. after the local step AND the exchange."""
        side = self.STEP_SIDE[stepType]
        t0 = time.perf_counter()
        info = self.backend.step(side)  # with a native communicator this includes the exchange
        if self.world > 1 and not self.native_exchange:
            self._exchange(side)
        self.stepTimes.append({"stepType": stepType, "iter": self.trainIter, "info": info,
                               "wall": time.perf_counter() - t0})
        return info

    def _exchange(self, side):
        """Exchange for backends WITHOUT a native one (the CPU oracle behind the gloo tests): a padded
        all-gather through torch.distributed.  The product backend exchanges inside the library."""
        if self.world == 1:
            return
        dist = self._dist
        b = self.shards[side]
        sizes = (b[1:] - b[:-1]).astype(np.int64)
        mx = int(sizes.max())
        fac = self.backend.factors(side)  # torch tensor [rows, k]
        k = fac.shape[1]
        torch = _torch()
        if not hasattr(self, "_xbuf") or self._xbuf.get(side) is None or self._xbuf[side].shape[1] != mx:
            self._xbuf = getattr(self, "_xbuf", {})
            self._xbuf[side] = torch.zeros(self.world, mx, k, dtype=fac.dtype, device=fac.device)
        buf = self._xbuf[side]
        lo, hi = int(b[self.rank]), int(b[self.rank + 1])
        buf[self.rank, : hi - lo].copy_(fac[lo:hi])
        dist.all_gather_into_tensor(buf.view(-1), buf[self.rank].reshape(-1))
        for r in range(self.world):
            if r == self.rank:
                continue
            lo, hi = int(b[r]), int(b[r + 1])
            if hi > lo:
                fac[lo:hi].copy_(buf[r, : hi - lo])

    # -- RMSE ----------------------------------------------------------------------------
    def calcRmse(self, stepType, useGlobalAvgShift):
        """EmfLord.calcRmse + EmfMaster._startCalcRmse / m_completedPortion
        (lib/emf/EmfLord.js:1043-1081, EmfMaster.js:389-412,757-786)."""
        distr = self.options["dataSetDistr"]
        if distr[1] == 0 and stepType == "rmseValidate":
            return None
        if distr[2] == 0 and stepType == "rmseTest":
            return None
        if stepType not in self.portionsRowIdTo:
            return None
        calcGlobalAvgShift = not useGlobalAvgShift
        if calcGlobalAvgShift:
            self.globalAvgShift = 0
        parts = self.backend.rmse(stepType, self.globalAvgShift, self.portionsRowIdTo[stepType])
        if self.world > 1 and self.native_exchange:
            parts = self.backend.allreduce_sum(parts.reshape(-1)).reshape(parts.shape)  # 'rmseSaveCalcs', EmfMaster.js:726-736
        elif self.world > 1:
            torch = _torch()
            t = torch.from_numpy(parts).to(self.backend.factors(0).device)
            self._dist.all_reduce(t)
            parts = t.cpu().numpy()
        rSumDiff2, rCnt, rSum = parts[:, 0].sum(), parts[:, 1].sum(), parts[:, 2].sum()
        self.rSumDiff2, self.rCnt, self.rSum = float(rSumDiff2), float(rCnt), float(rSum)
        self.rmse = math.sqrt(1.0 * rSumDiff2 / rCnt) if rCnt > 0 else float("nan")
        # reference quirk (EmfMaster.js:779): predAvg comes from the LAST completed portion's
        # sums, not the totals.  Portions complete in any order there; here "last" is the
        # highest-numbered non-empty portion.
        nz = np.nonzero(parts[:, 1] > 0)[0]
        if len(nz):
            last = parts[nz[-1]]
            self.predAvg = float(last[2] / last[1])
            if calcGlobalAvgShift:
                self.globalAvgShift = self.totalRatingsAvg - self.predAvg
        return self.rmse

    # -- results (EmfManager.js:158-176,463-570) ---------------------------------------------
    def getCalcInfo(self):
        o = self.options
        return {
            "alg": o["alg"],
            "algOptions": o[o["alg"]],
            "useDoublePrecision": o["useDoublePrecision"],
            "factorsCount": self.factorsCount,
            "dataSetDistr": o["dataSetDistr"],
            "totalUsersCount": self.totalUsersCount,
            "totalItemsCount": self.totalItemsCount,
            "dbType": o["dbType"],
            "calcDate": self.calcDate,
            "calcCnt": self.calcCnt,
            "globalAvgShift": self.globalAvgShift,
            "globalBias": self.globalBias,
        }

    def saveCalcResults(self, calcInfo):
        """Headerless raw dumps user_factors / item_factors + calc_info.json, written to
        <dbType>_factors_tmp and renamed to <dbType>_factors_ready
        (EmfManager._saveCalcResultsToRecommender, lib/emf/EmfManager.js:531-568)."""
        tmp, ready = self.factorsTempPath, self.factorsReadyPath
        os.makedirs(tmp, exist_ok=True)
        self.backend.get_factors(0).tofile(os.path.join(tmp, self.userFactorsFilename))
        self.backend.get_factors(1).tofile(os.path.join(tmp, self.itemFactorsFilename))
        with open(os.path.join(tmp, self.calcInfoFilename), "w") as f:
            f.write(json.dumps(calcInfo, indent=2))
        if os.path.isdir(ready):
            shutil.rmtree(ready)
        os.rename(tmp, ready)

    def loadCalcResults(self):
        """(calc_info, userFactors, itemFactors) from <dbType>_factors_ready, or None when the
        files cannot be reused (EmfManager._canReuseCalcResults, lib/emf/EmfManager.js:179-191)."""
        p = os.path.join(self.factorsReadyPath, self.calcInfoFilename)
        if not os.path.exists(p):
            return None
        ci = json.load(open(p))
        o = self.options
        if not (ci.get("alg") == o["alg"] and ci.get("dbType") == o["dbType"]
                and ci.get("factorsCount") == self.factorsCount
                and ci.get("useDoublePrecision") == o["useDoublePrecision"]):
            return None
        dt = np.float64 if ci["useDoublePrecision"] else np.float32
        k = ci["factorsCount"]
        U = np.fromfile(os.path.join(self.factorsReadyPath, self.userFactorsFilename), dt).reshape(-1, k)
        V = np.fromfile(os.path.join(self.factorsReadyPath, self.itemFactorsFilename), dt).reshape(-1, k)
        return ci, U, V

    def destroy(self):
        if self.backend is not None:
            self.backend.destroy()
            self.backend = None
        self._status = "destroyed"


def _to_np(x):
    return x if isinstance(x, np.ndarray) else x.cpu().numpy()


def _columns_between(csr, lo, hi):
    """The ratings of a Csr (numpy or torch) whose column id lies in [lo, hi): same shape, fewer ratings."""
    if isinstance(csr.indx, np.ndarray):
        mask = (csr.indx >= lo) & (csr.indx < hi)
        rows_of = np.repeat(np.arange(csr.rows), np.diff(csr.rowPtr))
        rp = np.zeros(csr.rows + 1, np.int64)
        rp[1:] = np.cumsum(np.bincount(rows_of[mask], minlength=csr.rows))
        return Csr(csr.rows, csr.cols, rp, np.ascontiguousarray(csr.indx[mask]), np.ascontiguousarray(csr.vals[mask]))
    from .data import select_csr
    return select_csr(csr, (csr.indx >= lo) & (csr.indx < hi))


def _torch():
    import torch
    return torch
