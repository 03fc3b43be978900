/**
 * Step execution: where the reference forks workers, fetches portions from PostgreSQL and
 * collects 'completedPortion' messages (lib/emf/EmfMaster.js), this class hands a whole
 * half-step or RMSE pass to the GPU and applies the same reductions.
 */
'use strict';

const EmfManager = require('./EmfManager');
const { als } = require('./EmfBase');

class EmfMaster extends EmfManager {
  constructor() {
    super();
    this.workType = null;
    this.stepType = null;
    this.trainIter = 0;
    this.lastStepInfo = null;
    this.lastStepInfoBySide = {};
    this.itersRun = 0;
    this.rebalanced = null;
  }

  /** prepareSharedFactors (EmfMaster.js:347-358) */
  prepareSharedFactors() {
    const [recreated, extended] = this._loadSharedFactorsForTrain();
    if (recreated) {
      this.initSharedFactorsRandom(0, 0);
    } else if (extended) {
      this.initSharedFactorsRandom(this.lastCalcInfo.totalUsersCount, this.lastCalcInfo.totalItemsCount);
    }
    return Promise.resolve();
  }

  /**
   * Modelled cost of re-solving a row with n ratings (the constants of python/ycnr_als/emf.py row_cost,
   * DESIGN.md 6): shards are cut so that they finish together, not so that they hold equal ratings.
   */
  static rowCost(n, k, double) {
    if (n <= 0) return 0;
    if (!double) k = Math.ceil(k / 4) * 4;  // float32 sizes that are not multiples of 4 run padded to the next one (the library's kPad)
    const nb = Math.ceil(k / 16);
    const tiles = nb * (nb + 1) / 2.0;
    const dualMax = double ? 0 : 16 * Math.min(k > 128 ? 12 : 5, nb - 1);
    if (n <= dualMax) return 2700.0 * Math.pow(Math.ceil(n / 16), 1.36) * (k / 100);
    const edge4 = !double && k <= 128 && nb >= 2 && k % 16 == 4;
    const nbs = edge4 ? nb - 1 : nb;
    const mfmas = 4.0 * (nbs * (nbs - 1) / 2.0 + (nbs - 1) * nbs * (nbs + 1) / 6.0) + (edge4 ? nbs * (nbs + 1) / 2.0 : 0.0);
    let perRating, perRow;
    if (double) {
      perRating = k > 128 ? 40.0 * tiles : 16.0 * tiles;
      perRow = k <= 128 ? 2.0 * (1500.0 * nbs + 35.0 * mfmas) : 0.1 * k * k * k;
    } else if (k <= 128) {
      perRating = 4.9 * (tiles - (edge4 ? nb / 2.0 : 0.0));
      perRow = 1500.0 * nbs + 35.0 * mfmas;
    } else if (k <= 256) {
      perRating = 5.4 * tiles;
      perRow = 0.0153 * k * k * k;
    } else {
      perRating = 40.0 * tiles;
      perRow = 0.1 * k * k * k;
    }
    return n * perRating + perRow;
  }

  /**
   * parts + 1 ascending row ids cutting rows [lo0, hi0) into contiguous ranges of equal modelled cost: the
   * greedy cumulative cut of splitToPortions (lib/emf/EmfLord.js:571-592) with one portion per GPU (or per
   * pipelined piece of a GPU's shard).  Same rule as shard_ranges of the Python mirror.
   */
  static shardRanges(rowPtr, lo0, hi0, parts, k, double) {
    const n = hi0 - lo0, cum = new Float64Array(n + 1);
    for (let i = 0; i < n; i++) cum[i + 1] = cum[i] + EmfMaster.rowCost(rowPtr[lo0 + i + 1] - rowPtr[lo0 + i], k, double);
    const b = [lo0];
    for (let p = 1; p < parts; p++) {
      const target = cum[n] * p / parts;
      let lo = 0, hi = n + 1;  // first index whose cumulative cost reaches the target
      while (lo < hi) {
        const mid = (lo + hi) >> 1;
        if (cum[mid] < target) lo = mid + 1; else hi = mid;
      }
      b.push(Math.max(b[b.length - 1], lo0 + Math.min(lo, n)));
    }
    b.push(hi0);
    return b;
  }

  /**
   * Row ranges for the iterations that follow, from the time every rank's shard just took (same rule as
   * rebalanced_ranges of the Python mirror): the modelled cost of the rows of shard r is scaled by
   * ms[r] / (modelled cost of r) and the ranges are cut again at equal scaled cost.  The feedback that stands in
   * for the reference's work-stealing portion dispenser (EmfLord.m_incrNextPortion, lib/emf/EmfLord.js:996-1006).
   */
  static rebalancedRanges(rowPtr, bounds, ms, k, double) {
    const world = bounds.length - 1, rows = bounds[world];
    for (let r = 0; r < world; r++) if (!(ms[r] > 0) || !isFinite(ms[r])) return bounds.slice();
    const cum = new Float64Array(rows + 1);
    for (let r = 0; r < world; r++) {
      let c = 0;
      for (let i = bounds[r]; i < bounds[r + 1]; i++) c += EmfMaster.rowCost(rowPtr[i + 1] - rowPtr[i], k, double);
      const scale = c > 0 ? ms[r] / c : 1;
      for (let i = bounds[r]; i < bounds[r + 1]; i++) cum[i + 1] = cum[i] + EmfMaster.rowCost(rowPtr[i + 1] - rowPtr[i], k, double) * scale;
    }
    const b = [0];
    for (let p = 1; p < world; p++) {
      const target = cum[rows] * p / world;
      let lo = 0, hi = rows + 1;
      while (lo < hi) {
        const mid = (lo + hi) >> 1;
        if (cum[mid] < target) lo = mid + 1; else hi = mid;
      }
      b.push(Math.max(b[b.length - 1], Math.min(lo, rows)));
    }
    b.push(rows);
    return b;
  }

  /** sharded upload of one side for the row ranges `shards` (world + 1 ascending ids), in pipelined pieces */
  _uploadSharded(side, csr, rows, shards) {
    const n = als.native, o = this.options, k = this.factorsCount, dbl = o.useDoublePrecision, s1 = this.TypedArraySize1;
    const nch = rows * k * s1 / o.world >= (8 << 20) ? Math.max(1, o.exchangeChunks) : 1;
    const bounds = new Float64Array(o.world * (nch + 1));
    for (let r = 0; r < o.world; r++)
      bounds.set(EmfMaster.shardRanges(csr.rowPtr, shards[r], shards[r + 1], nch, k, dbl), r * (nch + 1));
    n.setRatingsSharded(this.handle, side, csr.rowPtr, csr.indx, csr.vals, nch, bounds);
  }

  /**
   * Cut both sides' shards again from the compute time every rank measured in its last half-steps and upload
   * the ratings for the new cuts.  Collective (every per-GPU process calls it after the same iteration).
   */
  /** itemStepSharding 'bands': bring every replica of the USER matrix up to date (the half-steps never all-gather it). Collective. */
  finishExchange() {
    if (this.bands && this.options.world > 1) als.native.exchange(this.handle, als.BY_USER);
  }

  rebalance() {
    const n = als.native, o = this.options, ds = this.dataset;
    const out = {};
    if (this.bands) return out;  // (the bands are the shards: fixed for every world size)
    for (const [name, side, csr, rows] of [['byUser', als.BY_USER, ds.trainByUser, this.totalUsersCount],
      ['byItem', als.BY_ITEM, ds.trainByItem, this.totalItemsCount]]) {
      const info = this.lastStepInfoBySide[name];
      if (!info) continue;
      const ms = new Float64Array(o.world);
      ms[o.rank] = info.totalMs;
      n.allreduceSum(this.handle, ms);
      const nb = EmfMaster.rebalancedRanges(csr.rowPtr, this.shards[side], ms, this.factorsCount, o.useDoublePrecision);
      let mx = 0, mean = 0;
      for (let r = 0; r < o.world; r++) { mx = Math.max(mx, ms[r]); mean += ms[r] / o.world; }
      out[name] = { msByRank: Array.from(ms), bounds: nb };
      if (mx > 1.03 * mean && nb.some((v, i) => v != this.shards[side][i])) {
        this.shards[side] = nb;
        this._uploadSharded(side, csr, rows, nb);
      }
    }
    this.rebalanced = out;
    return out;
  }

  /**
   * itemStepSharding 'bands' (any world of 1, 2, 4, 8): the users in 8 cost-balanced bands, rank r holding 8 / world of them; the user
   * side as row shards whose half-steps exchange nothing, the item side through ycnr_als_set_ratings_banded.
   */
  _uploadBanded() {
    const ds = this.dataset, n = als.native, o = this.options, k = this.factorsCount, dbl = o.useDoublePrecision;
    if (![1, 2, 4, 8].includes(o.world)) throw new Error("itemStepSharding 'bands' needs 1, 2, 4 or 8 GPUs (8 user bands)");
    const bands = EmfMaster.shardRanges(ds.trainByUser.rowPtr, 0, this.totalUsersCount, 8, k, dbl);
    const rankBands = [], su = [];
    for (let r = 0; r <= o.world; r++) { rankBands.push(r * (8 / o.world)); su.push(bands[r * (8 / o.world)]); }
    const owners = EmfMaster.shardRanges(ds.trainByItem.rowPtr, 0, this.totalItemsCount, o.world, k, dbl);
    this.shards = {};
    this.shards[als.BY_USER] = su;
    this.shards[als.BY_ITEM] = owners;
    this.bands = bands;
    if (o.world > 1) {
      // nobody reads a user row outside its band until finishExchange()
      const ub = new Float64Array(o.world * 2);
      for (let r = 0; r < o.world; r++) { ub[2 * r] = su[r]; ub[2 * r + 1] = su[r + 1]; }
      n.setRatingsSharded(this.handle, als.BY_USER, ds.trainByUser.rowPtr, ds.trainByUser.indx, ds.trainByUser.vals, 1, ub);
      n.deferExchange(this.handle, als.BY_USER, 1);
    } else {
      n.setRatings(this.handle, als.BY_USER, ds.trainByUser.rowPtr, ds.trainByUser.indx, ds.trainByUser.vals);
    }
    // the items: every row, only this rank's users' ratings
    const sub = o.world > 1 ? ds.trainByItem.columnsBetween(su[o.rank], su[o.rank + 1]) : ds.trainByItem;
    n.setRatingsBanded(this.handle, als.BY_ITEM, sub.rowPtr, sub.indx, sub.vals, Float64Array.from(bands), Float64Array.from(rankBands), Float64Array.from(owners));
    this.shardUsers = [su[o.rank], su[o.rank + 1]];
  }

  /** Upload the ratings once: replaces createWorkPortionBuffers + per-portion fetches (EmfMaster.js:156-234,501-614) */
  prepareWorkersToTrain() {
    const n = als.native, o = this.options;
    if (o.dropLastRatingPerPortion && !this.dataset.portionQuirkApplied) {
      // opt-in (SURVEY.md 8f N2): the ratings the REFERENCE's packer hands to its workers -- every portion of every pass without
      // its last rating (lib/emf/EmfMaster.js:594-603); the stats stay those of the full data, as the reference's come from the db
      this.dataset = this.dataset.withoutLastRatingPerPortion(this.portionsRowIdTo);
    }
    const ds = this.dataset;
    this.shardUsers = [0, this.totalUsersCount];
    if (o.itemStepSharding == 'bands') {
      if (o.world > 1) n.commInit(this.handle, als.commTransport[o.commTransport], Uint8Array.from(Buffer.from(o.commId, 'base64')), o.rank, o.world);
      this._uploadBanded();
    } else if (o.world > 1) {
      // one process per GPU: this rank's place in the exchange, then the sharded upload of both sides
      n.commInit(this.handle, als.commTransport[o.commTransport], Uint8Array.from(Buffer.from(o.commId, 'base64')), o.rank, o.world);
      const k = this.factorsCount, dbl = o.useDoublePrecision;
      this.shards = {};
      const sharded = (side, csr, rows) => {
        const shards = EmfMaster.shardRanges(csr.rowPtr, 0, rows, o.world, k, dbl);
        this.shards[side] = shards;
        this._uploadSharded(side, csr, rows, shards);
        return shards;
      };
      const su = sharded(als.BY_USER, ds.trainByUser, this.totalUsersCount);
      sharded(als.BY_ITEM, ds.trainByItem, this.totalItemsCount);
      // the RMSE sets keep this cut of the users whatever rebalance() does later: their partial sums are added
      // over the ranks, any tiling of the rows will do
      this.shardUsers = [su[o.rank], su[o.rank + 1]];
    } else {
      n.setRatings(this.handle, als.BY_USER, ds.trainByUser.rowPtr, ds.trainByUser.indx, ds.trainByUser.vals);
      n.setRatings(this.handle, als.BY_ITEM, ds.trainByItem.rowPtr, ds.trainByItem.indx, ds.trainByItem.vals);
    }
    const [ub, ue] = this.shardUsers;
    if (ds.validate) n.setRmseRatings(this.handle, als.RMSE_VALIDATE, ds.validate.rowPtr, ds.validate.indx, ds.validate.vals, ub, ue);
    if (ds.test) n.setRmseRatings(this.handle, als.RMSE_TEST, ds.test.rowPtr, ds.test.indx, ds.test.vals, ub, ue);
    return Promise.resolve();
  }

  /** _startAlsTrainStep (EmfMaster.js:364-383): the whole step is one native call */
  _startAlsTrainStep(stepType) {
    this.workType = 'train';
    this.stepType = stepType;
    this.lastStepInfo = als.native.step(this.handle, als.stepSide[stepType]);
    this.lastStepInfoBySide[stepType] = this.lastStepInfo;
    this.emit('stepComplete');
  }

  /**
   * _startCalcRmse + m_completedPortion (EmfMaster.js:389-412,757-786): per-portion partial
   * sums come back from the GPU and are reduced exactly like the 'completedPortion' messages,
   * including predAvg = LAST portion's rSum / rCnt (EmfMaster.js:779).
   */
  _startCalcRmse(stepType, useGlobalAvgShift) {
    useGlobalAvgShift = useGlobalAvgShift && this.options.alg == 'als';
    this.calcGlobalAvgShift = !useGlobalAvgShift;
    this.workType = 'rmse';
    this.stepType = stepType;
    this.globalAvgShift = this.calcGlobalAvgShift ? 0 : this.globalAvgShift;
    const ends = Float64Array.from(this.portionsRowIdTo[stepType]);
    const parts = als.native.rmse(this.handle, als.rmseSet[stepType], this.globalAvgShift, ends);
    // 'rmseSaveCalcs' (EmfMaster.js:726-736): every rank's partial sums of every portion, summed
    if (this.options.world > 1) als.native.allreduceSum(this.handle, parts);
    this.rSum = 0; this.rSumDiff2 = 0; this.rCnt = 0;
    let last = -1;
    for (let p = 0; p < ends.length; p++) {
      this.rSumDiff2 += parts[3 * p];
      this.rCnt += parts[3 * p + 1];
      this.rSum += parts[3 * p + 2];
      if (parts[3 * p + 1] > 0) last = p;
    }
    this.rmse = Math.sqrt(1.0 * this.rSumDiff2 / this.rCnt);
    if (last >= 0) {
      this.predAvg = parts[3 * last + 2] / parts[3 * last + 1];
      if (this.calcGlobalAvgShift) {
        this.globalAvgShift = this.stats.totalRatingsAvg - this.predAvg;
      }
    }
    this.emit('stepComplete');
  }
}

module.exports = EmfMaster;
