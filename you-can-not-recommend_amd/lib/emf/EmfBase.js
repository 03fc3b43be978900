/**
 * Base of the trainer classes: options, stats, factor matrices, predict.
 *
 * Mirrors lib/emf/EmfBase.js of the reference for the ALS path only. The factor matrices
 * live in HBM behind the native handle (the role of the SysV shm arrays of
 * createSharedFactors, EmfBase.js:399-425); host typed arrays are staging copies in the
 * reference's layout: dense row-major [rows x factorsCount].
 */
'use strict';

const os = require('os');
const fs = require('fs');
const path = require('path');
const assert = require('assert');
const EventEmitter = require('events');
const als = require('../ycnr_als');
const { rng } = require('../Dataset');

const numCPUs = os.cpus().length;

function isObject(x) { return x !== null && typeof x === 'object' && !Array.isArray(x) && !ArrayBuffer.isView(x); }

/** deepmerge.all([...]) as used by EmfBase.init (EmfBase.js:284-287) */
function deepmergeAll(list) {
  const out = {};
  for (const d of list) {
    if (!d) continue;
    for (const k of Object.keys(d)) {
      if (isObject(d[k]) && isObject(out[k])) out[k] = deepmergeAll([out[k], d[k]]);
      else if (isObject(d[k])) out[k] = deepmergeAll([d[k]]);
      else if (Array.isArray(d[k])) out[k] = d[k].slice();
      else out[k] = d[k];
    }
  }
  return out;
}

class EmfBase extends EventEmitter {
  /** The ALS-path subset of EmfBase.DefaultOptions (EmfBase.js:52-140), same names. */
  static get DefaultOptions() {
    return {
      dbType: 'ml', //'ml', 'mal'
      maxRating: { mal: 10, ml: 5 },
      als: {
        userFactReg: 0.05,
        itemFactReg: 0.05,
        initFirstFactorAsAvgRating: false,
      },
      factorsCount: 100,
      trainIters: 10,
      saveCalcResultsEveryIter: false, // checkpoint after every iteration (reference todo, lib/YcnrController.js:288)
      alg: 'als',
      dataSetDistr: [85, 10, 5],
      ratingsInPortionForRmse: 10 * 1000,
      ratingsInPortionForAls: { byUser: 10 * 1000, byItem: 10 * 1000 },
      numThreadsForTrain: { als: numCPUs, sgd: 1 },
      numThreadsForRmse: numCPUs,
      useDoublePrecision: false,
      // not in the reference (opt-in): consume the ratings as the reference's portion packer hands them to its workers -- the
      // last rating of every portion dropped (lib/emf/EmfMaster.js:594-603) -- for bug-for-bug replays of a reference run
      dropLastRatingPerPortion: false,
      // not in the reference: HIP device ordinal, where the factor directories live
      // (reference: <repo>/data), GPU work-unit size (0 = library default), seed of the
      // initial factors (reference: unseeded)
      device: 0,
      // multi-GPU (not in the reference, whose cluster is TCP, lib/emf/EmfLord.js:668-747): `gpus` > 1 is what
      // EmfLord.trainOnGpus() forks one process per GPU for (train() itself always drives ONE GPU); inside such a process `rank` / `world` /
      // `commId` / `commTransport` describe its place, and a large side is solved in `exchangeChunks`
      // pieces so that the exchange of one piece overlaps with the solve of the next
      gpus: 1,
      rank: 0,
      world: 1,
      commId: null,
      commTransport: 'rccl', // 'rccl' | 'ipc' (mapped peer replicas + copy engines; also several ranks on one GPU) | 'shm' (host-staged stand-in)
      strictTransport: false, // true: trainOnGpus fails when commTransport cannot be set up in every GPU process (default: rccl <-> ipc are tried in turn)
      exchangeChunks: 4,
      rebalanceAfterIters: 2,   // multi-GPU: cut the row shards again from the measured compute times after each of the first N iterations (0 = never)
      // multi-GPU: 'rows' = items over the ranks, the user matrix all-gathered after every user half-step; 'bands' = the users in 8
      // cost-balanced bands (the same for every world size), every rank accumulates the Gramians of ALL items over its own users'
      // ratings, the per-band sums travel to the items' owners, the user matrix is never all-gathered (ycnr_als_set_ratings_banded;
      // world 1, 2, 4 or 8, transports rccl / ipc; the shards are not re-cut in this mode)
      itemStepSharding: 'rows',
      gpuDevices: 0,            // devices the per-GPU processes are spread over (0 = what the library reports)
      gpuProcessScript: null,   // entry point of a per-GPU process (default lib/emf/EmfGpuProcess.js)
      gpuProcessTimeoutMs: 0,   // trainOnGpus gives up after this long (0 = no limit)
      dataDir: path.join(__dirname, '..', '..', 'data'),
      chunkRatings: 0,
      seed: 1,
    };
  }

  static get InitialStats() {
    return {
      totalUsersCount: -1, totalItemsCount: -1, totalRatingsAvg: 0,
      trainUsersCount: -1, trainItemsCount: -1,
      trainUsersRatingsCount: 0, trainItemsRatingsCount: 0,
      maxRatingsPerUser: 0, maxRatingsPerItem: 0,
      ratingsCntPerUser: [], ratingsAvgPerUser: [], ratingsCntPerItem: [], ratingsAvgPerItem: [],
      ratingsCountTrain: 0, ratingsCountValidate: 0, ratingsCountTest: 0,
      portionsCount: {}, maxRowsInPortion: {}, maxRatingsInPortion: {}, portionsRowIdTo: {},
    };
  }

  get TypedArrayKey() { return this.options.useDoublePrecision ? 'Float64Array' : 'Float32Array'; }
  get TypedArrayClass() { return this.options.useDoublePrecision ? Float64Array : Float32Array; }
  get TypedArraySize1() { return this.options.useDoublePrecision ? 8 : 4; }
  get factorsWorkingPath() { return path.join(this.options.dataDir, this.options.dbType + '_factors_working'); }
  get factorsReadyPath() { return path.join(this.options.dataDir, this.options.dbType + '_factors_ready'); }
  get factorsTempPath() { return path.join(this.options.dataDir, this.options.dbType + '_factors_tmp'); }
  get factorsPath() { return this.factorsReadyPath; }
  get status() { return this._status; }
  get factorsCount() { return this.options.factorsCount; }
  set factorsCount(v) { this.options.factorsCount = v; }
  get maxRating() { return this.options.maxRating[this.options.dbType]; }
  get totalUsersCount() { return this.stats.totalUsersCount; }
  set totalUsersCount(v) { this.stats.totalUsersCount = v; }
  get totalItemsCount() { return this.stats.totalItemsCount; }
  set totalItemsCount(v) { this.stats.totalItemsCount = v; }
  get trainUsersCount() { return this.stats.trainUsersCount; }
  set trainUsersCount(v) { this.stats.trainUsersCount = v; }
  get trainItemsCount() { return this.stats.trainItemsCount; }
  set trainItemsCount(v) { this.stats.trainItemsCount = v; }
  get portionsCount() { return this.stats.portionsCount; }
  get portionsRowIdTo() { return this.stats.portionsRowIdTo; }
  set portionsRowIdTo(v) { this.stats.portionsRowIdTo = v; }
  get maxRowsInPortion() { return this.stats.maxRowsInPortion; }
  get maxRatingsInPortion() { return this.stats.maxRatingsInPortion; }

  constructor() {
    super();
    this._status = '';
    this.stats = Object.assign({}, EmfBase.InitialStats);
    this.userFactors = null; // host staging copies (TypedArray), valid after syncFactorsFromDevice()
    this.itemFactors = null;
    this.handle = null;      // native trainer: owns the device copies
    this.globalAvgShift = 0;
    this.globalBias = 0;
  }

  init(config, options = {}) {
    config = config || {};
    this.options = deepmergeAll([EmfBase.DefaultOptions, config.common, config.emf, options]);
    this.userFactorsFilename = 'user_factors';
    this.itemFactorsFilename = 'item_factors';
    this.calcInfoFilename = 'calc_info.json';
    this.userFactorsPath = path.join(this.factorsPath, this.userFactorsFilename);
    this.itemFactorsPath = path.join(this.factorsPath, this.itemFactorsFilename);
    this.calcInfoPath = path.join(this.factorsPath, this.calcInfoFilename);
    if (this.options.alg != 'als')
      throw new Error("Only alg 'als' is implemented (sgd is obsolete in the reference, README.md:13)");
    return Promise.resolve();
  }

  // -----------------------  factors  -----------------------

  areSharedFactorsOpened() { return this.handle !== null; }

  /** createSharedFactors (EmfBase.js:399-425): device matrices + host staging arrays */
  createSharedFactors() {
    this.detachSharedFactors();
    this.handle = als.native.create({
      device: this.options.device,
      useDoublePrecision: this.options.useDoublePrecision,
      factorsCount: this.factorsCount,
      totalUsersCount: this.totalUsersCount,
      totalItemsCount: this.totalItemsCount,
      userFactReg: this.options.als.userFactReg,
      itemFactReg: this.options.als.itemFactReg,
      chunkRatings: this.options.chunkRatings,
    });
    this.userFactors = new this.TypedArrayClass(this.totalUsersCount * this.factorsCount);
    this.itemFactors = new this.TypedArrayClass(this.totalItemsCount * this.factorsCount);
  }

  /** detachSharedFactors (EmfBase.js:351-376) */
  detachSharedFactors() {
    if (this.handle !== null) {
      als.native.destroy(this.handle);
      this.handle = null;
    }
    this.userFactors = null;
    this.itemFactors = null;
  }

  /**
   * initSharedFactorsRandom (EmfBase.js:457-513): N(0, 1/factorsCount) for rows >= oldCnt,
   * optionally first factor = the row's average rating.
   */
  initSharedFactorsRandom(oldUsersCnt = 0, oldItemsCnt = 0) {
    const k = this.factorsCount, sd = 1 / k;
    const fill = (arr, from, rows, seed) => {
      const rnd = rng(seed);
      for (let i = from * k; i < rows * k; i++) arr[i] = rnd.normal() * sd;
    };
    fill(this.userFactors, oldUsersCnt, this.totalUsersCount, this.options.seed * 2);
    fill(this.itemFactors, oldItemsCnt, this.totalItemsCount, this.options.seed * 2 + 1);
    if (this.options.als.initFirstFactorAsAvgRating) {
      for (let u = oldUsersCnt; u < this.totalUsersCount; u++) {
        const avg = this.stats.ratingsAvgPerUser[u];
        if (avg !== undefined) this.userFactors[u * k] = avg;
      }
      for (let i = oldItemsCnt; i < this.totalItemsCount; i++) {
        const avg = this.stats.ratingsAvgPerItem[i];
        if (avg !== undefined) this.itemFactors[i * k] = avg;
      }
    }
    this.syncFactorsToDevice();
  }

  syncFactorsToDevice() {
    als.native.setFactors(this.handle, als.BY_USER, this.userFactors);
    als.native.setFactors(this.handle, als.BY_ITEM, this.itemFactors);
  }

  syncFactorsFromDevice() {
    als.native.getFactors(this.handle, als.BY_USER, this.userFactors);
    als.native.getFactors(this.handle, als.BY_ITEM, this.itemFactors);
  }

  /** getFactorsRowSync (EmfBase.js:702-718): a view into the host staging copy */
  getFactorsRowSync(type, rowId) {
    const f = (type == 'byUser' ? this.userFactors : this.itemFactors);
    return f.subarray(rowId * this.factorsCount, (rowId + 1) * this.factorsCount);
  }

  // -----------------------  predict  -----------------------

  getCanPredictError() {
    if (this._status != 'ready') return 'Status is not ready';
    else if (!this.areSharedFactorsOpened()) return 'Factors not loaded';
    else return null;
  }

  /** userId, itemId are 0-based (EmfBase.js:815-827) */
  alsPredictSync(userId, itemId, uF = null) {
    if (!uF) uF = this.getFactorsRowSync('byUser', userId);
    const iF = this.getFactorsRowSync('byItem', itemId);
    return this._alsPredict(uF, iF);
  }

  predictSync(userId, itemId, userFactors = null) { return this.alsPredictSync(userId, itemId, userFactors); }

  _alsPredict(uF, iF) {
    let dot = 0;
    for (let f = 0; f < uF.length; f++) dot += uF[f] * iF[f];
    return dot + this.globalAvgShift;
  }

  destroy() {
    this.detachSharedFactors();
    this._status = 'destroyed';
  }
}

EmfBase.deepmergeAll = deepmergeAll;
module.exports = { EmfBase, als, fs, path, assert, numCPUs };
