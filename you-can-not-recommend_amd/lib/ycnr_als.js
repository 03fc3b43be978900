/**
 * JS face of the native ALS library: the counterpart of the reference's
 * cpp_utils/cpp_utils.js (s/d dispatch by typed-array class, Error('invalid type!')).
 * All arithmetic happens in libycnr_als.so on the GPU; there is no JS fallback.
 */
(function () {
  'use strict';

  // read by the HIP runtime when it starts (first HIP call of the process): the row kernel and every dual class of a
  // half-step on a hardware queue of their own (python/ycnr_als/_lib.py does the same)
  if (!process.env.GPU_MAX_HW_QUEUES) process.env.GPU_MAX_HW_QUEUES = '8';
  var native = require('../addon/ycnr_als.node');

  // cpp_utils/cpp_utils.js:6-13
  function typeCheck(array) {
    if (array.constructor === Float64Array)
      return true;
    else if (array.constructor === Float32Array)
      return false;

    throw new Error('invalid type!');
  }

  var als = {
    native: native,
    BY_USER: 0, BY_ITEM: 1,
    RMSE_VALIDATE: 0, RMSE_TEST: 1,
    FLAG_LDS_SOLVER: 1, FLAG_NO_DUAL: 2,
    COMM_RCCL: 1, COMM_SHM: 2, COMM_IPC: 3, COMM_STUB: 4,
    commTransport: { rccl: 1, shm: 2, ipc: 3, stub: 4 },
    stepSide: { byUser: 0, byItem: 1 },
    rmseSet: { rmseValidate: 0, rmseTest: 1 },
  };

  /**
   * Drop-in for the body of EmfWorker.mw_calcTrainAlsPortion (lib/emf/EmfWorker.js:176-251):
   * solves every row of the portion against fixedFactors and writes it in place into
   * solvedFactors. Returns ratingsInPortion.
   */
  als.alsCalcPortion = function (lambda, factorsCount, alsRows, alsIndx, alsVals, fixedFactors, solvedFactors) {
    return typeCheck(alsVals) ?
      native.dAlsCalcPortion(lambda, factorsCount, alsRows, alsIndx, alsVals, fixedFactors, solvedFactors) :
      native.sAlsCalcPortion(lambda, factorsCount, alsRows, alsIndx, alsVals, fixedFactors, solvedFactors);
  };

  /**
   * Optional, once per half-step (where the worker handles 'startTrainStep', lib/emf/EmfWorker.js:43-51): keep the
   * step's fixed factor matrix on the device, so that the portion calls that follow with the same typed array
   * skip the per-portion gather + upload -- the replacement of the per-rating BLAS.BufCopy of
   * EmfBase.copySubFixedFactors (lib/emf/EmfBase.js:537-555).  Pin again after the matrix has changed.
   */
  als.alsPinFixedFactors = function (fixedFactors, factorsCount) {
    return typeCheck(fixedFactors) ? native.dAlsPinFixedFactors(fixedFactors, factorsCount) :
      native.sAlsPinFixedFactors(fixedFactors, factorsCount);
  };

  /** Forget the pinned matrix.  A portion call that solves INTO the pinned typed array does this by itself. */
  als.alsUnpinFixedFactors = function () { return native.alsUnpinFixedFactors(); };

  /**
   * Drop-in for EmfWorker.mw_calcRmsePortion (lib/emf/EmfWorker.js:266-315).
   * Returns {rSumDiff2, rCnt, rSum}.
   */
  als.rmsePortion = function (factorsCount, rmseRows, rmseIndx, rmseVals, userFactors, itemFactors, globalAvgShift) {
    return typeCheck(rmseVals) ?
      native.dRmsePortion(factorsCount, rmseRows, rmseIndx, rmseVals, userFactors, itemFactors, globalAvgShift || 0) :
      native.sRmsePortion(factorsCount, rmseRows, rmseIndx, rmseVals, userFactors, itemFactors, globalAvgShift || 0);
  };

  /**
   * EmfLord.doSplitToSets (lib/emf/EmfLord.js:402-505) for ratings held as CSR by user:
   * completes `types` (Int8Array, 0 = unassigned, 1/2/3 = train/validate/test, others untouched)
   * in place with the reference's per-user counts and a keyed, seedable shuffle.
   * Returns the kernel time in ms.
   */
  als.splitToSets = function (rowPtr, types, dataSetDistr, seed) {
    if (types.constructor !== Int8Array) throw new Error('invalid type!');
    return native.splitToSets(rowPtr, types, dataSetDistr || [85, 10, 5], seed === undefined ? 1 : seed);
  };

  /**
   * ratings_count / avg_rating of every row over dataset_type IN (1, 2, 3)
   * (EmfLord.updateUsersStats / updateItemsStats, lib/emf/EmfLord.js:252-396).
   * types may be null (every rating counts). Returns {cnt: Int32Array, avg: Float64Array,
   * max, total, totalRatingsAvg}.
   */
  als.ratingStats = function (rowPtr, vals, types) {
    typeCheck(vals);
    var rows = rowPtr.length - 1, cnt = new Int32Array(rows), sum = new Float64Array(rows);
    native.ratingStats(rowPtr, vals, types || null, cnt, sum);
    var avg = new Float64Array(rows), max = 0, total = 0, totalSum = 0;
    for (var r = 0; r < rows; r++) {
      avg[r] = cnt[r] ? sum[r] / cnt[r] : 0;
      if (cnt[r] > max) max = cnt[r];
      total += cnt[r];
      totalSum += sum[r];
    }
    return { cnt: cnt, avg: avg, max: max, total: total, totalRatingsAvg: total ? totalSum / total : 0 };
  };

  /**
   * CSR of (row, col, value) triplets, rows by id and entries of a row by column id (the
   * ORDER BY of lib/emf/EmfMaster.js:511-529), sorted on the GPU. Returns {rowPtr, indx, vals}.
   */
  als.csrFromTriplets = function (rowIdx, colIdx, vals, rows, cols) {
    typeCheck(vals);
    var n = rowIdx.length, rowPtr = new Float64Array(rows + 1), indx = new Int32Array(n), out = new vals.constructor(n);
    native.csrFromTriplets(rowIdx, colIdx, vals, rows, cols, rowPtr, indx, out);
    return { rowPtr: rowPtr, indx: indx, vals: out };
  };

  /** The same ratings by column: CSR by user -> CSR by item. Returns {rowPtr, indx, vals}. */
  als.csrTranspose = function (rows, cols, rowPtr, indx, vals) {
    typeCheck(vals);
    var n = rowPtr[rows], outPtr = new Float64Array(cols + 1), outIndx = new Int32Array(n), out = new vals.constructor(n);
    native.csrTranspose(rows, cols, rowPtr, indx, vals, outPtr, outIndx, out);
    return { rowPtr: outPtr, indx: outIndx, vals: out };
  };

  /**
   * The item loop of YcnrController.recommendItemsForUser (lib/YcnrController.js:255-274) for one
   * user on the GPU. userFactors: the user's k factors; skipItemIds: 1-based ids to leave out
   * (rated + unrated_items, :244-250). Returns recItems: [{predict, id}] (id 1-based), best
   * first -- like the reference at most limit - 1 of them.
   */
  als.recommendItemsForUser = function (userFactors, itemFactors, factorsCount, skipItemIds, globalAvgShift, minRecommendRating, limit) {
    typeCheck(itemFactors);
    if (userFactors.constructor !== itemFactors.constructor) throw new Error('invalid type!');
    limit = limit || 20;
    var skip = Int32Array.from(skipItemIds || [], function (id1) { return id1 - 1; }).sort();
    var uniq = skip.filter(function (v, i) { return i === 0 || v !== skip[i - 1]; });
    var ids = new Int32Array(limit), pred = new Float64Array(limit), cnt = new Int32Array(1);
    native.recommendItems(userFactors, itemFactors, factorsCount, Float64Array.of(0, uniq.length), uniq, globalAvgShift || 0,
      minRecommendRating, limit, ids, pred, cnt);
    var recItems = [];
    for (var i = 0; i < cnt[0]; i++) recItems.push({ predict: pred[i], id: ids[i] + 1 });
    return recItems;
  };

  module.exports = als;
}());
