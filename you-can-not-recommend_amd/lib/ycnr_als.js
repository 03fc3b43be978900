/**
 * JS face of the native ALS library: the counterpart of the reference's
 * cpp_utils/cpp_utils.js (s/d dispatch by typed-array class, Error('invalid type!')).
 * All arithmetic happens in libycnr_als.so on the GPU; there is no JS fallback.
 */
(function () {
  'use strict';

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
   * Drop-in for EmfWorker.mw_calcRmsePortion (lib/emf/EmfWorker.js:266-315).
   * Returns {rSumDiff2, rCnt, rSum}.
   */
  als.rmsePortion = function (factorsCount, rmseRows, rmseIndx, rmseVals, userFactors, itemFactors, globalAvgShift) {
    return typeCheck(rmseVals) ?
      native.dRmsePortion(factorsCount, rmseRows, rmseIndx, rmseVals, userFactors, itemFactors, globalAvgShift || 0) :
      native.sRmsePortion(factorsCount, rmseRows, rmseIndx, rmseVals, userFactors, itemFactors, globalAvgShift || 0);
  };

  module.exports = als;
}());
