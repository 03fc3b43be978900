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

  /** Upload the ratings once: replaces createWorkPortionBuffers + per-portion fetches (EmfMaster.js:156-234,501-614) */
  prepareWorkersToTrain() {
    const ds = this.dataset, n = als.native;
    n.setRatings(this.handle, als.BY_USER, ds.trainByUser.rowPtr, ds.trainByUser.indx, ds.trainByUser.vals);
    n.setRatings(this.handle, als.BY_ITEM, ds.trainByItem.rowPtr, ds.trainByItem.indx, ds.trainByItem.vals);
    if (ds.validate) n.setRmseRatings(this.handle, als.RMSE_VALIDATE, ds.validate.rowPtr, ds.validate.indx, ds.validate.vals);
    if (ds.test) n.setRmseRatings(this.handle, als.RMSE_TEST, ds.test.rowPtr, ds.test.indx, ds.test.vals);
    return Promise.resolve();
  }

  /** _startAlsTrainStep (EmfMaster.js:364-383): the whole step is one native call */
  _startAlsTrainStep(stepType) {
    this.workType = 'train';
    this.stepType = stepType;
    this.lastStepInfo = als.native.step(this.handle, als.stepSide[stepType]);
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
