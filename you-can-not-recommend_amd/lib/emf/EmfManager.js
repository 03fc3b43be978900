/**
 * Persistence of calc results: calc_info.json + the two raw factor files, written to
 * <dbType>_factors_tmp and renamed to <dbType>_factors_ready.
 * Mirrors lib/emf/EmfManager.js:158-191,241-255,463-570 without the db / redis parts.
 */
'use strict';

const { EmfBase, fs, path } = require('./EmfBase');

class EmfManager extends EmfBase {
  constructor() {
    super();
    this.calcCnt = 0;
    this.calcDate = null;
    this.lastCalcInfo = null;
  }

  /** getCalcInfo (EmfManager.js:158-176), same keys in the same order */
  getCalcInfo() {
    return {
      //calc options
      alg: this.options.alg,
      algOptions: this.options[this.options.alg],
      useDoublePrecision: this.options.useDoublePrecision,
      factorsCount: this.factorsCount,
      dataSetDistr: this.options.dataSetDistr,
      //data state @ calc start moment
      totalUsersCount: this.totalUsersCount,
      totalItemsCount: this.totalItemsCount,
      dbType: this.options.dbType,
      calcDate: this.calcDate,
      calcCnt: this.calcCnt,
      //results
      globalAvgShift: this.globalAvgShift,
      globalBias: this.globalBias,
    };
  }

  /** EmfManager.js:179-191 */
  _canReuseCalcResults(ci1) {
    const ci2 = this.getCalcInfo();
    return (ci1 !== null
      && ci1.alg == ci2.alg
      && ci1.dbType == ci2.dbType
      && ci1.factorsCount == ci2.factorsCount
      && ci1.useDoublePrecision == ci2.useDoublePrecision);
  }

  _didUsersItemsCntsChanged(ci1) {
    const ci2 = this.getCalcInfo();
    return !(ci1 !== null
      && ci1.totalUsersCount == ci2.totalUsersCount
      && ci1.totalItemsCount == ci2.totalItemsCount);
  }

  /** calc_info.json of the ready directory, or null */
  readLastCalcInfo() {
    try {
      return JSON.parse(fs.readFileSync(this.calcInfoPath, 'utf8'));
    } catch (e) {
      return null;
    }
  }

  /**
   * _loadSharedFactorsForTrain (EmfManager.js:405-457): reuse the ready files when they are
   * compatible, extend with random rows when users/items were added.
   * @return [recreated, extended]
   */
  _loadSharedFactorsForTrain() {
    const ci = this.lastCalcInfo = this.readLastCalcInfo();
    this.createSharedFactors();
    if (!this._canReuseCalcResults(ci) || ci.totalUsersCount > this.totalUsersCount || ci.totalItemsCount > this.totalItemsCount)
      return [true, false];
    const k = this.factorsCount, s = this.TypedArraySize1;
    const load = (file, arr, rows) => {
      const buf = fs.readFileSync(file);
      if (buf.length != rows * k * s) return false;
      new Uint8Array(arr.buffer, arr.byteOffset, buf.length).set(buf); // headerless raw dump, EmfBase.js:384
      return true;
    };
    if (!load(this.userFactorsPath, this.userFactors, ci.totalUsersCount)
      || !load(this.itemFactorsPath, this.itemFactors, ci.totalItemsCount))
      return [true, false];
    this.calcCnt = ci.calcCnt || 0;
    this.globalAvgShift = ci.globalAvgShift || 0;
    const extended = this._didUsersItemsCntsChanged(ci);
    if (!extended) this.syncFactorsToDevice();
    return [false, extended];
  }

  /** saveCalcResults + _saveCalcResultsToRecommender (EmfManager.js:463-570) */
  saveCalcResults(calcInfo) {
    this.syncFactorsFromDevice();
    const tmp = this.factorsTempPath, ready = this.factorsReadyPath;
    fs.mkdirSync(tmp, { recursive: true });
    const dump = (name, arr) => fs.writeFileSync(path.join(tmp, name), Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength));
    dump(this.userFactorsFilename, this.userFactors);
    dump(this.itemFactorsFilename, this.itemFactors);
    fs.writeFileSync(path.join(tmp, this.calcInfoFilename), JSON.stringify(calcInfo, null, 2));
    //critical section - move /factors_temp to /factors_ready
    this.deleteCalcResultsSync(ready);
    if (fs.existsSync(ready)) fs.rmdirSync(ready);
    fs.renameSync(tmp, ready);
    this.lastCalcInfo = calcInfo;
    return Promise.resolve();
  }

  deleteCalcResultsSync(dir) {
    if (!fs.existsSync(dir)) return;
    for (const f of [this.userFactorsFilename, this.itemFactorsFilename, this.calcInfoFilename]) {
      const p = path.join(dir, f);
      if (fs.existsSync(p)) fs.unlinkSync(p);
    }
  }
}

module.exports = EmfManager;
