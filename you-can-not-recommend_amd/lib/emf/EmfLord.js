/**
 * Train orchestration: train(), alsTrainIter(), alsTrainStep(), calcRmse(), stats and the
 * row partitioner -- lib/emf/EmfLord.js:48-128,510-612,864-1081 without PostgreSQL and the
 * TCP cluster.
 */
'use strict';

const EmfMaster = require('./EmfMaster');
const { als } = require('./EmfBase');
const child_process = require('child_process');
const path = require('path');

class EmfLord extends EmfMaster {
  /**
   * getStats (EmfLord.js:48-128) from the data set instead of SQL: matrix row count is
   * max(id), not count (EmfLord.js:81-82); rows without ratings are holes in the sparse
   * arrays (EmfLord.js:99-100).
   */
  getStats() {
    const ds = this.dataset, st = this.stats;
    this.totalUsersCount = ds.totalUsersCount;
    this.totalItemsCount = ds.totalItemsCount;
    st.totalRatingsAvg = ds.totalRatingsAvg;
    const one = (csr, cntArr, avgArr) => {
      let used = 0, total = 0, max = 0;
      for (let r = 0; r < csr.rows; r++) {
        const b = csr.rowPtr[r], e = csr.rowPtr[r + 1], cnt = e - b;
        if (cnt == 0) continue;
        let s = 0;
        for (let p = b; p < e; p++) s += csr.vals[p];
        cntArr[r] = cnt; avgArr[r] = s / cnt;
        used++; total += cnt;
        if (cnt > max) max = cnt;
      }
      return [used, total, max];
    };
    st.ratingsCntPerUser = []; st.ratingsAvgPerUser = []; st.ratingsCntPerItem = []; st.ratingsAvgPerItem = [];
    [this.trainUsersCount, st.trainUsersRatingsCount, st.maxRatingsPerUser] = one(ds.trainByUser, st.ratingsCntPerUser, st.ratingsAvgPerUser);
    [this.trainItemsCount, st.trainItemsRatingsCount, st.maxRatingsPerItem] = one(ds.trainByItem, st.ratingsCntPerItem, st.ratingsAvgPerItem);
    return Promise.resolve();
  }

  /**
   * Row portions of every pass: contiguous id ranges whose (scaled) rating counts stay within a
   * budget.  Same rule and same results as EmfLord.splitToPortions of the reference
   * (lib/emf/EmfLord.js:510-612; checked against the oracle's restatement in
   * tests/test_node_host.py); portionsRowIdTo[pass][p] is the 1-based last row id of portion p.
   */
  splitToPortions() {
    const st = this.stats, opt = this.options;
    if (!st.trainUsersRatingsCount || !st.trainItemsRatingsCount) return Promise.resolve();
    const minPortions = opt.numThreadsForTrain[opt.alg];
    // the RMSE passes walk the validate / test share of every user's ratings: (pct + 1) %
    const share = { byUser: 0, byItem: 0, rmseValidate: opt.dataSetDistr[1] + 1, rmseTest: opt.dataSetDistr[2] + 1 };
    const scaled = (n, pct) => (pct ? Math.ceil(n * (pct / 100)) : n);

    // ratings one portion may hold: the configured size, lowered so that there are at least
    // `minPortions` portions (never fewer rows than one per portion), raised to the heaviest row
    const budgetOf = (ratings, rows, heaviest, wanted) => {
      let portions = Math.ceil(ratings / wanted), budget = wanted;
      if (portions < minPortions) {
        portions = minPortions;
        budget = Math.ceil(ratings / portions);
      }
      if (Math.floor(rows / portions) < 1) budget = Math.ceil(ratings / rows);
      return Math.max(budget, heaviest);
    };

    this.portionsRowIdTo = {};
    for (const pass of Object.keys(share)) {
      const items = pass == 'byItem', pct = share[pass];
      const counts = items ? st.ratingsCntPerItem : st.ratingsCntPerUser;
      const budget = budgetOf(
        scaled(items ? st.trainItemsRatingsCount : st.trainUsersRatingsCount, pct),
        items ? this.trainItemsCount : this.trainUsersCount,
        scaled(items ? st.maxRatingsPerItem : st.maxRatingsPerUser, pct),
        pct ? opt.ratingsInPortionForRmse : opt.ratingsInPortionForAls[pass]);
      // greedy cut over the rows that have ratings (the stats arrays are sparse), in id order
      const ends = [];
      let held = 0, rowsHere = 0, mostRows = 0;
      Object.keys(counts).forEach((key) => {
        const id0 = parseInt(key), cnt = scaled(counts[key], pct);
        if (held + cnt > budget) {
          ends.push(0);
          held = 0;
          rowsHere = 0;
        }
        if (ends.length == 0) ends.push(0);
        held += cnt;
        rowsHere++;
        mostRows = Math.max(mostRows, rowsHere);
        ends[ends.length - 1] = id0 + 1;
      });
      this.portionsRowIdTo[pass] = ends;
      this.portionsCount[pass] = ends.length;
      this.maxRatingsInPortion[pass] = budget;
      this.maxRowsInPortion[pass] = mostRows;
    }
    return Promise.resolve();
  }

  /**
   * prepareToTrain (EmfLord.js:617-653): stats -> portions -> factors -> ratings upload.
   * @param dataset Dataset (lib/Dataset.js) standing in for the db tables
   */
  prepareToTrain(dataset) {
    if (dataset) this.dataset = dataset;
    if (!this.dataset) return Promise.reject({ code: 'no_data', error: 'No data to train' });
    this._status = 'preparing';
    this.calcDate = new Date().toISOString();
    return this.getStats().then(() => {
      if (this.trainUsersCount == 0 && this.trainItemsCount == 0)
        return Promise.reject({ code: 'no_data', error: 'No data to train' });
      return this.splitToPortions();
    }).then(() => this.prepareSharedFactors())
      .then(() => this.prepareWorkersToTrain())
      .then(() => { this._status = 'ready'; });
  }

  getCanTrainError() {
    let err = null;
    if (this.status == 'training') {
      err = 'Training is already in progress';
    } else if (this.status != 'ready') {
      err = 'Not ready to train. Status is ' + this.status;
    }
    return err;
  }

  /** train (EmfLord.js:864-926) */
  train(dataset) {
    const prep = (dataset || this.status != 'ready') ? this.prepareToTrain(dataset) : Promise.resolve();
    return prep.then(() => {
      const err = this.getCanTrainError();
      if (err !== null) return Promise.reject(err);
      this._status = 'training';
      this.trainIter = 0;
      this.history = [];
      const loop = () => {
        if (!(this.trainIter < this.options.trainIters)) return Promise.resolve();
        const rec = { iter: this.trainIter };
        return this.alsTrainIter()
          .then(() => this.calcRmse('rmseValidate', false)).then((r) => { rec.rmseValidate = r; })
          .then(() => this.calcRmse('rmseTest', false)).then((r) => { rec.rmseTest = r; })
          .then(() => this.calcRmse('rmseTest', true)).then((r) => { rec.rmseTestShifted = r; })
          .then(() => {
            rec.globalAvgShift = this.globalAvgShift;
            this.history.push(rec);
            this.trainIter++;
            // the reference's open todo "saveCalcResults every iter!" (lib/YcnrController.js:288): a
            // checkpoint the next train() warm-starts from (_loadSharedFactorsForTrain)
            if (this.options.saveCalcResultsEveryIter && this.trainIter < this.options.trainIters) {
              this.finishExchange();  // (itemStepSharding 'bands': the user matrix is brought up to date only where it is read whole)
              if (this.options.rank == 0) return Promise.resolve(this.saveCalcResults(this.getCalcInfo())).then(loop);
            }
            return loop();
          });
      };
      return loop();
    }).then(() => {
      this.calcCnt++;
      this.finishExchange();
      if (this.options.rank != 0) return Promise.resolve();  // one writer of the result files
      return this.saveCalcResults(this.getCalcInfo());
    }).then(() => {
      this._status = 'ready';
      return this.history;
    });
  }

  /**
   * Multi-GPU train(): the Lord forks one process per GPU (cf. EmfMaster.createWorkers, lib/emf/EmfMaster.js:44-98:
   * child_process.fork of the worker entry point), hands every one the same options, its rank and the
   * communicator id the native library made (the role of the TCP registration of EmfLord.initClusterLord /
   * EmfChief.initClusterChief, lib/emf/EmfLord.js:668-747, EmfChief.js:87-231), and waits for 'trained'.
   * The per-GPU processes run in lockstep through the library's exchange; rank 0 writes the result files.
   * @param datasetSpec {inline: {users, items, user[], item[], rating[], type[]}} or {dir, validate, test, totalRatingsAvg}
   * @return Promise of rank 0's {history, calcInfo, stepInfo}
   */
  trainOnGpus(datasetSpec, config) {
    // A device-to-device transport that cannot be set up in every GPU process is followed by the other one (rccl -> ipc, ipc -> rccl)
    // unless options.strictTransport is set: a train on a node nobody has seen before should produce its factors, and the result says
    // over which transport it ran (commTransport) and why not over the one asked for (commFallback).  Only failures of the
    // set-up phase are retried (the processes are started afresh); a failure while training is an error as before.
    const wanted = this.options.commTransport;
    const d2d = ['rccl', 'ipc'];
    const others = (!this.options.strictTransport && d2d.includes(wanted)) ? d2d.filter((t) => t != wanted) : [];
    const why = [];
    const attempt = (transport, rest) => this._trainOnGpusOver(transport, datasetSpec, config).then((res) => {
      if (res && typeof res == 'object') {
        res.commTransport = transport;
        if (why.length) res.commFallback = why;
      }
      return res;
    }, (e) => {
      if (!(e && e.duringSetup) || !rest.length) return Promise.reject(e);
      why.push(transport + ': ' + e.message);
      return attempt(rest[0], rest.slice(1));
    });
    return attempt(wanted, others);
  }

  /** trainOnGpus over one transport; a rejection before every process reported 'ready' carries duringSetup = true */
  _trainOnGpusOver(transport, datasetSpec, config) {
    const world = this.options.gpus;
    if (!(world > 1)) return Promise.reject(new Error('trainOnGpus needs options.gpus > 1'));
    let commId;
    try {
      commId = Buffer.from(als.native.commUniqueId(als.commTransport[transport])).toString('base64');
    } catch (e) {
      e.duringSetup = true;
      return Promise.reject(e);
    }
    let settingUp = true;
    let devices = this.options.gpuDevices || 0;  // 0: ask the library (fails loudly without a HIP device)
    if (!devices) {
      try { devices = als.native.deviceCount(); } catch (e) { return Promise.reject(e); }
    }
    const kids = [];
    const stop = () => kids.forEach((k) => { try { k.send({ cmd: 'destroy' }); } catch (e) { /* already gone */ } });
    // A GPU process that dies (non-zero code, or a signal: SIGSEGV, the OOM killer, a GPU fault report code === null)
    // or cannot be started must end the whole train: its peers would otherwise wait for it inside the exchange.
    let died = null;
    const deaths = [];
    const onDeath = (k, why) => {
      if (died === null) died = new Error('GPU process of rank ' + kids.indexOf(k) + ' ' + why);
      deaths.forEach((f) => f(died));
    };
    const all = (evt, after) => Promise.all(kids.map((k) => new Promise((resolve, reject) => {
      const onMsg = (m) => {
        if (m.evt == evt) { k.removeListener('message', onMsg); resolve(m); }
        else if (m.evt == 'error') { k.removeListener('message', onMsg); reject(new Error(m.error)); }
      };
      if (died !== null) return reject(died);
      deaths.push(reject);
      k.on('message', onMsg);
      after(k);
    })));
    for (let r = 0; r < world; r++) {
      const k = child_process.fork(this.options.gpuProcessScript || path.join(__dirname, 'EmfGpuProcess.js'), [], { stdio: 'inherit' });
      k.finished = false;
      k.once('exit', (code, signal) => {
        if (!k.finished && (code !== 0 || signal)) onDeath(k, signal ? 'was killed by ' + signal : 'exited with code ' + code);
      });
      k.once('error', (e) => onDeath(k, 'failed: ' + (e && e.message)));
      kids.push(k);
    }
    const opts = Object.assign({}, this.options, { gpus: 1, commTransport: transport });
    // no train runs longer than this without a message (options.gpuProcessTimeoutMs, 0 = no limit)
    const limitMs = this.options.gpuProcessTimeoutMs || 0;
    let timer = null;
    const guarded = limitMs > 0 ? new Promise((_, reject) => {
      timer = setTimeout(() => reject(new Error('trainOnGpus: no result within ' + limitMs + ' ms')), limitMs);
    }) : null;
    const run = all('ready', (k) => {
      const rank = kids.indexOf(k);
      k.send({ cmd: 'init', config: config || {}, options: opts, rank, world, commId, device: rank % devices, dataset: datasetSpec });
    }).then(() => { settingUp = false; return all('trained', (k) => k.send({ cmd: 'train' })); });
    const done = (x) => { if (timer) clearTimeout(timer); kids.forEach((k) => { k.finished = true; }); return x; };
    return (guarded ? Promise.race([run, guarded]) : run)
      .then((res) => { done(); stop(); return res[0]; },
            (e) => {
              done(); stop(); kids.forEach((k) => { try { k.kill('SIGKILL'); } catch (e2) { /* gone */ } });
              if (settingUp && e && typeof e == 'object') e.duringSetup = true;
              return Promise.reject(e);
            });
  }

  /** 2 steps - first fix item vectors and calc user vectors, then vice versa (EmfLord.js:954-958) */
  alsTrainIter() {
    return this.alsTrainStep('byUser')
      .then(() => this.alsTrainStep('byItem'))
      .then(() => {
        // feedback for the static shards (options.rebalanceAfterIters, EmfMaster.rebalance)
        if (this.options.world > 1 && ++this.itersRun <= this.options.rebalanceAfterIters) this.rebalance();
      });
  }

  /** @param string stepType 'byUser', 'byItem' (EmfLord.js:963-984) */
  alsTrainStep(stepType) {
    return new Promise((resolve, reject) => {
      this.once('stepComplete', () => resolve(this.lastStepInfo));
      try {
        this._startAlsTrainStep(stepType);
      } catch (e) {
        this.removeAllListeners('stepComplete');
        reject(e);
      }
    });
  }

  /** @param string stepType 'rmseValidate', 'rmseTest' (EmfLord.js:1043-1081) */
  calcRmse(stepType, useGlobalAvgShift) {
    if (useGlobalAvgShift && this.options.alg != 'als')
      return Promise.resolve();
    if (this.options.dataSetDistr[1] == 0 && stepType == 'rmseValidate')
      return Promise.resolve();
    if (this.options.dataSetDistr[2] == 0 && stepType == 'rmseTest')
      return Promise.resolve();
    if (!this.dataset[stepType == 'rmseValidate' ? 'validate' : 'test'])
      return Promise.resolve();
    return new Promise((resolve, reject) => {
      this.once('stepComplete', () => resolve(this.rmse));
      try {
        this._startCalcRmse(stepType, useGlobalAvgShift);
      } catch (e) {
        this.removeAllListeners('stepComplete');
        reject(e);
      }
    });
  }
}

module.exports = EmfLord;
