/**
 * Entry point of one per-GPU process of a multi-GPU training run (forked by EmfLord.trainOnGpus) -- the
 * counterpart of the reference's lib/emf/EmfWorkerProcess.js (child bootstrap + message bridge,
 * EmfWorkerProcess.js:20-45, EmfProcess.js:38-62), with a GPU instead of a BLAS thread behind it.
 *
 * Every process runs the SAME train() on its own GPU over its own row shards; the native library keeps
 * the replicas of the factor matrices in step (the exchange inside every half-step, the all-reduce of
 * the RMSE partial sums), so the only messages are 'init' / 'train' / 'destroy' from the Lord and
 * 'ready' / 'trained' / 'error' back.  Rank 0 writes the result files.
 */
'use strict';

const path = require('path');
const EmfLord = require('./EmfLord');
const { Dataset } = require('../Dataset');
const { readCsr } = require('../CsrFile');

let lord = null;

function datasetOf(spec, F) {
  if (spec.inline) {
    const t = { user: Int32Array.from(spec.inline.user), item: Int32Array.from(spec.inline.item), rating: F.from(spec.inline.rating) };
    return new Dataset(spec.inline.users, spec.inline.items, t, spec.inline.type ? Int8Array.from(spec.inline.type) : null, F);
  }
  // a directory of YCSR files (lib/CsrFile.js): train_by_user, train_by_item[, validate, test]
  const ds = Object.create(Dataset.prototype);
  const rd = (name) => readCsr(path.join(spec.dir, name));
  ds.trainByUser = rd('train_by_user');
  ds.trainByItem = rd('train_by_item');
  ds.validate = spec.validate ? rd('validate') : null;
  ds.test = spec.test ? rd('test') : null;
  ds.totalUsersCount = ds.trainByUser.rows;
  ds.totalItemsCount = ds.trainByUser.cols;
  ds.totalRatingsAvg = spec.totalRatingsAvg;
  return ds;
}

function fail(e) {
  process.send({ evt: 'error', error: String((e && (e.stack || e.message || e.error)) || e) });
}

process.on('message', (m) => {
  try {
    if (m.cmd == 'init') {
      lord = new EmfLord();
      lord.init(m.config, Object.assign({}, m.options, { rank: m.rank, world: m.world, commId: m.commId, device: m.device }));
      const ds = datasetOf(m.dataset, lord.TypedArrayClass);
      lord.prepareToTrain(ds).then(() => process.send({ evt: 'ready' })).catch(fail);
    } else if (m.cmd == 'train') {
      lord.train().then((history) => {
        process.send({ evt: 'trained', history, calcInfo: lord.getCalcInfo(), stepInfo: lord.lastStepInfo });
      }).catch(fail);
    } else if (m.cmd == 'destroy') {
      if (lord) lord.destroy();
      process.exit(0);
    }
  } catch (e) {
    fail(e);
  }
});
