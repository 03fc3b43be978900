// GPU run of the NodeJS host on several per-GPU processes (tests/test_node_host.py, -m gpu): the Lord forks
// `world` processes that share cuda:0 through the shared-memory transport and trains; the result files in
// <dir> must equal those of the single-process run of the same options.
'use strict';
const path = require('path');
const fs = require('fs');
const root = path.join(__dirname, '..', '..', 'you-can-not-recommend_amd');
const Emf = require(path.join(root, 'lib', 'emf', 'Emf'));

const input = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const lord = Emf.createLord();
lord.init({}, { factorsCount: input.k, trainIters: input.iters, dataDir: input.dir, dbType: 'ml', useDoublePrecision: input.useDoublePrecision,
  ratingsInPortionForRmse: input.rip, numThreadsForTrain: { als: input.threads }, gpus: input.world, commTransport: input.transport || 'shm',
  exchangeChunks: 3, itemStepSharding: input.sharding || 'rows', strictTransport: !!input.strict });
const spec = { inline: { users: input.users, items: input.items, user: input.user, item: input.item, rating: input.rating, type: input.type } };
const { Dataset } = require(path.join(root, 'lib', 'Dataset'));
const F = input.useDoublePrecision ? Float64Array : Float32Array;
const run = input.world > 1 ? lord.trainOnGpus(spec) :
  lord.train(new Dataset(input.users, input.items, { user: Int32Array.from(input.user), item: Int32Array.from(input.item), rating: F.from(input.rating) },
    Int8Array.from(input.type), F)).then((history) => ({ history, calcInfo: lord.getCalcInfo(), stepInfo: lord.lastStepInfo }));
run.then((res) => {
  console.log(JSON.stringify({ history: res.history, calcInfo: res.calcInfo, stepInfo: res.stepInfo, commTransport: res.commTransport, commFallback: res.commFallback }));
  process.exit(0);
}).catch((e) => { console.error(e); process.exit(1); });
