// GPU run of the NodeJS host (tests/test_node_host.py, -m gpu): warm-starts from the factor
// files Python wrote into <dir>/ml_factors_ready, trains, and leaves its own result files.
'use strict';
const path = require('path');
const fs = require('fs');
const root = path.join(__dirname, '..', '..', 'you-can-not-recommend_amd');
const Emf = require(path.join(root, 'lib', 'emf', 'Emf'));
const als = require(path.join(root, 'lib', 'ycnr_als'));
const { Dataset } = require(path.join(root, 'lib', 'Dataset'));

const input = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const F = input.useDoublePrecision ? Float64Array : Float32Array;
const t = { user: Int32Array.from(input.user), item: Int32Array.from(input.item), rating: F.from(input.rating) };
const ds = new Dataset(input.users, input.items, t, Int8Array.from(input.type), F);
const lord = Emf.createLord();
lord.init({}, { factorsCount: input.k, trainIters: input.iters, dataDir: input.dir, dbType: 'ml', useDoublePrecision: input.useDoublePrecision,
  ratingsInPortionForRmse: input.rip, numThreadsForTrain: { als: input.threads },
  saveCalcResultsEveryIter: true });  // N4: a checkpoint after every iteration must not change the result
let checkpoints = 0;
const save = lord.saveCalcResults.bind(lord);
lord.saveCalcResults = (ci) => { checkpoints++; return save(ci); };
lord.train(ds).then((history) => {
  // level-1 portion op on the first 5 users, against the final item factors
  const k = input.k, bu = ds.trainByUser;
  const rows = [0], cnt = Math.min(5, bu.rows);
  for (let u = 0; u < cnt; u++) if (bu.count(u) > 0) { rows.push(u, bu.count(u)); rows[0]++; }
  const e = bu.rowPtr[cnt];
  const solved = new F(cnt * k);
  const ratings = als.alsCalcPortion(0.05, k, Int32Array.from(rows), bu.indx.slice(0, e), bu.vals.slice(0, e), lord.itemFactors, solved);
  console.log(JSON.stringify({ checkpoints, history, calcInfo: lord.getCalcInfo(), stepInfo: lord.lastStepInfo, portionRatings: ratings,
    portionSolved: Array.from(solved) }));
  lord.destroy();
}).catch((e) => { console.error(e); process.exit(1); });
