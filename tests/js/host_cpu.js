// CPU-side checks of the NodeJS host (run by tests/test_node_host.py): addon loads and
// exports the reference-shaped API, type errors and missing-GPU errors are JS Errors,
// splitToPortions reproduces the expected partitions passed in as JSON.
'use strict';
const assert = require('assert');
const path = require('path');
const root = path.join(__dirname, '..', '..', 'you-can-not-recommend_amd');
const als = require(path.join(root, 'lib', 'ycnr_als'));
const Emf = require(path.join(root, 'lib', 'emf', 'Emf'));
const { Dataset, rng } = require(path.join(root, 'lib', 'Dataset'));

const n = als.native;
for (const f of ['sAlsCalcPortion', 'dAlsCalcPortion', 'sRmsePortion', 'dRmsePortion', 'create', 'destroy', 'setRatings',
  'setRmseRatings', 'setFactors', 'getFactors', 'step', 'rmse', 'deviceCount', 'lastError', 'version',
  'alsUnpinFixedFactors', 'commUniqueId', 'commInit', 'setRatingsSharded', 'allreduceSum', 'broadcastFactors', 'exchange',
  'setRatingsBanded', 'deferExchange', 'commInfo', 'lastRmseMs'])
  assert.strictEqual(typeof n[f], 'function', f);
assert.strictEqual(n.version(), 4);
// the communicator id of the shared-memory stand-in needs no GPU: 128 bytes, not all zero, new every time
const id1 = n.commUniqueId(als.COMM_SHM), id2 = n.commUniqueId(als.COMM_SHM);
assert.strictEqual(id1.length, 128);
assert.ok(id1.some((b) => b != 0) && Buffer.compare(Buffer.from(id1), Buffer.from(id2)) != 0);

// s/d dispatch and Error('invalid type!') exactly like cpp_utils/cpp_utils.js:6-19
assert.throws(() => als.alsCalcPortion(0.05, 4, new Int32Array([1, 0, 1]), new Int32Array([0]), new Int16Array([1]),
  new Float32Array(4), new Float32Array(4)), /invalid type!/);
assert.throws(() => als.alsCalcPortion(0.05, 4, new Int32Array([1, 0, 1]), new Int32Array([0]), new Float32Array([1]),
  new Float64Array(4), new Float32Array(4)), /invalid type!/);

const input = JSON.parse(process.argv[2]);
const lord = Emf.createLord();
lord.init({ emf: { factorsCount: 8 } }, { dataDir: input.dir, ratingsInPortionForRmse: input.rip, ratingsInPortionForAls: { byUser: input.rip, byItem: input.rip },
  numThreadsForTrain: { als: input.threads } });
assert.strictEqual(lord.options.als.userFactReg, 0.05);
assert.strictEqual(lord.options.factorsCount, 8);
assert.strictEqual(lord.TypedArrayClass, Float32Array);
const t = { user: Int32Array.from(input.user), item: Int32Array.from(input.item), rating: Float32Array.from(input.rating) };
const ds = new Dataset(input.users, input.items, t, Int8Array.from(input.type));
lord.dataset = ds;
const out = {};
lord.getStats().then(() => lord.splitToPortions()).then(() => {
  out.portionsRowIdTo = lord.stats.portionsRowIdTo;
  out.maxRatingsInPortion = lord.stats.maxRatingsInPortion;
  out.maxRowsInPortion = lord.stats.maxRowsInPortion;
  out.trainNnz = ds.trainByUser.nnz;
  out.byItemIndxHead = Array.from(ds.trainByItem.indx.slice(0, 10));
  out.totalRatingsAvg = ds.totalRatingsAvg;
  // cost-balanced shard cuts (checked against the Python mirror's shard_ranges)
  const EmfMaster = require(path.join(root, 'lib', 'emf', 'EmfMaster'));
  out.shards = {};
  for (const [w, k] of [[2, 8], [3, 100], [8, 256]])
    out.shards[w + '_' + k] = EmfMaster.shardRanges(ds.trainByUser.rowPtr, 0, ds.trainByUser.rows, w, k, false);
  // the feedback re-cut from measured times (checked against the Python mirror's rebalanced_ranges)
  out.recut = {};
  for (const [w, k] of [[2, 8], [3, 100]]) {
    const b = out.shards[w + '_' + k], ms = b.slice(1).map((_, r) => 1.0 + 0.5 * r);
    out.recut[w + '_' + k] = EmfMaster.rebalancedRanges(ds.trainByUser.rowPtr, b, ms, k, false);
  }
  // the reference packer's end-of-data quirk, opt-in (checked against the Python mirror and the oracle's PackPortion)
  out.quirk = { byUser: [], byItemRowPtr: null };
  let begin = 0;
  for (const end of lord.stats.portionsRowIdTo.byUser) {
    out.quirk.byUser.push(Array.from(ds.trainByUser.toPortion(begin, end, true).alsRows));
    begin = end;
  }
  const dq = ds.withoutLastRatingPerPortion(lord.stats.portionsRowIdTo);
  out.quirk.byUserRowPtr = Array.from(dq.trainByUser.rowPtr);
  out.quirk.byItemRowPtr = Array.from(dq.trainByItem.rowPtr);
  out.quirk.byItemIndx = Array.from(dq.trainByItem.indx);
  out.quirk.validateRowPtr = dq.validate ? Array.from(dq.validate.rowPtr) : null;
  let gpu = true;
  try { n.deviceCount(); } catch (e) { gpu = false; out.deviceCountError = e.message; }
  if (gpu) return;
  // without a GPU the trainer must fail loudly, not fall back
  return lord.prepareToTrain(ds).then(() => { throw new Error('prepareToTrain succeeded without a GPU'); },
    (e) => { out.prepareError = String(e.message || e); });
}).then(() => { console.log(JSON.stringify(out)); }).catch((e) => { console.error(e); process.exit(1); });
