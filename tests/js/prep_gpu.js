// GPU run of N1 through the N-API addon (tests/test_node_host.py, -m gpu): split + stats of the
// CSR handed over in a JSON file; prints the completed types and the statistics.
'use strict';
const path = require('path');
const fs = require('fs');
const als = require(path.join(__dirname, '..', '..', 'you-can-not-recommend_amd', 'lib', 'ycnr_als'));
const input = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const rowPtr = Float64Array.from(input.rowPtr);   // exact below 2^53, as setRatings accepts
const types = Int8Array.from(input.types);
const vals = Float32Array.from(input.vals);
const ms = als.splitToSets(rowPtr, types, input.dataSetDistr, input.seed);
const st = als.ratingStats(rowPtr, vals, types);
let bad = null;
try { als.splitToSets(rowPtr, new Int32Array(types.length), input.dataSetDistr, input.seed); } catch (e) { bad = e.message; }
// N3 for one user: ids 1-based in and out, as the controller uses them
let rec = null;
if (input.rec) {
  const F = Float32Array;
  rec = als.recommendItemsForUser(F.from(input.rec.user), F.from(input.rec.items), input.rec.k, input.rec.skip1, input.rec.shift, input.rec.min, input.rec.limit);
}
console.log(JSON.stringify({ rec, ms, types: Array.from(types), cnt: Array.from(st.cnt), avg: Array.from(st.avg), max: st.max,
  total: st.total, totalRatingsAvg: st.totalRatingsAvg, wrongTypeMessage: bad }));
