// CPU side of the YCSR file format test (tests/test_ingest.py): reads the file Python wrote,
// writes it back with lib/CsrFile.js, and parses a MovieLens file with Dataset.readMovieLens.
'use strict';
const path = require('path');
const root = path.join(__dirname, '..', '..', 'you-can-not-recommend_amd');
const { readCsr, writeCsr } = require(path.join(root, 'lib', 'CsrFile'));
const { Dataset } = require(path.join(root, 'lib', 'Dataset'));
const [inFile, outFile, mlFile] = process.argv.slice(2);
const a = readCsr(inFile);
writeCsr(outFile, a);
const ml = Dataset.readMovieLens(mlFile);
console.log(JSON.stringify({ rows: a.rows, cols: a.cols, nnz: a.nnz, firstIndx: Array.from(a.indx.slice(0, 5)),
  lastVal: a.vals[a.vals.length - 1], double: a.vals.constructor === Float64Array,
  ml: { users: ml.totalUsersCount, items: ml.totalItemsCount, user: Array.from(ml.triplets.user), item: Array.from(ml.triplets.item),
        rating: Array.from(ml.triplets.rating) } }));
