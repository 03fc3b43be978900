// GPU side of N2 through the addon (tests/test_ingest.py, -m gpu): triplets -> CSR by user ->
// CSR by item, written as a YCSR file pair.
'use strict';
const path = require('path');
const fs = require('fs');
const root = path.join(__dirname, '..', '..', 'you-can-not-recommend_amd');
const als = require(path.join(root, 'lib', 'ycnr_als'));
const { Csr } = require(path.join(root, 'lib', 'Dataset'));
const { writeCsr } = require(path.join(root, 'lib', 'CsrFile'));
const input = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const u = als.csrFromTriplets(Int32Array.from(input.user), Int32Array.from(input.item), Float32Array.from(input.rating), input.users, input.items);
const i = als.csrTranspose(input.users, input.items, u.rowPtr, u.indx, u.vals);
writeCsr(path.join(input.dir, 'ratings_by_user.ycsr'), new Csr(input.users, input.items, u.rowPtr, u.indx, u.vals));
writeCsr(path.join(input.dir, 'ratings_by_item.ycsr'), new Csr(input.items, input.users, i.rowPtr, i.indx, i.vals));
console.log(JSON.stringify({ ok: true }));
