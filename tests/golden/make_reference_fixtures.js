// make_reference_fixtures.js -- executes the REFERENCE's own hot-path source under Node and records inputs and
// outputs as fixtures (SURVEY.md 8c, "partial oracle that is possible here").
//
//   node tests/golden/make_reference_fixtures.js /root/reference tests/golden      (build container only)
//
// What runs verbatim, loaded from <reference>/lib/emf/EmfWorker.js and EmfBase.js (never copied):
//   EmfWorker.prototype.mw_calcTrainAlsPortion (EmfWorker.js:169-261), mw_calcRmsePortion (:266-315),
//   EmfBase.getLatentFactorsPartData / copySubFixedFactors / getFactorsRowSync / predictSync / _alsPredict
//   (EmfBase.js:518-555,702-718,785-788,825-827) -- i.e. the portion-buffer parsing, the row offsets, the
//   lambda * n regularisation, the in-place placement of the solved row and the RMSE accumulation.
// What is a STAND-IN (the reference's arithmetic lives in the un-vendored forks vectorious-plus / nblas-plus,
// package.json:26): Matrix / Vector / BLAS below, written from the call sites alone -- gemm(Trans, NoTrans),
// diagonal(), add(), transposed(), multiply(), solveSquare() as LU with partial pivoting, transpose(out),
// BufCopy, row(), dot -- in float64, or rounding every operation to float32 for Float32Array data.
// So these fixtures pin the reference's DATA FLOW, not its BLAS: parity stays "unpinned" at that boundary
// (oracle/als_oracle.c says the same), but the restated buffer handling is now checked against an
// execution of the reference's source instead of a reading of it.
//
// Output: <out>/reference_harness.json -- data only (inputs, the solved matrix after the call, the
// 'completedPortion' messages).  Nothing of the reference's text is written anywhere.
'use strict';
const Module = require('module');
const path = require('path');
const fs = require('fs');

const refRoot = path.resolve(process.argv[2] || '/root/reference');
const outDir = path.resolve(process.argv[3] || __dirname);

// ---- stand-in for vectorious-plus / nblas-plus ------------------------------------------------------------
const rnd = (T, x) => (T === Float32Array ? Math.fround(x) : x);
class Matrix {
  constructor(data, opts) {
    opts = opts || {};
    this.type = opts.type || (data ? data.constructor : Float64Array);
    this._shape = opts.shape ? opts.shape.slice() : [0, 0];
    this.data = data || new this.type(this._shape[0] * this._shape[1]);
    this._t = false;
  }
  get shape() { return this._t ? [this._shape[1], this._shape[0]] : this._shape; }
  set shape(s) { this._shape = s.slice(); this._t = false; }
  diagonal(v) {  // sets the diagonal (call site: lambda.diagonal(_lambda * _n), EmfWorker.js:233-234)
    const n = this._shape[0], m = this._shape[1];
    for (let i = 0; i < Math.min(n, m); i++) this.data[i * m + i] = v;
    return this;
  }
  add(o) {  // in place
    for (let i = 0; i < this.data.length; i++) this.data[i] = rnd(this.type, this.data[i] + o.data[i]);
    return this;
  }
  transposed() { this._t = true; return this; }  // a view flag: "factorsCount x cols" (EmfWorker.js:243)
  at(i, j) { return this._t ? this.data[j * this._shape[1] + i] : this.data[i * this._shape[1] + j]; }
  multiply(o) {
    const [n, kk] = this.shape, m = o.shape[1], T = this.type;
    const out = new Matrix(null, { shape: [n, m], type: T });
    for (let i = 0; i < n; i++)
      for (let j = 0; j < m; j++) {
        let s = 0;
        for (let l = 0; l < kk; l++) s = rnd(T, s + rnd(T, this.at(i, l) * o.at(l, j)));
        out.data[i * m + j] = s;
      }
    return out;
  }
  static solveSquare(A, B, X) {  // gesv-class: LU with partial pivoting, one right-hand side; A is overwritten
    const n = A._shape[0], T = A.type, a = A.data, b = new T(B.data);
    for (let p = 0; p < n; p++) {
      let piv = p, best = Math.abs(a[p * n + p]);
      for (let i = p + 1; i < n; i++) if (Math.abs(a[i * n + p]) > best) { best = Math.abs(a[i * n + p]); piv = i; }
      if (piv != p) {
        for (let j = 0; j < n; j++) { const t = a[p * n + j]; a[p * n + j] = a[piv * n + j]; a[piv * n + j] = t; }
        const t = b[p]; b[p] = b[piv]; b[piv] = t;
      }
      for (let i = p + 1; i < n; i++) {
        const f = rnd(T, a[i * n + p] / a[p * n + p]);
        a[i * n + p] = f;
        for (let j = p + 1; j < n; j++) a[i * n + j] = rnd(T, a[i * n + j] - rnd(T, f * a[p * n + j]));
        b[i] = rnd(T, b[i] - rnd(T, f * b[p]));
      }
    }
    for (let i = n - 1; i >= 0; i--) {
      let s = b[i];
      for (let j = i + 1; j < n; j++) s = rnd(T, s - rnd(T, a[i * n + j] * b[j]));
      b[i] = rnd(T, s / a[i * n + i]);
    }
    X.data.set(b);
    return X;
  }
  transpose(out) {  // writes this^T into out (call site: tmp.transpose(latentFactorsPart), EmfWorker.js:247)
    const [n, m] = this.shape;
    for (let i = 0; i < n; i++) for (let j = 0; j < m; j++) out.data[j * n + i] = this.at(i, j);
    return out;
  }
  row(i, copy) {
    const m = this._shape[1];
    const d = copy === false ? this.data.subarray(i * m, (i + 1) * m) : this.data.slice(i * m, (i + 1) * m);
    return new Vector(d, { length: m });
  }
}
class Vector {
  constructor(data, opts) { this.data = data || new ((opts && opts.type) || Float64Array)((opts && opts.length) || 0); this.length = this.data.length; }
  dot(o) {
    const T = this.data.constructor;
    let s = 0;
    for (let i = 0; i < this.data.length; i++) s = rnd(T, s + rnd(T, this.data[i] * o.data[i]));
    return s;
  }
}
const BLAS = {
  Trans: 112, NoTrans: 111,
  // row-major C[m x n] = A^T B with A [k x m], B [k x n] (the only form the path uses, EmfWorker.js:231-232)
  gemm(a, b, c, m, n, k, transA, transB) {
    if (transA != BLAS.Trans || transB != BLAS.NoTrans) throw new Error('stub gemm: only (Trans, NoTrans)');
    const T = c.constructor;
    for (let i = 0; i < m; i++)
      for (let j = 0; j < n; j++) {
        let s = 0;
        for (let l = 0; l < k; l++) s = rnd(T, s + rnd(T, a[l * m + i] * b[l * n + j]));
        c[i * n + j] = s;
      }
  },
  BufCopy(dst, dstOff, src, srcOff, bytes) {
    new Uint8Array(dst.buffer, dst.byteOffset + dstOff, bytes).set(new Uint8Array(src.buffer, src.byteOffset + srcOff, bytes));
  },
};
const vectorious = { Matrix, Vector, SpMatrix: class {}, SpVector: class {}, BLAS };

// ---- everything else the reference's module graph asks for, as inert stubs -------------------------------------
const inert = new Proxy(function () { return inert; }, { get: (t, p) => (p === 'then' ? undefined : inert), apply: () => inert, construct: () => inert });
const stubs = {
  'vectorious-plus': vectorious, 'quick-tcp-socket': { TcpSocket: class {}, ReadBufferStream: class {}, WriteBufferStream: class {} },
  'shm-typed-array': inert, deepmerge: Object.assign((a, b) => Object.assign({}, a, b), { all: (l) => Object.assign({}, ...l) }),
  underscore: { _: inert }, 'pg-promise': () => inert, co: inert, progress: class {}, redis: inert, 'node-cleanup': () => {},
  'knuth-shuffle': inert,
};
const realLoad = Module._load;
Module._load = function (request, parent, isMain) {
  if (stubs[request]) return stubs[request];
  if (request.endsWith('cpp_utils/cpp_utils')) return inert;  // the dead native gather (cpp_utils/als_utils.cc), never called
  return realLoad.apply(this, arguments);
};
const EmfWorker = require(path.join(refRoot, 'lib', 'emf', 'EmfWorker.js'));

// ---- drive the reference's methods on synthetic portions ---------------------------------------------------
function lcg(seed) { let s = seed >>> 0; return () => ((s = (Math.imul(s, 1664525) + 1013904223) >>> 0) / 4294967296); }

function makeCase(name, k, users, items, double, stepType, seed) {
  const T = double ? Float64Array : Float32Array, r = lcg(seed);
  const U = new T(users * k), V = new T(items * k);
  for (let i = 0; i < U.length; i++) U[i] = (r() - 0.5) * 0.6;
  for (let i = 0; i < V.length; i++) V[i] = (r() - 0.5) * 0.6;
  const solvedRows = stepType == 'byUser' ? users : items, fixedRows = stepType == 'byUser' ? items : users;
  // a portion: some rows of the solved side (ids with gaps, ascending), each with 1 .. 2k ratings
  const rows = [0], indx = [], vals = [];
  for (let id = 1; id < solvedRows; id += 1 + Math.floor(r() * 3)) {
    const cols = 1 + Math.floor(r() * Math.min(fixedRows, 2 * k));
    const picked = new Set();
    while (picked.size < cols) picked.add(Math.floor(r() * fixedRows));
    const ids = Array.from(picked).sort((a, b) => a - b);
    rows.push(id, cols);
    rows[0]++;
    for (const c of ids) { indx.push(c); vals.push(1 + Math.floor(r() * 5)); }
  }
  const w = Object.create(EmfWorker.prototype);
  w.options = { useDoublePrecision: double, factorsCount: k, lowmem: false, alg: 'als', als: { userFactReg: 0.05, itemFactReg: 0.07 } };
  w.stats = { totalUsersCount: users, totalItemsCount: items };
  w.userFactors = new Matrix(U, { shape: [users, k] });
  w.itemFactors = new Matrix(V, { shape: [items, k] });
  w.globalAvgShift = 0.25;
  w.stepType = stepType;
  const msgs = [];
  w.process = { emit: (evt, m) => msgs.push(Object.assign({ evt }, m, { time: undefined, memoryUsage: undefined })) };
  Object.defineProperty(w, 'memoryUsage', { value: null });
  const alsRows = Int32Array.from(rows), alsIndx = Int32Array.from(indx), alsVals = T.from(vals);
  const before = { U: Array.from(U), V: Array.from(V) };
  w.portionBuffer = { alsRows, alsIndx, alsVals, factorsBuffer: null };
  w.mw_calcTrainAlsPortion({ portionNo: 3 });
  const after = { U: Array.from(U), V: Array.from(V) };
  // the RMSE pass over the same buffers read as a by-user portion (ids must be users / items)
  let rmse = null;
  if (stepType == 'byUser') {
    w.portionBuffer = { rmseRows: alsRows, rmseIndx: alsIndx, rmseVals: alsVals };
    w.mw_calcRmsePortion({ portionNo: 5 });
    rmse = msgs[msgs.length - 1];
  }
  return { name, k, users, items, useDoublePrecision: double, stepType, lambda: stepType == 'byUser' ? 0.05 : 0.07, globalAvgShift: 0.25,
    alsRows: rows, alsIndx: indx, alsVals: vals, before, after, completedPortion: msgs[0], rmseCompletedPortion: rmse };
}

const cases = [
  makeCase('byUser_f64_k7', 7, 23, 31, true, 'byUser', 11),
  makeCase('byItem_f64_k20', 20, 40, 27, true, 'byItem', 12),
  makeCase('byUser_f32_k20', 20, 35, 50, false, 'byUser', 13),
  makeCase('byItem_f32_k12', 12, 30, 22, false, 'byItem', 14),
];
fs.writeFileSync(path.join(outDir, 'reference_harness.json'), JSON.stringify({
  generator: 'tests/golden/make_reference_fixtures.js (runs <reference>/lib/emf/EmfWorker.js verbatim; BLAS/LAPACK stand-in in plain JS)',
  node: process.version, cases }));
console.log('wrote', cases.length, 'cases:', cases.map((c) => c.name + ' rows=' + c.alsRows[0] + ' ratings=' + c.alsIndx.length).join(', '));
