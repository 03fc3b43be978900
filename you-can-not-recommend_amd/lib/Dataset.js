/**
 * Ratings container for the trainer: replaces the PostgreSQL tables the reference reads
 * portion by portion (lib/emf/EmfMaster.js:501-541).  A data set is the same matrix in CSR by
 * user and by item (train = dataset_type 1+2), plus the validate (2) and test (3) parts by
 * user.  Ids are 0-based (db id - 1, EmfMaster.js:584-586), columns ascend within a row.
 */
'use strict';

const fs = require('fs');

class Csr {
  constructor(rows, cols, rowPtr, indx, vals) {
    this.rows = rows; this.cols = cols;
    this.rowPtr = rowPtr; this.indx = indx; this.vals = vals;
  }
  get nnz() { return this.rowPtr[this.rows]; }
  count(r) { return this.rowPtr[r + 1] - this.rowPtr[r]; }

  /**
   * The ratings a pass of the REFERENCE really consumes when this matrix is fed to it in the portions that end at the
   * 1-based inclusive row ids `portionsRowIdTo` (lib/emf/EmfLord.js:571-592): its packer records the row that is open when a
   * portion's last rating arrives BEFORE counting that rating (lib/emf/EmfMaster.js:594-603), so every portion loses its last
   * rating.  Returns a new Csr; a row left without ratings is not solved.  Opt-in (options.dropLastRatingPerPortion).
   */
  withoutLastRatingPerPortion(portionsRowIdTo) {
    const drop = new Set();
    let begin = 0;
    for (const end of portionsRowIdTo) {
      if (!(end > 0) || end > this.rows) continue;
      if (this.rowPtr[end] > this.rowPtr[begin]) drop.add(this.rowPtr[end] - 1);
      begin = end;
    }
    const n = this.nnz - drop.size;
    const rowPtr = new Float64Array(this.rows + 1), indx = new Int32Array(n), vals = new this.vals.constructor(n);
    let m = 0;
    for (let r = 0; r < this.rows; r++) {
      for (let p = this.rowPtr[r]; p < this.rowPtr[r + 1]; p++) {
        if (drop.has(p)) continue;
        indx[m] = this.indx[p]; vals[m] = this.vals[p]; m++;
      }
      rowPtr[r + 1] = m;
    }
    return new Csr(this.rows, this.cols, rowPtr, indx, vals);
  }

  /** The ratings whose column id lies in [lo, hi): same shape, fewer ratings (the item side of a rank that holds the users [lo, hi)). */
  columnsBetween(lo, hi) {
    let n = 0;
    for (let p = 0; p < this.nnz; p++) if (this.indx[p] >= lo && this.indx[p] < hi) n++;
    const rowPtr = new Float64Array(this.rows + 1), indx = new Int32Array(n), vals = new this.vals.constructor(n);
    let m = 0;
    for (let r = 0; r < this.rows; r++) {
      for (let p = this.rowPtr[r]; p < this.rowPtr[r + 1]; p++) {
        if (this.indx[p] >= lo && this.indx[p] < hi) { indx[m] = this.indx[p]; vals[m] = this.vals[p]; m++; }
      }
      rowPtr[r + 1] = m;
    }
    return new Csr(this.rows, this.cols, rowPtr, indx, vals);
  }

  /**
   * Rows [rowBegin, rowEnd) in the reference's portion-buffer format (alsRows / alsIndx / alsVals, lib/emf/EmfMaster.js:589-609),
   * rows without ratings omitted.  dropLast: the row table the reference's packer writes for these ratings, bug for bug (the
   * last row one rating short; a last row of one rating not recorded, or recorded with cols = 0 when it is the portion's only
   * rating); alsIndx / alsVals hold every rating either way.
   */
  toPortion(rowBegin, rowEnd, dropLast) {
    const b = this.rowPtr[rowBegin], e = this.rowPtr[rowEnd];
    const ids = [], cols = [];
    for (let r = rowBegin; r < rowEnd; r++) if (this.count(r) > 0) { ids.push(r); cols.push(this.count(r)); }
    if (dropLast && ids.length) {
      if (cols[cols.length - 1] > 1 || e - b == 1) cols[cols.length - 1]--;
      else { ids.pop(); cols.pop(); }
    }
    const alsRows = new Int32Array(1 + 2 * ids.length);
    alsRows[0] = ids.length;
    ids.forEach((id, i) => { alsRows[1 + 2 * i] = id; alsRows[2 + 2 * i] = cols[i]; });
    return { alsRows, alsIndx: this.indx.slice(b, e), alsVals: this.vals.slice(b, e) };
  }
}

/** Counting sort of (r, c, v) triplets into CSR; triplets must be unique per (r, c). */
function csrFromTriplets(rows, cols, r, c, v, ValClass, n) {
  n = (n === undefined) ? r.length : n;
  const rowPtr = new Float64Array(rows + 1);
  for (let i = 0; i < n; i++) rowPtr[r[i] + 1]++;
  for (let i = 0; i < rows; i++) rowPtr[i + 1] += rowPtr[i];
  const next = Float64Array.from(rowPtr);
  const indx = new Int32Array(n), vals = new ValClass(n);
  // columns ascend within a row when the input is visited in ascending column order
  const order = new Int32Array(n);
  for (let i = 0; i < n; i++) order[i] = i;
  order.sort((a, b) => (c[a] - c[b]));
  for (let j = 0; j < n; j++) {
    const i = order[j], p = next[r[i]]++;
    indx[p] = c[i]; vals[p] = v[i];
  }
  return new Csr(rows, cols, rowPtr, indx, vals);
}

/**
 * Deterministic 32-bit PRNG (mulberry32) + Box-Muller: the reference's randomNormal and
 * knuth-shuffle are unseeded (EmfBase.js:486-493, EmfLord.js:463); seeding them makes a GPU
 * run and a CPU run start from identical bytes.
 */
function rng(seed) {
  let a = seed >>> 0;
  const next = function () {
    a = (a + 0x6D2B79F5) >>> 0;
    let t = a;
    t = Math.imul(t ^ (t >>> 15), t | 1);
    t ^= t + Math.imul(t ^ (t >>> 7), t | 61);
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
  next.normal = function () {
    let u = 0;
    while (u === 0) u = next();
    return Math.sqrt(-2.0 * Math.log(u)) * Math.cos(2.0 * Math.PI * next());
  };
  return next;
}

class Dataset {
  /**
   * triplets: {user: Int32Array, item: Int32Array, rating: Float32Array|Float64Array} 0-based,
   * type: optional Int8Array of dataset_type 1 / 2 / 3 per triplet (default: all train).
   */
  constructor(totalUsersCount, totalItemsCount, triplets, type, ValClass) {
    ValClass = ValClass || Float32Array;
    const n = triplets.user.length;
    const pick = (pred) => {
      const u = new Int32Array(n), it = new Int32Array(n), v = new ValClass(n);
      let m = 0;
      for (let i = 0; i < n; i++) if (pred(type ? type[i] : 1)) { u[m] = triplets.user[i]; it[m] = triplets.item[i]; v[m] = triplets.rating[i]; m++; }
      return { u, it, v, m };
    };
    const tr = pick((t) => t === 1 || t === 2);  // EmfMaster.js:502-503
    this.totalUsersCount = totalUsersCount;
    this.totalItemsCount = totalItemsCount;
    this.trainByUser = csrFromTriplets(totalUsersCount, totalItemsCount, tr.u, tr.it, tr.v, ValClass, tr.m);
    this.trainByItem = csrFromTriplets(totalItemsCount, totalUsersCount, tr.it, tr.u, tr.v, ValClass, tr.m);
    const va = pick((t) => t === 2), te = pick((t) => t === 3);
    this.validate = va.m ? csrFromTriplets(totalUsersCount, totalItemsCount, va.u, va.it, va.v, ValClass, va.m) : null;
    this.test = te.m ? csrFromTriplets(totalUsersCount, totalItemsCount, te.u, te.it, te.v, ValClass, te.m) : null;
    let s = 0;
    for (let i = 0; i < n; i++) s += triplets.rating[i];
    this.totalRatingsAvg = n ? s / n : 0;
  }

  /** options.dropLastRatingPerPortion: every pass cut into its portions (EmfLord.splitToPortions), each without its last rating */
  withoutLastRatingPerPortion(portionsRowIdTo) {
    const out = Object.create(Dataset.prototype);
    Object.assign(out, this);
    out.trainByUser = this.trainByUser.withoutLastRatingPerPortion(portionsRowIdTo.byUser || []);
    out.trainByItem = this.trainByItem.withoutLastRatingPerPortion(portionsRowIdTo.byItem || []);
    if (this.validate) out.validate = this.validate.withoutLastRatingPerPortion((portionsRowIdTo.rmseValidate || []).map((e, i, a) => (i == a.length - 1 ? this.totalUsersCount : e)));
    if (this.test) out.test = this.test.withoutLastRatingPerPortion((portionsRowIdTo.rmseTest || []).map((e, i, a) => (i == a.length - 1 ? this.totalUsersCount : e)));
    out.portionQuirkApplied = true;
    return out;
  }

  /**
   * Per-user split into train / validate / test by a seeded shuffle, in the spirit of
   * EmfLord.doSplitToSets (lib/emf/EmfLord.js:402-505). Returns Int8Array of 1 / 2 / 3.
   */
  static splitToSets(triplets, totalUsersCount, dataSetDistr, seed) {
    const n = triplets.user.length, rnd = rng(seed || 1);
    const byUser = new Array(totalUsersCount);
    for (let i = 0; i < n; i++) (byUser[triplets.user[i]] || (byUser[triplets.user[i]] = [])).push(i);
    const type = new Int8Array(n);
    for (let u = 0; u < totalUsersCount; u++) {
      const a = byUser[u];
      if (!a) continue;
      for (let i = a.length - 1; i > 0; i--) { const j = Math.floor(rnd() * (i + 1)); const t = a[i]; a[i] = a[j]; a[j] = t; }
      for (let i = 0; i < a.length; i++) {
        const frac = (i + 0.5) / a.length * 100;
        type[a[i]] = frac < dataSetDistr[0] ? 1 : (frac < dataSetDistr[0] + dataSetDistr[1] ? 2 : 3);
      }
    }
    return type;
  }

  /**
   * MovieLens files: 'u.data' (tab separated user item rating ts, data/db-schema.sql:459-476)
   * or 'ratings.dat' (user::item::rating::ts, lib/YcnrController.js:126-133). Ids 1-based.
   */
  static readMovieLens(path) {
    const text = fs.readFileSync(path, 'utf8');
    const sep = path.endsWith('.dat') ? '::' : '\t';
    const lines = text.split('\n').filter((l) => l.length > 0);
    const user = new Int32Array(lines.length), item = new Int32Array(lines.length), rating = new Float32Array(lines.length);
    let mu = 0, mi = 0;
    lines.forEach((l, i) => {
      const p = l.split(sep);
      user[i] = parseInt(p[0]) - 1; item[i] = parseInt(p[1]) - 1; rating[i] = parseFloat(p[2]);
      if (user[i] + 1 > mu) mu = user[i] + 1;
      if (item[i] + 1 > mi) mi = item[i] + 1;
    });
    return { totalUsersCount: mu, totalItemsCount: mi, triplets: { user, item, rating } };
  }
}

module.exports = { Csr, Dataset, csrFromTriplets, rng };
