/**
 * Binary CSR files (SURVEY.md 8f, N2): the on-disk input of a training run, replacing the
 * per-portion SQL fetch + packer of the reference (lib/emf/EmfMaster.js:501-614).
 *
 * Layout (include/ycnr_als.h), little endian, no padding:
 *   "YCSR", uint32 version = 1, uint32 dtype (0 float32 / 1 float64), uint32 flags (bit 0: rows sorted
 *   by column id), int64 rows, cols, nnz, int64 rowPtr[rows + 1], int32 indx[nnz], T vals[nnz]
 * python/ycnr_als/csrfile.py writes the same bytes.
 */
'use strict';
const fs = require('fs');
const { Csr } = require('./Dataset');

const HEADER_BYTES = 40;

function writeCsr(path, csr, sortedRows) {
  const isDouble = csr.vals.constructor === Float64Array;
  if (!isDouble && csr.vals.constructor !== Float32Array) throw new Error('invalid type!');
  const nnz = csr.vals.length;
  const head = Buffer.alloc(HEADER_BYTES);
  head.write('YCSR', 0, 'latin1');
  head.writeUInt32LE(1, 4);
  head.writeUInt32LE(isDouble ? 1 : 0, 8);
  head.writeUInt32LE(sortedRows === false ? 0 : 1, 12);
  head.writeBigInt64LE(BigInt(csr.rows), 16);
  head.writeBigInt64LE(BigInt(csr.cols), 24);
  head.writeBigInt64LE(BigInt(nnz), 32);
  const rp = new BigInt64Array(csr.rows + 1);
  for (let r = 0; r <= csr.rows; r++) rp[r] = BigInt(csr.rowPtr[r]);
  const fd = fs.openSync(path, 'w');
  try {
    fs.writeSync(fd, head);
    fs.writeSync(fd, Buffer.from(rp.buffer));
    fs.writeSync(fd, Buffer.from(csr.indx.buffer, csr.indx.byteOffset, nnz * 4));
    fs.writeSync(fd, Buffer.from(csr.vals.buffer, csr.vals.byteOffset, nnz * (isDouble ? 8 : 4)));
  } finally {
    fs.closeSync(fd);
  }
}

function readCsr(path) {
  const buf = fs.readFileSync(path);
  if (buf.length < HEADER_BYTES || buf.toString('latin1', 0, 4) !== 'YCSR' || buf.readUInt32LE(4) !== 1)
    throw new Error(path + ': not a YCSR version 1 file');
  const dtype = buf.readUInt32LE(8);
  const rows = Number(buf.readBigInt64LE(16)), cols = Number(buf.readBigInt64LE(24)), nnz = Number(buf.readBigInt64LE(32));
  const ts = dtype ? 8 : 4;
  if (dtype > 1 || buf.length !== HEADER_BYTES + 8 * (rows + 1) + 4 * nnz + ts * nnz) throw new Error(path + ': truncated or inconsistent');
  let off = HEADER_BYTES;
  const ab = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.length);  // aligned copy
  const rp64 = new BigInt64Array(ab.slice(off, off + 8 * (rows + 1)));
  off += 8 * (rows + 1);
  const rowPtr = new Float64Array(rows + 1);  // exact below 2^53 (the addon accepts it as row pointers)
  for (let r = 0; r <= rows; r++) rowPtr[r] = Number(rp64[r]);
  const indx = new Int32Array(ab.slice(off, off + 4 * nnz));
  off += 4 * nnz;
  const vals = dtype ? new Float64Array(ab.slice(off, off + 8 * nnz)) : new Float32Array(ab.slice(off, off + 4 * nnz));
  if (rowPtr[0] !== 0 || rowPtr[rows] !== nnz) throw new Error(path + ': truncated or inconsistent');
  return new Csr(rows, cols, rowPtr, indx, vals);
}

module.exports = { writeCsr, readCsr };
