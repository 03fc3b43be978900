// ycnr_als_napi.cc -- thin N-API shim over the C ABI of libycnr_als.so (include/ycnr_als.h).
//
// It replaces the reference's node-gyp addon cpp_utils (binding.gyp:3-16,
// cpp_utils/cpp_utils.cc:3-8): same shape -- s/d-prefixed functions that borrow the raw
// memory of JS typed arrays for the duration of one synchronous call
// (cpp_utils/cpp_utils.h:6-7 GET_CONTENTS) -- but written against N-API instead of the V8 API
// the reference used (which no longer compiles on Node 12).  No arithmetic
// happens here; every failure of the library is rethrown as a JS Error carrying
// ycnr_last_error().
#include <node_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ycnr_als.h"

namespace {

#define NAPI_OK(call)                                            \
  do {                                                           \
    if ((call) != napi_ok) {                                     \
      napi_throw_error(env, nullptr, "N-API call failed: " #call); \
      return nullptr;                                            \
    }                                                            \
  } while (0)

napi_value throw_msg(napi_env env, const std::string &msg) {
  napi_throw_error(env, nullptr, msg.c_str());
  return nullptr;
}

napi_value throw_last(napi_env env, const char *where, long long code) {
  char buf[1200];
  snprintf(buf, sizeof buf, "%s failed (%lld): %s", where, code, ycnr_last_error());
  napi_throw_error(env, nullptr, buf);
  return nullptr;
}

struct View {
  void *data = nullptr;
  size_t length = 0;
  napi_typedarray_type type = napi_int8_array;
  bool ok = false;
};

// raw pointer of a typed array (the N-API spelling of GET_CONTENTS, cpp_utils/cpp_utils.h:6-7)
View view_of(napi_env env, napi_value v) {
  View w;
  bool is = false;
  if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return w;
  napi_value ab;
  size_t off = 0;
  if (napi_get_typedarray_info(env, v, &w.type, &w.length, &w.data, &ab, &off) != napi_ok) return w;
  w.ok = true;
  return w;
}

bool get_double(napi_env env, napi_value v, double *out) { return napi_get_value_double(env, v, out) == napi_ok; }
bool get_int(napi_env env, napi_value v, int64_t *out) {
  double d;
  if (napi_get_value_double(env, v, &d) != napi_ok) return false;
  *out = (int64_t)d;
  return true;
}

napi_value num(napi_env env, double d) {
  napi_value v;
  napi_create_double(env, d, &v);
  return v;
}

void set_num(napi_env env, napi_value obj, const char *key, double d) { napi_set_named_property(env, obj, key, num(env, d)); }

// rowPtr may arrive as Int32Array, Float64Array (exact below 2^53) or BigInt64Array
bool to_i64(const View &w, std::vector<int64_t> &out) {
  out.resize(w.length);
  switch (w.type) {
    case napi_int32_array:
      for (size_t i = 0; i < w.length; ++i) out[i] = static_cast<const int32_t *>(w.data)[i];
      return true;
    case napi_float64_array:
      for (size_t i = 0; i < w.length; ++i) out[i] = (int64_t) static_cast<const double *>(w.data)[i];
      return true;
    case napi_bigint64_array:
      memcpy(out.data(), w.data, sizeof(int64_t) * w.length);
      return true;
    default:
      return false;
  }
}

template <bool DOUBLE>
napi_value AlsCalcPortion(napi_env env, napi_callback_info info) {
  size_t argc = 7;
  napi_value a[7];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 7) return throw_msg(env, "AlsCalcPortion(lambda, k, alsRows, alsIndx, alsVals, fixedFactors, solvedFactors)");
  double lambda;
  int64_t k;
  if (!get_double(env, a[0], &lambda) || !get_int(env, a[1], &k)) return throw_msg(env, "invalid type!");
  const napi_typedarray_type ft = DOUBLE ? napi_float64_array : napi_float32_array;
  View rows = view_of(env, a[2]), indx = view_of(env, a[3]), vals = view_of(env, a[4]), fixed = view_of(env, a[5]),
       solved = view_of(env, a[6]);
  if (!rows.ok || !indx.ok || !vals.ok || !fixed.ok || !solved.ok || rows.type != napi_int32_array ||
      indx.type != napi_int32_array || vals.type != ft || fixed.type != ft || solved.type != ft)
    return throw_msg(env, "invalid type!");  // cpp_utils/cpp_utils.js:12
  if (k < 1 || rows.length < 1) return throw_msg(env, "invalid portion");
  const int32_t *r = static_cast<const int32_t *>(rows.data);
  if (r[0] < 0 || (size_t)(1 + 2 * (int64_t)r[0]) > rows.length) return throw_msg(env, "alsRows shorter than its row count");
  int64_t total = 0;
  for (int i = 0; i < r[0]; ++i) total += r[2 + 2 * i];
  if ((size_t)total > indx.length || (size_t)total > vals.length) return throw_msg(env, "alsIndx / alsVals shorter than the portion");
  int64_t n;
  if (DOUBLE)
    n = ycnr_dAlsCalcPortion(lambda, (int)k, r, static_cast<const int32_t *>(indx.data), static_cast<const double *>(vals.data),
                             static_cast<const double *>(fixed.data), (int64_t)(fixed.length / k),
                             static_cast<double *>(solved.data), (int64_t)(solved.length / k));
  else
    n = ycnr_sAlsCalcPortion(lambda, (int)k, r, static_cast<const int32_t *>(indx.data), static_cast<const float *>(vals.data),
                             static_cast<const float *>(fixed.data), (int64_t)(fixed.length / k),
                             static_cast<float *>(solved.data), (int64_t)(solved.length / k));
  if (n < 0) return throw_last(env, DOUBLE ? "dAlsCalcPortion" : "sAlsCalcPortion", n);
  return num(env, (double)n);
}

// s/dAlsPinFixedFactors(fixedFactors, k): keep the step's fixed matrix on the device for the portion calls that follow
template <bool DOUBLE>
napi_value AlsPinFixedFactors(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value a[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  int64_t k;
  if (argc < 2 || !get_int(env, a[1], &k) || k < 1) return throw_msg(env, "AlsPinFixedFactors(fixedFactors, k)");
  View fixed = view_of(env, a[0]);
  if (!fixed.ok || fixed.type != (DOUBLE ? napi_float64_array : napi_float32_array)) return throw_msg(env, "invalid type!");
  int rc = DOUBLE ? ycnr_dAlsPinFixedFactors(static_cast<const double *>(fixed.data), (int64_t)(fixed.length / k), (int)k)
                  : ycnr_sAlsPinFixedFactors(static_cast<const float *>(fixed.data), (int64_t)(fixed.length / k), (int)k);
  if (rc) return throw_last(env, "AlsPinFixedFactors", rc);
  return nullptr;
}
napi_value AlsUnpinFixedFactors(napi_env env, napi_callback_info) {
  ycnr_AlsUnpinFixedFactors();
  return nullptr;
}
napi_value AlsReleasePortionState(napi_env env, napi_callback_info) {
  ycnr_AlsReleasePortionState();
  (void)env;
  return nullptr;
}

template <bool DOUBLE>
napi_value RmsePortion(napi_env env, napi_callback_info info) {
  size_t argc = 7;
  napi_value a[7];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 7) return throw_msg(env, "RmsePortion(k, rmseRows, rmseIndx, rmseVals, userFactors, itemFactors, globalAvgShift)");
  int64_t k;
  double shift;
  if (!get_int(env, a[0], &k) || !get_double(env, a[6], &shift)) return throw_msg(env, "invalid type!");
  const napi_typedarray_type ft = DOUBLE ? napi_float64_array : napi_float32_array;
  View rows = view_of(env, a[1]), indx = view_of(env, a[2]), vals = view_of(env, a[3]), U = view_of(env, a[4]), I = view_of(env, a[5]);
  if (!rows.ok || !indx.ok || !vals.ok || !U.ok || !I.ok || rows.type != napi_int32_array || indx.type != napi_int32_array ||
      vals.type != ft || U.type != ft || I.type != ft)
    return throw_msg(env, "invalid type!");
  if (k < 1) return throw_msg(env, "invalid k");
  double out[3];
  int rc;
  if (DOUBLE)
    rc = ycnr_dRmsePortion((int)k, static_cast<const int32_t *>(rows.data), static_cast<const int32_t *>(indx.data),
                           static_cast<const double *>(vals.data), static_cast<const double *>(U.data), (int64_t)(U.length / k),
                           static_cast<const double *>(I.data), (int64_t)(I.length / k), shift, out);
  else
    rc = ycnr_sRmsePortion((int)k, static_cast<const int32_t *>(rows.data), static_cast<const int32_t *>(indx.data),
                           static_cast<const float *>(vals.data), static_cast<const float *>(U.data), (int64_t)(U.length / k),
                           static_cast<const float *>(I.data), (int64_t)(I.length / k), shift, out);
  if (rc) return throw_last(env, "RmsePortion", rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  set_num(env, o, "rSumDiff2", out[0]);  // the 'completedPortion' fields, EmfWorker.js:311-313
  set_num(env, o, "rCnt", out[1]);
  set_num(env, o, "rSum", out[2]);
  return o;
}

napi_value DeviceCount(napi_env env, napi_callback_info) {
  int n = ycnr_device_count();
  if (n < 0) return throw_last(env, "deviceCount", n);
  return num(env, n);
}

napi_value LastError(napi_env env, napi_callback_info) {
  napi_value s;
  napi_create_string_utf8(env, ycnr_last_error(), NAPI_AUTO_LENGTH, &s);
  return s;
}

napi_value Version(napi_env env, napi_callback_info) { return num(env, ycnr_version()); }

// ---------------------------------------------------------------- resident trainer

struct Handle {
  ycnr_als *h = nullptr;
  int dtype = YCNR_F32;
  int k = 0;
  int64_t rows[2] = {0, 0};
};

void finalize_handle(napi_env, void *data, void *) {
  Handle *hd = static_cast<Handle *>(data);
  if (hd) {
    if (hd->h) ycnr_als_destroy(hd->h);
    delete hd;
  }
}

Handle *handle_of(napi_env env, napi_value v) {
  void *p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) return nullptr;
  Handle *hd = static_cast<Handle *>(p);
  return hd->h ? hd : nullptr;
}

bool named_double(napi_env env, napi_value obj, const char *key, double *out, double dflt) {
  napi_value v;
  bool has = false;
  *out = dflt;
  if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return true;
  if (napi_get_named_property(env, obj, key, &v) != napi_ok) return false;
  napi_valuetype t;
  napi_typeof(env, v, &t);
  if (t == napi_boolean) {
    bool b;
    napi_get_value_bool(env, v, &b);
    *out = b ? 1 : 0;
    return true;
  }
  if (t == napi_undefined || t == napi_null) return true;
  return napi_get_value_double(env, v, out) == napi_ok;
}

// create({device, useDoublePrecision, factorsCount, totalUsersCount, totalItemsCount,
//         userFactReg, itemFactReg, chunkRatings, flags}) -> handle
napi_value Create(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value a[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 1) return throw_msg(env, "create(options)");
  double device, dbl, k, users, items, ur, ir, chunk, flags;
  if (!named_double(env, a[0], "device", &device, 0) || !named_double(env, a[0], "useDoublePrecision", &dbl, 0) ||
      !named_double(env, a[0], "factorsCount", &k, 100) || !named_double(env, a[0], "totalUsersCount", &users, 0) ||
      !named_double(env, a[0], "totalItemsCount", &items, 0) || !named_double(env, a[0], "userFactReg", &ur, 0.05) ||
      !named_double(env, a[0], "itemFactReg", &ir, 0.05) || !named_double(env, a[0], "chunkRatings", &chunk, 0) ||
      !named_double(env, a[0], "flags", &flags, 0))
    return throw_msg(env, "create: bad option value");
  ycnr_als_options o;
  memset(&o, 0, sizeof o);
  o.struct_size = (int32_t)sizeof o;
  o.device = (int32_t)device;
  o.dtype = dbl != 0 ? YCNR_F64 : YCNR_F32;
  o.factorsCount = (int32_t)k;
  o.totalUsersCount = (int64_t)users;
  o.totalItemsCount = (int64_t)items;
  o.userFactReg = ur;
  o.itemFactReg = ir;
  o.chunkRatings = (int32_t)chunk;
  o.flags = (int32_t)flags;
  Handle *hd = new Handle();
  int rc = ycnr_als_create(&o, &hd->h);
  if (rc) {
    delete hd;
    return throw_last(env, "create", rc);
  }
  hd->dtype = o.dtype;
  hd->k = o.factorsCount;
  hd->rows[0] = o.totalUsersCount;
  hd->rows[1] = o.totalItemsCount;
  napi_value ext;
  if (napi_create_external(env, hd, finalize_handle, nullptr, &ext) != napi_ok) {
    finalize_handle(env, hd, nullptr);
    return throw_msg(env, "napi_create_external failed");
  }
  return ext;
}

napi_value Destroy(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value a[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  void *p = nullptr;
  if (argc < 1 || napi_get_value_external(env, a[0], &p) != napi_ok || !p) return throw_msg(env, "destroy(handle)");
  Handle *hd = static_cast<Handle *>(p);
  if (hd->h) {
    ycnr_als_destroy(hd->h);
    hd->h = nullptr;
  }
  return nullptr;
}

template <bool RMSE>
napi_value SetRatings(napi_env env, napi_callback_info info) {
  size_t argc = 7;
  napi_value a[7];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 5) return throw_msg(env, "setRatings(handle, side, rowPtr, indx, vals[, rowBegin, rowEnd])");
  Handle *hd = handle_of(env, a[0]);
  int64_t side;
  if (!hd || !get_int(env, a[1], &side)) return throw_msg(env, "bad handle or side");
  View rp = view_of(env, a[2]), indx = view_of(env, a[3]), vals = view_of(env, a[4]);
  const napi_typedarray_type ft = hd->dtype == YCNR_F64 ? napi_float64_array : napi_float32_array;
  if (!rp.ok || !indx.ok || !vals.ok || indx.type != napi_int32_array || vals.type != ft) return throw_msg(env, "invalid type!");
  std::vector<int64_t> rowPtr;
  if (!to_i64(rp, rowPtr) || rowPtr.empty()) return throw_msg(env, "invalid type!");
  const int64_t totalRows = RMSE ? hd->rows[0] : hd->rows[side == YCNR_BY_ITEM ? 1 : 0];
  if ((int64_t)rowPtr.size() != totalRows + 1) return throw_msg(env, "rowPtr must have rows + 1 entries");
  int64_t rb = 0, re = totalRows;
  if (argc >= 7) {
    if (!get_int(env, a[5], &rb) || !get_int(env, a[6], &re)) return throw_msg(env, "bad shard range");
  }
  if (rowPtr.back() < 0 || (size_t)rowPtr.back() > indx.length || (size_t)rowPtr.back() > vals.length)
    return throw_msg(env, "indx / vals shorter than rowPtr says");
  int rc = RMSE ? ycnr_als_set_rmse_ratings(hd->h, (int)side, rowPtr.data(), static_cast<const int32_t *>(indx.data), vals.data, rb,
                                            re, YCNR_MEM_HOST)
                : ycnr_als_set_ratings(hd->h, (int)side, rowPtr.data(), static_cast<const int32_t *>(indx.data), vals.data, rb, re,
                                       YCNR_MEM_HOST);
  if (rc) return throw_last(env, RMSE ? "setRmseRatings" : "setRatings", rc);
  return nullptr;
}

napi_value SetFactors(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value a[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 3) return throw_msg(env, "setFactors(handle, side, array)");
  Handle *hd = handle_of(env, a[0]);
  int64_t side;
  if (!hd || !get_int(env, a[1], &side) || (side != 0 && side != 1)) return throw_msg(env, "bad handle or side");
  View v = view_of(env, a[2]);
  const napi_typedarray_type ft = hd->dtype == YCNR_F64 ? napi_float64_array : napi_float32_array;
  if (!v.ok || v.type != ft) return throw_msg(env, "invalid type!");
  if ((int64_t)v.length != hd->rows[side] * hd->k) return throw_msg(env, "factor array has the wrong length");
  int rc = ycnr_als_set_factors(hd->h, (int)side, v.data, YCNR_MEM_HOST);
  if (rc) return throw_last(env, "setFactors", rc);
  return nullptr;
}

napi_value GetFactors(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value a[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 3) return throw_msg(env, "getFactors(handle, side, array[, rowBegin, rowCount])");
  Handle *hd = handle_of(env, a[0]);
  int64_t side;
  if (!hd || !get_int(env, a[1], &side) || (side != 0 && side != 1)) return throw_msg(env, "bad handle or side");
  View v = view_of(env, a[2]);
  const napi_typedarray_type ft = hd->dtype == YCNR_F64 ? napi_float64_array : napi_float32_array;
  if (!v.ok || v.type != ft) return throw_msg(env, "invalid type!");
  int64_t rb = 0, rc_ = hd->rows[side];
  if (argc >= 5 && (!get_int(env, a[3], &rb) || !get_int(env, a[4], &rc_))) return throw_msg(env, "bad row range");
  if ((int64_t)v.length < rc_ * hd->k) return throw_msg(env, "factor array too short");
  int rc = ycnr_als_get_factors(hd->h, (int)side, v.data, rb, rc_, YCNR_MEM_HOST);
  if (rc) return throw_last(env, "getFactors", rc);
  return nullptr;
}

napi_value Step(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value a[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Handle *hd = argc >= 2 ? handle_of(env, a[0]) : nullptr;
  int64_t side;
  if (!hd || !get_int(env, a[1], &side)) return throw_msg(env, "step(handle, side)");
  int rc = ycnr_als_step(hd->h, (int)side);
  if (rc) return throw_last(env, "step", rc);
  ycnr_als_step_info si;
  rc = ycnr_als_last_step_info(hd->h, &si);
  if (rc) return throw_last(env, "lastStepInfo", rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  set_num(env, o, "rows", (double)si.rows);
  set_num(env, o, "ratings", (double)si.ratings);  // ratingsInPortion of 'completedPortion', summed
  set_num(env, o, "units", (double)si.units);
  set_num(env, o, "splitRows", (double)si.splitRows);
  set_num(env, o, "dualRows", (double)si.dualRows);
  set_num(env, o, "gramSlabMs", si.gramSlabMs);
  set_num(env, o, "gramSolveMs", si.gramSolveMs);
  set_num(env, o, "dualSolveMs", si.dualSolveMs);
  set_num(env, o, "reduceSolveMs", si.reduceSolveMs);
  set_num(env, o, "dualOverlapped", (double)si.dualOverlapped);
  set_num(env, o, "time", si.totalMs);  // 'time' of 'completedPortion', EmfWorker.js:258
  set_num(env, o, "parts", (double)si.parts);
  set_num(env, o, "exchangeBytes", (double)si.exchangeBytes);
  set_num(env, o, "exchangeMs", si.exchangeMs);
  set_num(env, o, "exposedExchangeMs", si.exposedExchangeMs);
  return o;
}

// ---- multi-GPU exchange (include/ycnr_als.h "multi-GPU"): the Lord process creates the id and hands it to
// the per-GPU processes it forks (process.send), replacing the TCP cluster of EmfLord.initClusterLord /
// EmfChief.initClusterChief (lib/emf/EmfLord.js:668-747, lib/emf/EmfChief.js:87-231)

// commUniqueId(transport) -> Uint8Array(128)
napi_value CommUniqueId(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value a[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  int64_t transport;
  if (argc < 1 || !get_int(env, a[0], &transport)) return throw_msg(env, "commUniqueId(transport)");
  napi_value ab, out;
  void *buf = nullptr;
  NAPI_OK(napi_create_arraybuffer(env, YCNR_COMM_ID_BYTES, &buf, &ab));
  NAPI_OK(napi_create_typedarray(env, napi_uint8_array, YCNR_COMM_ID_BYTES, ab, 0, &out));
  int rc = ycnr_comm_unique_id((int)transport, buf);
  if (rc) return throw_last(env, "commUniqueId", rc);
  return out;
}

// commInit(handle, transport, id: Uint8Array(128), rank, world)
napi_value CommInit(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value a[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Handle *hd = argc >= 5 ? handle_of(env, a[0]) : nullptr;
  int64_t transport, rank, world;
  if (!hd || !get_int(env, a[1], &transport) || !get_int(env, a[3], &rank) || !get_int(env, a[4], &world))
    return throw_msg(env, "commInit(handle, transport, id, rank, world)");
  View id = view_of(env, a[2]);
  if (!id.ok || id.type != napi_uint8_array || id.length != YCNR_COMM_ID_BYTES) return throw_msg(env, "commInit: id must be a Uint8Array(128)");
  int rc = ycnr_als_comm_init(hd->h, (int)transport, id.data, (int)rank, (int)world);
  if (rc) return throw_last(env, "commInit", rc);
  return nullptr;
}

// setRatingsSharded(handle, side, rowPtr, indx, vals, nChunks, bounds: Float64Array(world * (nChunks + 1)))
napi_value SetRatingsSharded(napi_env env, napi_callback_info info) {
  size_t argc = 7;
  napi_value a[7];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 7) return throw_msg(env, "setRatingsSharded(handle, side, rowPtr, indx, vals, nChunks, bounds)");
  Handle *hd = handle_of(env, a[0]);
  int64_t side, nChunks;
  if (!hd || !get_int(env, a[1], &side) || !get_int(env, a[5], &nChunks) || (side != 0 && side != 1)) return throw_msg(env, "bad handle, side or nChunks");
  View rp = view_of(env, a[2]), indx = view_of(env, a[3]), vals = view_of(env, a[4]), bv = view_of(env, a[6]);
  const napi_typedarray_type ft = hd->dtype == YCNR_F64 ? napi_float64_array : napi_float32_array;
  if (!rp.ok || !indx.ok || !vals.ok || !bv.ok || indx.type != napi_int32_array || vals.type != ft) return throw_msg(env, "invalid type!");
  std::vector<int64_t> rowPtr, bounds;
  if (!to_i64(rp, rowPtr) || rowPtr.empty() || !to_i64(bv, bounds)) return throw_msg(env, "invalid type!");
  if ((int64_t)rowPtr.size() != hd->rows[side] + 1) return throw_msg(env, "rowPtr must have rows + 1 entries");
  if (nChunks < 1 || bounds.empty() || bounds.size() % (size_t)(nChunks + 1) != 0) return throw_msg(env, "bounds must hold world * (nChunks + 1) row ids");
  if (rowPtr.back() < 0 || (size_t)rowPtr.back() > indx.length || (size_t)rowPtr.back() > vals.length)
    return throw_msg(env, "indx / vals shorter than rowPtr says");
  // the library checks the rank count the array describes against its communicator's world
  int rc = ycnr_als_set_ratings_sharded(hd->h, (int)side, rowPtr.data(), static_cast<const int32_t *>(indx.data), vals.data, YCNR_MEM_HOST,
                                        (int)nChunks, (int)(bounds.size() / (size_t)(nChunks + 1)), bounds.data());
  if (rc) return throw_last(env, "setRatingsSharded", rc);
  return nullptr;
}

// setRatingsBanded(handle, side, rowPtr, indx, vals, bandBounds: Float64Array(nBands + 1), rankBands: Float64Array(world + 1),
//                  ownerBounds: Float64Array(world + 1)): the side's half-step sharded by bands of columns (ycnr_als_set_ratings_banded)
napi_value SetRatingsBanded(napi_env env, napi_callback_info info) {
  size_t argc = 8;
  napi_value a[8];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 8) return throw_msg(env, "setRatingsBanded(handle, side, rowPtr, indx, vals, bandBounds, rankBands, ownerBounds)");
  Handle *hd = handle_of(env, a[0]);
  int64_t side;
  if (!hd || !get_int(env, a[1], &side) || (side != 0 && side != 1)) return throw_msg(env, "bad handle or side");
  View rp = view_of(env, a[2]), indx = view_of(env, a[3]), vals = view_of(env, a[4]), bb = view_of(env, a[5]), rb = view_of(env, a[6]), ob = view_of(env, a[7]);
  const napi_typedarray_type ft = hd->dtype == YCNR_F64 ? napi_float64_array : napi_float32_array;
  if (!rp.ok || !indx.ok || !vals.ok || !bb.ok || !rb.ok || !ob.ok || indx.type != napi_int32_array || vals.type != ft) return throw_msg(env, "invalid type!");
  std::vector<int64_t> rowPtr, bandBounds, rankBands, ownerBounds;
  if (!to_i64(rp, rowPtr) || rowPtr.empty() || !to_i64(bb, bandBounds) || !to_i64(rb, rankBands) || !to_i64(ob, ownerBounds)) return throw_msg(env, "invalid type!");
  if ((int64_t)rowPtr.size() != hd->rows[side] + 1) return throw_msg(env, "rowPtr must have rows + 1 entries");
  if (bandBounds.size() < 2 || rankBands.size() < 2 || rankBands.size() != ownerBounds.size()) return throw_msg(env, "bandBounds: nBands + 1 ids; rankBands, ownerBounds: world + 1 each");
  if (rowPtr.back() < 0 || (size_t)rowPtr.back() > indx.length || (size_t)rowPtr.back() > vals.length)
    return throw_msg(env, "indx / vals shorter than rowPtr says");
  // (the library checks rankBands / ownerBounds against its communicator's world: it reads world + 1 entries of each)
  int rc = ycnr_als_set_ratings_banded(hd->h, (int)side, rowPtr.data(), static_cast<const int32_t *>(indx.data), vals.data, YCNR_MEM_HOST,
                                       (int)bandBounds.size() - 1, bandBounds.data(), rankBands.data(), ownerBounds.data());
  if (rc) return throw_last(env, "setRatingsBanded", rc);
  return nullptr;
}

// commInfo(handle) -> {transport, rank, world, rcclRanks}: ycnr_als_comm_info (rcclRanks: what ncclCommCount says, -1 off RCCL)
napi_value CommInfo(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value a[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Handle *hd = argc >= 1 ? handle_of(env, a[0]) : nullptr;
  if (!hd) return throw_msg(env, "commInfo(handle)");
  int32_t v[4] = {0, 0, 1, -1};
  int rc = ycnr_als_comm_info(hd->h, v);
  if (rc) return throw_last(env, "commInfo", rc);
  napi_value o;
  NAPI_OK(napi_create_object(env, &o));
  set_num(env, o, "transport", (double)v[0]);
  set_num(env, o, "rank", (double)v[1]);
  set_num(env, o, "world", (double)v[2]);
  set_num(env, o, "rcclRanks", (double)v[3]);
  return o;
}

// lastRmseMs(handle) -> device time of the handle's last rmse() pass (ycnr_als_last_rmse_ms)
napi_value LastRmseMs(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value a[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Handle *hd = argc >= 1 ? handle_of(env, a[0]) : nullptr;
  if (!hd) return throw_msg(env, "lastRmseMs(handle)");
  double ms = 0;
  int rc = ycnr_als_last_rmse_ms(hd->h, &ms);
  if (rc) return throw_last(env, "lastRmseMs", rc);
  return num(env, ms);
}

// deferExchange(handle, side, deferred): ycnr_als_defer_exchange
napi_value DeferExchange(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value a[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Handle *hd = argc >= 3 ? handle_of(env, a[0]) : nullptr;
  int64_t side, on;
  if (!hd || !get_int(env, a[1], &side) || !get_int(env, a[2], &on)) return throw_msg(env, "deferExchange(handle, side, deferred)");
  int rc = ycnr_als_defer_exchange(hd->h, (int)side, (int)on);
  if (rc) return throw_last(env, "deferExchange", rc);
  return nullptr;
}

// allreduceSum(handle, Float64Array) -> the same array, summed over the ranks in place
napi_value AllreduceSum(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value a[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Handle *hd = argc >= 2 ? handle_of(env, a[0]) : nullptr;
  if (!hd) return throw_msg(env, "allreduceSum(handle, Float64Array)");
  View v = view_of(env, a[1]);
  if (!v.ok || v.type != napi_float64_array) return throw_msg(env, "invalid type!");
  int rc = ycnr_als_allreduce_sum(hd->h, static_cast<double *>(v.data), (int64_t)v.length);
  if (rc) return throw_last(env, "allreduceSum", rc);
  return a[1];
}

// broadcastFactors(handle, side, root); exchange(handle, side)
template <bool BCAST>
napi_value CommSide(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value a[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Handle *hd = argc >= 2 ? handle_of(env, a[0]) : nullptr;
  int64_t side, root = 0;
  if (!hd || !get_int(env, a[1], &side) || (BCAST && (argc < 3 || !get_int(env, a[2], &root)))) return throw_msg(env, BCAST ? "broadcastFactors(handle, side, root)" : "exchange(handle, side)");
  int rc = BCAST ? ycnr_als_broadcast_factors(hd->h, (int)side, (int)root) : ycnr_als_exchange(hd->h, (int)side);
  if (rc) return throw_last(env, BCAST ? "broadcastFactors" : "exchange", rc);
  return nullptr;
}

// rmse(handle, which, globalAvgShift, portionRowEnd) -> Float64Array [rSumDiff2, rCnt, rSum] * nPortions
napi_value Rmse(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value a[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  Handle *hd = argc >= 3 ? handle_of(env, a[0]) : nullptr;
  int64_t which;
  double shift;
  if (!hd || !get_int(env, a[1], &which) || !get_double(env, a[2], &shift)) return throw_msg(env, "rmse(handle, which, shift[, portionRowEnd])");
  std::vector<int64_t> ends;
  if (argc >= 4) {
    napi_valuetype t;
    napi_typeof(env, a[3], &t);
    if (t != napi_undefined && t != napi_null) {
      View e = view_of(env, a[3]);
      if (!e.ok || !to_i64(e, ends)) return throw_msg(env, "invalid type!");
    }
  }
  const size_t np = ends.empty() ? 1 : ends.size();
  napi_value ab, out;
  void *buf = nullptr;
  NAPI_OK(napi_create_arraybuffer(env, np * 3 * sizeof(double), &buf, &ab));
  NAPI_OK(napi_create_typedarray(env, napi_float64_array, np * 3, ab, 0, &out));
  int rc = ycnr_als_rmse(hd->h, (int)which, shift, (int)ends.size(), ends.empty() ? nullptr : ends.data(), static_cast<double *>(buf));
  if (rc) return throw_last(env, "rmse", rc);
  return out;
}

// N3: recommendItems(userRows, itemFactors, k, skipPtr, skipIds: Int32Array, globalAvgShift, minRecommendRating, limit,
//                    outIds: Int32Array(nUsers * limit), outPredict: Float64Array(nUsers * limit), outCount: Int32Array(nUsers)) -> kernel ms
napi_value RecommendItems(napi_env env, napi_callback_info info) {
  size_t argc = 11;
  napi_value a[11];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 11) return throw_msg(env, "recommendItems(userRows, itemFactors, k, skipPtr, skipIds, shift, minRating, limit, outIds, outPredict, outCount)");
  View ur = view_of(env, a[0]), it = view_of(env, a[1]), sp = view_of(env, a[3]), sk = view_of(env, a[4]), oi = view_of(env, a[8]),
       op = view_of(env, a[9]), oc = view_of(env, a[10]);
  int64_t k, limit;
  double shift, minRating;
  std::vector<int64_t> skipPtr;
  if (!ur.ok || !it.ok || !sp.ok || !sk.ok || !oi.ok || !op.ok || !oc.ok || !get_int(env, a[2], &k) || !get_double(env, a[5], &shift) ||
      !get_double(env, a[6], &minRating) || !get_int(env, a[7], &limit) || !to_i64(sp, skipPtr) || skipPtr.empty() ||
      (it.type != napi_float32_array && it.type != napi_float64_array) || ur.type != it.type || sk.type != napi_int32_array ||
      oi.type != napi_int32_array || op.type != napi_float64_array || oc.type != napi_int32_array)
    return throw_msg(env, "invalid type!");  // cpp_utils/cpp_utils.js:12
  if (k < 1 || limit < 1 || ur.length % (size_t)k || it.length % (size_t)k) return throw_msg(env, "factor arrays are not multiples of k");
  const size_t nUsers = ur.length / (size_t)k;
  if (skipPtr.size() != nUsers + 1 || skipPtr.back() < 0 || (size_t)skipPtr.back() > sk.length || oi.length < nUsers * (size_t)limit ||
      op.length < nUsers * (size_t)limit || oc.length < nUsers)
    return throw_msg(env, "array lengths do not match");
  double ms = 0;
  if (ycnr_recommend_items(it.type == napi_float64_array ? YCNR_F64 : YCNR_F32, (int32_t)k, (int64_t)nUsers, ur.data, (int64_t)(it.length / (size_t)k),
                           it.data, skipPtr.data(), static_cast<const int32_t *>(sk.data), shift, minRating, (int32_t)limit,
                           static_cast<int32_t *>(oi.data), static_cast<double *>(op.data), static_cast<int32_t *>(oc.data), &ms))
    return throw_msg(env, ycnr_last_error());
  return num(env, ms);
}

// N2: csrFromTriplets(rowIdx: Int32Array, colIdx: Int32Array, vals, rows, cols, rowPtr: Float64Array(rows+1) out,
//                     indx: Int32Array(n) out, outVals out) -> kernel ms
napi_value CsrFromTriplets(napi_env env, napi_callback_info info) {
  size_t argc = 8;
  napi_value a[8];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 8) return throw_msg(env, "csrFromTriplets(rowIdx, colIdx, vals, rows, cols, rowPtr, indx, outVals)");
  View ri = view_of(env, a[0]), ci = view_of(env, a[1]), va = view_of(env, a[2]), rp = view_of(env, a[5]), ix = view_of(env, a[6]),
       ov = view_of(env, a[7]);
  int64_t rows, cols;
  if (!ri.ok || !ci.ok || !va.ok || !rp.ok || !ix.ok || !ov.ok || !get_int(env, a[3], &rows) || !get_int(env, a[4], &cols) ||
      ri.type != napi_int32_array || ci.type != napi_int32_array || ix.type != napi_int32_array || rp.type != napi_float64_array ||
      (va.type != napi_float32_array && va.type != napi_float64_array) || ov.type != va.type)
    return throw_msg(env, "invalid type!");
  const size_t n = ri.length;
  if (ci.length != n || va.length != n || ix.length < n || ov.length < n || rows < 0 || rp.length != (size_t)rows + 1)
    return throw_msg(env, "array lengths do not match");
  std::vector<int64_t> rowPtr((size_t)rows + 1);
  double ms = 0;
  if (ycnr_csr_from_triplets(va.type == napi_float64_array ? YCNR_F64 : YCNR_F32, (int64_t)n, static_cast<const int32_t *>(ri.data),
                             static_cast<const int32_t *>(ci.data), va.data, rows, cols, rowPtr.data(), static_cast<int32_t *>(ix.data),
                             ov.data, &ms))
    return throw_msg(env, ycnr_last_error());
  for (size_t r = 0; r < rowPtr.size(); ++r) static_cast<double *>(rp.data)[r] = (double)rowPtr[r];
  return num(env, ms);
}

// N2: csrTranspose(rows, cols, rowPtr, indx, vals, outPtr: Float64Array(cols+1), outIndx, outVals) -> kernel ms
napi_value CsrTranspose(napi_env env, napi_callback_info info) {
  size_t argc = 8;
  napi_value a[8];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 8) return throw_msg(env, "csrTranspose(rows, cols, rowPtr, indx, vals, outPtr, outIndx, outVals)");
  int64_t rows, cols;
  View rp = view_of(env, a[2]), ix = view_of(env, a[3]), va = view_of(env, a[4]), op = view_of(env, a[5]), oi = view_of(env, a[6]),
       ov = view_of(env, a[7]);
  std::vector<int64_t> rowPtr;
  if (!get_int(env, a[0], &rows) || !get_int(env, a[1], &cols) || !rp.ok || !ix.ok || !va.ok || !op.ok || !oi.ok || !ov.ok ||
      !to_i64(rp, rowPtr) || ix.type != napi_int32_array || oi.type != napi_int32_array || op.type != napi_float64_array ||
      (va.type != napi_float32_array && va.type != napi_float64_array) || ov.type != va.type)
    return throw_msg(env, "invalid type!");
  if (rows < 0 || cols < 0 || rowPtr.size() != (size_t)rows + 1 || op.length != (size_t)cols + 1) return throw_msg(env, "array lengths do not match");
  const int64_t n = rowPtr.back();
  if (n < 0 || (size_t)n > ix.length || (size_t)n > va.length || (size_t)n > oi.length || (size_t)n > ov.length)
    return throw_msg(env, "indx / vals shorter than rowPtr says");
  std::vector<int64_t> outPtr((size_t)cols + 1);
  double ms = 0;
  if (ycnr_csr_transpose(va.type == napi_float64_array ? YCNR_F64 : YCNR_F32, rows, cols, rowPtr.data(), static_cast<const int32_t *>(ix.data),
                         va.data, outPtr.data(), static_cast<int32_t *>(oi.data), ov.data, &ms))
    return throw_msg(env, ycnr_last_error());
  for (size_t c = 0; c < outPtr.size(); ++c) static_cast<double *>(op.data)[c] = (double)outPtr[c];
  return num(env, ms);
}

// N1: splitToSets(rowPtr, types: Int8Array (in/out), dataSetDistr: [train, validate, test], seed) -> kernel ms
// (EmfLord.doSplitToSets, lib/emf/EmfLord.js:402-505)
napi_value SplitToSets(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value a[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 4) return throw_msg(env, "splitToSets(rowPtr, types, dataSetDistr, seed)");
  View rp = view_of(env, a[0]), ty = view_of(env, a[1]);
  std::vector<int64_t> rowPtr;
  if (!rp.ok || !ty.ok || ty.type != napi_int8_array || !to_i64(rp, rowPtr) || rowPtr.empty()) return throw_msg(env, "invalid type!");
  if (rowPtr.back() < 0 || (size_t)rowPtr.back() > ty.length) return throw_msg(env, "types shorter than rowPtr says");
  int32_t pcts[3];
  for (uint32_t i = 0; i < 3; ++i) {
    napi_value e;
    int64_t v;
    if (napi_get_element(env, a[2], i, &e) != napi_ok || !get_int(env, e, &v)) return throw_msg(env, "dataSetDistr must be [train, validate, test]");
    pcts[i] = (int32_t)v;
  }
  int64_t seed;
  if (!get_int(env, a[3], &seed)) return throw_msg(env, "invalid type!");
  double ms = 0;
  if (ycnr_split_to_sets((int64_t)rowPtr.size() - 1, rowPtr.data(), static_cast<int8_t *>(ty.data), pcts, (uint32_t)seed, &ms))
    return throw_msg(env, ycnr_last_error());
  return num(env, ms);
}

// N1: ratingStats(rowPtr, vals, types | null, cnt: Int32Array, sum: Float64Array) -> kernel ms
// (ratings_count / avg_rating of EmfLord.js:252-396: avg = sum / cnt)
napi_value RatingStats(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value a[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, a, nullptr, nullptr));
  if (argc < 5) return throw_msg(env, "ratingStats(rowPtr, vals, types, cnt, sum)");
  View rp = view_of(env, a[0]), vals = view_of(env, a[1]), ty = view_of(env, a[2]), cnt = view_of(env, a[3]), sum = view_of(env, a[4]);
  std::vector<int64_t> rowPtr;
  if (!rp.ok || !vals.ok || !cnt.ok || !sum.ok || !to_i64(rp, rowPtr) || rowPtr.empty() || cnt.type != napi_int32_array ||
      sum.type != napi_float64_array || (vals.type != napi_float32_array && vals.type != napi_float64_array) ||
      (ty.ok && ty.type != napi_int8_array))
    return throw_msg(env, "invalid type!");
  const size_t rows = rowPtr.size() - 1;
  if (cnt.length < rows || sum.length < rows) return throw_msg(env, "cnt / sum shorter than the row count");
  if (rowPtr.back() < 0 || (size_t)rowPtr.back() > vals.length || (ty.ok && (size_t)rowPtr.back() > ty.length))
    return throw_msg(env, "vals / types shorter than rowPtr says");
  double ms = 0;
  if (ycnr_rating_stats(vals.type == napi_float64_array ? YCNR_F64 : YCNR_F32, (int64_t)rows, rowPtr.data(), vals.data,
                        ty.ok ? static_cast<const int8_t *>(ty.data) : nullptr, static_cast<int32_t *>(cnt.data),
                        static_cast<double *>(sum.data), &ms))
    return throw_msg(env, ycnr_last_error());
  return num(env, ms);
}

napi_value Init(napi_env env, napi_value exports) {
  const napi_property_descriptor props[] = {
      {"sAlsCalcPortion", nullptr, AlsCalcPortion<false>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"dAlsCalcPortion", nullptr, AlsCalcPortion<true>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"sRmsePortion", nullptr, RmsePortion<false>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"dRmsePortion", nullptr, RmsePortion<true>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"deviceCount", nullptr, DeviceCount, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"lastError", nullptr, LastError, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"version", nullptr, Version, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"create", nullptr, Create, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"destroy", nullptr, Destroy, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"setRatings", nullptr, SetRatings<false>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"setRmseRatings", nullptr, SetRatings<true>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"setFactors", nullptr, SetFactors, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"getFactors", nullptr, GetFactors, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"step", nullptr, Step, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"rmse", nullptr, Rmse, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"splitToSets", nullptr, SplitToSets, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"ratingStats", nullptr, RatingStats, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"csrFromTriplets", nullptr, CsrFromTriplets, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"csrTranspose", nullptr, CsrTranspose, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"recommendItems", nullptr, RecommendItems, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"sAlsPinFixedFactors", nullptr, AlsPinFixedFactors<false>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"dAlsPinFixedFactors", nullptr, AlsPinFixedFactors<true>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"alsUnpinFixedFactors", nullptr, AlsUnpinFixedFactors, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"alsReleasePortionState", nullptr, AlsReleasePortionState, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"commUniqueId", nullptr, CommUniqueId, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"commInit", nullptr, CommInit, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"setRatingsSharded", nullptr, SetRatingsSharded, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"setRatingsBanded", nullptr, SetRatingsBanded, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"deferExchange", nullptr, DeferExchange, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"commInfo", nullptr, CommInfo, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"lastRmseMs", nullptr, LastRmseMs, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"allreduceSum", nullptr, AllreduceSum, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"broadcastFactors", nullptr, CommSide<true>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
      {"exchange", nullptr, CommSide<false>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
  };
  napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
  return exports;
}

}  // namespace

NAPI_MODULE(ycnr_als, Init)
