/*
 * ycnr_als.h -- C ABI of libycnr_als.so, the MI355X (gfx950) replacement for the
 * per-iteration ALS user/item factor solve of ukrbublik/You-Can-Not-Recommend.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repo).  Plain pointers and sizes only; no C++ or torch types cross this
 * boundary.  All functions are synchronous unless named *_async, are not re-entrant per
 * handle, never abort(), and return 0 / a non-negative count on success or a negative
 * YCNR_ERR_* code, with a human-readable message available from ycnr_last_error().
 *
 * Two levels:
 *   level 1 -- portion ops on caller-owned HOST buffers: bit-for-bit the argument meaning
 *              of EmfWorker.mw_calcTrainAlsPortion / mw_calcRmsePortion;
 *   level 2 -- a resident trainer that keeps CSR ratings and both factor matrices in HBM
 *              and runs whole half-steps (EmfLord.alsTrainStep) and RMSE passes.
 *
 * The reference's native boundary is the node-gyp addon cpp_utils (binding.gyp:3-16,
 * cpp_utils/cpp_utils.cc:3-8) with s/d-prefixed functions over typed arrays
 * (cpp_utils/cpp_utils.js:15-19); the same s/d naming is kept here.
 */
#ifndef YCNR_ALS_H
#define YCNR_ALS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YCNR_ALS_ABI_VERSION 4 /* 4: + ycnr_als_comm_info, ycnr_als_last_rmse_ms (nothing changed or removed) */

/* error codes */
#define YCNR_OK 0
#define YCNR_ERR_INVALID (-1)     /* bad argument: null pointer, k out of range, index >= rows ... */
#define YCNR_ERR_HIP (-2)         /* a HIP runtime call failed (message carries hipGetErrorString) */
#define YCNR_ERR_NOMEM (-3)       /* host or device allocation failed */
#define YCNR_ERR_UNSUPPORTED (-4) /* valid request this build cannot serve (e.g. factorsCount > 4096) */
#define YCNR_ERR_NUMERIC (-5)     /* a row's normal matrix was not positive definite (NaN/Inf input) */
#define YCNR_ERR_STATE (-6)       /* call order violated (e.g. step before set_ratings) */

/* stepType of EmfLord.alsTrainStep(stepType), lib/emf/EmfLord.js:963: 'byUser' | 'byItem' */
#define YCNR_BY_USER 0
#define YCNR_BY_ITEM 1
/* options.useDoublePrecision, lib/emf/EmfBase.js:112,174-182 */
#define YCNR_F32 0
#define YCNR_F64 1
/* where a caller's buffer lives */
#define YCNR_MEM_HOST 0
#define YCNR_MEM_DEVICE 1
/* stepType of EmfLord.calcRmse(stepType, ...), lib/emf/EmfLord.js:1043: 'rmseValidate' | 'rmseTest' */
#define YCNR_RMSE_VALIDATE 0
#define YCNR_RMSE_TEST 1

/* Message of the last failing call on this thread ("" if none). */
const char *ycnr_last_error(void);
/* YCNR_ALS_ABI_VERSION the library was built with. */
int ycnr_version(void);
/* Number of visible HIP devices, or YCNR_ERR_HIP. Does not create a context. */
int ycnr_device_count(void);

/* ------------------------------------------------------------------ level 1: portion ops
 *
 * ycnr_{s,d}AlsCalcPortion replaces the body of EmfWorker.mw_calcTrainAlsPortion,
 * lib/emf/EmfWorker.js:176-251, for one portion buffer:
 *   lambda        als.userFactReg / als.itemFactReg of the step (EmfWorker.js:180-181);
 *                 applied as lambda * cols on the diagonal (EmfWorker.js:233-235)
 *   k             options.factorsCount
 *   alsRows       Int32 [nRows, rowId0, cols0, rowId1, cols1, ...]   (EmfMaster.js:597-598,609)
 *   alsIndx       Int32 column ids, rows concatenated                 (EmfMaster.js:590)
 *   alsVals       Float32/64 ratings, rows concatenated               (EmfMaster.js:589)
 *   fixedFactors  [fixedRows x k] row-major: the opposite side (item factors for 'byUser');
 *                 read only (EmfBase.copySubFixedFactors, EmfBase.js:537-555)
 *   solvedFactors [solvedRows x k] row-major: row rowId is overwritten in place
 *                 (EmfBase.getLatentFactorsPartData, EmfBase.js:518-532); other rows untouched
 * Buffers are borrowed for the duration of the call (as cpp_utils/als_utils.cc:4-20 does).
 * Unlike the reference, ids are bounds-checked (YCNR_ERR_INVALID) and rows recorded with
 * cols == 0 are skipped instead of solving a singular system.
 * Returns ratingsInPortion (the 'completedPortion' message field, EmfWorker.js:257) or < 0. */
int64_t ycnr_sAlsCalcPortion(double lambda, int k, const int32_t *alsRows, const int32_t *alsIndx,
                             const float *alsVals, const float *fixedFactors, int64_t fixedRows,
                             float *solvedFactors, int64_t solvedRows);
int64_t ycnr_dAlsCalcPortion(double lambda, int k, const int32_t *alsRows, const int32_t *alsIndx,
                             const double *alsVals, const double *fixedFactors, int64_t fixedRows,
                             double *solvedFactors, int64_t solvedRows);

/* Optional, once per half-step: keep the step's fixed factor matrix resident on the device for the
 * portion calls that follow on this thread (call it where the worker handles 'startTrainStep',
 * lib/emf/EmfWorker.js:43-51 / EmfMaster._startAlsTrainStep, lib/emf/EmfMaster.js:364-383).  A later
 * ycnr_{s,d}AlsCalcPortion whose fixedFactors / fixedRows / k are the pinned ones skips the per-portion
 * gather + upload of the rows it refers to -- the replacement of the per-rating BLAS.BufCopy of
 * EmfBase.copySubFixedFactors (lib/emf/EmfBase.js:537-555).  The host must pin again after it has
 * changed the matrix (every half-step does).  Stream and device buffers of the portion ops are
 * cached per thread either way; ycnr_AlsReleasePortionState frees them. */
int ycnr_sAlsPinFixedFactors(const float *fixedFactors, int64_t fixedRows, int k);
int ycnr_dAlsPinFixedFactors(const double *fixedFactors, int64_t fixedRows, int k);
/* Forget the pinned matrix (the device copy stays allocated for the next pin).  A portion call whose
 * solvedFactors overlaps the pinned host range does this by itself: the reference's host updates its matrices
 * in place (the matrix pinned for 'byUser' is the one 'byItem' writes, EmfBase.js:518-532), so such a call
 * has just made the device copy stale. */
int ycnr_AlsUnpinFixedFactors(void);
int ycnr_AlsReleasePortionState(void);

/* ycnr_{s,d}RmsePortion replaces EmfWorker.mw_calcRmsePortion, lib/emf/EmfWorker.js:266-315:
 * pred = userFactors[u] . itemFactors[i] + globalAvgShift (EmfBase._alsPredict, EmfBase.js:825-827);
 * out = {rSumDiff2, rCnt, rSum} accumulated in double (EmfWorker.js:297-299). */
int ycnr_sRmsePortion(int k, const int32_t *rmseRows, const int32_t *rmseIndx, const float *rmseVals,
                      const float *userFactors, int64_t usersRows, const float *itemFactors,
                      int64_t itemsRows, double globalAvgShift, double *out3);
int ycnr_dRmsePortion(int k, const int32_t *rmseRows, const int32_t *rmseIndx, const double *rmseVals,
                      const double *userFactors, int64_t usersRows, const double *itemFactors,
                      int64_t itemsRows, double globalAvgShift, double *out3);

/* ------------------------------------------------------------------ level 2: resident trainer */

typedef struct ycnr_als ycnr_als; /* opaque; owns device memory (cf. createSharedFactors, EmfBase.js:399-425) */

/* The subset of EmfBase.DefaultOptions (lib/emf/EmfBase.js:52-140) and stats
 * (EmfBase.js:142-169) the path depends on. */
typedef struct ycnr_als_options {
  int32_t struct_size;     /* = sizeof(ycnr_als_options) */
  int32_t device;          /* HIP device ordinal (one process per GPU: LOCAL_RANK) */
  int32_t dtype;           /* YCNR_F32 | YCNR_F64 = options.useDoublePrecision */
  int32_t factorsCount;    /* options.factorsCount: 1..4096 in either precision.  Kernel families by size: one wave per
                            * row up to 128, a workgroup per row up to 256 (float32), the any-k path beyond (float64: beyond
                            * 128); float32 values that are not multiples of 4 run on copies padded to the next multiple of 4
                            * (the matrices the host sees keep factorsCount columns) */
  int64_t totalUsersCount; /* stats.totalUsersCount = max user id (EmfLord.js:81) */
  int64_t totalItemsCount; /* stats.totalItemsCount = max item id (EmfLord.js:82) */
  double userFactReg;      /* options.als.userFactReg (EmfBase.js:67) */
  double itemFactReg;      /* options.als.itemFactReg (EmfBase.js:69) */
  int32_t chunkRatings;    /* ratings per work unit for split rows; 0 = default */
  int32_t flags;           /* YCNR_FLAG_* bits, normally 0 */
} ycnr_als_options;

/* options.flags: use the plain LDS Cholesky (the first solver) instead of the
 * register-resident MFMA block Cholesky. For A/B tests. */
#define YCNR_FLAG_LDS_SOLVER 1
/* options.flags: never use the dual (n x n) form for rows with fewer ratings than factors */
#define YCNR_FLAG_NO_DUAL 2
/* options.flags: accepted and ignored (band-major chunks, its successor, are the default;
 * see YCNR_FLAG_NO_BANDS) */
#define YCNR_FLAG_LOCALITY_SORT 4
/* options.flags: keep the last 4 Gramian columns of k = 16 m + 4 on the matrix cores (padded
 * tile column) instead of accumulating them on the VALU */
#define YCNR_FLAG_NO_VALU_EDGE 8
/* options.flags: keep the chunk Gramian on v_mfma_f32_16x16x4_f32 instead of the exact
 * 3-way bf16 split on the bf16 matrix pipe */
#define YCNR_FLAG_NO_BF16X6 16
/* options.flags: cut split rows every chunkRatings ratings only.  Default: when the fixed matrix
 * is much larger than the last-level cache, split rows are cut at common column-id boundaries
 * (bands of 96 MB of the fixed matrix) and their chunks run band by band, so the waves in flight
 * gather from one cache-resident band. */
#define YCNR_FLAG_NO_BANDS 32
/* options.flags: launch every kernel of a half-step on the handle's stream, one after the other.
 * Default: the dual-form kernels (rows with fewer ratings than factors) run on two side streams
 * of the handle, next to the row kernel, when a half-step has at least 1024 such rows; they fork
 * from and join the handle's stream, so the half-step still begins and ends in stream order. */
#define YCNR_FLAG_NO_OVERLAP 64
/* options.flags: launch every half-step kernel by kernel.  Default: uploads below 2 M ratings (the ML-100k / ML-1M
 * shapes), whose half-step is a dozen launches of kernels that each fill a fraction of the chip, are captured
 * once into a hipGraph -- chunk Gramians -> reduce, the row kernel and the dual classes as parallel branches -- and
 * replayed with one launch per half-step; ycnr_als_step_info then reports one interval (gramSolveMs, dualOverlapped). */
#define YCNR_FLAG_NO_GRAPH 128

/* Timing / accounting of the last ycnr_als_step, measured with HIP events on the
 * handle's stream around each kernel (DESIGN.md "Measurement"). */
typedef struct ycnr_als_step_info {
  int32_t struct_size;
  int32_t side;
  int64_t rows;          /* rows solved (rows with >= 1 rating in the local shard) */
  int64_t ratings;       /* ratings consumed */
  int64_t units;         /* wave-level work units launched by the Gramian kernel */
  int64_t splitRows;     /* rows whose Gramian was split over several units */
  int64_t fusedRows;     /* rows handled whole by als_gram_solve (= rows - splitRows) */
  int64_t fusedRatings;  /* their ratings (the rest went through als_gram_slab) */
  int64_t dualRows;      /* of the fused rows, those short enough for the dual-form kernel */
  int64_t dualRatings;   /* and their ratings */
  float gramSlabMs;      /* als_gram_slab kernel: Gramian chunks of split rows */
  float gramSolveMs;     /* als_gram_solve kernel: whole rows, Gramian + solve fused */
  float dualSolveMs;     /* als_dual_solve kernels: whole rows with fewer ratings than factors */
  float reduceSolveMs;   /* als_reduce_solve kernel: slab sum + solve of split rows */
  float totalMs;         /* first kernel start -> last kernel end */
  int32_t numericErrors; /* rows whose matrix was not positive definite */
  int32_t dualOverlapped; /* 1: the dual kernels ran on the side streams next to als_gram_solve
                           * (see YCNR_FLAG_NO_OVERLAP): gramSolveMs is the time until that kernel
                           * ended, dualSolveMs what the dual kernels still needed after it; only
                           * their sum is the time of a kernel group */
  /* multi-GPU (ycnr_als_comm_init + ycnr_als_set_ratings_sharded); the kernel times above are sums
   * over the pieces of the local shard, totalMs spans the first kernel start to the last kernel end */
  int32_t parts;           /* pieces the local shard was solved in (exchange of piece c overlaps the solve of c + 1) */
  int32_t reserved0;
  int64_t exchangeBytes;   /* bytes this rank sent + received in the half-step's exchange */
  float exchangeMs;        /* sum of the exchanges' durations on the communicator's stream */
  float exposedExchangeMs; /* what the step still waited for after its own last kernel had ended */
  double dualFlops;        /* flops the dual form executes for the dualRows: n(n+1)k + n^3/3 + 2n^2 + 2nk per row */
} ycnr_als_step_info;

int ycnr_als_create(const ycnr_als_options *opts, ycnr_als **out);
/* Frees all device memory of the handle (cf. detachSharedFactors, EmfBase.js:351-376). */
int ycnr_als_destroy(ycnr_als *h);

/* Use an existing hipStream_t (e.g. torch's current stream) for all work of the handle, so
 * that the caller's copies and collectives are ordered against it.  NULL is the device's
 * default (null) stream; YCNR_OWN_STREAM restores the handle's own non-blocking stream. */
#define YCNR_OWN_STREAM ((void *)(intptr_t)-1)
int ycnr_als_set_stream(ycnr_als *h, void *hipStream);

/* Ratings of one side in CSR form, replacing the per-portion SQL fetch + packer
 * (EmfMaster.m_fetchPortionTrainAlsOrRmse / m_processFetchedPortionAlsOrRmse,
 * lib/emf/EmfMaster.js:501-614) by a one-time upload:
 *   side       YCNR_BY_USER: rows = users, indx = item ids; YCNR_BY_ITEM: rows = items, indx = user ids
 *   rowPtr     int64[totalRows + 1] offsets into indx / vals (ids 0-based = db id - 1, EmfMaster.js:584-586)
 *   vals       float or double per options.dtype
 *   rowBegin,rowEnd  the shard [rowBegin, rowEnd) this handle solves (whole matrix: 0, totalRows);
 *              only that slice of indx / vals is copied to the device
 *   memKind    YCNR_MEM_HOST or YCNR_MEM_DEVICE for all three arrays
 * Column ids are validated against the opposite side's row count. */
int ycnr_als_set_ratings(ycnr_als *h, int side, const int64_t *rowPtr, const int32_t *indx,
                         const void *vals, int64_t rowBegin, int64_t rowEnd, int memKind);

/* Ratings for an RMSE pass (always by user; dataset_type 2 / 3 rows of
 * EmfMaster.js:502-503), same layout as above. */
int ycnr_als_set_rmse_ratings(ycnr_als *h, int which, const int64_t *rowPtr, const int32_t *indx,
                              const void *vals, int64_t rowBegin, int64_t rowEnd, int memKind);

/* Whole factor matrix in / a row range out, in the reference's layout: dense row-major
 * [rows x factorsCount] of float/double = the raw content of the user_factors /
 * item_factors files (EmfBase.js:384-390, EmfManager.js:241-255). */
int ycnr_als_set_factors(ycnr_als *h, int side, const void *src, int memKind);
int ycnr_als_get_factors(ycnr_als *h, int side, void *dst, int64_t rowBegin, int64_t rowCount,
                         int memKind);
/* Borrowed device pointer of a side's matrix, so the host harness can all-gather shards in
 * place with RCCL (replaces 'alsSaveCalcedFactors' streaming, EmfMaster.js:711-723). */
int ycnr_als_factors_ptr(ycnr_als *h, int side, void **devicePtr);
/* Adopt caller-owned device memory ([rows x k], e.g. a torch tensor) as a side's matrix. */
int ycnr_als_bind_factors(ycnr_als *h, int side, void *devicePtr);

/* One half-step = EmfLord.alsTrainStep(stepType), lib/emf/EmfLord.js:963-984: every local
 * row with >= 1 rating is re-solved against the CURRENT opposite factors and written in
 * place; rows without ratings are untouched.  Returns after the stream has drained. */
int ycnr_als_step(ycnr_als *h, int side);
/* Enqueue only; pair with ycnr_als_sync before reading results or step info.  Several half-steps may be
 * enqueued back to back.  With a communicator the call is collective like ycnr_als_step; on the YCNR_COMM_IPC
 * transport a half-step enqueued behind one that ycnr_als_sync has not completed first completes that one
 * (stream drained + the end-of-step barrier: its fixed side is the matrix the peers were still pushing into),
 * so back-to-back calls are correct there too, only not asynchronous. */
int ycnr_als_step_async(ycnr_als *h, int side);
int ycnr_als_sync(ycnr_als *h);
int ycnr_als_last_step_info(ycnr_als *h, ycnr_als_step_info *info);
/* Half-steps of both sides may be in flight at once (ycnr_als_step_async(BY_USER), ycnr_als_step_async(BY_ITEM),
 * ycnr_als_sync: EmfLord.alsTrainIter, lib/emf/EmfLord.js:954-958, without the host in between); ycnr_als_sync completes
 * every half-step in flight and keeps the step info of each side: this returns that of the last completed half-step of ONE
 * side (ycnr_als_last_step_info: of the last one enqueued).  Rows that were not positive definite are counted since the
 * last sync and reported with the last half-step completed.  (The hosts of this repository await each step: measured,
 * csrc/devtest/tried/NOTES_r04.md.) */
int ycnr_als_step_info_of(ycnr_als *h, int side, ycnr_als_step_info *info);

/* ---- multi-GPU: row shards + exchange -------------------------------------------
 *
 * One process drives one GPU.  Both factor matrices are replicated on every GPU; each rank solves a
 * contiguous range of rows of the side being solved and every rank's replica is brought up to date
 * before the next half-step.  This replaces the reference's cluster path:
 *   'alsSaveCalcedFactors' -- after each portion the solved rows are streamed to every other node
 *       (EmfMaster.wm_completedPortion, lib/emf/EmfMaster.js:711-723; receivers EmfLord.js:727-732,
 *       EmfChief.js:207-212)                                    -> the exchange inside ycnr_als_step
 *   getFactors / setFactors when a node joins (EmfChief.loadFactorsFromLord, EmfChief.js:55-71,
 *       EmfLord.js:738-740)                                     -> ycnr_als_broadcast_factors
 *   'rmseSaveCalcs' partial sums to the Lord (EmfMaster.js:726-736, EmfLord.js:734-736)
 *                                                               -> ycnr_als_allreduce_sum
 * Transports (YCNR_COMM_IPC and YCNR_COMM_STUB are described at their definitions): YCNR_COMM_RCCL is the product path -- one group of point-to-point ncclSend / ncclRecv
 * between all pairs of ranks straight into the replicated matrix at each shard's row offset (xGMI
 * is a point-to-point mesh: every link carries only what its two ends owe each other, all links at
 * once; uneven shards need no padding or staging).  YCNR_COMM_SHM is a functional stand-in for
 * tests (what gloo is to nccl): ranks of one node stage rows through POSIX shared memory, so that
 * several ranks can share one GPU, which RCCL refuses.
 *
 * ycnr_comm_unique_id: 128 bytes created by ONE rank and handed to all ranks by the host's own
 * control plane (the NodeJS Lord's process.send, torch.distributed's store ...), cf. ncclGetUniqueId.
 * ycnr_als_comm_init is collective: every rank calls it with the same id. */
#define YCNR_COMM_NONE 0
#define YCNR_COMM_RCCL 1
#define YCNR_COMM_SHM 2
/* device-to-device without RCCL: peers' replicas mapped with hipIpcOpenMemHandle (handles through a shared-memory
 * control segment), solved rows pushed with hipMemcpyAsync on the communicator's stream (copy engines, no
 * compute units), one host barrier at the end of each half-step.  Works with several ranks on one device.
 * Bind the factor matrices (ycnr_als_bind_factors) before ycnr_als_comm_init or before the collective
 * ycnr_als_set_ratings_sharded, on every rank alike. */
#define YCNR_COMM_IPC 3
/* rank r of a world of N with the exchange left out (timing events only): one GPU can solve every rank's shard
 * in turn and report the compute time per rank (bench.py --emulate-world) */
#define YCNR_COMM_STUB 4
#define YCNR_COMM_ID_BYTES 128
int ycnr_comm_unique_id(int transport, void *id128);
int ycnr_als_comm_init(ycnr_als *h, int transport, const void *id128, int rank, int world);
int ycnr_als_comm_destroy(ycnr_als *h);
/* What the handle's communicator is: out = {transport (YCNR_COMM_*), rank, world, ranks RCCL itself counts in its
 * communicator (ncclCommCount; -1 for the other transports)} -- lets a host or a benchmark line state which path ran and
 * that RCCL saw every rank (the reference's Lord logs the nodes it gathered, lib/emf/EmfLord.js:752-828). */
int ycnr_als_comm_info(ycnr_als *h, int32_t out[4]);
/* Sharded upload of one side for all ranks of the communicator (world = 1 without one):
 *   boundsWorld  the number of ranks `bounds` describes; must equal the communicator's world
 *   bounds   int64[boundsWorld * (nChunks + 1)], ascending; rank r solves rows [bounds[r (nChunks+1)],
 *            bounds[r (nChunks+1) + nChunks]) in nChunks pieces cut at the values in between; the
 *            shards tile the side's rows [0, rows) in rank order.  The host chooses the cuts (cost-balanced:
 *            the counterpart of EmfLord.splitToPortions, lib/emf/EmfLord.js:510-612).
 * rowPtr / indx / vals describe the WHOLE side (only this rank's slice is copied to the device).
 * After this call ycnr_als_step(side) includes the exchange: piece c's rows travel while piece
 * c + 1 is being solved, and the step returns when every replica holds every solved row -- the
 * meaning of 'stepComplete' in the reference's cluster (EmfLord.alsTrainStep, EmfLord.js:963-984). */
int ycnr_als_set_ratings_sharded(ycnr_als *h, int side, const int64_t *rowPtr, const int32_t *indx, const void *vals,
                                 int memKind, int nChunks, int boundsWorld, const int64_t *bounds);
/* A side whose half-step is sharded by BANDS OF COLUMNS instead of by rows (meant for the item side, whose fixed
 * matrix -- the users' -- is the large one): the columns are cut into nBands bands (nBands <= 8, the same cut for every world
 * size), rank r holds the bands [rankBands[r], rankBands[r + 1]) and accumulates, for EVERY row of the side, the Gramian over the
 * ratings whose column lies in its bands -- so the fixed matrix never has to be current outside its own band: with the user side
 * uploaded as a plain shard (ycnr_als_set_ratings: no exchange) the user matrix is never all-gathered.  Every (row, band) sum
 * travels to the rank that owns the row (rows [ownerBounds[r], ownerBounds[r + 1])), which adds a row's bands in band order,
 * solves, and the solved rows are exchanged as after a sharded upload.  The order of every sum is fixed by the bands, not by the
 * ranks: 1, 2, 4 or 8 ranks give the same bits.  This replaces, for that side, the reference's per-portion row broadcast
 * (EmfMaster.wm_completedPortion, lib/emf/EmfMaster.js:711-723) by a reduce-scatter of Gramians + the broadcast of the (small) solved side.
 *   rowPtr / indx / vals   CSR of the side with ALL its rows, holding only the ratings whose column id lies in this rank's bands
 *                          (the host filters; ascending column ids per row)
 *   bandBounds             int64[nBands + 1] column ids, ascending from 0 to the column count
 *   rankBands              int64[world + 1] band indices, ascending from 0 to nBands
 *   ownerBounds            int64[world + 1] row ids, ascending from 0 to the row count
 * Collective (every rank of the communicator; world = 1 without one).  factorsCount <= 128 (float32: multiples of 4);
 * transports rccl, ipc and stub.  A later ycnr_als_set_ratings[_sharded] of the side returns it to row shards. */
int ycnr_als_set_ratings_banded(ycnr_als *h, int side, const int64_t *rowPtr, const int32_t *indx, const void *vals, int memKind,
                                int nBands, const int64_t *bandBounds, const int64_t *rankBands, const int64_t *ownerBounds);
/* deferred != 0: the half-steps of this (sharded) side no longer exchange their solved rows -- every rank's replica is current in
 * its own rows only -- until ycnr_als_exchange brings all replicas up to date.  For a side whose matrix nobody reads outside its
 * own shard between two exchanges: the user side when the item side is sharded by user bands (ycnr_als_set_ratings_banded).
 * A new upload of the side (ycnr_als_set_ratings[_sharded]) resets it. */
int ycnr_als_defer_exchange(ycnr_als *h, int side, int deferred);
/* The exchange alone, whole shards, synchronous (e.g. after ycnr_als_set_factors of local rows). */
int ycnr_als_exchange(ycnr_als *h, int side);
/* root's whole matrix to every rank. */
int ycnr_als_broadcast_factors(ycnr_als *h, int side, int root);
/* vals[i] <- sum over ranks (host doubles; the RMSE partials {rSumDiff2, rCnt, rSum} per portion). */
int ycnr_als_allreduce_sum(ycnr_als *h, double *vals, int64_t n);
/* A self-addressed ncclSend / ncclRecv pair of nFloats floats and a 3-value all-reduce through the
 * handle's communicator; YCNR_ERR_STATE if the bytes or sums come back wrong. */
int ycnr_als_comm_selftest(ycnr_als *h, int64_t nFloats);

/* RMSE partial sums = EmfLord.calcRmse / EmfWorker.mw_calcRmsePortion
 * (lib/emf/EmfLord.js:1043-1081, EmfWorker.js:266-315) over the local rows of set `which`:
 *   nPortions, portionRowEnd[p]  exclusive 0-based upper row of portion p (ascending, clipped to
 *              the local shard), so the caller can reproduce the reference's per-portion
 *              reduce including predAvg = LAST portion's rSum / rCnt (EmfMaster.js:779);
 *              nPortions == 0: one portion covering the shard
 *   out        double[3 * max(nPortions,1)] = {rSumDiff2, rCnt, rSum} per portion
 * How the set is cut into portions does not decide how much of the GPU works: every portion is
 * summed in pieces of at most ceil(rows / 4096) rows, one workgroup each, added in row order. */
int ycnr_als_rmse(ycnr_als *h, int which, double globalAvgShift, int nPortions,
                  const int64_t *portionRowEnd, double *out);
/* Device time of the last ycnr_als_rmse on this handle (its kernel between two HIP events on the handle's stream): the
 * counterpart of the per-portion `time` the reference's workers report (lib/emf/EmfWorker.js:266-315). */
int ycnr_als_last_rmse_ms(ycnr_als *h, double *ms);

/* ---- Preprocessing directly before the path ------------------------------
 *
 * ycnr_split_to_sets replaces EmfLord.doSplitToSets (lib/emf/EmfLord.js:402-505): every user's
 * unassigned ratings (types[q] == 0) are shuffled and cut into train / validate / test
 * (dataset_type 1 / 2 / 3) so that the row ends up with
 *   targetCnts[0] = ceil(total * pcts[0] / 100), targetCnts[1] = ceil(total * (pcts[0] + pcts[1]) / 100) - targetCnts[0],
 *   targetCnts[2] = total - targetCnts[0] - targetCnts[1]
 * where total counts the row's ratings of type 0..3 (EmfLord.js:447-457, including its rule for
 * ratings that are already assigned and for left-overs: they go to train).  Types outside 0..3
 * (the reference's 4 = excluded) are left alone.  The reference's shuffle is unseeded; here the
 * shuffled order of a row's free ratings is ascending (key, j) with
 *   key = fmix32(fmix32(seed + 0x9e3779b9 * row) ^ j),  fmix32 = MurmurHash3's 32-bit finalizer,
 * j the rating's position inside the row, so every implementation produces the same split.
 * rowPtr / types are host arrays (rowPtr[0] == 0); deviceMs (may be NULL) receives the kernel time.
 *
 * ycnr_rating_stats replaces the SQL of updateUsersStats / updateItemsStats
 * (EmfLord.js:252-396): per row the count and the (double) sum of the ratings whose type is
 * 1, 2 or 3 (every rating when types == NULL); avg = sum / cnt, maxRatingsPer* = max cnt and
 * totalRatingsAvg = sum of sums / sum of counts are left to the caller. */
int ycnr_split_to_sets(int64_t rows, const int64_t *rowPtr, int8_t *types, const int32_t pcts[3], uint32_t seed,
                       double *deviceMs);
int ycnr_rating_stats(int dtype, int64_t rows, const int64_t *rowPtr, const void *vals, const int8_t *types,
                      int32_t *cnt, double *sum, double *deviceMs);

/* ---- Ratings ingestion ------------------------------------------------------
 *
 * The reference feeds the path from PostgreSQL, portion by portion (EmfMaster.js:501-614), after
 * importing MovieLens files into it (YcnrController.importML, lib/YcnrController.js:94-150).  Here
 * the input of a training run is a pair of binary CSR files -- the train ratings by user and the
 * same ratings by item -- in the layout below, produced once from (user, item, rating) triplets.
 *
 * ycnr_csr_from_triplets: CSR of n triplets (0-based ids), rows ordered by id, entries of a row
 * by column id (the ORDER BY of EmfMaster.js:511-529), equal (row, col) pairs in input order.
 * ycnr_csr_transpose: the same matrix by column (CSR by user -> CSR by item); entries of an
 * output row are ordered by the input row id.  Host arrays in and out; the sort runs on the GPU.
 *
 * File layout (little endian, no padding): char magic[4] = "YCSR"; uint32 version = 1;
 * uint32 dtype (YCNR_F32 / YCNR_F64); uint32 flags (bit 0: entries of every row ascending by
 * column id); int64 rows, cols, nnz; int64 rowPtr[rows + 1]; int32 indx[nnz]; T vals[nnz].
 * Readers and writers: python/ycnr_als/csrfile.py, lib/CsrFile.js (byte-identical output). */
#define YCNR_CSR_MAGIC "YCSR"
#define YCNR_CSR_VERSION 1
int ycnr_csr_from_triplets(int dtype, int64_t n, const int32_t *rowIdx, const int32_t *colIdx, const void *vals,
                           int64_t rows, int64_t cols, int64_t *rowPtr, int32_t *indx, void *outVals, double *deviceMs);
int ycnr_csr_transpose(int dtype, int64_t rows, int64_t cols, const int64_t *rowPtr, const int32_t *indx, const void *vals,
                       int64_t *outPtr, int32_t *outIndx, void *outVals, double *deviceMs);

/* ---- Top-N recommend ---------------------------------------------------------
 *
 * ycnr_recommend_items replaces the loop of YcnrController.recommendItemsForUser
 * (lib/YcnrController.js:255-274) for a batch of users: for every item id 0 .. totalItems-1 that is
 * not in the user's skip list (rated + "unrated_items" ids, :244-250),
 *   predict = userRow . itemFactors[item] + globalAvgShift        (EmfBase._alsPredict, EmfBase.js:825-827)
 * items with predict >= minRecommendRating compete, best first.  The reference pushes, sorts and
 * pops so that its list never holds more than limit - 1 items (:264-269: the pop happens when the
 * length REACHES limit); outCount[u] <= limit - 1 reproduces that.  Equal predicts rank by item
 * id (the reference's Array.sort leaves them unspecified on the Node versions of its time).
 * userRows: nUsers x k; itemFactors: totalItems x k; skipPtr / skipIds: CSR of 0-based item ids,
 * ascending per user; outIds / outPredict: nUsers x limit (unused slots -1 / 0).  Host arrays. */
int ycnr_recommend_items(int dtype, int32_t k, int64_t nUsers, const void *userRows, int64_t totalItems, const void *itemFactors,
                         const int64_t *skipPtr, const int32_t *skipIds, double globalAvgShift, double minRecommendRating,
                         int32_t limit, int32_t *outIds, double *outPredict, int32_t *outCount, double *deviceMs);

#ifdef __cplusplus
}
#endif
#endif /* YCNR_ALS_H */
