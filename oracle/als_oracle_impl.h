/*
 * als_oracle_impl.h -- type-generic body of the CPU oracle.  Included twice by
 * als_oracle.c, once with T=float / PFX(x)=oracle_s##x and once with
 * T=double / PFX(x)=oracle_d##x, mirroring the reference's s/d naming
 * (cpp_utils/cpp_utils.js:15-19).
 *
 * TEST INFRASTRUCTURE ONLY.  See the header of als_oracle.c.
 */

/* Gather: Y[c,:] = fixed[indx[c],:]
 * follows EmfBase.copySubFixedFactors, lib/emf/EmfBase.js:537-555
 * (and its dead native twin cpp_utils/als_utils.cc:4-38). */
static void PFX(gather)(T *sub, const T *fixed, const int32_t *indx, int cols, int k)
{
  for (int c = 0; c < cols; c++)
    memcpy(sub + (size_t)c * k, fixed + (size_t)indx[c] * k, (size_t)k * sizeof(T));
}

/* A = Y^T * Y, full k x k product with beta = 0 (not a symmetric rank-k update)
 * follows BLAS.gemm(Y, Y, A, k, k, n, Trans, NoTrans), lib/emf/EmfWorker.js:231-232.
 * Summation over the n gathered rows in ascending order, in T. */
static void PFX(gram)(T *A, const T *Y, int n, int k)
{
  for (int i = 0; i < k * k; i++) A[i] = (T)0;
  for (int c = 0; c < n; c++) {
    const T *y = Y + (size_t)c * k;
    for (int i = 0; i < k; i++) {
      T yi = y[i];
      T *Ai = A + (size_t)i * k;
      for (int j = 0; j < k; j++) Ai[j] += yi * y[j];
    }
  }
}

/* General square solve, one right-hand side, LU with partial (row) pivoting:
 * the gesv-class routine behind Matrix.solveSquare(A, tmp, tmp),
 * lib/emf/EmfWorker.js:246.  A (row-major k x k) is destroyed, b becomes x.
 * Returns 0, or j+1 when the j-th pivot is exactly zero (LAPACK info > 0). */
static int PFX(gesv)(T *A, T *b, int k)
{
  for (int j = 0; j < k; j++) {
    int p = j;
    T amax = A[(size_t)j * k + j] < 0 ? -A[(size_t)j * k + j] : A[(size_t)j * k + j];
    for (int i = j + 1; i < k; i++) {
      T v = A[(size_t)i * k + j];
      if (v < 0) v = -v;
      if (v > amax) { amax = v; p = i; }
    }
    if (amax == (T)0) return j + 1;
    if (p != j) {
      for (int c = 0; c < k; c++) {
        T t = A[(size_t)j * k + c]; A[(size_t)j * k + c] = A[(size_t)p * k + c]; A[(size_t)p * k + c] = t;
      }
      T t = b[j]; b[j] = b[p]; b[p] = t;
    }
    T piv = A[(size_t)j * k + j];
    for (int i = j + 1; i < k; i++) {
      T l = A[(size_t)i * k + j] / piv;
      A[(size_t)i * k + j] = l;
      if (l != (T)0) {
        T *Ai = A + (size_t)i * k;
        const T *Aj = A + (size_t)j * k;
        for (int c = j + 1; c < k; c++) Ai[c] -= l * Aj[c];
      }
    }
  }
  /* forward substitution with the unit lower factor */
  for (int i = 1; i < k; i++) {
    T s = b[i];
    const T *Ai = A + (size_t)i * k;
    for (int c = 0; c < i; c++) s -= Ai[c] * b[c];
    b[i] = s;
  }
  /* back substitution with the upper factor */
  for (int i = k - 1; i >= 0; i--) {
    T s = b[i];
    const T *Ai = A + (size_t)i * k;
    for (int c = i + 1; c < k; c++) s -= Ai[c] * b[c];
    b[i] = s / Ai[i];
  }
  return 0;
}

/* One row of the hot loop, lib/emf/EmfWorker.js:214-248:
 *   gather -> A = Y^T Y -> A += diag(lambda*n) -> b = Y^T r -> solve -> store row.
 * scratch: Y[maxCols*k] | A[k*k] | b[k].  Returns the gesv info. */
static int PFX(solve_row)(double lambda, int k, int cols, const int32_t *indx, const T *vals,
                          const T *fixed, T *out_row, T *Y, T *A, T *b)
{
  PFX(gather)(Y, fixed, indx, cols, k);
  PFX(gram)(A, Y, cols, k);
  /* lambda.diagonal(_lambda * _n); A.add(lambda)  (EmfWorker.js:233-235):
   * the product is a JS double, the store into the typed array rounds it to T,
   * the add is a dense k x k add in T (off-diagonal terms add 0). */
  T lam = (T)(lambda * (double)cols);
  for (int i = 0; i < k; i++) A[(size_t)i * k + i] += lam;
  /* tmp = Y^T * rat  (EmfWorker.js:243-245) */
  for (int i = 0; i < k; i++) b[i] = (T)0;
  for (int c = 0; c < cols; c++) {
    const T *y = Y + (size_t)c * k;
    T r = vals[c];
    for (int i = 0; i < k; i++) b[i] += y[i] * r;
  }
  int info = PFX(gesv)(A, b, k);
  /* tmp.transpose(latentFactorsPart): in-place write of the factor row (EmfWorker.js:247) */
  for (int i = 0; i < k; i++) out_row[i] = b[i];
  return info;
}

/* Portion op = the body of EmfWorker.mw_calcTrainAlsPortion, lib/emf/EmfWorker.js:176-251.
 *   alsRows = [nRows, rowId0, cols0, rowId1, cols1, ...], alsIndx / alsVals concatenated
 *   (format written by EmfMaster.m_processFetchedPortionAlsOrRmse, lib/emf/EmfMaster.js:571-614).
 * fixed: the opposite side's factor matrix (read only); solved: this side's matrix,
 * row rowId overwritten in place (EmfBase.getLatentFactorsPartData, EmfBase.js:518-532).
 * Rows are independent within a portion, so the row loop is OpenMP-parallel when
 * threads > 1 (the reference gets the same parallelism from forked workers).
 * Rows recorded with cols == 0 (packer quirk) would give a singular A in the reference;
 * they are skipped here, as in the HIP path.
 * Returns the number of ratings consumed, or -(info) of the first failed solve. */
int64_t PFX(AlsCalcPortion)(double lambda, int k, const int32_t *alsRows, const int32_t *alsIndx,
                            const T *alsVals, const T *fixed, T *solved, int threads)
{
  int nRows = alsRows[0];
  if (nRows <= 0) return 0;
  int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nRows + 1));
  int maxCols = 0;
  offs[0] = 0;
  for (int r = 0; r < nRows; r++) {
    int cols = alsRows[1 + 2 * r + 1];
    offs[r + 1] = offs[r] + cols;
    if (cols > maxCols) maxCols = cols;
  }
  int64_t total = offs[nRows];
  int bad = 0;
  if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
  {
    T *Y = (T *)malloc(sizeof(T) * ((size_t)(maxCols > 0 ? maxCols : 1) * k));
    T *A = (T *)malloc(sizeof(T) * (size_t)k * k);
    T *b = (T *)malloc(sizeof(T) * (size_t)k);
#pragma omp for schedule(dynamic, 1)
    for (int r = 0; r < nRows; r++) {
      int rowId = alsRows[1 + 2 * r];
      int cols = alsRows[1 + 2 * r + 1];
      if (cols <= 0) continue;
      int info = PFX(solve_row)(lambda, k, cols, alsIndx + offs[r], alsVals + offs[r], fixed,
                                solved + (size_t)rowId * k, Y, A, b);
      if (info) {
#pragma omp critical
        if (!bad) bad = info;
      }
    }
    free(Y); free(A); free(b);
  }
  free(offs);
  return bad ? -(int64_t)bad : total;
}

/* Same row op on a plain CSR (rowPtr) description of rows [rowBegin, rowEnd): what a
 * whole half-step does when every portion of the step is processed
 * (EmfLord.alsTrainStep, lib/emf/EmfLord.js:963-984).  Rows with no rating are not
 * touched. */
int64_t PFX(AlsStepCsr)(double lambda, int k, int64_t rowBegin, int64_t rowEnd, const int64_t *rowPtr,
                        const int32_t *indx, const T *vals, const T *fixed, T *solved, int threads)
{
  int64_t maxCols = 0, total = 0;
  for (int64_t r = rowBegin; r < rowEnd; r++) {
    int64_t c = rowPtr[r + 1] - rowPtr[r];
    if (c > maxCols) maxCols = c;
    total += c;
  }
  int bad = 0;
  if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
  {
    T *Y = (T *)malloc(sizeof(T) * ((size_t)(maxCols > 0 ? maxCols : 1) * k));
    T *A = (T *)malloc(sizeof(T) * (size_t)k * k);
    T *b = (T *)malloc(sizeof(T) * (size_t)k);
#pragma omp for schedule(dynamic, 1)
    for (int64_t r = rowBegin; r < rowEnd; r++) {
      int64_t cols = rowPtr[r + 1] - rowPtr[r];
      if (cols <= 0) continue;
      int info = PFX(solve_row)(lambda, k, (int)cols, indx + rowPtr[r], vals + rowPtr[r], fixed,
                                solved + (size_t)r * k, Y, A, b);
      if (info) {
#pragma omp critical
        if (!bad) bad = info;
      }
    }
    free(Y); free(A); free(b);
  }
  return bad ? -(int64_t)bad : total;
}

/* RMSE partial sums of one portion: EmfWorker.mw_calcRmsePortion, lib/emf/EmfWorker.js:266-315,
 * with pred = uF.dot(iF) + globalAvgShift (EmfBase._alsPredict, lib/emf/EmfBase.js:825-827).
 * The dot product is accumulated in T in ascending factor order (an sdot/ddot-class call),
 * the three sums in double exactly as the JS numbers rSumDiff2 / rCnt / rSum.
 * out = {rSumDiff2, rCnt, rSum}. */
void PFX(RmsePortion)(int k, const int32_t *rmseRows, const int32_t *rmseIndx, const T *rmseVals,
                      const T *userFactors, const T *itemFactors, double globalAvgShift, double *out)
{
  double rSumDiff2 = 0, rCnt = 0, rSum = 0;
  int nRows = rmseRows[0];
  int64_t off = 0;
  for (int r = 0; r < nRows; r++) {
    int userId = rmseRows[1 + 2 * r];
    int cols = rmseRows[1 + 2 * r + 1];
    const T *uF = userFactors + (size_t)userId * k;
    for (int i = 0; i < cols; i++) {
      const T *iF = itemFactors + (size_t)rmseIndx[off + i] * k;
      T dot = (T)0;
      for (int f = 0; f < k; f++) dot += uF[f] * iF[f];
      double pred = (double)dot + globalAvgShift;
      double d = (double)rmseVals[off + i] - pred;
      rSumDiff2 += d * d;
      rSum += pred;
      rCnt += 1;
    }
    off += cols;
  }
  out[0] = rSumDiff2; out[1] = rCnt; out[2] = rSum;
}

/* The same sums over CSR rows [rowBegin, rowEnd) (one call = one portion). */
void PFX(RmseCsr)(int k, int64_t rowBegin, int64_t rowEnd, const int64_t *rowPtr, const int32_t *indx,
                  const T *vals, const T *userFactors, const T *itemFactors, double globalAvgShift,
                  double *out)
{
  double rSumDiff2 = 0, rCnt = 0, rSum = 0;
  for (int64_t u = rowBegin; u < rowEnd; u++) {
    const T *uF = userFactors + (size_t)u * k;
    for (int64_t p = rowPtr[u]; p < rowPtr[u + 1]; p++) {
      const T *iF = itemFactors + (size_t)indx[p] * k;
      T dot = (T)0;
      for (int f = 0; f < k; f++) dot += uF[f] * iF[f];
      double pred = (double)dot + globalAvgShift;
      double d = (double)vals[p] - pred;
      rSumDiff2 += d * d;
      rSum += pred;
      rCnt += 1;
    }
  }
  out[0] = rSumDiff2; out[1] = rCnt; out[2] = rSum;
}

/* Portion packer: EmfMaster.m_processFetchedPortionAlsOrRmse, lib/emf/EmfMaster.js:571-614.
 * Input: nData triplets sorted by row as the SQL returns them, ids 1-based.
 * compat != 0 reproduces the reference literally, including its end-of-data branch
 * (EmfMaster.js:594-603): the row that is open when the last triplet arrives is recorded
 * BEFORE that triplet is counted, so the last rating of every portion is dropped from the
 * row table.  compat == 0 records every rating.
 * Returns the number of rows written to bufRows. */
int PFX(PackPortion)(int nData, const int32_t *r1, const int32_t *c1, const T *rating,
                     int32_t *bufRows, int32_t *bufIndx, T *bufVals, int compat)
{
  int last_r = 0, r = 0, cols = 0;
  for (int i = 0; i < nData; i++) {
    int dr = r1[i] - 1, dc = c1[i] - 1; /* 1-based in db, 0-based in matrix (EmfMaster.js:584-586) */
    bufVals[i] = rating[i];
    bufIndx[i] = dc;
    if (i == 0) last_r = dr;
    if (compat) {
      if (last_r != dr || i == nData - 1) {
        bufRows[1 + r * 2] = last_r;
        bufRows[1 + r * 2 + 1] = cols;
        last_r = dr; cols = 0; r++;
      }
      cols++;
    } else {
      if (last_r != dr) {
        bufRows[1 + r * 2] = last_r;
        bufRows[1 + r * 2 + 1] = cols;
        last_r = dr; cols = 0; r++;
      }
      cols++;
      if (i == nData - 1) {
        bufRows[1 + r * 2] = last_r;
        bufRows[1 + r * 2 + 1] = cols;
        r++;
      }
    }
  }
  bufRows[0] = r;
  return r;
}
