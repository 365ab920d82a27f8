/*
 * oracle/kn_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into, imported by, or called from the product path).
 *
 * CPU restatement of the arithmetic the reference's keyed forward ends in.  The reference
 * (visym/keynet, keynet/sparse.py:488-492 SparseMatrix.torchdot and :603-612 TiledMatrix.torchdot) delegates the
 * product W.dot(X) to a third-party dependency that is NOT under /root/reference:
 *
 *     scipy.sparse  (setup.py:22-31 lists `scipy`, unpinned; this image: scipy 1.15.3, plain x86-64 gcc build)
 *     _cs_matrix._matmul_multivector -> _sparsetools.csr_matvecs   (scipy/sparse/sparsetools/csr.h)
 *
 * Published algorithm restated here:  for a CSR matrix (Ap, Aj, Ax) with n_row rows and a dense row-major block
 * of n_vecs vectors X[n_col][n_vecs], Y[n_row][n_vecs] (pre-zeroed by the caller, as scipy does with np.zeros):
 *
 *     for i in rows:  for jj in Ap[i]..Ap[i+1] (STORED order, indices may be unsorted / non-canonical):
 *         y[i, :] += Ax[jj] * x[Aj[jj], :]            (axpy: one rounding for the product, one for the sum; no FMA)
 *
 * Parity pinning: tests/test_oracle_golden.py checks this function bit-for-bit against outputs produced by the
 * reference itself in the build container (the .npz files under tests/golden, generator tests/golden/make_golden.py), including
 * the demo/challenge.ipynb known answer.  Build with -ffp-contract=off (see oracle/Makefile).
 */
#include <stdint.h>
#include <stddef.h>

/* f32 matrix, f32 vectors, f32 accumulate (every layer the reference builds itself: keynet/layer.py:32-70) */
void kn_oracle_csr_matvecs_f32(int64_t n_row, int64_t n_col, int64_t n_vecs,
                               const int32_t* Ap, const int32_t* Aj, const float* Ax,
                               const float* Xx, float* Yx)
{
    (void)n_col;
    for (int64_t i = 0; i < n_row; i++) {
        float* y = Yx + (size_t)n_vecs * (size_t)i;
        for (int32_t jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            const float a = Ax[jj];
            const float* x = Xx + (size_t)n_vecs * (size_t)Aj[jj];
            for (int64_t k = 0; k < n_vecs; k++) {
                const float p = a * x[k];   /* separate multiply ... */
                y[k] = y[k] + p;            /* ... then add (compiled with -ffp-contract=off) */
            }
        }
    }
}

/* f64 matrix x (f32 vectors upcast to f64) -> f64: numpy's upcast rule, hit by the pickled public challenge keynet
 * whose conv/pool operators carry float64 data (demo/keynet_challenge_lenet_10AUG20.pkl). */
void kn_oracle_csr_matvecs_f64(int64_t n_row, int64_t n_col, int64_t n_vecs,
                               const int32_t* Ap, const int32_t* Aj, const double* Ax,
                               const double* Xx, double* Yx)
{
    (void)n_col;
    for (int64_t i = 0; i < n_row; i++) {
        double* y = Yx + (size_t)n_vecs * (size_t)i;
        for (int32_t jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            const double a = Ax[jj];
            const double* x = Xx + (size_t)n_vecs * (size_t)Aj[jj];
            for (int64_t k = 0; k < n_vecs; k++) {
                const double p = a * x[k];
                y[k] = y[k] + p;
            }
        }
    }
}

/* torch.nn.functional.relu on the whole [D+1, N] block, homogeneous row included (keynet/system.py:92, layer.py:93) */
void kn_oracle_relu_f32(int64_t n, float* y)
{
    for (int64_t i = 0; i < n; i++) {
        /* torch clamp_min semantics: NaN propagates, -0.0 -> 0.0 compares equal */
        y[i] = (y[i] < 0.0f) ? 0.0f : y[i];
    }
}
