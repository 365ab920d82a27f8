// kn_csr_f64.hip -- order-preserving CSR x dense-block product for a FLOAT64 operator (gfx950).
//
// SparseMatrix.torchdot (keynet/sparse.py:488-492) hands `self._matrix` to scipy whatever its dtype; a float64 operator against the float32
// activations (`x_torch.type(torch.FloatTensor)`, :489-491) makes numpy up-cast: scipy's csr_matvecs runs on (f64 values, x up-cast to f64
// element by element -- exact), accumulates in f64 and returns a float64 block; the NEXT layer's coercion rounds it to f32 once.  The only
// key-net the reference ships (demo/keynet_challenge_lenet_10AUG20.pkl, demo/challenge.ipynb cell 5) carries such conv / pool operators.
//     for each row i, for jj in STORED order:  y[i,:] = y[i,:] + (a_jj * (double)x[col_jj,:])     (f64 mul, then f64 add; no FMA)
// One wavefront per (row, 64 * VEC batch columns); strictly serial over the stored entries of a row.  The output is either the float64 block
// itself (kn_spmm_f64: what the reference's torchdot returns) or that block rounded to f32 once (kn_spmm: what the next layer consumes; ReLU
// commutes with the rounding).  f64 vector multiply / add issue at the f32 no-FMA rate on MI355X, so no second formulation is kept for it.
#include "kn_internal.h"

#pragma clang fp contract(off)

namespace kn {

static constexpr int WAVES64 = 4;

template <int VEC, typename TOUT>
__global__ __launch_bounds__(256) void csr_rows_f64_kernel(int64_t n_rows, const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                           const double* __restrict__ data, const float* __restrict__ X, int64_t ldx,
                                                           TOUT* __restrict__ Y, int64_t ldy, int64_t n_vecs, int relu, int64_t n_rb) {
    const int64_t n_ct = (n_vecs + 64 * VEC - 1) / (64 * VEC);
    // XCD lane x = bid & 7 owns a contiguous range of (column tile, row block) items: the blocks resident on one XCD share gathered rows in its L2
    const int64_t n_items = n_ct * n_rb;
    const int64_t chunk = (n_items + 7) >> 3;
    const int64_t xcd = blockIdx.x & 7;
    const int64_t item = xcd * chunk + (blockIdx.x >> 3);          // (blockIdx.x >> 3) < chunk by the grid size
    if (item >= n_items) return;
    const int64_t ct = item / n_rb;
    const int64_t rb = item - ct * n_rb;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t row = rb * WAVES64 + wave;
    if (row >= n_rows) return;
    const int64_t c = ct * (64 * VEC) + (int64_t)lane * VEC;
    const bool active = c < n_vecs;                     // n_vecs % VEC == 0 is guaranteed by the launcher
    const int start = indptr[row];
    const int end = indptr[row + 1];
    const float* xc = X + (active ? c : 0);

    double acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = 0.0;

    for (int base = start; base < end; base += 64) {
        const int n = (end - base) < 64 ? (end - base) : 64;          // wave-uniform
        int mycol = 0;
        double myval = 0.0;
        if (lane < n) {
            mycol = indices[base + lane];
            myval = data[base + lane];
        }
        const int vlo = __builtin_bit_cast(int2, myval).x, vhi = __builtin_bit_cast(int2, myval).y;
        int i = 0;
        for (; i + 8 <= n; i += 8) {                                   // eight gathers in flight, then the ordered arithmetic
            float xv[8][VEC];
            double a[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int col = __builtin_amdgcn_readlane(mycol, i + u);
                a[u] = __builtin_bit_cast(double, make_int2(__builtin_amdgcn_readlane(vlo, i + u), __builtin_amdgcn_readlane(vhi, i + u)));
                const float* p = xc + (int64_t)col * ldx;
                if constexpr (VEC == 1) {
                    xv[u][0] = *p;
                } else if constexpr (VEC == 2) {
                    const float2 t = *reinterpret_cast<const float2*>(p);
                    xv[u][0] = t.x; xv[u][1] = t.y;
                } else {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][3] = t.w;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    const double p = a[u] * (double)xv[u][v];          // separate multiply ...
                    acc[v] = acc[v] + p;                               // ... then add
                }
            }
        }
        for (; i < n; i++) {
            const int col = __builtin_amdgcn_readlane(mycol, i);
            const double a = __builtin_bit_cast(double, make_int2(__builtin_amdgcn_readlane(vlo, i), __builtin_amdgcn_readlane(vhi, i)));
            const float* p = xc + (int64_t)col * ldx;
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                const double pr = a * (double)p[v];
                acc[v] = acc[v] + pr;
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int v = 0; v < VEC; v++) {
        double t = acc[v];
        if (relu) t = (t < 0.0) ? 0.0 : t;                             // torch relu on the f64 block: NaN stays NaN
        Y[row * ldy + c + v] = (TOUT)t;                                // TOUT = float: ONE round-to-nearest-even, the next layer's x.type(FloatTensor)
    }
}

template <typename TOUT>
int csr_f64_spmm(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, TOUT* y, int64_t ldy, uint32_t flags, hipStream_t s) {
    const int relu = (flags & KN_FLAG_RELU) ? 1 : 0;
    const int64_t n_rb = (A.rows + WAVES64 - 1) / WAVES64;
    auto aligned = [&](int v) { return (n_vecs % v == 0) && (ldx % v == 0) && (((uintptr_t)x) % (4 * v) == 0) && n_vecs >= 64 * v; };
    auto grid = [&](int v) { return dim3((unsigned)((((n_vecs + 64 * v - 1) / (64 * v)) * n_rb + 7) / 8 * 8)); };
    const char* out = std::is_same<TOUT, double>::value ? "f64" : "f32";
    if (aligned(4) && n_rb * ((n_vecs + 255) / 256) >= 1024) {
        KN_LAUNCH(std::string("csr_rows_f64_kernel<vec=4,out=") + out + ">", (csr_rows_f64_kernel<4, TOUT>), grid(4), dim3(256), 0, s, A.rows, A.indptr, A.indices, A.data64, x, ldx, y, ldy,
                  n_vecs, relu, n_rb);
    } else if (aligned(2) && n_rb * ((n_vecs + 127) / 128) >= 1024) {
        KN_LAUNCH(std::string("csr_rows_f64_kernel<vec=2,out=") + out + ">", (csr_rows_f64_kernel<2, TOUT>), grid(2), dim3(256), 0, s, A.rows, A.indptr, A.indices, A.data64, x, ldx, y, ldy,
                  n_vecs, relu, n_rb);
    } else {
        KN_LAUNCH(std::string("csr_rows_f64_kernel<vec=1,out=") + out + ">", (csr_rows_f64_kernel<1, TOUT>), grid(1), dim3(256), 0, s, A.rows, A.indptr, A.indices, A.data64, x, ldx, y, ldy,
                  n_vecs, relu, n_rb);
    }
    KN_HIP(hipGetLastError());
    return KN_OK;
}

template int csr_f64_spmm<float>(const CsrDev&, const float*, int64_t, int64_t, float*, int64_t, uint32_t, hipStream_t);
template int csr_f64_spmm<double>(const CsrDev&, const float*, int64_t, int64_t, double*, int64_t, uint32_t, hipStream_t);

}  // namespace kn
