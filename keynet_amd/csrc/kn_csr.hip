// kn_csr.hip -- order-preserving CSR x dense-block product for gfx950 (MI355X).
//
// Replaces scipy's csr_matvecs behind keynet.sparse.SparseMatrix.torchdot (keynet/sparse.py:488-492):
//     for each row i, for jj in STORED order:  y[i,:] = y[i,:] + (a_jj * x[col_jj,:])      (f32 mul, then f32 add; no FMA)
// Parallel over rows and over batch columns, strictly serial over the stored non-zeros of a row, so every output
// element sees exactly the rounding sequence of the reference (bit-exact; tests/test_parity_gpu.py).
//
// HBM/L2 layout: X is feature-major [cols, n_vecs] (batch contiguous): one gathered row of X is one coalesced
// 64*VEC*4-byte segment per wavefront.  Two kernels:
//   * csr_rows_kernel   one wavefront per row ("loose" rows): a wave streams the row's (col,val) pairs 64 at a time
//                       with one coalesced load, broadcasts them with v_readlane, gathers X rows.
//   * csr_group_kernel  rows that share one column sequence (the Cout rows of a conv output pixel; every row of a
//                       dense Linear) are processed RB at a time from ONE gather of X: RB*VEC accumulators per lane,
//                       values streamed as wave-uniform scalars.  Cuts vector-memory traffic per MAC by RB.
// Work items are dealt to XCDs in contiguous chunks (blockIdx%8 labels the XCD) so that the blocks resident on one
// XCD share a batch-column tile of X in that XCD's 4 MiB L2.
#include "kn_internal.h"
#include <type_traits>
#include <unordered_map>
#include <algorithm>
#include <cstring>

#pragma clang fp contract(off)

namespace kn {

static constexpr int WAVES = 4;  // wavefronts (rows / row-bundles) per 256-thread workgroup
static constexpr int RB = 16;    // rows per stored bundle of a pattern group (the kernels take RBK <= RB of them per wavefront)

template <int VEC> struct VecT;
template <> struct VecT<1> { using T = float; };
template <> struct VecT<2> { using T = float2; };
template <> struct VecT<4> { using T = float4; };

template <int VEC>
__device__ __forceinline__ void load_vec(float (&d)[VEC], const float* p) {
    if constexpr (VEC == 1) {
        d[0] = *p;
    } else if constexpr (VEC == 2) {
        float2 v = *reinterpret_cast<const float2*>(p);
        d[0] = v.x; d[1] = v.y;
    } else {
        float4 v = *reinterpret_cast<const float4*>(p);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&d)[VEC]) {
    if constexpr (VEC == 1) {
        *p = d[0];
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(d[0], d[1]);
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]);
    }
}

__device__ __forceinline__ float relu_f(float v) { return (v < 0.0f) ? 0.0f : v; }  // torch relu: NaN stays NaN

// item -> (column tile, row block): XCD lane x = bid&7 owns the contiguous item range [x*chunk, (x+1)*chunk)
__device__ __forceinline__ bool decode_item(int64_t n_items, int64_t& item) {
    const int64_t chunk = (n_items + 7) >> 3;
    const int64_t x = blockIdx.x & 7;
    const int64_t q = blockIdx.x >> 3;
    item = x * chunk + q;
    const int64_t hi = (x + 1) * chunk < n_items ? (x + 1) * chunk : n_items;
    return item < hi;
}

template <int VEC>
__global__ __launch_bounds__(256) void csr_rows_kernel(const int32_t* __restrict__ row_list, int64_t n_rows,
                                                       const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                       const float* __restrict__ data, const float* __restrict__ X, int64_t ldx,
                                                       float* __restrict__ Y, int64_t ldy, int64_t n_vecs, int relu, int64_t n_rb, float* absmax = nullptr,
                                                       int64_t x_plane = 0, int64_t y_plane = 0) {
    X += (int64_t)blockIdx.y * x_plane;          // kn_spmm_planes: grid dimension y = independent activation blocks under ONE operator (a single launch: y = 0, nothing moves)
    Y += (int64_t)blockIdx.y * y_plane;
    const int64_t n_ct = (n_vecs + 64 * VEC - 1) / (64 * VEC);
    int64_t item;
    if (!decode_item(n_ct * n_rb, item)) return;
    const int64_t ct = item / n_rb;
    const int64_t rb = item - ct * n_rb;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t ri = rb * WAVES + wave;
    if (ri >= n_rows) return;
    const int row = row_list ? row_list[ri] : (int)ri;
    const int64_t c = ct * (64 * VEC) + (int64_t)lane * VEC;
    const bool active = c < n_vecs;        // n_vecs % VEC == 0 is guaranteed by the launcher
    const int start = indptr[row];
    const int end = indptr[row + 1];
    const float* xc = X + (active ? c : 0);

    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = 0.0f;

    for (int base = start; base < end; base += 64) {
        const int n = (end - base) < 64 ? (end - base) : 64;   // wave-uniform
        int mycol = 0;
        float myval = 0.0f;
        if (lane < n) {
            mycol = indices[base + lane];
            myval = data[base + lane];
        }
        int i = 0;
        for (; i + 8 <= n; i += 8) {
            float xv[8][VEC];
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int col = __builtin_amdgcn_readlane(mycol, i + u);
                a[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myval), i + u));
                load_vec<VEC>(xv[u], xc + (int64_t)col * ldx);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    const float p = a[u] * xv[u][v];
                    acc[v] = acc[v] + p;
                }
            }
        }
        for (; i < n; i++) {
            const int col = __builtin_amdgcn_readlane(mycol, i);
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myval), i));
            float xv[VEC];
            load_vec<VEC>(xv, xc + (int64_t)col * ldx);
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                const float p = a * xv[v];
                acc[v] = acc[v] + p;
            }
        }
    }
    if (active) {
        if (relu) {
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] = relu_f(acc[v]);
        }
        store_vec<VEC>(Y + (int64_t)row * ldy + c, acc);
    }
    if (absmax) {                              // kn_spmm_screen on an operator of loose rows only (keyed pooling): max |y| rides along, no extra pass
        float m = 0.0f;
        if (active) {
#pragma unroll
            for (int v = 0; v < VEC; v++) m = fmaxf(m, fabsf(acc[v]));
        }
        kn_wave_absmax_commit(m, absmax, lane);
    }
}

// Loose rows when the batch window is 128 columns wide (a half batch of the overlapped forward): one row per HALF wavefront, so
// all 64 lanes carry 16-byte accesses (a whole wave on one row would leave lanes 32..63 idle, 8-byte accesses would halve the
// bytes per request).  Each lane reads its row's (col, val) pairs itself -- the 32 lanes of a half read one address, i.e. one
// request -- which suits the short rows this path sees (keyed pooling: ~9 non-zeros).  Same serial, stored-order mul-then-add.
__global__ __launch_bounds__(256) void csr_rows_pair_kernel(const int32_t* __restrict__ row_list, int64_t n_rows, const int32_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices, const float* __restrict__ data,
                                                            const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, int64_t n_vecs,
                                                            int relu, int64_t n_rb, float* absmax = nullptr) {
    const int64_t n_ct = (n_vecs + 127) / 128;
    int64_t item;
    if (!decode_item(n_ct * n_rb, item)) return;
    const int64_t ct = item / n_rb;
    const int64_t rb = item - ct * n_rb;
    const int lane = threadIdx.x & 63;
    const int64_t ri = (rb * WAVES + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const int64_t c = ct * 128 + (int64_t)(lane & 31) * 4;
    const bool valid = ri < n_rows && c < n_vecs;                  // n_vecs % 4 == 0 is guaranteed by the launcher
    if (!valid && absmax == nullptr) return;                       // (with a screen slot the lane stays for the wave-wide reduction, walking an empty row)
    const int row = valid ? (row_list ? row_list[ri] : (int)ri) : 0;
    const int start = valid ? indptr[row] : 0;
    const int end = valid ? indptr[row + 1] : 0;
    const float* xc = X + (valid ? c : 0);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int k = start;
    for (; k + 4 <= end; k += 4) {
        int col[4];
        float a[4], xv[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            col[u] = indices[k + u];
            a[u] = data[k + u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) load_vec<4>(xv[u], xc + (int64_t)col[u] * ldx);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const float p = a[u] * xv[u][v];
                acc[v] = acc[v] + p;
            }
    }
    for (; k < end; k++) {
        const int col = indices[k];
        const float a = data[k];
        float xv[4];
        load_vec<4>(xv, xc + (int64_t)col * ldx);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const float p = a * xv[v];
            acc[v] = acc[v] + p;
        }
    }
    if (relu) {
#pragma unroll
        for (int v = 0; v < 4; v++) acc[v] = relu_f(acc[v]);
    }
    if (valid) store_vec<4>(Y + (int64_t)row * ldy + c, acc);
    if (absmax) kn_wave_absmax_commit(valid ? fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3]))) : 0.0f, absmax, lane);
}

// Grouped rows: work item w = RB member rows [r0, r0+RB) of group g; all share the column sequence grp_cols[colptr[g]..].
// grp_vals layout per group: [j][Rpad] (Rpad = members rounded up to RB), so the RB values of column j are contiguous.
template <int VEC, int RBK>
__global__ __launch_bounds__(256) void csr_group_kernel(int64_t n_work, const int32_t* __restrict__ work_grp, const int32_t* __restrict__ work_r0,
                                                        const int32_t* __restrict__ grp_colptr, const int32_t* __restrict__ grp_cols,
                                                        const int32_t* __restrict__ grp_rowptr, const int32_t* __restrict__ grp_rows,
                                                        const int64_t* __restrict__ grp_valptr, const float* __restrict__ grp_vals,
                                                        const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy,
                                                        int64_t n_vecs, int relu, int64_t n_rb, int64_t x_plane = 0, int64_t y_plane = 0) {
    X += (int64_t)blockIdx.y * x_plane;          // (kn_spmm_planes, as in csr_rows_kernel)
    Y += (int64_t)blockIdx.y * y_plane;
    const int64_t n_ct = (n_vecs + 64 * VEC - 1) / (64 * VEC);
    int64_t item;
    if (!decode_item(n_ct * n_rb, item)) return;
    const int64_t ct = item / n_rb;
    const int64_t rb = item - ct * n_rb;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // the stored work list is in bundles of RB rows; a wavefront takes RBK <= RB of them (more wavefronts for operators
    // with few rows, e.g. a 121-row Linear: the serial walk over its columns cannot be split without changing rounding)
    constexpr int SUB = RB / RBK;
    const int64_t w = rb * WAVES + wave;
    if (w >= n_work * SUB) return;
    const int g = work_grp[w / SUB];
    const int r0 = work_r0[w / SUB] + (int)(w % SUB) * RBK;
    const int cbeg = grp_colptr[g];
    const int ncol = grp_colptr[g + 1] - cbeg;
    const int rbeg = grp_rowptr[g];
    const int nmem = grp_rowptr[g + 1] - rbeg;
    const int rpad = (nmem + RB - 1) / RB * RB;
    if (r0 >= nmem) return;
    const float* vals = grp_vals + grp_valptr[g] + r0;
    const int32_t* cols = grp_cols + cbeg;
    const int64_t c = ct * (64 * VEC) + (int64_t)lane * VEC;
    const bool active = c < n_vecs;
    const float* xc = X + (active ? c : 0);

    float acc[RBK][VEC];
#pragma unroll
    for (int r = 0; r < RBK; r++)
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[r][v] = 0.0f;

    // U columns per step; the next step's U gathers are issued before this step's arithmetic (a dense Linear is ONE
    // group of ~500 bundles, i.e. 1-2 wavefronts per SIMD: latency must be hidden inside the wavefront).  Thin bundles
    // (RBK <= 2: a 121-row Linear walking 785 columns, or fc6 of VGG-16: 2 x 256 outputs per wave walking 25 089 columns at two waves
    // per SIMD) have registers to spare: 16 gathers in flight whatever the vector width (fc6 in exact mode 6.27 -> 5.30 ms; what
    // bounds it then is re-gathering the 25.7 MB activation block once per wave from the Infinity Cache, ~10 TB/s).
    constexpr int U = (RBK <= 2) ? 16 : ((VEC == 4) ? 4 : 8);
    int j = 0;
    float xcur[U][VEC], xnxt[U][VEC];
    if (ncol >= U) {
#pragma unroll
        for (int u = 0; u < U; u++) load_vec<VEC>(xcur[u], xc + (int64_t)cols[u] * ldx);
    }
    for (; j + U <= ncol; j += U) {
        if (j + 2 * U <= ncol) {
#pragma unroll
            for (int u = 0; u < U; u++) load_vec<VEC>(xnxt[u], xc + (int64_t)cols[j + U + u] * ldx);
        }
        // the step's U x RBK values first (wave-uniform scalar loads, ONE wait for all of them), then the ordered arithmetic
        float av[U][RBK];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float* a = vals + (int64_t)(j + u) * rpad;
#pragma unroll
            for (int r = 0; r < RBK; r++) av[u][r] = a[r];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int r = 0; r < RBK; r++) {
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    const float p = av[u][r] * xcur[u][v];
                    acc[r][v] = acc[r][v] + p;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int v = 0; v < VEC; v++) xcur[u][v] = xnxt[u][v];
    }
    for (; j < ncol; j++) {
        float xv[VEC];
        load_vec<VEC>(xv, xc + (int64_t)cols[j] * ldx);
        const float* a = vals + (int64_t)j * rpad;
#pragma unroll
        for (int r = 0; r < RBK; r++) {
            const float ar = a[r];
#pragma unroll
            for (int v = 0; v < VEC; v++) {
                const float p = ar * xv[v];
                acc[r][v] = acc[r][v] + p;
            }
        }
    }
    if (active) {
#pragma unroll
        for (int r = 0; r < RBK; r++) {
            if (r0 + r < nmem) {
                const int row = grp_rows[rbeg + r0 + r];
                if (relu) {
#pragma unroll
                    for (int v = 0; v < VEC; v++) acc[r][v] = relu_f(acc[r][v]);
                }
                store_vec<VEC>(Y + (int64_t)row * ldy + c, acc[r]);
            }
        }
    }
}

// Big pattern groups (a keyed nn.Linear under the bit-exact contract: fc6 of VGG-16 = 4 097 rows x 25 089 shared columns).  The
// per-wave kernel above re-gathers the whole activation block once per wave (fc6: 2 049 waves x 25.7 MB = 52 GB from the Infinity
// Cache) with two waves per SIMD to hide the gather latency.  Here a workgroup owns 32 member rows x 64 batch columns and walks the
// shared column sequence in chunks of KS: the chunk's KS activation rows (one coalesced 256-byte segment each) and its KS x 32 values
// are staged in LDS once for the four waves (double buffered, the next chunk's loads in flight in registers during the arithmetic),
// each wave then reads one activation dword per lane and its 8 row values as two broadcast ds_read_b128 per column.  Strictly serial
// over the stored columns per output element, mul then add: same rounding sequence as the reference.
constexpr int BIG_ROWS = 32, BIG_KS = 32;

__global__ __launch_bounds__(256) void csr_big_group_kernel(int64_t n_big, const int32_t* __restrict__ big_grp, const int32_t* __restrict__ big_r0,
                                                            const int32_t* __restrict__ grp_colptr, const int32_t* __restrict__ grp_cols,
                                                            const int32_t* __restrict__ grp_rowptr, const int32_t* __restrict__ grp_rows,
                                                            const int64_t* __restrict__ grp_valptr, const float* __restrict__ grp_vals,
                                                            const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, int64_t n_vecs, int relu,
                                                            int64_t grid_big, const int32_t* __restrict__ long_rows, int64_t n_long,
                                                            const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ data) {
    __shared__ __attribute__((aligned(16))) float lds[2 * BIG_KS * 64 + 2 * BIG_KS * BIG_ROWS];
    float* Xs = lds;                          // [2][KS][64]
    float* Vs = lds + 2 * BIG_KS * 64;        // [2][KS][32]
    const int64_t n_ct = (n_vecs + 63) / 64;
    if ((int64_t)blockIdx.x >= grid_big) {
        // ---- long loose rows: workgroup = one row x four 64-column tiles (one per wave); 32 gathers in flight per wave --------------
        const int64_t tiles4 = (n_ct + 3) / 4;
        const int64_t lb = (int64_t)blockIdx.x - grid_big;
        const int64_t ri = lb / tiles4;
        const int64_t ctl = (lb - ri * tiles4) * 4 + (threadIdx.x >> 6);
        if (ri >= n_long || ctl >= n_ct) return;
        const int lane = threadIdx.x & 63;
        const int row = long_rows[ri];
        const int start = indptr[row], end = indptr[row + 1];
        const int64_t c = ctl * 64 + lane;
        const float* xc = X + (c < n_vecs ? c : 0);
        float acc = 0.0f;
        float xa[32], xb[32];
        int mycol = 0, ncol_ = 0;
        float myval = 0.0f, nval_ = 0.0f;
        if (start + lane < end) {
            mycol = indices[start + lane];
            myval = data[start + lane];
        }
#pragma unroll
        for (int i = 0; i < 32; i++) xa[i] = xc[(int64_t)__builtin_amdgcn_readlane(mycol, i) * ldx];       // lanes past the end hold column 0: a valid row
        for (int base = start; base < end; base += 64) {
            const int n = (end - base) < 64 ? (end - base) : 64;
            if (base + 64 + lane < end) {
                ncol_ = indices[base + 64 + lane];
                nval_ = data[base + 64 + lane];
            } else {
                ncol_ = 0;
                nval_ = 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 32; i++) xb[i] = xc[(int64_t)__builtin_amdgcn_readlane(mycol, 32 + i) * ldx];
#pragma unroll
            for (int i = 0; i < 32; i++) {
                if (i < n) {
                    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myval), i));
                    const float pr = a * xa[i];
                    acc = acc + pr;
                }
            }
#pragma unroll
            for (int i = 0; i < 32; i++) xa[i] = xc[(int64_t)__builtin_amdgcn_readlane(ncol_, i) * ldx];
#pragma unroll
            for (int i = 0; i < 32; i++) {
                if (32 + i < n) {
                    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myval), 32 + i));
                    const float pr = a * xb[i];
                    acc = acc + pr;
                }
            }
            mycol = ncol_;
            myval = nval_;
        }
        if (c < n_vecs) Y[(int64_t)row * ldy + c] = relu ? relu_f(acc) : acc;
        return;
    }
    int64_t item;
    if (!decode_item(n_ct * n_big, item)) return;
    const int64_t ct = item / n_big;
    const int64_t wi = item - ct * n_big;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = big_grp[wi];
    const int r0 = big_r0[wi];
    const int cbeg = grp_colptr[g];
    const int ncol = grp_colptr[g + 1] - cbeg;
    const int rbeg = grp_rowptr[g];
    const int nmem = grp_rowptr[g + 1] - rbeg;
    const int rpad = (nmem + RB - 1) / RB * RB;
    const int32_t* cols = grp_cols + cbeg;
    const float* vals = grp_vals + grp_valptr[g] + r0;
    const int64_t c = ct * 64 + lane;
    const float* xc = X + (c < n_vecs ? c : 0);
    // staging roles: wave w loads activation rows w, w+4, ... of the chunk (KS/4 per wave); thread t loads values (step t/8, rows 4*(t%8)..+3)
    const int v_step = tid >> 3, v_r4 = (tid & 7) * 4;
    const bool v_ok = (r0 + v_r4 < rpad);     // rows beyond the padded member count are not stored
    // register ring: chunk k waits in set k % PD from its request until it is written to LDS, PD - 1 chunk-times later.  Two
    // workgroups per CU (fc6: 516 work items) cannot hide an Infinity-Cache round trip behind one chunk of arithmetic; four chunks can.
    constexpr int PD = 4;
    float xr0[BIG_KS / 4], xr1[BIG_KS / 4], xr2[BIG_KS / 4], xr3[BIG_KS / 4];      // named sets: an indexed 2-D array would live in scratch
    float vr0[4], vr1[4], vr2[4], vr3[4];
    auto gload = [&](float (&xr)[BIG_KS / 4], float (&vr)[4], int j0) {
#pragma unroll
        for (int u = 0; u < BIG_KS / 4; u++) {
            const int j = j0 + wave + 4 * u;
            xr[u] = (j < ncol) ? xc[(int64_t)cols[j] * ldx] : 0.0f;
        }
        const int j = j0 + v_step;
#pragma unroll
        for (int e = 0; e < 4; e++) vr[e] = 0.0f;
        if (j < ncol && v_ok) {
            const float4 t = *reinterpret_cast<const float4*>(vals + (int64_t)j * rpad + v_r4);      // rpad % 8 == 0, r0 % 32 == 0: 16-byte aligned
            vr[0] = t.x; vr[1] = t.y; vr[2] = t.z; vr[3] = t.w;
        }
    };
    auto lstore = [&](const float (&xr)[BIG_KS / 4], const float (&vr)[4], int buf) {
#pragma unroll
        for (int u = 0; u < BIG_KS / 4; u++) Xs[(buf * BIG_KS + wave + 4 * u) * 64 + lane] = xr[u];
        *reinterpret_cast<float4*>(Vs + (buf * BIG_KS + v_step) * BIG_ROWS + v_r4) = make_float4(vr[0], vr[1], vr[2], vr[3]);
    };
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; r++) acc[r] = 0.0f;
    const int n_chunks = (ncol + BIG_KS - 1) / BIG_KS;
    auto compute = [&](int q) {
        const int buf = q & 1;
        const float* xs = Xs + buf * BIG_KS * 64 + lane;
        const float* vs = Vs + buf * BIG_KS * BIG_ROWS + wave * 8;
        const int steps = (ncol - q * BIG_KS) < BIG_KS ? (ncol - q * BIG_KS) : BIG_KS;
        if (steps == BIG_KS) {
#pragma unroll
            for (int j = 0; j < BIG_KS; j++) {
                const float x = xs[j * 64];
                const float4 a = *reinterpret_cast<const float4*>(vs + j * BIG_ROWS);
                const float4 b = *reinterpret_cast<const float4*>(vs + j * BIG_ROWS + 4);
                const float av[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float pr = av[r] * x;
                    acc[r] = acc[r] + pr;
                }
            }
        } else {
            for (int j = 0; j < steps; j++) {
                const float x = xs[j * 64];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float pr = vs[j * BIG_ROWS + r] * x;
                    acc[r] = acc[r] + pr;
                }
            }
        }
    };
    // one step of the pipeline for chunk q: arithmetic on chunk q from LDS; chunk q+1 (waiting in the register set passed in) -> the
    // other LDS buffer; request chunk q+1+PD into that set
    auto step = [&](int q, float (&xn)[BIG_KS / 4], float (&vn)[4]) {
        if (q >= n_chunks) return;
        compute(q);
        if (q + 1 < n_chunks) {
            lstore(xn, vn, (q + 1) & 1);       // that buffer's last readers finished before the barrier that ended chunk q-1
            if (q + 1 + PD < n_chunks) gload(xn, vn, (q + 1 + PD) * BIG_KS);
        }
        __syncthreads();
    };
    gload(xr0, vr0, 0);
    gload(xr1, vr1, 1 * BIG_KS);
    gload(xr2, vr2, 2 * BIG_KS);
    gload(xr3, vr3, 3 * BIG_KS);
    lstore(xr0, vr0, 0);
    gload(xr0, vr0, PD * BIG_KS);
    __syncthreads();
    for (int q = 0; q < n_chunks; q += PD) {      // chunk k waits in set k % PD
        step(q, xr1, vr1);
        step(q + 1, xr2, vr2);
        step(q + 2, xr3, vr3);
        step(q + 3, xr0, vr0);
    }
    if (c < n_vecs) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int m = r0 + wave * 8 + r;
            if (m < nmem) {
                float t = acc[r];
                if (relu) t = relu_f(t);
                Y[(int64_t)grp_rows[rbeg + m] * ldy + c] = t;
            }
        }
    }
}

void csr_free(CsrDev& c) {
    void* ptrs[] = {c.indptr, c.indices, c.data, c.data64, c.grp_colptr, c.grp_cols, c.grp_rowptr, c.grp_rows, c.grp_valptr, c.grp_vals,
                    c.work_grp, c.work_r0, c.loose_rows, c.big_grp, c.big_r0, c.long_rows, c.patch_rows, c.patch_ptr, c.patch_cols,
                    c.mf_grp[0], c.mf_grp[1], c.mf_grp[2], c.mf_r0[0], c.mf_r0[1], c.mf_r0[2], c.ws_grp, c.ws_r0, c.mf16_grp, c.mf16_r0};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    c = CsrDev();
}

// Host analysis: bucket rows by their exact column sequence.  Groups need >= 2 members and >= 1 column.
int csr_build_groups(kn_operator* h, const int32_t* indptr, const int32_t* indices, const float* data) {
    CsrDev& A = h->csr;
    const int64_t rows = A.rows;
    std::unordered_map<uint64_t, std::vector<int32_t>> buckets;   // hash -> group ids with that hash
    std::vector<std::vector<int32_t>> members;                    // group -> rows (ascending)
    std::vector<int32_t> rep;                                     // group -> representative row
    buckets.reserve((size_t)rows);
    for (int64_t r = 0; r < rows; r++) {
        const int32_t s = indptr[r], e = indptr[r + 1];
        if (e <= s) continue;
        uint64_t hsh = 1469598103934665603ull ^ (uint64_t)(e - s);
        for (int32_t k = s; k < e; k++) {
            hsh ^= (uint64_t)(uint32_t)indices[k];
            hsh *= 1099511628211ull;
        }
        auto& cand = buckets[hsh];
        int32_t found = -1;
        for (int32_t g : cand) {
            const int32_t rs = indptr[rep[g]], re = indptr[rep[g] + 1];
            if (re - rs == e - s && std::memcmp(indices + rs, indices + s, sizeof(int32_t) * (size_t)(e - s)) == 0) {
                found = g;
                break;
            }
        }
        if (found < 0) {
            found = (int32_t)members.size();
            members.emplace_back();
            rep.push_back((int32_t)r);
            cand.push_back(found);
        }
        members[found].push_back((int32_t)r);
    }
    // PATCHED members.  A permutation-keyed conv layer whose filter holds an exact zero has rows that are their pixel group's column sequence minus
    // that entry (AllConvNet conv5: 752 of 49 153 rows; walked alone they cost 1.9 ms of the layer's 11.4 -- a serial gather walk per row against one
    // shared gather per sixteen rows).  Such a row joins the group with 0.0f at the missing positions: a running sum that started at +0.0 is never
    // -0.0, so adding +-0.0 = 0.0f * x leaves it unchanged bit for bit whenever x is finite; csr_patch_guard_kernel rewrites the output for batch
    // columns where x at a missing position is NOT finite (0 * Inf would leak a NaN the reference's row does not have).
    constexpr int MAX_PATCH = 4;
    std::vector<std::vector<std::pair<int32_t, std::vector<int32_t>>>> patched(members.size());    // group -> (row, missing positions in the group's sequence)
    if (!A.tune.no_patch) {
        std::unordered_map<int32_t, std::vector<int32_t>> by_col;      // one of a group's first MAX_PATCH + 1 columns -> group
        for (size_t g = 0; g < members.size(); g++) {
            const int32_t s = indptr[rep[g]], e = indptr[rep[g] + 1];
            if (members[g].size() < 8 || e - s < 32) continue;
            for (int32_t t = 0; t <= MAX_PATCH && s + t < e; t++) by_col[indices[s + t]].push_back((int32_t)g);
        }
        for (size_t g1 = 0; g1 < members.size() && !by_col.empty(); g1++) {
            if (members[g1].size() != 1) continue;
            const int32_t r = members[g1][0];
            const int32_t rs = indptr[r], re = indptr[r + 1];
            if (re - rs < 32 - MAX_PATCH) continue;
            auto it = by_col.find(indices[rs]);
            if (it == by_col.end()) continue;
            for (int32_t G : it->second) {
                const int32_t gs = indptr[rep[G]], ge = indptr[rep[G] + 1];
                const int32_t miss = (ge - gs) - (re - rs);
                if (miss < 1 || miss > MAX_PATCH) continue;
                std::vector<int32_t> pos;
                int32_t k = rs;
                for (int32_t j = gs; j < ge; j++) {
                    if (k < re && indices[k] == indices[j]) k++;
                    else {
                        pos.push_back(j - gs);
                        if ((int)pos.size() > miss) break;
                    }
                }
                if (k == re && (int)pos.size() == miss) {
                    patched[(size_t)G].emplace_back(r, std::move(pos));
                    members[g1].clear();
                    break;
                }
            }
        }
    }
    std::vector<int32_t> patch_rows, patch_ptr{0}, patch_cols;
    std::vector<int32_t> colptr{0}, cols, rowptr{0}, grows, wgrp, wr0, bgrp, br0, loose, mfg[3], mfr[3], wsg, wsr, m16g, m16r;
    int64_t mf_rows = 0, mf_nnz = 0;
    std::vector<int64_t> valptr{0};
    std::vector<float> vals;
    int64_t grouped_nnz = 0;
    std::vector<char> in_group((size_t)rows, 0);
    for (size_t g = 0; g < members.size(); g++) {
        const auto& m = members[g];
        if (m.size() < 2) continue;
        const int32_t s = indptr[rep[g]], e = indptr[rep[g] + 1];
        const int32_t ncol = e - s;
        const int32_t gid = (int32_t)(colptr.size() - 1);
        cols.insert(cols.end(), indices + s, indices + e);
        colptr.push_back((int32_t)cols.size());
        grows.insert(grows.end(), m.begin(), m.end());
        const auto& pm = patched[g];
        for (const auto& pr : pm) grows.push_back(pr.first);
        rowptr.push_back((int32_t)grows.size());
        const int64_t n_mem = (int64_t)m.size() + (int64_t)pm.size();
        const int64_t rpad = (n_mem + RB - 1) / RB * RB;
        const int64_t v0 = (int64_t)vals.size();
        vals.resize((size_t)(v0 + rpad * ncol), 0.0f);
        for (size_t mi = 0; mi < m.size(); mi++) {
            const int32_t rs = indptr[m[mi]];
            for (int32_t j = 0; j < ncol; j++) vals[(size_t)(v0 + (int64_t)j * rpad + (int64_t)mi)] = data[rs + j];
            in_group[(size_t)m[mi]] = 1;
        }
        for (size_t pi = 0; pi < pm.size(); pi++) {
            const int32_t r = pm[pi].first;
            const std::vector<int32_t>& pos = pm[pi].second;
            int32_t k = indptr[r];
            size_t pp = 0;
            for (int32_t j = 0; j < ncol; j++) {
                if (pp < pos.size() && pos[pp] == j) {                  // missing here: the value stays 0.0f
                    patch_cols.push_back(indices[s + j]);
                    pp++;
                } else {
                    vals[(size_t)(v0 + (int64_t)j * rpad + (int64_t)(m.size() + pi))] = data[k++];
                }
            }
            patch_rows.push_back(r);
            patch_ptr.push_back((int32_t)patch_cols.size());
            in_group[(size_t)r] = 1;
        }
        valptr.push_back((int64_t)vals.size());
        if (n_mem >= 256 && ncol >= 2048 && !A.tune.no_big_groups) {      // a keyed nn.Linear: LDS-staged kernel, 32 rows per workgroup
            for (int64_t r0 = 0; r0 < n_mem; r0 += BIG_ROWS) {
                bgrp.push_back(gid);
                br0.push_back((int32_t)r0);
            }
            for (int64_t r0 = 0; r0 < n_mem; r0 += 16) {                       // the same group for the 16-row matrix-pipe kernel (narrow batches)
                m16g.push_back(gid);
                m16r.push_back((int32_t)r0);
            }
        } else {
            for (int64_t r0 = 0; r0 < n_mem; r0 += RB) {
                wgrp.push_back(gid);
                wr0.push_back((int32_t)r0);
            }
            if (n_mem >= MF_MIN_MEMBERS) {
                // matrix-pipe kernel: chunks of ONE 32-row block (a last block may be partly filled): 125 registers, four wavefronts per SIMD -- the adds of a result
                // block wait for its matrix instruction and only other wavefronts fill that wait.  AllConvNet kept in CSR form, whole forward, chunks of 3 / 2 / 1
                // blocks (229 / 157 / 125 registers): 32.84 / 32.26 / 31.38 ms.  Tuning::mf_nrb (KN_MF_NRB when the operator is created) picks another.
                const int nrb_max = A.tune.mf_nrb;
                for (int64_t r0 = 0; r0 < n_mem;) {
                    const int64_t left = n_mem - r0;
                    const int nrb = left >= 32 * nrb_max ? nrb_max : (int)((left + 31) / 32);
                    mfg[nrb - 1].push_back(gid);
                    mfr[nrb - 1].push_back((int32_t)r0);
                    r0 += 32 * nrb;
                }
                mf_rows += n_mem;
                mf_nnz += n_mem * ncol;
            } else {
                for (int64_t r0 = 0; r0 < n_mem; r0 += RB) {
                    wsg.push_back(gid);
                    wsr.push_back((int32_t)r0);
                }
            }
        }
        grouped_nnz += n_mem * ncol;
    }
    std::vector<int32_t> longrows;
    const bool use_long = !A.tune.no_big_groups;
    for (int64_t r = 0; r < rows; r++)
        if (!in_group[(size_t)r]) {
            if (use_long && indptr[r + 1] - indptr[r] >= 1024) longrows.push_back((int32_t)r);   // one wave would walk them latency-bound: deep-queue role
            else loose.push_back((int32_t)r);                     // includes empty rows (they must still be zeroed)
        }
    // loose rows of a big operator (keyed pooling: ~9 non-zeros per row, windows overlap): order them for L2 reuse of the gathers
    if (loose.size() >= 4096 && !A.tune.no_row_order) {
        int64_t lnnz = 0;
        for (int32_t r : loose) lnnz += indptr[r + 1] - indptr[r];
        if (lnnz <= 64 * (int64_t)loose.size()) loose = locality_order(loose, indptr, indices, h->cols, 64, 64);
    }
    A.n_groups = (int64_t)colptr.size() - 1;
    A.n_work = (int64_t)wgrp.size();
    A.n_big = (int64_t)bgrp.size();
    A.n_long = (int64_t)longrows.size();
    A.n_loose = (int64_t)loose.size();
    A.grouped_nnz = grouped_nnz;
    A.n_patch = (int64_t)patch_rows.size();
    int rc;
    if ((rc = upload(&A.patch_rows, patch_rows.data(), patch_rows.size()))) return rc;
    if ((rc = upload(&A.patch_ptr, patch_ptr.data(), patch_ptr.size()))) return rc;
    if ((rc = upload(&A.patch_cols, patch_cols.data(), patch_cols.size()))) return rc;
    if ((rc = upload(&A.grp_colptr, colptr.data(), colptr.size()))) return rc;
    if ((rc = upload(&A.grp_cols, cols.data(), cols.size()))) return rc;
    if ((rc = upload(&A.grp_rowptr, rowptr.data(), rowptr.size()))) return rc;
    if ((rc = upload(&A.grp_rows, grows.data(), grows.size()))) return rc;
    if ((rc = upload(&A.grp_valptr, valptr.data(), valptr.size()))) return rc;
    vals.resize(vals.size() + 64, 0.0f);         // the matrix-pipe kernel reads whole 32-row blocks: a partly filled last block reads past its column's rpad values
    if ((rc = upload(&A.grp_vals, vals.data(), vals.size()))) return rc;
    for (int k = 0; k < 3; k++) {
        A.n_mf[k] = (int64_t)mfg[k].size();
        if ((rc = upload(&A.mf_grp[k], mfg[k].data(), mfg[k].size()))) return rc;
        if ((rc = upload(&A.mf_r0[k], mfr[k].data(), mfr[k].size()))) return rc;
    }
    A.mf_rows = mf_rows;
    A.mf_nnz = mf_nnz;
    A.n_mf16 = (int64_t)m16g.size();
    if ((rc = upload(&A.mf16_grp, m16g.data(), m16g.size()))) return rc;
    if ((rc = upload(&A.mf16_r0, m16r.data(), m16r.size()))) return rc;
    A.n_ws = (int64_t)wsg.size();
    if ((rc = upload(&A.ws_grp, wsg.data(), wsg.size()))) return rc;
    if ((rc = upload(&A.ws_r0, wsr.data(), wsr.size()))) return rc;
    if ((rc = upload(&A.work_grp, wgrp.data(), wgrp.size()))) return rc;
    if ((rc = upload(&A.work_r0, wr0.data(), wr0.size()))) return rc;
    if ((rc = upload(&A.big_grp, bgrp.data(), bgrp.size()))) return rc;
    if ((rc = upload(&A.big_r0, br0.data(), br0.size()))) return rc;
    if ((rc = upload(&A.long_rows, longrows.data(), longrows.size()))) return rc;
    if ((rc = upload(&A.loose_rows, loose.data(), loose.size()))) return rc;
    return KN_OK;
}

// ---- locality order (host) --------------------------------------------------------------------------------------------------
// A permutation key scatters rows that touch neighbouring inputs all over the row index space.  Rows that are resident on
// one XCD at the same time should share gathered X rows in that XCD's L2, so the processing order is rebuilt from the
// operator's own sparsity structure: balls of `patch` rows grown breadth-first over the "shares a column" relation,
// seeded along a global breadth-first sweep so consecutive balls are adjacent.  The key is never needed.
std::vector<int32_t> locality_order(const std::vector<int32_t>& row_ids, const int32_t* indptr, const int32_t* indices, int64_t n_cols, int patch,
                                    int max_degree) {
    const size_t n = row_ids.size();
    // column -> local rows (CSC of the selected rows)
    std::vector<int32_t> cptr((size_t)n_cols + 1, 0);
    for (size_t i = 0; i < n; i++)
        for (int32_t k = indptr[row_ids[i]]; k < indptr[row_ids[i] + 1]; k++) cptr[(size_t)indices[k] + 1]++;
    for (int64_t c = 0; c < n_cols; c++) cptr[(size_t)c + 1] += cptr[(size_t)c];
    std::vector<int32_t> crow((size_t)cptr[(size_t)n_cols]);
    {
        std::vector<int32_t> fill(cptr.begin(), cptr.end() - 1);
        for (size_t i = 0; i < n; i++)
            for (int32_t k = indptr[row_ids[i]]; k < indptr[row_ids[i] + 1]; k++) crow[(size_t)fill[(size_t)indices[k]]++] = (int32_t)i;
    }
    auto for_neighbours = [&](int32_t i, auto&& fn) {
        for (int32_t k = indptr[row_ids[(size_t)i]]; k < indptr[row_ids[(size_t)i] + 1]; k++) {
            const int32_t c = indices[k];
            if (cptr[(size_t)c + 1] - cptr[(size_t)c] > max_degree) continue;
            for (int32_t q = cptr[(size_t)c]; q < cptr[(size_t)c + 1]; q++) fn(crow[(size_t)q]);
        }
    };
    std::vector<int32_t> sweep(n);
    {
        std::vector<char> seen(n, 0);
        size_t head = 0, tail = 0;
        for (size_t seed = 0; seed < n; seed++) {
            if (seen[seed]) continue;
            seen[seed] = 1;
            sweep[tail++] = (int32_t)seed;
            while (head < tail) {
                const int32_t i = sweep[head++];
                for_neighbours(i, [&](int32_t j) {
                    if (!seen[(size_t)j]) {
                        seen[(size_t)j] = 1;
                        sweep[tail++] = j;
                    }
                });
            }
        }
    }
    std::vector<char> taken(n, 0);
    std::vector<int32_t> out;
    out.reserve(n);
    std::vector<int32_t> ball;
    ball.reserve((size_t)patch * 2);
    for (size_t si = 0; si < n; si++) {
        const int32_t seed = sweep[si];
        if (taken[(size_t)seed]) continue;
        ball.clear();
        ball.push_back(seed);
        taken[(size_t)seed] = 1;
        size_t head = 0;
        while (head < ball.size() && (int)ball.size() < patch) {
            const int32_t i = ball[head++];
            for_neighbours(i, [&](int32_t j) {
                if (!taken[(size_t)j] && (int)ball.size() < patch) {
                    taken[(size_t)j] = 1;
                    ball.push_back(j);
                }
            });
        }
        for (int32_t i : ball) out.push_back(row_ids[(size_t)i]);
    }
    return out;
}


// Software-pipelined instantiation of the grouped kernel for the wide case (4 batch columns per lane, RBX = 16 or 8 member rows per
// wavefront): the permutation-keyed conv layers of configs 2-3.  Same arithmetic and order as csr_group_kernel; what differs is how a
// step's operands arrive (compare convtaps_exact_pipe_kernel, kn_conv.hip): everything wave-uniform stays scalar.  Per stored column j
//   * the column index comes from a scalar load (s_load_dword) issued one step ahead of the row request that needs it,
//   * the activation row is a saddr-form global_load_dwordx4 (SGPR-pair base + the lane's constant byte offset), FOUR rows in flight
//     per wavefront (a gathered row of a big permuted operator is an HBM / Infinity-Cache miss, ~2 us: one step of cover -- 128 packed
//     instructions x the co-resident waves -- is not enough),
//   * the RBX values are one s_load_dwordx16 / x8 into an SGPR tuple that the packed multiplies read directly, one step ahead,
// so a step is 2*RBX packed multiplies and adds and no other vector instruction.  All loads of the loop are inline asm with
// explicit waits (vector loads return in order: vmcnt(3) = "the oldest of the four rows in flight has landed"; scalar loads do not, so
// the next step's scalar loads are issued right after this step's lgkmcnt(0)).
template <int RBX>
__global__ __launch_bounds__(256) void csr_group_pipe_kernel(int64_t n_work, const int32_t* __restrict__ work_grp, const int32_t* __restrict__ work_r0,
                                                             const int32_t* __restrict__ grp_colptr, const int32_t* __restrict__ grp_cols,
                                                             const int32_t* __restrict__ grp_rowptr, const int32_t* __restrict__ grp_rows,
                                                             const int64_t* __restrict__ grp_valptr, const float* __restrict__ grp_vals,
                                                             const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy,
                                                             int64_t n_vecs, int relu, int64_t n_rb) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float vals_t __attribute__((ext_vector_type(RBX)));
    const int64_t n_ct = (n_vecs + 255) / 256;
    int64_t item;
    if (!decode_item(n_ct * n_rb, item)) return;
    const int64_t ct = item / n_rb;
    const int64_t rb = item - ct * n_rb;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    constexpr int SUB = RB / RBX;
    const int64_t w = rb * WAVES + wave;
    if (w >= n_work * SUB) return;
    // wave-uniform by construction; pinned to SGPRs so that the whole walk below stays on the scalar ALU
    const int g = __builtin_amdgcn_readfirstlane(work_grp[w / SUB]);
    const int r0 = __builtin_amdgcn_readfirstlane(work_r0[w / SUB]) + (int)(w % SUB) * RBX;
    const int cbeg = __builtin_amdgcn_readfirstlane(grp_colptr[g]);
    const int ncol = __builtin_amdgcn_readfirstlane(grp_colptr[g + 1]) - cbeg;
    const int rbeg = __builtin_amdgcn_readfirstlane(grp_rowptr[g]);
    const int nmem = __builtin_amdgcn_readfirstlane(grp_rowptr[g + 1]) - rbeg;
    const int rpad = (nmem + RB - 1) / RB * RB;
    if (r0 >= nmem) return;
    const int64_t c = ct * 256 + (int64_t)lane * 4;
    const bool active = c < n_vecs;
    const uint32_t lane_off_bytes = 4u * (uint32_t)(active ? c : 0);

    f32x2 acc[RBX][2];                                                    // [member row][columns 0-1 | 2-3]
#pragma unroll
    for (int r = 0; r < RBX; r++) acc[r][0] = acc[r][1] = f32x2{0.f, 0.f};

    if (ncol > 0) {
        // readfirstlane on a value that is already scalar is free; it guarantees an SGPR pair for the "s" operands below (hipcc
        // silently hands an "s" constraint a VGPR pair when it chose to keep the value in vector registers)
        auto uni = [](const uint64_t v) {
            return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
        };
        const int32_t* cols = grp_cols + cbeg;
        const uint64_t cbase = uni(reinterpret_cast<uint64_t>(cols));
        const uint64_t xbase = uni(reinterpret_cast<uint64_t>(X));
        const int64_t v0 = grp_valptr[g];
        const uint64_t vp = uni(reinterpret_cast<uint64_t>(grp_vals + v0 + r0));
        const int ldx_i = (int)ldx;                                       // cols * ldx < 2^31 elements: checked by the launcher
        const int vstep = rpad * 4;
        constexpr int XD = 4;                                             // activation rows in flight
        int col_nxt = 0;
        auto fetch_x = [&](f32x4& xr, const int col) {
            const uint64_t xaddr = xbase + 4 * (uint64_t)(uint32_t)(col * ldx_i);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(xr) : "v"(lane_off_bytes), "s"(xaddr));
        };
        auto clampj = [&](const int j) { return j < ncol ? j : ncol - 1; };     // past the end: the last column again (results unused)
        auto fetch_a = [&](vals_t& ar, const int j) {                     // values of column j, and the index of column j + XD - 1 (its row is requested after step j - 1)
            const uint64_t va = vp + (uint64_t)(uint32_t)clampj(j) * (uint64_t)(uint32_t)vstep;
            if constexpr (RBX == 16) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(ar) : "s"(va));
            else asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(ar) : "s"(va));
            const uint64_t caddr = cbase + 4 * (uint64_t)(uint32_t)clampj(j + XD);
            asm volatile("s_load_dword %0, %1, 0x0" : "=&s"(col_nxt) : "s"(caddr));
        };
        auto scalars_landed = [&](vals_t& ar) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ar), "+s"(col_nxt)); };
        auto row_landed = [&](f32x4& xr, auto younger) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xr) : "n"(decltype(younger)::value)); };
        // acc[r] += x * a[r]: separate IEEE multiply and add; the packed multiply reads an aligned SGPR pair and broadcasts its low or high
        // half with op_sel.  Order pinned (volatile asm): four multiplies of row pair k, then the four adds of pair k-1.
        auto mul4 = [&](const f32x2& xlo, const f32x2& xhi, const f32x2& a2, f32x2 (&pr)[4]) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr[0]) : "v"(xlo), "s"(a2));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr[1]) : "v"(xhi), "s"(a2));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[2]) : "v"(xlo), "s"(a2));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[3]) : "v"(xhi), "s"(a2));
        };
        auto add4 = [&](int r, const f32x2 (&pr)[4]) {
            asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r][0]) : "v"(pr[0]));
            asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r][1]) : "v"(pr[1]));
            asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r + 1][0]) : "v"(pr[2]));
            asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r + 1][1]) : "v"(pr[3]));
        };
        auto mac = [&](const f32x4& xv, const vals_t& av) {
            const f32x2 xlo = {xv.x, xv.y}, xhi = {xv.z, xv.w};
            f32x2 pa[4], pb[4];
#pragma unroll
            for (int r = 0; r < RBX; r += 4) {
                mul4(xlo, xhi, f32x2{av[r], av[r + 1]}, pa);
                if (r > 0) add4(r - 2, pb);
                mul4(xlo, xhi, f32x2{av[r + 2], av[r + 3]}, pb);
                add4(r, pa);
            }
            add4(RBX - 2, pb);
        };
        f32x4 x0, x1, x2, x3;
        vals_t a0, a1;
        fetch_x(x0, __builtin_amdgcn_readfirstlane(cols[0]));
        fetch_x(x1, __builtin_amdgcn_readfirstlane(cols[clampj(1)]));
        fetch_x(x2, __builtin_amdgcn_readfirstlane(cols[clampj(2)]));
        fetch_x(x3, __builtin_amdgcn_readfirstlane(cols[clampj(3)]));
        fetch_a(a0, 0);
        // one step: this step's scalars have landed -> request the next step's -> wait for this step's row (three younger rows stay in
        // flight) -> arithmetic -> request the row four steps ahead into the register the arithmetic just released
        auto step = [&](const int q, f32x4& xq, vals_t& aq, vals_t& an) {
            scalars_landed(aq);
            const int col_far = col_nxt;                                  // index of column q + XD
            fetch_a(an, q + 1);
            row_landed(xq, std::integral_constant<int, XD - 1>());
            __builtin_amdgcn_sched_barrier(0);
            if (q < ncol) mac(xq, aq);
            __builtin_amdgcn_sched_barrier(0);
            fetch_x(xq, col_far);
        };
        for (int q = 0; q < ncol; q += 4) {
            step(q, x0, a0, a1);
            step(q + 1, x1, a1, a0);
            step(q + 2, x2, a0, a1);
            step(q + 3, x3, a1, a0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(a0), "+s"(col_nxt));
    }
    if (!active) return;
#pragma unroll
    for (int r = 0; r < RBX; r++) {
        if (r0 + r < nmem) {
            const int row = grp_rows[rbeg + r0 + r];
            f32x4 t = {acc[r][0].x, acc[r][0].y, acc[r][1].x, acc[r][1].y};
            if (relu) {
                t.x = relu_f(t.x);
                t.y = relu_f(t.y);
                t.z = relu_f(t.z);
                t.w = relu_f(t.w);
            }
            __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(Y + (int64_t)row * ldy + c));      // not re-read by this launch (AllConvNet +0.7 %)
        }
    }
}

struct Planes {            // kn_spmm_planes: n independent activation blocks (block p at x + p * x_stride floats, its result at y + p * y_stride) under one operator, one launch per kernel
    int64_t n = 1, x_stride = 0, y_stride = 0;
};

template <int VEC, int RBK>
static int launch_csr(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, int relu, hipStream_t s, float* absmax = nullptr, const Planes pl = Planes()) {
    const int64_t n_ct = (n_vecs + 64 * VEC - 1) / (64 * VEC);
    const std::string np = pl.n > 1 ? (" x " + std::to_string(pl.n) + " planes in one launch") : std::string();
    if (A.n_work > 0) {
        const int64_t n_rb = (A.n_work * (RB / RBK) + WAVES - 1) / WAVES;
        const int64_t items = n_ct * n_rb;
        const int64_t grid = ((items + 7) / 8) * 8;
        KN_LAUNCH("csr_group_kernel<vec=" + std::to_string(VEC) + ",rows=" + std::to_string(RBK) + ">" + np, (csr_group_kernel<VEC, RBK>), dim3((unsigned)grid, (unsigned)pl.n), dim3(256), 0, s, A.n_work, A.work_grp, A.work_r0, A.grp_colptr, A.grp_cols,
                           A.grp_rowptr, A.grp_rows, A.grp_valptr, A.grp_vals, x, ldx, y, ldy, n_vecs, relu, n_rb, pl.x_stride, pl.y_stride);
    }
    if (A.n_loose > 0) {
        const int64_t n_rb = (A.n_loose + WAVES - 1) / WAVES;
        const int64_t items = n_ct * n_rb;
        const int64_t grid = ((items + 7) / 8) * 8;
        KN_LAUNCH("csr_rows_kernel<vec=" + std::to_string(VEC) + ">" + np, csr_rows_kernel<VEC>, dim3((unsigned)grid, (unsigned)pl.n), dim3(256), 0, s, A.loose_rows, A.n_loose, A.indptr, A.indices, A.data, x, ldx,
                           y, ldy, n_vecs, relu, n_rb, absmax, pl.x_stride, pl.y_stride);
    }
    KN_HIP(hipGetLastError());
    return KN_OK;
}

// grouped rows through the software-pipelined kernel (RBX member rows per wavefront), loose rows as in launch_csr<4, .>
template <int RBX>
static int launch_csr_pipe(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, int relu, hipStream_t s) {
    const int64_t n_ct = (n_vecs + 255) / 256;
    {
        const int64_t n_rb = (A.n_work * (RB / RBX) + WAVES - 1) / WAVES;
        const int64_t grid = ((n_ct * n_rb + 7) / 8) * 8;
        KN_LAUNCH("csr_group_pipe_kernel<rows=" + std::to_string(RBX) + ">", (csr_group_pipe_kernel<RBX>), dim3((unsigned)grid), dim3(256), 0, s, A.n_work, A.work_grp, A.work_r0, A.grp_colptr, A.grp_cols,
                           A.grp_rowptr, A.grp_rows, A.grp_valptr, A.grp_vals, x, ldx, y, ldy, n_vecs, relu, n_rb);
    }
    if (A.n_loose > 0) {
        const int64_t n_rb = (A.n_loose + WAVES - 1) / WAVES;
        const int64_t grid = ((n_ct * n_rb + 7) / 8) * 8;
        KN_LAUNCH("csr_rows_kernel<vec=4>", csr_rows_kernel<4>, dim3((unsigned)grid), dim3(256), 0, s, A.loose_rows, A.n_loose, A.indptr, A.indices, A.data, x, ldx, y, ldy,
                           n_vecs, relu, n_rb);
    }
    KN_HIP(hipGetLastError());
    return KN_OK;
}

// Patched group members (csr_build_groups): one lane per (patched row, batch column).  Nothing to do where the activations at the row's missing
// positions are finite -- the group kernel's result is then the reference's, bit for bit.  Otherwise the row is walked again in its OWN stored
// sequence (separate multiply and add), which is what the reference computes.  Runs after the group kernels, on their stream.
__global__ __launch_bounds__(256) void csr_patch_guard_kernel(int64_t n_patch, const int32_t* __restrict__ patch_rows, const int32_t* __restrict__ patch_ptr,
                                                              const int32_t* __restrict__ patch_cols, const int32_t* __restrict__ indptr,
                                                              const int32_t* __restrict__ indices, const float* __restrict__ data, const float* __restrict__ X,
                                                              int64_t ldx, float* __restrict__ Y, int64_t ldy, int64_t n_vecs, int relu) {
    const int64_t n_ct = (n_vecs + 255) / 256;
    const int64_t p = (int64_t)blockIdx.x / n_ct;
    const int64_t c = ((int64_t)blockIdx.x % n_ct) * 256 + threadIdx.x;
    if (p >= n_patch || c >= n_vecs) return;
    bool bad = false;
    for (int32_t k = patch_ptr[p]; k < patch_ptr[p + 1]; k++) {
        const float xv = X[(int64_t)patch_cols[k] * ldx + c];
        bad = bad || !(__builtin_fabsf(xv) <= 3.4028234663852886e38f);       // Inf or NaN
    }
    if (!bad) return;
    const int32_t r = patch_rows[p];
    float acc = 0.0f;
    for (int32_t k = indptr[r]; k < indptr[r + 1]; k++) {
        const float t = data[k] * X[(int64_t)indices[k] * ldx + c];
        acc = acc + t;
    }
    if (relu) acc = (acc < 0.0f) ? 0.0f : acc;                              // torch relu: NaN stays NaN
    Y[(int64_t)r * ldy + c] = acc;
}

static int csr_spmm_groups(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, uint32_t flags, hipStream_t s, float* rows_only_absmax = nullptr);

int csr_spmm(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, uint32_t flags, hipStream_t s, float* absmax, bool* absmax_fused) {
    // an operator of loose rows only (keyed pooling) is ONE launch of a row kernel: max |Y| (kn_spmm_screen) rides in its epilogue
    const bool rows_only = A.n_work == 0 && A.n_big == 0 && A.n_long == 0 && A.n_patch == 0 && A.n_loose > 0;
    if (absmax_fused) *absmax_fused = absmax != nullptr && rows_only;
    const int rc = csr_spmm_groups(A, x, ldx, n_vecs, y, ldy, flags, s, rows_only ? absmax : nullptr);
    if (rc != KN_OK || A.n_patch == 0) return rc;
    const int64_t n_ct = (n_vecs + 255) / 256;
    KN_LAUNCH("csr_patch_guard_kernel<" + std::to_string(A.n_patch) + " patched rows>", csr_patch_guard_kernel, dim3((unsigned)(A.n_patch * n_ct)), dim3(256), 0, s, A.n_patch,
              A.patch_rows, A.patch_ptr, A.patch_cols, A.indptr, A.indices, A.data, x, ldx, y, ldy, n_vecs, (flags & KN_FLAG_RELU) ? 1 : 0);
    KN_HIP(hipGetLastError());
    return KN_OK;
}

static int csr_spmm_groups(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, uint32_t flags, hipStream_t s, float* rows_only_absmax) {
    const int relu = (flags & KN_FLAG_RELU) ? 1 : 0;
    // a vector width is usable when pointers / strides allow it AND it keeps the wavefronts filled (64 * v columns per wave): n_vecs = 128
    // (a half-batch column window) takes v = 2 with every lane active rather than v = 4 with lanes 32..63 idle
    auto aligned = [&](int v) {
        return (n_vecs % v == 0) && (ldx % v == 0) && (ldy % v == 0) && (((uintptr_t)x) % (4 * v) == 0) && (((uintptr_t)y) % (4 * v) == 0) &&
               (v == 1 || 4 * n_vecs >= 3 * ((n_vecs + 64 * v - 1) / (64 * v)) * (64 * v));     // >= 75 % of the lanes busy
    };
    // (rows per wavefront, batch columns per lane): the most accumulators per lane that still gives the chip >= 2048
    // wavefronts; operators with few rows (a 121-row Linear, a dense Linear at n_vecs = 256) fall through to thinner
    // bundles / narrower vectors -- the walk over a row's columns is serial by contract, so parallelism can only come
    // from rows and batch columns.
    // Long loose rows (>= 1024 stored entries: a row of a keyed Linear, or of a wide permutation-keyed conv, whose pattern lost an entry to
    // an exact zero).  The deep-queue role of the big-group launch (one wave per row and 64 batch columns, 4 bytes per lane) exists for
    // NARROW batches, where such a row would otherwise be walked by a single wave; with a wide batch there are enough (row, 256-column
    // tile) pairs to fill the chip with the plain row kernel at 16 bytes per lane (AllConvNet conv5 at 4 096 images: 752 such rows took
    // 1.80 ms in the deep-queue role -- 15 % of the layer -- against ~0.2 ms this way).
    const bool long_as_rows = A.n_long > 0 && (n_vecs % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)x) % 16 == 0) && (((uintptr_t)y) % 16 == 0) &&
                              A.n_long * ((n_vecs + 255) / 256) >= 1024;
    const int64_t n_long_deep = long_as_rows ? 0 : A.n_long;
    if (long_as_rows) {
        const int64_t n_ct4 = (n_vecs + 255) / 256;
        const int64_t n_rb = (A.n_long + WAVES - 1) / WAVES;
        KN_LAUNCH("csr_rows_kernel<vec=4> (long rows)", csr_rows_kernel<4>, dim3((unsigned)(((n_ct4 * n_rb + 7) / 8) * 8)), dim3(256), 0, s, A.long_rows, A.n_long, A.indptr, A.indices,
                  A.data, x, ldx, y, ldy, n_vecs, relu, n_rb);
        KN_HIP(hipGetLastError());
    }
    // Big pattern groups (a keyed Linear in the reference's order): 16-row chunks with their products on the matrix pipe, one wavefront per chunk and 64 batch
    // columns, when that gives every SIMD two wavefronts (>= 2048 of them).  Same-process A/B against the LDS-staged kernel below (ms, 4096 x 25088 at 512 /
    // 1024 / 2048 columns: 3.38 / 5.85 / 11.17 against 2.68 / 4.37 / 8.17; 4096 x 4096 at 512 / 1024: 0.59 / 1.02 against 0.49 / 0.77); with ONE wavefront per
    // SIMD (VGG-16 fc6 at 256 images: 1 024 chunks) it loses -- 2.16 against 1.74 ms, a lone wavefront pays ~9 cycles per instruction -- and the LDS-staged
    // kernel stays.  Tuning::big_mfma16 (KN_BIG_MFMA16=0|1 when the operator is created) forces either.
    bool big_done = false;
    if (A.n_big > 0 && A.n_mf16 > 0 && n_vecs >= 64 && (A.tune.big_mfma16 < 0 ? A.n_mf16 * ((n_vecs + 63) / 64) >= 2048 : A.tune.big_mfma16 > 0)) {
        const int rc = csr_group_mfma16_spmm(A, x, ldx, n_vecs, y, ldy, relu, s);
        if (rc) return rc;
        big_done = true;
    }
    if ((A.n_big > 0 && !big_done) || n_long_deep > 0) {
        const int64_t n_ct = (n_vecs + 63) / 64;
        const int64_t n_big = big_done ? 0 : A.n_big;
        const int64_t grid_big = ((n_ct * n_big + 7) / 8) * 8;
        const int64_t grid_long = n_long_deep * ((n_ct + 3) / 4);
        KN_LAUNCH("csr_big_group_kernel", csr_big_group_kernel, dim3((unsigned)(grid_big + grid_long)), dim3(256), 0, s, n_big, A.big_grp, A.big_r0, A.grp_colptr, A.grp_cols,
                           A.grp_rowptr, A.grp_rows, A.grp_valptr, A.grp_vals, x, ldx, y, ldy, n_vecs, relu, grid_big, A.long_rows, n_long_deep, A.indptr, A.indices, A.data);
        KN_HIP(hipGetLastError());
    }
    if ((A.n_big > 0 || A.n_long > 0) && A.n_work == 0 && A.n_loose == 0) return KN_OK;
    // Pattern groups with >= MF_MIN_MEMBERS members and long stored sequences on a wide batch: products on the matrix pipe, sums on the vector
    // ALU (kn_csr_mfma.hip: same bits; an activation row is fetched once per 96 member rows instead of once per 16 -- AllConvNet conv2 reads
    // 11.7 GB from the fabric per launch instead of 27.8).  Same-process A/B under sustained load (tools/ab_allconv.py): the two formulations
    // take the same ALU time -- the f32 matrix instruction runs on the vector ALU's FP32 lanes -- so the long layers are within 0-2.5 % of each
    // other, and layers with short sequences (AllConvNet conv1: 28 columns, conv8: 193) are faster on the vector-ALU pipeline, whose five
    // wavefronts per SIMD hide the ring's start-up better: those stay there (mean stored columns per member row < 256).  The remaining small
    // groups and the loose rows go through the kernels below.  Tuning::group_mfma (KN_GROUP_MFMA=0|1 when the operator is created) forces either.
    {
        const int64_t n_mf = A.n_mf[0] + A.n_mf[1] + A.n_mf[2];
        const bool long_rows = A.mf_rows > 0 && A.mf_nnz >= 256 * A.mf_rows;
        const bool want = A.tune.group_mfma < 0 ? long_rows : A.tune.group_mfma > 0;
        if (want && n_mf > 0 && n_vecs >= 128 && n_mf * ((n_vecs + 255) / 256) * WAVES >= 2048) {
            int rc = csr_group_mfma_spmm(A, x, ldx, n_vecs, y, ldy, relu, s);
            if (rc) return rc;
            if (A.n_ws == 0 && A.n_loose == 0) return KN_OK;
            CsrDev R = A;                                // (a view: same device arrays, the small groups' bundle list in place of all groups')
            R.work_grp = A.ws_grp;
            R.work_r0 = A.ws_r0;
            R.n_work = A.n_ws;
            R.n_mf[0] = R.n_mf[1] = R.n_mf[2] = 0;
            R.n_big = R.n_long = 0;
            return csr_spmm_groups(R, x, ldx, n_vecs, y, ldy, flags, s);
        }
    }
    // short loose rows over a batch window that fills only half of a 256-column wave tile: one row per half wavefront
    if (A.n_work == 0 && A.n_loose >= 4096 && A.nnz <= 32 * A.n_loose && n_vecs % 128 == 0 && (n_vecs / 128) % 2 == 1 && (ldx % 4 == 0) && (ldy % 4 == 0) &&
        (((uintptr_t)x) % 16 == 0) && (((uintptr_t)y) % 16 == 0)) {
        const int64_t n_rb = (A.n_loose + 2 * WAVES - 1) / (2 * WAVES);
        const int64_t items = ((n_vecs + 127) / 128) * n_rb;
        KN_LAUNCH("csr_rows_pair_kernel", csr_rows_pair_kernel, dim3((unsigned)(((items + 7) / 8) * 8)), dim3(256), 0, s, A.loose_rows, A.n_loose, A.indptr, A.indices, A.data, x, ldx, y,
                           ldy, n_vecs, relu, n_rb, rows_only_absmax);
        KN_HIP(hipGetLastError());
        return KN_OK;
    }
    const int64_t loose = (A.n_loose + WAVES - 1) / WAVES * WAVES;
    auto waves = [&](int v, int rbk) { return (A.n_work * (RB / rbk) + loose) * ((n_vecs + 64 * v - 1) / (64 * v)); };
    constexpr int64_t ENOUGH = 2048;
#define KN_TRY(V, R) \
    if (aligned(V) && waves(V, R) >= ENOUGH) return launch_csr<V, R>(A, x, ldx, n_vecs, y, ldy, relu, s, rows_only_absmax);
    // wide case: the software-pipelined grouped kernel, 16 member rows per wavefront when the groups fill such bundles, else 8
    // (Tuning::no_group_pipe: the plain grouped kernel, for the parity tests' side-by-side)
    if (A.n_work > 0 && aligned(4) && A.cols * ldx < ((int64_t)1 << 31) && !A.tune.no_group_pipe) {
        const int64_t grouped_rows = A.rows - A.n_loose - A.n_long - A.n_big * 32;           // upper bound (a big group's last bundle may be partial)
        if (waves(4, 16) >= ENOUGH && 10 * grouped_rows >= 7 * A.n_work * 16) return launch_csr_pipe<16>(A, x, ldx, n_vecs, y, ldy, relu, s);
        if (waves(4, 8) >= ENOUGH) return launch_csr_pipe<8>(A, x, ldx, n_vecs, y, ldy, relu, s);
    }
    KN_TRY(4, 8)
    KN_TRY(2, 8)
    KN_TRY(4, 2)
    KN_TRY(1, 8)
    KN_TRY(2, 2)
    KN_TRY(1, 2)
#undef KN_TRY
    if (A.n_work == 0) {   // loose rows only: the bundle height is irrelevant, take the widest aligned vector
        if (aligned(4) && n_vecs >= 256) return launch_csr<4, 8>(A, x, ldx, n_vecs, y, ldy, relu, s, rows_only_absmax);
        if (aligned(2) && n_vecs >= 128) return launch_csr<2, 8>(A, x, ldx, n_vecs, y, ldy, relu, s, rows_only_absmax);
        return launch_csr<1, 8>(A, x, ldx, n_vecs, y, ldy, relu, s, rows_only_absmax);
    }
    return launch_csr<1, 1>(A, x, ldx, n_vecs, y, ldy, relu, s, rows_only_absmax);
}

// kn_spmm_planes: Y_p = W . X_p for p = 0 .. n_planes - 1 with ONE launch per kernel (grid dimension y = plane) instead of n_planes launches.  What it is for: the split
// application of a filled-in conv (keynet_amd/sparse.py: the spatial CSR of all taps applied to every input channel's plane -- 64 .. 512 planes, each a launch too small to fill
// the chip).  Same kernels, same instruction sequence per output element: bit-identical to n_planes kn_spmm calls.  Operators that need more than the grouped / loose-row
// kernels (big groups, long rows, matrix-pipe groups, patched rows) are refused with KN_ERR_UNSUPPORTED: the caller loops over kn_spmm then.
int csr_spmm_planes(const CsrDev& A, const float* x, int64_t ldx, int64_t x_stride, int64_t n_planes, int64_t n_vecs, float* y, int64_t ldy, int64_t y_stride, uint32_t flags, hipStream_t s) {
    KN_REQUIRE(A.n_big == 0 && A.n_long == 0 && A.n_patch == 0, KN_ERR_UNSUPPORTED, "kn_spmm_planes: operator has big / long / patched rows (loop over kn_spmm)");
    KN_REQUIRE(n_planes >= 1 && n_planes <= 65535, KN_ERR_UNSUPPORTED, "kn_spmm_planes: 1 .. 65535 planes");
    const int relu = (flags & KN_FLAG_RELU) ? 1 : 0;
    Planes pl;
    pl.n = n_planes;
    pl.x_stride = x_stride;
    pl.y_stride = y_stride;
    auto aligned = [&](int v) {
        return (n_vecs % v == 0) && (ldx % v == 0) && (ldy % v == 0) && (x_stride % v == 0) && (y_stride % v == 0) && (((uintptr_t)x) % (4 * v) == 0) && (((uintptr_t)y) % (4 * v) == 0) &&
               (v == 1 || 4 * n_vecs >= 3 * ((n_vecs + 64 * v - 1) / (64 * v)) * (64 * v));
    };
    const int64_t loose = (A.n_loose + WAVES - 1) / WAVES * WAVES;
    auto waves = [&](int v, int rbk) { return (A.n_work * (RB / rbk) + loose) * ((n_vecs + 64 * v - 1) / (64 * v)) * n_planes; };
    constexpr int64_t ENOUGH = 2048;
#define KN_TRY(V, R) \
    if (aligned(V) && waves(V, R) >= ENOUGH) return launch_csr<V, R>(A, x, ldx, n_vecs, y, ldy, relu, s, nullptr, pl);
    KN_TRY(4, 8)
    KN_TRY(2, 8)
    KN_TRY(4, 2)
    KN_TRY(1, 8)
    KN_TRY(2, 2)
    KN_TRY(1, 2)
#undef KN_TRY
    if (aligned(4) && n_vecs >= 256) return launch_csr<4, 8>(A, x, ldx, n_vecs, y, ldy, relu, s, nullptr, pl);
    if (aligned(2) && n_vecs >= 128) return launch_csr<2, 8>(A, x, ldx, n_vecs, y, ldy, relu, s, nullptr, pl);
    return launch_csr<1, 8>(A, x, ldx, n_vecs, y, ldy, relu, s, nullptr, pl);
}

}  // namespace kn
