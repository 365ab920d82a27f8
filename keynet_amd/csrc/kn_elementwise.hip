// kn_elementwise.hip -- the memory-bound glue of the keyed forward (gfx950).
//   relu_inplace        nn.ReLU over [rows, n_vecs] incl. the homogeneous row (keynet/system.py:92) when it cannot be fused
//   affine_to_linear    keynet/torch.py:65-68 + the x.t() of keynet/layer.py:92: [n, d] images -> [d+1, n] feature-major
//   linear_to_affine    keynet/torch.py:71-77: [d+1, n] -> [n, d], plus max |last row - 1| for the host-side ValueError
// Transposes go through a padded 64x65 LDS tile so both the global read and the global write are coalesced.
#include "kn_internal.h"

namespace kn {

__global__ __launch_bounds__(256) void relu_kernel(float* __restrict__ y, int64_t rows, int64_t ld, int64_t n_vecs) {
    const int64_t total = rows * n_vecs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_vecs;
        const int64_t c = i - r * n_vecs;
        float v = y[r * ld + c];
        y[r * ld + c] = (v < 0.0f) ? 0.0f : v;
    }
}

__global__ __launch_bounds__(256) void relu_kernel_v4(float4* __restrict__ y, int64_t total4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = y[i];
        v.x = (v.x < 0.0f) ? 0.0f : v.x;
        v.y = (v.y < 0.0f) ? 0.0f : v.y;
        v.z = (v.z < 0.0f) ? 0.0f : v.z;
        v.w = (v.w < 0.0f) ? 0.0f : v.w;
        y[i] = v;
    }
}

int relu_inplace(float* y, int64_t rows, int64_t ld, int64_t n_vecs, hipStream_t s) {
    if (rows <= 0 || n_vecs <= 0) return KN_OK;
    if (ld == n_vecs && (rows * n_vecs) % 4 == 0 && ((uintptr_t)y) % 16 == 0) {
        const int64_t total4 = rows * n_vecs / 4;
        const int64_t grid = std::min<int64_t>((total4 + 255) / 256, 2048);
        hipLaunchKernelGGL(relu_kernel_v4, dim3((unsigned)grid), dim3(256), 0, s, reinterpret_cast<float4*>(y), total4);
    } else {
        const int64_t grid = std::min<int64_t>((rows * n_vecs + 255) / 256, 2048);
        hipLaunchKernelGGL(relu_kernel, dim3((unsigned)grid), dim3(256), 0, s, y, rows, ld, n_vecs);
    }
    KN_HIP(hipGetLastError());
    return KN_OK;
}

// in [R, C] row-major (ld_in) -> out [C, R] row-major (ld_out); 64x64 tiles, 256 threads (64 x 4)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int64_t R, int64_t C, int64_t ld_in, float* __restrict__ out,
                                                        int64_t ld_out) {
    __shared__ float tile[64][65];
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63;
    const int ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4) {
        const int64_t r = r0 + k, c = c0 + tx;
        if (r < R && c < C) tile[k][tx] = in[r * ld_in + c];
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        const int64_t c = c0 + k, r = r0 + tx;
        if (r < R && c < C) out[c * ld_out + r] = tile[tx][k];
    }
}

__global__ __launch_bounds__(256) void fill_row_kernel(float* __restrict__ row, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) row[i] = v;
}

// y[o, b] = (sum_s z[o*S + s, b], s ascending) + lastcol[o] * xlast[b]   (+ReLU);  y[outs, b] = lastcol[outs] * xlast[b]
__global__ __launch_bounds__(256) void dense_reduce_kernel(const float* __restrict__ z, int64_t ldz, int64_t outs, int splits, const float* __restrict__ lastcol,
                                                           const float* __restrict__ xlast, float* __restrict__ y, int64_t ldy, int64_t n_vecs, int relu) {
    const int64_t total = (outs + 1) * n_vecs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = i / n_vecs;
        const int64_t b = i - o * n_vecs;
        float acc = 0.0f;
        if (o < outs) {
            const float* zp = z + (o * splits) * ldz + b;
            for (int s = 0; s < splits; s++) acc = acc + zp[(int64_t)s * ldz];
        }
        const float bp = lastcol[o] * xlast[b];
        acc = acc + bp;
        if (relu) acc = (acc < 0.0f) ? 0.0f : acc;
        y[o * ldy + b] = acc;
    }
}

int dense_reduce(const float* z, int64_t ldz, int64_t outs, int64_t splits, const float* lastcol, const float* xlast, float* y, int64_t ldy, int64_t n_vecs,
                 int relu, hipStream_t s) {
    const int64_t total = (outs + 1) * n_vecs;
    const int64_t grid = std::min<int64_t>((total + 255) / 256, 4096);
    KN_LAUNCH("dense_reduce_kernel", dense_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, s, z, ldz, outs, (int)splits, lastcol, xlast, y, ldy, n_vecs, relu);
    KN_HIP(hipGetLastError());
    return KN_OK;
}

int affine_to_linear(const float* x, int64_t n, int64_t d, float* out, int64_t ldo, hipStream_t s) {
    if (n <= 0) return KN_OK;
    if (d > 0) {
        dim3 grid((unsigned)((d + 63) / 64), (unsigned)((n + 63) / 64));
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, x, n, d, d, out, ldo);
    }
    hipLaunchKernelGGL(fill_row_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, s, out + d * ldo, n, 1.0f);
    KN_HIP(hipGetLastError());
    return KN_OK;
}

// max over b of |y[d, b] - 1| ; single workgroup (n is a batch size)
__global__ __launch_bounds__(256) void lastrow_dev_kernel(const float* __restrict__ row, int64_t n, float* __restrict__ maxdev) {
    __shared__ float red[256];
    float m = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        float dv = fabsf(row[i] - 1.0f);
        m = (dv > m || dv != dv) ? dv : m;   // NaN wins so the host check fires
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) {
            float o = red[threadIdx.x + sft];
            float a = red[threadIdx.x];
            red[threadIdx.x] = (o > a || o != o) ? o : a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *maxdev = red[0];
}

// max |y| over [rows, n_vecs] (leading dimension ld) raised into *absmax (kn_absmax; kn_spmm_screen after kernels that do not fold it into
// their stores).  NaN entries are ignored (max of the finite and infinite ones); grid-stride, 16 bytes per lane when the block is dense.
__global__ __launch_bounds__(256) void absmax_kernel_v4(const float4* __restrict__ y, int64_t total4, float* __restrict__ absmax) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = y[i];
        m = fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y));
        m = fmaxf(fmaxf(m, fabsf(v.z)), fabsf(v.w));
    }
    kn_wave_absmax_commit(m, absmax, threadIdx.x & 63);
}

// a column window of a wider block (ld > n_vecs: the half-batch windows of the overlapped forward), 16 bytes per lane: `tpr` threads walk one
// row's n_vecs / 4 quads, 256 / tpr rows per workgroup step -- one 32-bit division per thread, none per element
__global__ __launch_bounds__(256) void absmax_kernel_win4(const float* __restrict__ y, int64_t rows, int64_t ld, int n4, int tpr, float* __restrict__ absmax) {
    const int rpb = 256 / tpr;
    const int tr = (int)threadIdx.x / tpr, tc = (int)threadIdx.x - tr * tpr;
    float m = 0.0f;
    if (tr < rpb) {
        for (int64_t r = (int64_t)blockIdx.x * rpb + tr; r < rows; r += (int64_t)gridDim.x * rpb) {
            const float4* row = reinterpret_cast<const float4*>(y + r * ld);
            for (int c = tc; c < n4; c += tpr) {
                const float4 v = row[c];
                m = fmaxf(fmaxf(m, fabsf(v.x)), fabsf(v.y));
                m = fmaxf(fmaxf(m, fabsf(v.z)), fabsf(v.w));
            }
        }
    }
    kn_wave_absmax_commit(m, absmax, threadIdx.x & 63);
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ y, int64_t rows, int64_t ld, int64_t n_vecs, float* __restrict__ absmax) {
    float m = 0.0f;
    const int64_t total = rows * n_vecs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n_vecs;
        m = fmaxf(m, fabsf(y[r * ld + (i - r * n_vecs)]));
    }
    kn_wave_absmax_commit(m, absmax, threadIdx.x & 63);
}

int absmax_pass(const float* y, int64_t rows, int64_t ld, int64_t n_vecs, float* absmax, hipStream_t s) {
    if (rows <= 0 || n_vecs <= 0 || absmax == nullptr) return KN_OK;
    const bool a16 = ((uintptr_t)y) % 16 == 0;
    if (ld == n_vecs && (rows * n_vecs) % 4 == 0 && a16) {
        const int64_t total4 = rows * n_vecs / 4;
        KN_LAUNCH("absmax_kernel_v4", absmax_kernel_v4, dim3((unsigned)std::min<int64_t>((total4 + 255) / 256, 4096)), dim3(256), 0, s, reinterpret_cast<const float4*>(y), total4, absmax);
    } else if (n_vecs % 4 == 0 && ld % 4 == 0 && a16) {
        const int n4 = (int)(n_vecs / 4);
        int tpr = 1;
        while (tpr < n4 && tpr < 256) tpr <<= 1;                  // threads per row: a power of two >= the row's quads (<= 256)
        const int rpb = 256 / tpr;
        KN_LAUNCH("absmax_kernel_win4", absmax_kernel_win4, dim3((unsigned)std::min<int64_t>((rows + rpb - 1) / rpb, 8192)), dim3(256), 0, s, y, rows, ld, n4, tpr, absmax);
    } else {
        KN_LAUNCH("absmax_kernel", absmax_kernel, dim3((unsigned)std::min<int64_t>((rows * n_vecs + 255) / 256, 4096)), dim3(256), 0, s, y, rows, ld, n_vecs, absmax);
    }
    KN_HIP(hipGetLastError());
    return KN_OK;
}

int linear_to_affine(const float* y, int64_t ldy, int64_t n, int64_t d, float* out, float* maxdev, hipStream_t s) {
    if (n <= 0) return KN_OK;
    if (d > 0) {
        dim3 grid((unsigned)((n + 63) / 64), (unsigned)((d + 63) / 64));
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, y, d, n, ldy, out, d);
    }
    if (maxdev) hipLaunchKernelGGL(lastrow_dev_kernel, dim3(1), dim3(256), 0, s, y + d * ldy, n, maxdev);
    KN_HIP(hipGetLastError());
    return KN_OK;
}

}  // namespace kn
