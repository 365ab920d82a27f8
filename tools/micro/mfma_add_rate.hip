// Issue rates behind kn_csr_mfma.hip, one workgroup per CU, W wavefronts per SIMD: cycles per matrix instruction for
//   mode 0: v_mfma_f32_32x32x1_2b_f32 alone, two result blocks alternating (independent instructions)
//   mode 1: the same + 16 v_pk_add_f32 of the PREVIOUS result block behind each (the kernel's inner loop without loads)
//   mode 2: 16 v_pk_add_f32 alone
//   mode 3: v_mfma_f32_32x32x2_f32 alone (the conv-taps kernel's instruction) for comparison
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/mfma_add_rate.hip -o /tmp/mfma_add_rate && /tmp/mfma_add_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, float a0, float x0) {
    f32x32 zero;
    for (int q = 0; q < 32; q++) zero[q] = 0.f;
    f32x2 acc[16];
    for (int q = 0; q < 16; q++) acc[q] = f32x2{0.f, 0.f};
    f32x32 d0 = zero, d1 = zero;
    f32x16 e0, e1;
    for (int q = 0; q < 16; q++) e0[q] = e1[q] = 0.f;
    float a = a0 + threadIdx.x, x = x0 - threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0 || MODE == 1) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x1f32(a, x, zero, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1)
                _Pragma("unroll") for (int q = 0; q < 16; q++) { const f32x2 p = {d1[2 * q], d1[2 * q + 1]}; asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[q]) : "v"(p)); }
            __builtin_amdgcn_sched_barrier(0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x1f32(x, a, zero, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1)
                _Pragma("unroll") for (int q = 0; q < 16; q++) { const f32x2 p = {d0[2 * q], d0[2 * q + 1]}; asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[q]) : "v"(p)); }
            __builtin_amdgcn_sched_barrier(0);
        } else if (MODE == 2) {
            _Pragma("unroll") for (int r = 0; r < 2; r++)
                _Pragma("unroll") for (int q = 0; q < 16; q++) { const f32x2 p = {a, x}; asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[q]) : "v"(p)); }
        } else {
            e0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, a, e1, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < 16; q++) s += acc[q].x + acc[q].y + e0[q] + e1[q];
    for (int q = 0; q < 32; q++) s += d0[q] + d1[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int waves_per_simd) {
    const int blocks = 256, threads = 256 * waves_per_simd, iters = 20000;
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 100, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c[4];
    hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz; use wall time and the nominal 2.4 GHz for a cycle estimate
    const double ns_per_pair = 1e6 * ms / iters;                     // per loop iteration = 2 matrix instructions (or 32 packed adds)
    printf("%-34s %d wave(s)/SIMD: %.1f ns per iteration per wave-slot -> %.1f cycles @2.4GHz per matrix instruction (or per 16 packed adds), per SIMD %.1f\n", name, waves_per_simd,
           ns_per_pair, ns_per_pair * 2.4 / 2, ns_per_pair * 2.4 / 2 / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 4; w++) {
        run<0>("mfma 32x32x1_2b alone", w);
        run<1>("mfma 32x32x1_2b + 16 pk_add", w);
        run<2>("16 pk_add alone", w);
        run<3>("mfma 32x32x2 alone (accumulating)", w);
    }
    return 0;
}
