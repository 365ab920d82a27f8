// Micro-benchmark: what one wavefront alone on a SIMD pays per instruction of the whole-net kernel's thin-layer walk.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dep_chain tools/micro/dep_chain.hip && /tmp/dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#pragma clang fp contract(off)
template <int MODE>
__global__ void k(float* out, long long* cyc, int n, const float* g) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)(i % 7) * 0.25f;
    __syncthreads();
    float acc = out[threadIdx.x];
    f32x2 a2 = {acc, acc + 1.f}, b2 = {acc * 2.f, acc * 3.f};
    float p0 = g[0], p1 = g[1], p2 = g[2], p3 = g[3];
    int idx = threadIdx.x & 15;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        if (MODE == 0) {                      // 4 dependent f32 adds
            acc = acc + p0; acc = acc + p1; acc = acc + p2; acc = acc + p3;
        } else if (MODE == 1) {               // 4 x (mul, dependent add)
            float m0 = p0 * acc, m1 = p1 * p2, m2 = p2 * p3, m3 = p3 * p0;
            acc = acc + m1; acc = acc + m2; acc = acc + m3; acc = acc + m0;
        } else if (MODE == 2) {               // 4 dependent packed adds on two accumulators
            a2 = a2 + f32x2{p0, p1}; b2 = b2 + f32x2{p2, p3}; a2 = a2 + f32x2{p1, p2}; b2 = b2 + f32x2{p3, p0};
            a2 = a2 + f32x2{p2, p3}; b2 = b2 + f32x2{p0, p1}; a2 = a2 + f32x2{p3, p0}; b2 = b2 + f32x2{p1, p2};
        } else if (MODE == 3) {               // LDS pointer chase: read -> address of next read (latency)
            idx = (int)lds[idx & 4095] + (idx & 15);
        } else if (MODE == 4) {               // 4 independent ds_read_b128 + dependent math on them (one quad of the walk, all in one iteration)
            f32x4 x0 = *(f32x4*)&lds[(idx * 4) & 4092], x1 = *(f32x4*)&lds[(idx * 4 + 64) & 4092], x2 = *(f32x4*)&lds[(idx * 4 + 128) & 4092], x3 = *(f32x4*)&lds[(idx * 4 + 192) & 4092];
            a2 = a2 + f32x2{x0.x, x0.y} * p0; b2 = b2 + f32x2{x0.z, x0.w} * p0;
            a2 = a2 + f32x2{x1.x, x1.y} * p1; b2 = b2 + f32x2{x1.z, x1.w} * p1;
            a2 = a2 + f32x2{x2.x, x2.y} * p2; b2 = b2 + f32x2{x2.z, x2.w} * p2;
            a2 = a2 + f32x2{x3.x, x3.y} * p3; b2 = b2 + f32x2{x3.z, x3.w} * p3;
            idx += 1;
        } else if (MODE == 5) {               // global load (L2 hit) pointer chase
            idx = (int)g[idx & 1023] + (idx & 15);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc + a2.x + a2.y + b2.x + b2.y + (float)idx;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float *out, *g;
    long long* cyc;
    hipMalloc(&out, 4096);
    hipMalloc(&g, 4096);
    hipMalloc(&cyc, 8);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; i++) h[i] = (float)((i * 37) % 1000);
    hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
    hipMemset(out, 0, 4096);
    const int n = 4096;
    const char* names[] = {"4 dependent v_add_f32", "4 x (v_mul + dependent v_add)", "8 packed adds, two chains", "LDS pointer chase (latency)", "quad: 4 ds_read_b128 + 16 packed mul/add", "global pointer chase (L2 hit latency)"};
    for (int waves = 1; waves <= 4; waves *= 4) {
        for (int m = 0; m < 6; m++) {
            for (int rep = 0; rep < 2; rep++) {
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n, g);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n, g);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n, g);
                if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n, g);
                if (m == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n, g);
                if (m == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n, g);
                hipDeviceSynchronize();
            }
            long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d wave(s) in the workgroup | %-46s | %.1f s_memtime ticks per iteration\n", waves, names[m], (double)c / n);
        }
    }
    return 0;
}
