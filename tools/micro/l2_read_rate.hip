// What can all 256 CUs together read from an L2-RESIDENT array?  The LeNet whole-net kernel (csrc/kn_chain.hip) streams its 2.6 MB of operators from L2 once per workgroup:
// 256 workgroups x 2.6 MB = ~0.7 GB per forward (SQ_INSTS_VMEM_RD x 1 KiB, profiles/r05_lenet_b1024_pmc.csv).  A formulation with twice the workgroups per forward (two co-resident
// workgroups of 2 batch columns per CU) reads twice that; this program measures the ceiling it would run into: every workgroup of 1024 threads walks the SAME array with 16-byte loads
// (each wavefront a contiguous 1 KiB per instruction, as the kernel's value quads), `reps` passes, wall time by events.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/l2_read_rate.hip -o /tmp/l2_read_rate && /tmp/l2_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(1024) void k(const f32x4* __restrict__ a, size_t n4, int reps, float* out, int lockstep) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; r++) {
        // staggered: every workgroup starts somewhere else, so the 32 CUs of an XCD do not all ask for the same line at once;
        // lockstep: all workgroups walk the array from its start together (what workgroups that stream one operator in the same order do until they drift apart)
        size_t base = lockstep ? 0 : ((size_t)blockIdx.x * 4099 + (size_t)r * 977) * 1024 % n4;
        for (size_t i = 0; i < n4; i += (size_t)1024 * UNROLL) {
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                size_t j = base + i + (size_t)u * 1024 + threadIdx.x;
                if (j >= n4) j -= n4;
                v[u] = a[j];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc += v[u];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;      // (never true: keeps the loads)
}

int main() {
    for (size_t mb10 : {26, 13, 52, 160}) {                              // 2.6 MB = LeNet's operators; 1.3 / 5.2 / 16 MB for scale (4 MB of L2 per XCD, 256 MB Infinity Cache)
        const size_t bytes = mb10 * 100 * 1024, n4 = bytes / 16 / 1024 * 1024;
        f32x4* a;
        float* out;
        hipMalloc(&a, n4 * 16);
        hipMalloc(&out, 16);
        hipMemset(a, 0, n4 * 16);
        for (int lockstep : {0, 1})
        for (int wgs : {256, 512}) {
            const int reps = lockstep ? 1 : 40;                          // (lockstep: ONE pass per launch -- behind the first pass the workgroups have drifted apart; averaged over 40 launches)
            const int launches = lockstep ? 40 : 1;
            hipLaunchKernelGGL(k<8>, dim3(wgs), dim3(1024), 0, 0, a, n4, 4, out, lockstep);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            for (int l = 0; l < launches; l++) hipLaunchKernelGGL(k<8>, dim3(wgs), dim3(1024), 0, 0, a, n4, reps, out, lockstep);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double total = (double)n4 * 16 * reps * launches * wgs;
            printf("array %5.1f MB | %3d workgroups x 1024 threads | %-9s | %6.2f TB/s from L2 (%.1f GB in %.3f ms) | one pass of all workgroups: %.2f us%s\n", n4 * 16 / 1e6, wgs, lockstep ? "lockstep" : "staggered",
                   total / ms / 1e9, total / 1e9, ms, 1e3 * ms / (reps * launches), lockstep ? " (incl. ~2 us of launch per pass)" : "");
        }
        hipFree(a);
        hipFree(out);
    }
    return 0;
}
