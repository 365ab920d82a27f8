// What does a SIMD's vector ALU issue per cycle under sustained load, and at what clock?  Blocks of 64 independent instructions (no operand shared between
// neighbours), 256 workgroups x 1024 threads (four wavefronts per SIMD), wall time by events; wavefront 0 of workgroup 0 also reads s_memtime (shader clock
// counter) and s_memrealtime (constant 100 MHz) around its loop: their ratio is the clock the loop actually ran at.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/valu_issue_rate.hip -o /tmp/valu_issue_rate && /tmp/valu_issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"
// four independent accumulators per group; operands in different bank pairs
#define PKADD "v_pk_add_f32 v[32:33], v[2:3], v[32:33]\n v_pk_add_f32 v[36:37], v[6:7], v[36:37]\n v_pk_add_f32 v[40:41], v[10:11], v[40:41]\n v_pk_add_f32 v[44:45], v[14:15], v[44:45]\n"
#define PKMUL "v_pk_mul_f32 v[32:33], v[2:3], v[18:19]\n v_pk_mul_f32 v[36:37], v[6:7], v[22:23]\n v_pk_mul_f32 v[40:41], v[10:11], v[26:27]\n v_pk_mul_f32 v[44:45], v[14:15], v[30:31]\n"
#define PKFMA "v_pk_fma_f32 v[32:33], v[2:3], v[18:19], v[32:33]\n v_pk_fma_f32 v[36:37], v[6:7], v[22:23], v[36:37]\n v_pk_fma_f32 v[40:41], v[10:11], v[26:27], v[40:41]\n v_pk_fma_f32 v[44:45], v[14:15], v[30:31], v[44:45]\n"
#define ADD32 "v_add_f32 v32, v2, v32\n v_add_f32 v36, v6, v36\n v_add_f32 v40, v10, v40\n v_add_f32 v44, v14, v44\n"
#define FMA32 "v_fma_f32 v32, v2, v18, v32\n v_fma_f32 v36, v6, v22, v36\n v_fma_f32 v40, v10, v26, v40\n v_fma_f32 v44, v14, v30, v44\n"
#define MOV32 "v_mov_b32 v32, v2\n v_mov_b32 v36, v6\n v_mov_b32 v40, v10\n v_mov_b32 v44, v14\n"

template <int V>
__global__ __launch_bounds__(1024) void k(int iters, unsigned long long* out) {
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(R16(PKADD) ::: CLOB);
        if (V == 1) asm volatile(R16(PKMUL) ::: CLOB);
        if (V == 2) asm volatile(R16(PKFMA) ::: CLOB);
        if (V == 3) asm volatile(R16(ADD32) ::: CLOB);
        if (V == 4) asm volatile(R16(FMA32) ::: CLOB);
        if (V == 5) asm volatile(R16(MOV32) ::: CLOB);
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
}

int main() {
    const int iters = 20000;
    const char* names[6] = {"v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_add_f32", "v_fma_f32", "v_mov_b32"};
    unsigned long long* d;
    hipMalloc(&d, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 6; v++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, iters, d);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, iters, d);
            if (v == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, iters, d);
            if (v == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, iters, d);
            if (v == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), 0, 0, iters, d);
            if (v == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(1024), 0, 0, iters, d);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long h[2];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const double n_inst = (double)iters * 64.0 * 4.0;             // per SIMD: 64 instructions per iteration x 4 wavefronts
        const double real_s = (double)h[1] / 100e6;
        printf("%-14s wall %.3f ms | wavefront 0: s_memtime delta %llu, s_memrealtime delta %llu (%.3f ms) -> s_memtime rate %.3f GHz | %.2f ns per instruction and SIMD = %.2f cycles @2.4 GHz\n",
               names[v], ms, h[0], h[1], real_s * 1e3, (double)h[0] / real_s / 1e9, (double)ms * 1e6 / n_inst, (double)ms * 1e6 / n_inst * 2.4);
    }
    return 0;
}
