// Does v_pk_add_f32 pay for VGPR bank conflicts between its two 64-bit source operands?  (bank = register index mod 4)
// Sixteen packed adds acc[q] += d[q] with hard-coded registers: d pairs at v[0:31], acc pairs at v[32 + S : ...] with S = 0 (acc pair q and d pair q start
// in the SAME bank pair: all sixteen conflict) or S = 2 (opposite bank pairs: none conflict), or alternating.  Cycles per block of 16 and SIMD at 1..4 wavefronts.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/pk_add_banks.hip -o /tmp/pk_add_banks && /tmp/pk_add_banks
#include <hip/hip_runtime.h>
#include <cstdio>

#define PK(a, d) "v_pk_add_f32 v[" #a ":" #a "+1], v[" #d ":" #d "+1], v[" #a ":" #a "+1]\n"
// all sixteen: acc base 32 (same bank pair as d) / 34 (opposite) / mixed (even q same, odd q opposite)
#define BLOCK_SAME PK(32, 0) PK(34, 2) PK(36, 4) PK(38, 6) PK(40, 8) PK(42, 10) PK(44, 12) PK(46, 14) PK(48, 16) PK(50, 18) PK(52, 20) PK(54, 22) PK(56, 24) PK(58, 26) PK(60, 28) PK(62, 30)
#define BLOCK_OPP PK(34, 0) PK(36, 2) PK(38, 4) PK(40, 6) PK(42, 8) PK(44, 10) PK(46, 12) PK(48, 14) PK(50, 16) PK(52, 18) PK(54, 20) PK(56, 22) PK(58, 24) PK(60, 26) PK(62, 28) PK(64, 30)
#define BLOCK_MIX PK(32, 0) PK(36, 2) PK(36, 4) PK(40, 6) PK(40, 8) PK(44, 10) PK(44, 12) PK(48, 14) PK(48, 16) PK(52, 18) PK(52, 20) PK(56, 22) PK(56, 24) PK(60, 26) PK(60, 28) PK(64, 30)
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65"

template <int V>
__global__ __launch_bounds__(1024) void k(int iters) {
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(BLOCK_SAME BLOCK_SAME BLOCK_SAME BLOCK_SAME ::: CLOB);
        if (V == 1) asm volatile(BLOCK_OPP BLOCK_OPP BLOCK_OPP BLOCK_OPP ::: CLOB);
        if (V == 2) asm volatile(BLOCK_MIX BLOCK_MIX BLOCK_MIX BLOCK_MIX ::: CLOB);
    }
}

int main() {
    const int iters = 20000;
    const char* names[3] = {"same bank pair (16 conflicts)", "opposite bank pairs (none)", "mixed (8 conflicts)"};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 4; waves++)
        for (int v = 0; v < 3; v++) {
            // one workgroup per CU with `waves` wavefronts on each of its 4 SIMDs; wall time by events (sustained: second launch timed)
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves), 0, 0, iters);
                if (v == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves), 0, 0, iters);
                if (v == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * waves), 0, 0, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double cyc = (double)ms * 1e-3 * 2.4e9 / ((double)iters * 4.0);
            printf("waves/SIMD %d  %-32s %.3f ms -> %.1f cycles @2.4GHz per block of 16 per wavefront, %.1f per SIMD\n", waves, names[v], ms, cyc, cyc / waves);
        }
    return 0;
}
