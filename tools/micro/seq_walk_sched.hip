// What does ONE wavefront per SIMD pay for a quad of the whole-net kernel's sequential thin walk (csrc/kn_chain.hip: chain_rows_thin_seq) -- four dependent packed adds, four packed
// multiplies, two LDS reads, one value request -- depending on how the instructions are ordered?  Each MODE is one iteration body written in inline assembly (the compiler keeps
// the order); time = s_memtime ticks per iteration of wavefront 0, workgroups of 1 / 4 / 8 wavefronts (4 = one per SIMD as in the kernel, 8 = two per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/seq_walk_sched.hip -o /tmp/seq_walk_sched && /tmp/seq_walk_sched
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ADD(P) "v_pk_add_f32 %0, %0, " P "\n\t"
#define MUL(P, X) "v_pk_mul_f32 " P ", %5, " X " op_sel_hi:[0,1]\n\t"
#define OPS : "+v"(acc), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(vp), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(q0), "v"(q1)

template <int MODE>
__global__ void k(float* out, long long* cyc, int n, const f32x4* g) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)(i % 7) * 0.25f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x2 acc = {out[lane], 1.f}, p0 = {1.f, 2.f}, p1 = {0.5f, 0.25f}, p2 = {3.f, 1.f}, p3 = {0.125f, 2.f}, vp = {1.0001f, 0.9999f};
    f32x2 x0 = {1.f, 1.f}, x1 = x0, x2 = x0, x3 = x0, q0 = {2.f, 2.f}, q1 = q0;
    float a0 = out[lane], a1 = 1.f;            // two unpacked chains
    f32x4 v[6];
    const f32x4* gp = g + lane;
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(g), 0, -1, 0x00020000);
    for (int i = 0; i < 6; i++) v[i] = gp[64 * i];
    f32x2 xq[3][4];
    uint32_t la = 0;
    auto xr = [&](f32x2 (&x)[4], uint32_t a) {
        for (int e = 0; e < 4; e++) x[e] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(lds) + a + 16u * e);
    };
    xr(xq[0], 0);
    xr(xq[1], 64);
    xr(xq[2], 128);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; it += (MODE >= 6 ? 6 : 1)) {
        if (MODE == 0) {              // [add, mul] x 4
            asm volatile(ADD("%1") MUL("%1", "%6") ADD("%2") MUL("%2", "%7") ADD("%3") MUL("%3", "%8") ADD("%4") MUL("%4", "%9") OPS);
        } else if (MODE == 1) {       // [add, s_nop] x 4: the bare chain
            asm volatile(ADD("%1") "s_nop 0\n\t" ADD("%2") "s_nop 0\n\t" ADD("%3") "s_nop 0\n\t" ADD("%4") "s_nop 0\n\t" OPS);
        } else if (MODE == 2) {       // [add, mul, mul] x 4
            asm volatile(ADD("%1") MUL("%1", "%6") MUL("%10", "%6") ADD("%2") MUL("%2", "%7") MUL("%11", "%7") ADD("%3") MUL("%3", "%8") MUL("%10", "%8") ADD("%4") MUL("%4", "%9") MUL("%11", "%9") OPS);
        } else if (MODE == 3) {       // [add, mul, mul, mul] x 4
            asm volatile(ADD("%1") MUL("%1", "%6") MUL("%10", "%6") MUL("%11", "%6") ADD("%2") MUL("%2", "%7") MUL("%10", "%7") MUL("%11", "%7") ADD("%3") MUL("%3", "%8") MUL("%10", "%8") MUL("%11", "%8")
                         ADD("%4") MUL("%4", "%9") MUL("%10", "%9") MUL("%11", "%9") OPS);
        } else if (MODE == 4) {       // unpacked adds, two chains, packed multiplies: [add_a, add_b, pk_mul] x 4
            asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_mul_f32 %2, %6, %7\n\tv_mul_f32 %3, %6, %8\n\t"
                         "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\tv_mul_f32 %4, %6, %7\n\tv_mul_f32 %5, %6, %8\n\t"
                         "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_mul_f32 %2, %6, %7\n\tv_mul_f32 %3, %6, %8\n\t"
                         "v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\tv_mul_f32 %4, %6, %7\n\tv_mul_f32 %5, %6, %8\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(p0.x), "+v"(p0.y), "+v"(p1.x), "+v"(p1.y) : "v"(vp.x), "v"(x0.x), "v"(x1.x));
        } else if (MODE == 5) {       // [add_a, add_b] x 4 only
            asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"
                         "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(p0.x), "+v"(p0.y), "+v"(p1.x), "+v"(p1.y) : "v"(vp.x), "v"(x0.x), "v"(x1.x));
        } else if (MODE >= 6) {
            // the kernel's quad, six per iteration (ring of six value quads, activations two quads ahead): counted as six iterations below
#pragma unroll
            for (int i = 0; i < 6; i++) {
                f32x2 (&xc)[4] = xq[i % 3];
                f32x2 (&xn)[4] = xq[(i + 2) % 3];
                la = (la + 64u) & 16383u;
                const f32x2 vlo = {v[i].x, v[i].y};
                auto RD = [&]() { xr(xn, la); __builtin_amdgcn_sched_barrier(0); };
                auto RD1 = [&](int e0) {
                    xn[e0] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(lds) + la + 16u * e0);
                    xn[e0 + 1] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(lds) + la + 16u * e0 + 16u);
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto RQ = [&](int slot) {
                    v[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res, 16u * (uint32_t)lane, 1024u * (uint32_t)((it + i) & 63), 0));
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto A4 = [&]() {
                    asm volatile(ADD("%1") MUL("%1", "%6") ADD("%2") MUL("%2", "%7") ADD("%3") MUL("%3", "%8") ADD("%4") MUL("%4", "%9")
                                 : "+v"(acc), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(vlo), "v"(xc[0]), "v"(xc[1]), "v"(xc[2]), "v"(xc[3]));
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto A2a = [&]() { asm volatile(ADD("%1") MUL("%1", "%3") ADD("%2") MUL("%2", "%4") : "+v"(acc), "+v"(p0), "+v"(p1) : "v"(xc[0]), "v"(xc[1]), "v"(vlo)); __builtin_amdgcn_sched_barrier(0); };
                auto A2b = [&]() { asm volatile(ADD("%1") MUL("%1", "%3") ADD("%2") MUL("%2", "%4") : "+v"(acc), "+v"(p2), "+v"(p3) : "v"(xc[2]), "v"(xc[3]), "v"(vlo)); __builtin_amdgcn_sched_barrier(0); };
                auto A1 = [&](f32x2& p, const f32x2& x) { asm volatile(ADD("%1") MUL("%1", "%2") : "+v"(acc), "+v"(p) : "v"(x), "v"(vlo), "v"(vlo), "v"(vlo)); __builtin_amdgcn_sched_barrier(0); };
                const int prev = (i + 5) % 6;                       // the slot the previous quad consumed
                if (MODE == 6) { RD(); A4(); RQ(i); }               // grouped: reads, arithmetic, request (what ships)
                else if (MODE == 7) { A1(p0, xc[0]); RD1(0); A1(p1, xc[1]); RD1(2); A2b(); RQ(i); }     // reads dealt into the gaps
                else if (MODE == 8) { A4(); RQ(i); }                // no LDS reads
                else if (MODE == 9) { RD(); A4(); }                 // no request
                else if (MODE == 10) { RQ(prev); RD(); A4(); }      // the previous quad's slot requested first
                else if (MODE == 11) { RD(); A2a(); RQ(prev); A2b(); }                                 // request in the middle
                else if (MODE == 12) { RD(); RQ(prev); A4(); }      // reads, request, arithmetic
                else if (MODE == 13) { A4(); RD(); RQ(i); }         // arithmetic first, then all three memory instructions
                else if (MODE == 14) { A2a(); RD(); A2b(); RQ(i); }
                else if (MODE == 15) { RD(); asm volatile("s_waitcnt lgkmcnt(4)"); __builtin_amdgcn_sched_barrier(0); A4(); RQ(i); }                // the LDS wait on its own
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float vs = 0.f;
    for (int i = 0; i < 6; i++) vs += v[i].x + v[i].y + v[i].z + v[i].w;
    for (int e = 0; e < 4; e++) vs += xq[0][e].x + xq[0][e].y + xq[1][e].x + xq[1][e].y + xq[2][e].x + xq[2][e].y;
    out[threadIdx.x] = acc.x + acc.y + p0.x + p1.x + p2.x + p3.x + a0 + a1 + vs + q0.x + q1.x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out;
    f32x4* g;
    long long* cyc;
    hipMalloc(&out, 4096);
    hipMalloc(&g, 16 * 64 * 128);
    hipMalloc(&cyc, 8);
    hipMemset(g, 0, 16 * 64 * 128);
    hipMemset(out, 0, 4096);
    const int n = 8190;
    const char* names[] = {"[pk_add, pk_mul] x 4", "[pk_add, s_nop] x 4 (bare chain)", "[pk_add, pk_mul, pk_mul] x 4", "[pk_add, pk_mul x 3] x 4", "[v_add a, v_add b, v_mul a, v_mul b] x 4 (unpacked)",
                           "[v_add a, v_add b] x 4", "quad: 2 LDS reads | [add, mul] x 4 | request (grouped)", "quad: reads dealt into the gaps", "quad without the LDS reads", "quad without the request", "quad: request (previous slot) | reads | arithmetic", "quad: reads | half | request | half",
                           "quad: reads | request | arithmetic", "quad: arithmetic | reads | request", "quad: half | reads | half | request", "quad: grouped, LDS wait separate"};
    for (int waves : {1, 4}) {
        for (int m = 0; m < 16; m++) {
            for (int rep = 0; rep < 2; rep++) {
#define L(M) if (m == M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n, g);
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15)
                hipDeviceSynchronize();
            }
            long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d wave(s) in the workgroup | %-58s | %6.1f s_memtime ticks per iteration\n", waves, names[m], (double)c / n);
        }
    }
    return 0;
}
