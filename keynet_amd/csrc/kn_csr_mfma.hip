// kn_csr_mfma.hip -- grouped rows of an order-preserving CSR product with the MULTIPLIES on the matrix pipe (gfx950).
//
// The reference's arithmetic (scipy csr_matvecs behind keynet.sparse.SparseMatrix.torchdot, keynet/sparse.py:488-492) is, per output element,
//     for jj in stored order:  y = fl(y + fl(a_jj * x_jj))          -- an f32 multiply rounded, then an f32 add rounded; never an FMA.
// kn_csr.hip keeps that on the vector ALU: v_pk_mul_f32 + v_pk_add_f32, i.e. two vector instructions per two MACs -- the "no-FMA roof" of
// 39.3 T MAC/s.  But a matrix instruction with K = 1 and a ZERO accumulator is exactly the rounded product:
//     v_mfma_f32_32x32x1_2b_f32  D = 0 + a (x) b      =>      D[i][j] = fl(a_i * b_j)
// (one product per output element, one rounding; verified bit for bit against v_mul_f32 on 33.5 M products incl. denormal operands and
// results, overflow, Inf and NaN by tools/micro/mfma_product.hip -- the only difference is the SIGN OF A ZERO product, +0 from the matrix pipe
// where v_mul_f32 gives -0, and a running sum that starts at +0.0 is never -0.0, so adding either zero leaves it unchanged bit for bit).
// So the 32 x 64 products of one stored column -- 32 member rows of a pattern group x 64 batch columns -- come out of ONE matrix instruction
// (64 cycles of the matrix pipe), and the vector ALU only adds them to the running sums, in stored order: 16 v_pk_add_f32 (64 cycles).
// MEASURED (tools/micro/mfma_add_rate.hip, profiles/r05_micro_*.txt): on gfx950 the f32 matrix instruction and the packed f32 adds do NOT
// overlap -- the f32 MFMA runs on the vector ALU's own FP32 lanes -- so the pair costs 145-155 cycles per stored column and wavefront and the
// no-FMA roof stays 39.3 T MAC/s.  What this formulation buys is traffic, not issue rate: an activation row is fetched once per 32-96 member
// rows instead of once per 16, and (TAPS) the values come from the 0.3 MB tap table instead of per-row copies.
//
// Tile: a 256-thread workgroup owns NRB <= 3 row blocks (32 member rows each) of ONE pattern group x 256 batch columns; wavefront w owns
// columns 64w .. 64w+63 for all NRB row blocks (96 x 64 running sums = 96 VGPRs).  Per stored column j a wavefront loads its activation
// row segment once (256 B, saddr-form dword load: the B operand, lane = batch column) and the NRB x 32 values of the column (128 B each:
// the A operand, lane & 31 = member row; the four wavefronts of the workgroup read the same values: L1 hits), issues NRB matrix
// instructions and adds each result block while the next one is computed.  Loads run PF = 6 columns ahead in a register ring (a gathered
// row of a big operator misses L2; 8 columns ahead for the three-block instantiation), counted vmcnt waits, the column index one more step ahead through a scalar load.  Compared with the
// 16-row x 256-column wavefront tiles of csr_group_pipe_kernel an activation row is fetched once per 96 member rows instead of once per 16.
#include "kn_internal.h"
#include <type_traits>
#include <cstdlib>

#pragma clang fp contract(off)

namespace kn {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

static __device__ __forceinline__ float mfma_relu_f(float v) { return (v < 0.0f) ? 0.0f : v; }  // torch relu: NaN stays NaN

template <int I, int N, class F>
static __device__ __forceinline__ void mfma_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        mfma_static_for<I + 1, N>(f);
    }
}

// TAPS: the same kernel on a FACTORED conv operator (kn_convtaps_drop_zero_entries: an untiled keyed conv CSR proven to be the ascending-column expansion
// of taps x slots): group = output pixel, members = output channels, stored column j of pixel o = ex_tab[ex_ptr[o] + j] = (activation row, index of the
// value row in tapsT) -- the A operand is 128 contiguous bytes of tapsT[tap][ci][co0 ..] instead of a slice of the CSR's per-row value copies (AllConvNet
// conv2: 0.33 MB of taps, L2-resident, instead of 326 MB streamed once per column tile); the bias column is added last in the epilogue, like the conv kernels.
struct MfTaps {
    const int32_t* ex_ptr;        // [HoWo + 1]
    const int32_t* ex_tab;        // [total][2] = (activation row, value row)
    const float* tapsT;           // [value rows][cout_pad]
    const float* lastcol;         // [Cout * HoWo + 1] or null
    const int32_t* pix_order;     // [HoWo] processing order of the pixels
    int64_t last_in_row;          // activation row of the homogeneous coordinate
    int32_t cout_pad, HoWo, Cout, n_cc;   // n_cc = channel chunks (32 * NRB channels) per pixel
};

// (TAPS, measured and dropped: the four wavefronts of a workgroup on four pixels of a 2 x 2 block x the same 64 batch columns, so that the block's 16 input
// pixels would be shared through the CU's vector cache -- HBM fetch of the AllConvNet forward 12.6 -> 19.5 GB and 2 % slower: the 256-byte row segments
// cut the reuse between workgroups in L2 by more than the vector cache gives back.)
template <int NRB, int PF, int NW = 4, bool TAPS = false>       // NW wavefronts per workgroup = 64 * NW batch columns per gathered value block (NW = 8, one workgroup per CU, halves the value
                                            // re-reads but measured 8.5 % slower on the AllConvNet forward: 35.9 against 33.0 ms; not instantiated)
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void csr_group_mfma_kernel(int64_t n_work, const int32_t* __restrict__ work_grp, const int32_t* __restrict__ work_r0,
                                                                const int32_t* __restrict__ grp_colptr, const int32_t* __restrict__ grp_cols,
                                                                const int32_t* __restrict__ grp_rowptr, const int32_t* __restrict__ grp_rows,
                                                                const int64_t* __restrict__ grp_valptr, const float* __restrict__ grp_vals,
                                                                const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, int64_t n_vecs, int relu,
                                                                const MfTaps tp = MfTaps()) {
    static_assert((PF * NRB) % 2 == 0, "two result blocks alternate: an even number of matrix instructions per unrolled loop body");   // PF = stored columns in flight per wavefront (ring of operand registers)
    constexpr int LPS = NRB + 1;                           // vector loads per stored column
    const int64_t n_ct = (n_vecs + 64 * NW - 1) / (64 * NW);
    // item -> (column tile, work item): XCD x = blockIdx & 7 owns the contiguous item range [x * chunk, (x + 1) * chunk), work item fastest
    const int64_t n_items = n_ct * n_work;
    const int64_t chunk = (n_items + 7) >> 3;
    const int64_t item = (int64_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (item >= n_items || (blockIdx.x >> 3) >= chunk) return;
    const int64_t ct = item / n_work;
    const int64_t w = item - ct * n_work;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int g, r0, cbeg, ncol, rbeg, nmem, rpad;
    if constexpr (TAPS) {
        g = __builtin_amdgcn_readfirstlane(tp.pix_order[w / tp.n_cc]);                   // the output pixel
        r0 = __builtin_amdgcn_readfirstlane((int)(w % tp.n_cc) * (32 * NRB));            // first output channel of this chunk
        cbeg = __builtin_amdgcn_readfirstlane(tp.ex_ptr[g]);
        ncol = __builtin_amdgcn_readfirstlane(tp.ex_ptr[g + 1]) - cbeg;
        rbeg = 0;
        nmem = tp.Cout;
        rpad = tp.cout_pad;
    } else {
        g = __builtin_amdgcn_readfirstlane(work_grp[w]);
        r0 = __builtin_amdgcn_readfirstlane(work_r0[w]);
        cbeg = __builtin_amdgcn_readfirstlane(grp_colptr[g]);
        ncol = __builtin_amdgcn_readfirstlane(grp_colptr[g + 1]) - cbeg;
        rbeg = __builtin_amdgcn_readfirstlane(grp_rowptr[g]);
        nmem = __builtin_amdgcn_readfirstlane(grp_rowptr[g + 1]) - rbeg;
        rpad = (nmem + 15) / 16 * 16;                      // kn_csr.hip: values of one stored column = rpad floats (members padded to bundles of 16)
    }
    const int64_t c0 = ct * (64 * NW) + (int64_t)wave * 64;
    if (c0 >= n_vecs) return;                              // (wave-uniform)
    const int64_t c = c0 + lane;
    const bool active = c < n_vecs;

    // running sums as independent register PAIRS (element e of a row block's 32 = pair e / 2, half e % 2): a 32-register tuple per row block
    // would have to be copied whenever the allocator cannot update it in place
    f32x2 acc[NRB][16];
#pragma unroll
    for (int b = 0; b < NRB; b++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[b][q] = f32x2{0.0f, 0.0f};

    if (ncol > 0) {
        auto uni = [](const uint64_t v) {
            return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
        };
        const uint64_t cbase = TAPS ? uni(reinterpret_cast<uint64_t>(tp.ex_tab + 2 * (int64_t)cbeg)) : uni(reinterpret_cast<uint64_t>(grp_cols + cbeg));
        const uint64_t xbase = uni(reinterpret_cast<uint64_t>(X));
        const uint64_t vbase = TAPS ? uni(reinterpret_cast<uint64_t>(tp.tapsT + r0)) : uni(reinterpret_cast<uint64_t>(grp_vals + grp_valptr[g] + r0));
        const uint32_t b_off = 4u * (uint32_t)(active ? c : c0);                 // lane's byte offset inside an activation row (inactive lanes: a valid address, result unused)
        const uint32_t a_off = 4u * (uint32_t)(lane & 31);                       // lane's byte offset inside a row block's 32 values
        const uint64_t ldx_b = 4ull * (uint64_t)ldx;
        const uint32_t vstep = 4u * (uint32_t)rpad;                              // (TAPS: rpad = cout_pad, one value row of tapsT)
        auto clampj = [&](const int j) { return j < ncol ? j : ncol - 1; };      // past the end: the last column again (loaded, never used)
        float xa[PF][3], xb[PF];                  // (row blocks beyond NRB: never loaded; their registers only appear in the waits' operand lists)
#pragma unroll
        for (int q = 0; q < PF; q++) xa[q][0] = xa[q][1] = xa[q][2] = xb[q] = 0.0f;
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        int col_nxt = 0;
        i32x2 cv_nxt = {0, 0};                                                  // TAPS: (activation row, value row) of the column whose index is on its way
        auto fetch_col = [&](const int j) {                                     // scalar: index of stored column j (TAPS: and of its value row)
            if constexpr (TAPS) {
                const uint64_t caddr = cbase + 8ull * (uint64_t)(uint32_t)clampj(j);
                asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(cv_nxt) : "s"(caddr));
            } else {
                const uint64_t caddr = cbase + 4ull * (uint64_t)(uint32_t)clampj(j);
                asm volatile("s_load_dword %0, %1, 0x0" : "=&s"(col_nxt) : "s"(caddr));
            }
        };
        auto col_landed = [&]() {
            if constexpr (TAPS) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(cv_nxt));
                col_nxt = cv_nxt.x;
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(col_nxt));
            }
        };
        // (the operand registers are passed by reference: clang refuses a captured array element as an asm operand inside a generic lambda)
        auto fetch = [&](float& rb, float& ra0, float& ra1, float& ra2, const int vrow, const int col) {   // the LPS vector loads of one stored column (value row `vrow`) into one ring slot
            const uint64_t xaddr = xbase + (uint64_t)(uint32_t)col * ldx_b;
            asm volatile("global_load_dword %0, %1, %2" : "=&v"(rb) : "v"(b_off), "s"(xaddr));
            const uint64_t va = vbase + (uint64_t)(uint32_t)vrow * (uint64_t)vstep;
            asm volatile("global_load_dword %0, %1, %2" : "=&v"(ra0) : "v"(a_off), "s"(va));
            if (NRB > 1) asm volatile("global_load_dword %0, %1, %2 offset:128" : "=&v"(ra1) : "v"(a_off), "s"(va));
            if (NRB > 2) asm volatile("global_load_dword %0, %1, %2 offset:256" : "=&v"(ra2) : "v"(a_off), "s"(va));
        };
        auto landed = [&](float& rb, float& ra0, float& ra1, float& ra2) {      // the oldest column in flight has landed (PF - 1 younger ones stay in flight)
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rb), "+v"(ra0), "+v"(ra1), "+v"(ra2) : "n"(LPS * (PF - 1)));
        };
        // prologue: columns 0 .. PF-1 in flight, the index of column PF on its way
        fetch_col(0);
        mfma_static_for<0, PF>([&](auto I_) {
            constexpr int I = decltype(I_)::value;
            col_landed();
            const int cj = col_nxt;
            const int vj = TAPS ? (int)cv_nxt.y : clampj(I);
            fetch_col(I + 1);
            fetch(xb[I], xa[I][0], xa[I][1], xa[I][2], vj, cj);
        });
        f32x32 zero;
#pragma unroll
        for (int q = 0; q < 32; q++) zero[q] = 0.0f;
        // running sums: acc += d, 16 packed adds (the products of a 32 x 64 block, one register pair per instruction)
        auto add_into = [&](f32x2 (&a)[16], const f32x32& d) {
#pragma unroll
            for (int q = 0; q < 16; q++) {                                     // (asm: left alone, the compiler splits about half of the pairs into two v_add_f32)
                const f32x2 p2 = {d[2 * q], d[2 * q + 1]};
                asm("v_pk_add_f32 %0, %1, %0" : "+v"(a[q]) : "v"(p2));
            }
        };
        auto add_into_c = [&](f32x2 (&a)[16], const f32x32& d) {              // the same sums with the dependency visible to the compiler (contraction is off: an add is an add)
#pragma unroll
            for (int q = 0; q < 16; q++) a[q] = a[q] + f32x2{d[2 * q], d[2 * q + 1]};
        };
        // Products of one row block on the matrix pipe (zero accumulator: D = fl(a * x) exactly) into one of TWO result blocks, alternating;
        // behind each matrix instruction the vector ALU adds the PREVIOUS instruction's block to its running sums -- so a wavefront keeps both
        // pipes busy by itself (a single result block would put ~18 idle issue slots between every matrix instruction and its adds; the
        // one-result-block variant at five wavefronts per SIMD was measured -- 31.42 against 31.58 ms per AllConvNet forward but 16.2 -> 22.2 GB of HBM
        // reads -- and removed).  The
        // block pending at the first instruction is all zeros: +0.0 added to sums that are still +0.0.
        f32x32 d0 = zero, d1 = zero;
        auto product = [&](auto T_, const float a, const float x, f32x2 (&pending_sum)[16]) {
            constexpr int T = decltype(T_)::value;
            // The adds are inline asm, which the compiler's hazard recognizer does not cover: a vector-ALU read of a matrix instruction's result
            // needs passes + 2 = 18 wait states behind the 16-pass v_mfma_f32_32x32x1 (what LLVM inserts on gfx950: the tails below).  The reader of d0 sits behind
            // [s_nop 1, 16 adds of d1, the next matrix instruction, s_nop 1] = 21; tests/test_isa_lint.py measures that distance in the ISA.
            if constexpr ((T & 1) == 0) {
                d0 = __builtin_amdgcn_mfma_f32_32x32x1f32(a, x, zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 1");
                add_into(pending_sum, d1);
            } else {
                d1 = __builtin_amdgcn_mfma_f32_32x32x1f32(a, x, zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 1");
                add_into(pending_sum, d0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto step = [&](auto slot, const int j) {
            constexpr int S = decltype(slot)::value;
            col_landed();
            const int col_far = col_nxt;                                        // index of stored column j + PF
            const int vrow_far = TAPS ? (int)cv_nxt.y : clampj(j + PF);
            fetch_col(j + PF + 1);
            landed(xb[S], xa[S][0], xa[S][1], xa[S][2]);
            __builtin_amdgcn_sched_barrier(0);
            product(std::integral_constant<int, S * NRB>(), xa[S][0], xb[S], acc[NRB - 1]);
            if constexpr (NRB > 1) product(std::integral_constant<int, S * NRB + 1>(), xa[S][1], xb[S], acc[0]);
            if constexpr (NRB > 2) product(std::integral_constant<int, S * NRB + 2>(), xa[S][2], xb[S], acc[1]);
            fetch(xb[S], xa[S][0], xa[S][1], xa[S][2], vrow_far, col_far);
        };
        int j = 0;
        for (; j + PF <= ncol; j += PF) mfma_static_for<0, PF>([&](auto S_) { step(S_, j + decltype(S_)::value); });
        // the block still pending (PF * NRB is even: the last one written is d1; zeros if the loop never ran) -- through plain C++ adds: the compiler sees
        // the dependency on the matrix instruction and spaces them itself (here and in the tail below nothing else fills the gap)
        add_into_c(acc[NRB - 1], d1);
        // everything in flight lands (the ring holds the last ncol % PF stored columns and, behind them, harmless re-loads of the last column)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+s"(col_nxt), "+s"(cv_nxt));
#pragma unroll
        for (int q = 0; q < PF; q++) asm volatile("" : "+v"(xb[q]), "+v"(xa[q][0]), "+v"(xa[q][1]), "+v"(xa[q][2]));
        // the last ncol % PF stored columns, one result block (the compiler spaces each matrix instruction and its adds)
        auto tail = [&](auto slot) {
            constexpr int S = decltype(slot)::value;
#pragma unroll
            for (int b = 0; b < NRB; b++) {
                const f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(xa[S][b], xb[S], zero, 0, 0, 0);
                add_into_c(acc[b], d);
            }
        };
        mfma_static_for<0, PF - 1>([&](auto S_) {
            if (j + decltype(S_)::value < ncol) tail(S_);
        });
    }
    // D layout of v_mfma_f32_32x32x1_2b_f32: register 16 * blk + r of lane l = element (row 8 * (r / 4) + 4 * (l / 32) + r % 4, column l % 32) of block blk;
    // block blk = batch columns 32 * blk .. 32 * blk + 31 of this wavefront's 64.  A store instruction writes two 128-byte row segments.
    // (a lane stores the columns c0 + (lane & 31) + 32 * blk -- not the column it loaded its B operand for)
    const int half = lane >> 5;
    const int64_t colo = c0 + (lane & 31);
    float xl[2] = {0.0f, 0.0f};                                                  // TAPS: the homogeneous coordinate of this lane's two columns (bias last)
    if constexpr (TAPS) {
        if (tp.lastcol) {
#pragma unroll
            for (int blk = 0; blk < 2; blk++)
                if (colo + 32 * blk < n_vecs) xl[blk] = X[tp.last_in_row * ldx + colo + 32 * blk];
        }
    }
#pragma unroll
    for (int b = 0; b < NRB; b++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int mi = r0 + 32 * b + 8 * (r / 4) + 4 * half + (r % 4);
            if (mi < nmem) {
                const int64_t row = TAPS ? ((int64_t)mi * tp.HoWo + g) : (int64_t)grp_rows[rbeg + mi];
                float lc = 0.0f;
                if constexpr (TAPS) lc = tp.lastcol ? tp.lastcol[row] : 0.0f;
#pragma unroll
                for (int blk = 0; blk < 2; blk++) {
                    const int64_t cc = colo + 32 * blk;
                    if (cc < n_vecs) {
                        float v = acc[b][(16 * blk + r) / 2][(16 * blk + r) % 2];
                        if (TAPS && lc != 0.0f) {                                // the bias entry, where the reference stores one: separate multiply and add
                            const float bp = xl[blk] * lc;
                            v = v + bp;
                        }
                        if (relu) v = mfma_relu_f(v);
                        __builtin_nontemporal_store(v, Y + row * ldy + cc);
                    }
                }
            }
        }
    }
}

// The 16-row form for BIG pattern groups (a keyed nn.Linear in the reference's order: 4 096 rows x 25 089 stored columns): the walk over a row's stored columns
// is serial by contract, so parallelism can only come from rows x batch columns.  v_mfma_f32_16x16x1_4b_f32 with a zero accumulator = the rounded products of
// 16 member rows x 4 blocks of 16 batch columns (lane l supplies ITS OWN column as the B operand and value l % 16 as the A operand; bit-identical to v_mul_f32:
// tools/micro/mfma16_product.hip), 8 packed adds per stored column.  Taken when it puts two wavefronts on every SIMD (csr_spmm_groups: 21-27 % faster than the
// LDS-staged big-group kernel there; slower with one).
// One wavefront = 16 rows x 64 columns; a workgroup = four 64-column blocks of the same 16 rows.  Same operand ring as above (2 loads per stored column).
template <int PF>
__global__ __launch_bounds__(256, 2) void csr_group_mfma16_kernel(int64_t n_work, const int32_t* __restrict__ work_grp, const int32_t* __restrict__ work_r0,
                                                                  const int32_t* __restrict__ grp_colptr, const int32_t* __restrict__ grp_cols,
                                                                  const int32_t* __restrict__ grp_rowptr, const int32_t* __restrict__ grp_rows,
                                                                  const int64_t* __restrict__ grp_valptr, const float* __restrict__ grp_vals,
                                                                  const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy, int64_t n_vecs, int relu) {
    static_assert(PF % 2 == 0, "two result blocks alternate");
    const int64_t n_ct = (n_vecs + 255) / 256;
    const int64_t n_items = n_ct * n_work;
    const int64_t chunk = (n_items + 7) >> 3;
    const int64_t item = (int64_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (item >= n_items || (blockIdx.x >> 3) >= chunk) return;
    const int64_t ct = item / n_work;
    const int64_t w = item - ct * n_work;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int g = __builtin_amdgcn_readfirstlane(work_grp[w]);
    const int r0 = __builtin_amdgcn_readfirstlane(work_r0[w]);
    const int cbeg = __builtin_amdgcn_readfirstlane(grp_colptr[g]);
    const int ncol = __builtin_amdgcn_readfirstlane(grp_colptr[g + 1]) - cbeg;
    const int rbeg = __builtin_amdgcn_readfirstlane(grp_rowptr[g]);
    const int nmem = __builtin_amdgcn_readfirstlane(grp_rowptr[g + 1]) - rbeg;
    const int rpad = (nmem + 15) / 16 * 16;
    const int64_t c0 = ct * 256 + (int64_t)wave * 64;
    if (c0 >= n_vecs) return;                              // (wave-uniform; no barriers in this kernel)
    const int64_t c = c0 + lane;
    const bool active = c < n_vecs;
    f32x2 acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = f32x2{0.0f, 0.0f};
    if (ncol > 0) {
        auto uni = [](const uint64_t v) {
            return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
        };
        const uint64_t cbase = uni(reinterpret_cast<uint64_t>(grp_cols + cbeg));
        const uint64_t xbase = uni(reinterpret_cast<uint64_t>(X));
        const uint64_t vbase = uni(reinterpret_cast<uint64_t>(grp_vals + grp_valptr[g] + r0));
        const uint32_t b_off = 4u * (uint32_t)(active ? c : c0);
        const uint32_t a_off = 4u * (uint32_t)(lane & 15);
        const uint64_t ldx_b = 4ull * (uint64_t)ldx;
        const uint64_t vstep = 4ull * (uint64_t)(uint32_t)rpad;
        auto clampj = [&](const int j) { return j < ncol ? j : ncol - 1; };
        float xa[PF], xb[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) xa[q] = xb[q] = 0.0f;
        int col_nxt = 0;
        auto fetch_col = [&](const int j) {
            const uint64_t caddr = cbase + 4ull * (uint64_t)(uint32_t)clampj(j);
            asm volatile("s_load_dword %0, %1, 0x0" : "=&s"(col_nxt) : "s"(caddr));
        };
        auto col_landed = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(col_nxt)); };
        auto fetch = [&](float& rb, float& ra, const int vrow, const int col) {
            const uint64_t xaddr = xbase + (uint64_t)(uint32_t)col * ldx_b;
            asm volatile("global_load_dword %0, %1, %2" : "=&v"(rb) : "v"(b_off), "s"(xaddr));
            const uint64_t va = vbase + (uint64_t)(uint32_t)vrow * vstep;
            asm volatile("global_load_dword %0, %1, %2" : "=&v"(ra) : "v"(a_off), "s"(va));
        };
        auto landed = [&](float& rb, float& ra) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rb), "+v"(ra) : "n"(2 * (PF - 1))); };
        fetch_col(0);
        mfma_static_for<0, PF>([&](auto I_) {
            constexpr int I = decltype(I_)::value;
            col_landed();
            const int cj = col_nxt;
            fetch_col(I + 1);
            fetch(xb[I], xa[I], clampj(I), cj);
        });
        f32x16 zero;
#pragma unroll
        for (int q = 0; q < 16; q++) zero[q] = 0.0f;
        auto add_into = [&](const f32x16& d) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f32x2 p2 = {d[2 * q], d[2 * q + 1]};
                asm("v_pk_add_f32 %0, %1, %0" : "+v"(acc[q]) : "v"(p2));
            }
        };
        f32x16 d0 = zero, d1 = zero;                        // two result blocks alternate: the adds of one run behind the matrix instruction of the other
        auto step = [&](auto slot, const int j) {
            constexpr int S = decltype(slot)::value;
            col_landed();
            const int col_far = col_nxt;
            fetch_col(j + PF + 1);
            landed(xb[S], xa[S]);
            __builtin_amdgcn_sched_barrier(0);
            // (inline-asm adds: the 8-pass v_mfma_f32_16x16x1 needs 8 + 2 = 10 wait states before a vector-ALU read of its result; the reader of d0 sits
            // behind [s_nop 2, 8 adds of d1, the next matrix instruction, s_nop 2] = 15; tests/test_isa_lint.py measures it)
            if constexpr ((S & 1) == 0) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x1f32(xa[S], xb[S], zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 2");
                add_into(d1);
            } else {
                d1 = __builtin_amdgcn_mfma_f32_16x16x1f32(xa[S], xb[S], zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 2");
                add_into(d0);
            }
            __builtin_amdgcn_sched_barrier(0);
            fetch(xb[S], xa[S], clampj(j + PF), col_far);
        };
        int j = 0;
        for (; j + PF <= ncol; j += PF) mfma_static_for<0, PF>([&](auto S_) { step(S_, j + decltype(S_)::value); });
        auto add_into_c = [&](const f32x16& d) {            // the same sums with the dependency visible to the compiler (it spaces them behind the matrix instruction itself)
#pragma unroll
            for (int q = 0; q < 8; q++) acc[q] = acc[q] + f32x2{d[2 * q], d[2 * q + 1]};
        };
        add_into_c(d1);                                     // the block still pending (PF is even: the last one written is d1; zeros if the loop never ran)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+s"(col_nxt));
#pragma unroll
        for (int q = 0; q < PF; q++) asm volatile("" : "+v"(xb[q]), "+v"(xa[q]));
        mfma_static_for<0, PF - 1>([&](auto S_) {
            constexpr int S = decltype(S_)::value;
            if (j + S < ncol) add_into_c(__builtin_amdgcn_mfma_f32_16x16x1f32(xa[S], xb[S], zero, 0, 0, 0));
        });
    }
    // D layout: register 4 * blk + r of lane l = element (row 4 * (l / 16) + r, column 16 * blk + l % 16)
    const int quarter = lane >> 4;
    const int64_t colo = c0 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int mi = r0 + 4 * quarter + r;
        if (mi < nmem) {
            const int64_t row = (int64_t)grp_rows[rbeg + mi];
#pragma unroll
            for (int blk = 0; blk < 4; blk++) {
                const int64_t cc = colo + 16 * blk;
                if (cc < n_vecs) {
                    float v = acc[(4 * blk + r) / 2][(4 * blk + r) % 2];
                    if (relu) v = mfma_relu_f(v);
                    Y[row * ldy + cc] = v;
                }
            }
        }
    }
}

// The order-preserving product of a factored conv operator through the kernel above (convtaps_spmm, KN_FLAG_EXACT, operators that carry the table).
static int exact_table_launch(const MfTaps& tp, int64_t n_pix, bool strided, int force_nrb, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, int relu, hipStream_t s) {
    // Row blocks (32 output channels each) per workgroup.  ONE (two result blocks alternating): 123 registers, four wavefronts per SIMD instead of two with three blocks (227 registers) -- the
    // matrix instruction and the packed adds behind it are dependent work of ONE wavefront, and the pair costs 145 cycles with four wavefronts interleaved
    // against 155 with two (tools/micro/mfma_add_rate.hip).  Same-process A/B on the AllConvNet forward (tools/ab_allconv.py): 33.9 -> 31.6 ms with the strided layers kept at three blocks, every
    // layer faster (conv2 10.23 -> 9.60, conv5 9.81 -> 8.97 ms), bit-equal.  The price is paid in L2: the activation rows of a pixel are now requested by
    // Cout / 32 workgroups instead of Cout / 96, and the siblings find each other's rows only while they stay within an XCD's 4 MB of each other: HBM reads
    // of the seven launches 13.1 -> 21.3 GB with one block everywhere.  Strided layers (neighbouring pixels share 3 of 9 input pixels instead of 6: conv3 +2.5 GB, conv6 +2.6 GB for
    // 0.14 / 0.25 ms) keep three blocks: 16.2 GB.
    int nrb = strided ? (tp.Cout % 96 == 0 ? 3 : (tp.Cout % 64 == 0 ? 2 : 1)) : 1;
    if (force_nrb >= 1 && force_nrb <= 3 && tp.Cout % (32 * force_nrb) == 0) nrb = force_nrb;      // Tuning::table_nrb (recorded at create)
    MfTaps t = tp;
    t.n_cc = tp.Cout / (32 * nrb);
    const int64_t n_work = n_pix * t.n_cc;
    const int64_t items = ((n_vecs + 255) / 256) * n_work;
    const int64_t grid = ((items + 7) / 8) * 8;
    const std::string d = "csr_group_mfma_kernel<row blocks=" + std::to_string(nrb) + ",taps> (factored operator: products on the matrix pipe from the tap table)";
    if (nrb == 3) KN_LAUNCH(d, (csr_group_mfma_kernel<3, 8, 4, true>), dim3((unsigned)grid), dim3(256), 0, s, n_work, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, x, ldx, y, ldy, n_vecs, relu, t);
    else if (nrb == 2) KN_LAUNCH(d, (csr_group_mfma_kernel<2, 6, 4, true>), dim3((unsigned)grid), dim3(256), 0, s, n_work, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, x, ldx, y, ldy, n_vecs, relu, t);
    else KN_LAUNCH(d, (csr_group_mfma_kernel<1, 6, 4, true>), dim3((unsigned)grid), dim3(256), 0, s, n_work, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, x, ldx, y, ldy, n_vecs, relu, t);
    KN_HIP(hipGetLastError());
    return KN_OK;
}

int convtaps_exact_table_spmm(const ConvTapsDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, int relu, hipStream_t s) {
    MfTaps t;
    t.ex_ptr = A.ex_ptr;
    t.ex_tab = A.ex_tab;
    t.tapsT = A.tapsT;
    t.lastcol = A.has_last ? A.lastcol : nullptr;
    t.pix_order = A.ex_order ? A.ex_order : A.pix_order;
    t.last_in_row = A.Cin * A.Hin * A.Win;
    t.cout_pad = (int32_t)A.cout_pad;
    t.HoWo = (int32_t)(A.Hout * A.Wout);
    t.Cout = (int32_t)A.Cout;
    t.n_cc = 1;
    return exact_table_launch(t, A.Hout * A.Wout, A.Hin * A.Win > A.Hout * A.Wout, A.tune.table_nrb, x, ldx, n_vecs, y, ldy, relu, s);
}

// work lists per NRB (CsrDev::mf_*): chunks of 32 * NRB member rows of the pattern groups with >= MF_MIN_MEMBERS members
int csr_group_mfma_spmm(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, int relu, hipStream_t s) {
    const int64_t n_ct = (n_vecs + 255) / 256;
    for (int k = 0; k < 3; k++) {
        if (A.n_mf[k] == 0) continue;
        const int64_t items = n_ct * A.n_mf[k];
        const int64_t grid = ((items + 7) / 8) * 8;
        const std::string d = "csr_group_mfma_kernel<row blocks=" + std::to_string(k + 1) + "> (products on the matrix pipe, K = 1, zero accumulator)";
        if (k == 0) KN_LAUNCH(d, (csr_group_mfma_kernel<1, 6>), dim3((unsigned)grid), dim3(256), 0, s, A.n_mf[k], A.mf_grp[k], A.mf_r0[k], A.grp_colptr, A.grp_cols, A.grp_rowptr, A.grp_rows,
                              A.grp_valptr, A.grp_vals, x, ldx, y, ldy, n_vecs, relu);
        if (k == 1) KN_LAUNCH(d, (csr_group_mfma_kernel<2, 6>), dim3((unsigned)grid), dim3(256), 0, s, A.n_mf[k], A.mf_grp[k], A.mf_r0[k], A.grp_colptr, A.grp_cols, A.grp_rowptr, A.grp_rows,
                              A.grp_valptr, A.grp_vals, x, ldx, y, ldy, n_vecs, relu);
        if (k == 2) {
            const int pf = A.tune.mf_pf;      // operand columns in flight (diagnostic build: A/B knob) (AllConvNet forward, same box: 33.79 / 33.38 / 33.53 ms at 6 / 8 / 10)
            if (pf == 10) KN_LAUNCH(d + " pf=10", (csr_group_mfma_kernel<3, 10>), dim3((unsigned)grid), dim3(256), 0, s, A.n_mf[k], A.mf_grp[k], A.mf_r0[k], A.grp_colptr, A.grp_cols, A.grp_rowptr,
                                    A.grp_rows, A.grp_valptr, A.grp_vals, x, ldx, y, ldy, n_vecs, relu);
            else if (pf == 8) KN_LAUNCH(d + " pf=8", (csr_group_mfma_kernel<3, 8>), dim3((unsigned)grid), dim3(256), 0, s, A.n_mf[k], A.mf_grp[k], A.mf_r0[k], A.grp_colptr, A.grp_cols, A.grp_rowptr,
                                        A.grp_rows, A.grp_valptr, A.grp_vals, x, ldx, y, ldy, n_vecs, relu);
            else KN_LAUNCH(d, (csr_group_mfma_kernel<3, 6>), dim3((unsigned)grid), dim3(256), 0, s, A.n_mf[k], A.mf_grp[k], A.mf_r0[k], A.grp_colptr, A.grp_cols, A.grp_rowptr, A.grp_rows,
                                        A.grp_valptr, A.grp_vals, x, ldx, y, ldy, n_vecs, relu);
        }
    }
    KN_HIP(hipGetLastError());
    return KN_OK;
}

// big pattern groups (CsrDev::mf16_*: chunks of 16 member rows)
int csr_group_mfma16_spmm(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, int relu, hipStream_t s) {
    const int64_t items = ((n_vecs + 255) / 256) * A.n_mf16;
    const int64_t grid = ((items + 7) / 8) * 8;
    KN_LAUNCH("csr_group_mfma16_kernel (big pattern groups: products of 16 rows x 64 columns on the matrix pipe, K = 1, zero accumulator)", (csr_group_mfma16_kernel<8>),
              dim3((unsigned)grid), dim3(256), 0, s, A.n_mf16, A.mf16_grp, A.mf16_r0, A.grp_colptr, A.grp_cols, A.grp_rowptr, A.grp_rows, A.grp_valptr, A.grp_vals, x, ldx, y, ldy, n_vecs, relu);
    KN_HIP(hipGetLastError());
    return KN_OK;
}

}  // namespace kn
