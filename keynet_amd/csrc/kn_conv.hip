// kn_conv.hip -- keyed block-Toeplitz conv operator on gfx950 matrix cores (f32-input MFMA, exact f32 products).
//
// Replaces TiledMatrix.torchdot for a Conv2dTiledMatrix (keynet/sparse.py:603-612 + 781-835).  The reference rebuilds
// the full CSR from (blocks, tiles) on every call and runs csr_matvecs; the operator it applies is
//     W[o + co*HoWo, i + ci*HiWi] = sum_{entries e=(o,i)} coef_e * taps[tap_e][co][ci]     (+ last column, + e_last row)
// i.e. a permuted/keyed im2col convolution.  Here it is an implicit GEMM per OUTPUT PIXEL o:
//     Y[co, o, b] = sum_{slots s of o} sum_ci  tapsT[tap_s][ci][co] * ( coef_s * X[ci*HiWi + in_s, b] )
//   M = Cout (tile MT), N = batch columns (tile NB, contiguous in HBM: one gathered X row = one coalesced segment),
//   K = slots(o) x Cin, walked in chunks of KC input channels of one slot.
// Per workgroup (4 wavefronts): the tap tile [KC][MT] and the gathered X tile [KC][NB] are staged in LDS (double
// buffered, one barrier per chunk; 3-stage software pipeline: chunk q in LDS, chunk q+1 in registers and written to the
// other LDS buffer mid-chunk, chunk q+2's global loads issued right after); each wavefront owns a
// (TM*32)x(TN*32) sub-tile as TMxTN accumulators of v_mfma_f32_32x32x2_f32; A/B fragments are conflict-free LDS reads
// (lane&31 -> consecutive dwords, lane>>5 -> k) at base-register + immediate addresses (permuted tile columns, chunk loop
// unrolled by two).  The chunk loop is kept almost free of VALU work (5 instructions per 32 MFMAs): VALU issue competes with
// MFMA issue on a SIMD.  The bias column (homogeneous coordinate) is one extra MFMA k-step; the epilogue applies ReLU and
// streams the tile out 16 bytes per lane through a per-wavefront LDS transposition (kn_store_tile).
// Work items (pixel, batch tile, Cout tile) are dealt to the 8 XCDs in contiguous chunks with the Cout tile fastest,
// so the workgroups that share one gathered X tile run on one XCD and hit its L2; the last partial round of workgroups is
// split into quarter tiles (TAIL), and launches with only a few rounds of resident workgroups take the occupancy whose partial
// round is smaller.  Other kernels in this file: convtaps_smallk_pipe_kernel (first layer of an image net: whole contraction
// <= 28 rows, write-bound: persistent workgroups, the next pixel's gathers and MFMAs run under the current pixel's stores;
// convtaps_smallk_kernel = its one-shot predecessor and fallback), convtaps_exact_pipe_kernel / convtaps_exact_kernel
// (KN_FLAG_EXACT: the reference's accumulation order and rounding on the VALU, bit-exact), conv_lastrow_kernel (homogeneous row).
#include "kn_internal.h"
#include <type_traits>
#include <cstdio>
#include <cstring>
#include <vector>

namespace kn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
static constexpr int MAX_FAST_SLOTS = 64;   // slots per output pixel the FAST path keeps in LDS (3x3 .. 7x7 windows)
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: arrays of HIP's float4 struct are not promoted to registers

// Ordering point for LDS traffic that stays inside one wavefront (no s_barrier, no vmcnt wait on in-flight global stores).
__device__ __forceinline__ void kn_wave_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// Ordering point for the machine scheduler (holds inside one basic block).
__device__ __forceinline__ void kn_order() { __builtin_amdgcn_sched_barrier(0); }

// Wide store of a wavefront's (TM*32) x (TN*32) accumulator sub-tile.  An accumulator register holds one output row per 32-lane
// half (4 bytes per lane), so storing it directly would cost 64 scalar stores per lane.  Instead the wavefront transposes 8 rows
// at a time through its own LDS slice `stage` ([8][TN*32] floats; private, and a wavefront's LDS operations execute in order) and
// streams 16 bytes per lane: whole row segments, nontemporal (the output is not re-read by this launch).  `yp` is this lane's
// pointer into the first row it stores (sub-tile row `lane / (TN*8)`); the rows a lane visits are a constant RPI apart, so the
// address is ONE running pointer plus a wave-uniform byte step -- no per-store 64-bit multiplies.  `rows_ok` (wave-uniform) says
// that every row of the sub-tile exists (< Cout): the per-row test disappears for channel counts that fill the tile.
template <int TM, int TN, bool ALL_ROWS = false>
__device__ __forceinline__ void kn_store_tile(const f32x16 (&acc)[TM][TN], float* stage, const int lane, float* yp, const int64_t row_step_bytes,
                                              const int m_first, const int Cout, const bool rows_ok, const int relu, float* absmax = nullptr, float* carry = nullptr) {
    // ReLU without a branch: clamp from below at 0, or at -inf (a no-op; NaN stays NaN either way).  ALL_ROWS (compile time: the
    // tile lies inside Cout) additionally removes the per-row test, so the 16 stores are straight-line code and a caller that
    // keeps loads in flight across them gets an exact counted vmcnt from the compiler instead of a drain.
    const float lo = relu ? 0.0f : -__builtin_inff();
    constexpr int COLS = TN * 32;                 // columns of this wave's sub-tile
    constexpr int LPR = COLS / 4;                 // lanes per row (16 B each)
    constexpr int RPI = 64 / LPR;                 // rows per wave-instruction
    const int rl = lane / LPR, c4 = lane % LPR;
    char* ypb = reinterpret_cast<char*>(yp);
    // running max |y| of what this lane stores (kn_spmm_screen: the float-key contract re-screens every forward on the NEXT layer's
    // max |x|): two v_max3_f32 with |.| source modifiers per 16-byte store, NaN operands ignored; rows beyond Cout are zero (zero-padded taps)
    float am = 0.0f;
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {             // rows 8g .. 8g+7 of the 32-row MFMA tile i
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) stage[(rr + 4 * (lane >> 5)) * COLS + j * 32 + (lane & 31)] = acc[i][j][4 * g + rr];
            kn_wave_sync();
#pragma unroll
            for (int h = 0; h < 8 / RPI; h++) {
                const int rloc = rl + h * RPI;
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + rloc * COLS + c4 * 4);
                v.x = (v.x < lo) ? lo : v.x;
                v.y = (v.y < lo) ? lo : v.y;
                v.z = (v.z < lo) ? lo : v.z;
                v.w = (v.w < lo) ? lo : v.w;
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(am) : "v"(v.x), "v"(v.y));
                asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(am) : "v"(v.z), "v"(v.w));
                if (ALL_ROWS || rows_ok || m_first + i * 32 + 8 * g + rloc < Cout) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(ypb));
                ypb += row_step_bytes;            // next visited row: RPI rows further (also across g and i: 8 and 32 are multiples of RPI steps)
            }
            kn_wave_sync();
        }
    }
    if (carry) *carry = (am > *carry) ? am : *carry;          // a persistent caller commits once, after its last tile (kn_wave_absmax_commit: why)
    else if (absmax) kn_wave_absmax_commit(am, absmax, lane);      // wave-uniform branch; one relaxed load (+ rarely an atomic) per tile
}

struct ConvArgs {
    const float* tapsT;
    const int32_t* pix_ptr;
    const int32_t* slot_in;
    const int32_t* slot_tap;
    const float* slot_coef;
    const int32_t* pix_order;
    const float* lastcol;
    const float* X;
    float* Y;
    int64_t ldx, ldy;
    int32_t cin_pad, cout_pad, Cin, Cout, HiWi, HoWo;
    int32_t n_vecs, relu, unit_coef, vec_ok;
    int32_t n_mt, n_bt, n_pix, max_slots, ntaps, wide_store;
    int64_t last_in_row;   // Cin*HiWi (row of X holding the homogeneous coordinate)
    int32_t tail_main;     // work items per XCD chunk computed as full tiles; the rest of the chunk runs as 4 quarter tiles each
    int64_t* stamps;          // diagnostic build only (KN_STAMPS): per-workgroup {start, end} of s_memrealtime, XCC id, kind; null otherwise
    const int32_t* sk_desc;   // small-K pipeline: per-pixel descriptors in processing order (ConvTapsDev::sk_desc), or null
    int32_t sk_stride, sk_tab_rows;
    const uint16_t* tapsB;    // bf16x3 path: taps split into three bf16 planes (ConvTapsDev::tapsB), or null
    int64_t tapsB_plane;      // bytes between planes
    float* absmax;            // kn_spmm_screen: device slot raised to max |Y| by the wide-store epilogue, or null
#ifdef KN_ABLATION
    int32_t abl;           // diagnostic build only (tools/ablate_conv.sh): bit 0 no chunk barrier, 1 no LDS stores, 2 no global loads, 4 no pointer walk, 5 / 6 no tap / activation loads; bf16x3 kernel (tools/ablate_bf16x3.sh): 8 no split + activation stores, 9 no global loads, 10 no barrier, 11 no tap stores
#endif
};

// Loop-piece ablation (timing only, results are garbage): compiled in with -DKN_ABLATION into a separate library by
// tools/ablate_conv.sh; the shipped library has none of these tests in its loops.
#ifdef KN_ABLATION
#define KN_ABL(p, bit) (((p).abl >> (bit)) & 1)
#else
#define KN_ABL(p, bit) 0
#endif

// item -> (Cout tile, position in the pixel order, batch tile): the Cout tile is fastest (its workgroups share one gathered X tile),
// the batch tile slowest (measured: batch tile inner, i.e. an XCD owning a pixel range for every batch tile, is neutral)
__device__ __forceinline__ void decode_conv_item(const ConvArgs& p, const int64_t item, int& mt, int& pi, int& bt) {
    mt = (int)(item % p.n_mt);
    const int64_t t1 = item / p.n_mt;
    pi = (int)(t1 % p.n_pix);
    bt = (int)(t1 / p.n_pix);
}

// FAST = (batch 16-byte aligned and a multiple of the batch tile NB) && (all coefficients 1: identity / permutation keys) && (Cin % KC == 0):
// the loaders are straight-line code, so the next chunk's global loads stay in flight in registers during the MFMAs.
// MODE 2 = FAST with wave-uniform tile pointers (SPTR): the chunk's four tile loads are saddr-form global_load_dwordx4 (SGPR-pair base +
// one constant 32-bit offset per lane) and the pointer walk over (slot, channel chunk) runs on the scalar ALU.  What it replaces: four
// 64-bit per-lane pointers advanced with eight VALU adds per chunk.  Measured on the way (loads left out of the loop, results
// discarded): every register-destination dwordx4 load costs the matrix pipe of its SIMD ~29 cycles, the per-lane pointer walk 1.6 %
// of the launch; the same tiles fetched by global_load_lds_dwordx4 (no VGPR destination, no ds_write pass, two stages with a
// vmcnt(0) in front of each chunk's barrier) ran 3.5 % SLOWER than register staging and was dropped.
template <int MT, int NB, int KC, int WM, int WN, int MODE>
__device__ __forceinline__ void convtaps_mfma_tile(const ConvArgs& p, const int o, const int m0, const int b0, float* lds) {
    constexpr bool FAST = MODE >= 1;
    constexpr bool SPTR = MODE >= 2;
    constexpr bool GROUPS = MODE == 3;               // SPTR over pixels with more than 64 slots: slot groups (its own instantiation: the single-group walk pays nothing for it)
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    static_assert(!SPTR || KC == 16, "the scalar-pointer loader is built for 16-row chunks of whole channels");
    constexpr int TM = MT / WM / 32;
    constexpr int TN = NB / WN / 32;
    constexpr int A4 = KC * MT / 4;                 // float4s in the A tile
    constexpr int B4 = KC * NB / 4;
    constexpr int AL = (A4 + 255) / 256;            // float4 loads per thread
    constexpr int BL = (B4 + 255) / 256;
    constexpr int LS_AT = (KC >= 8) ? KC / 2 - 2 : 0;   // k-step at which chunk q+1 is written to LDS ...
    constexpr int GL_AT = KC / 2;                        // ... and the one at which chunk q+2's global loads are issued
    float* As = lds;                  // [2][KC][MT]
    float* Bs = lds + 2 * KC * MT;    // [2][KC][NB]
    int64_t* s_da = reinterpret_cast<int64_t*>(lds + 2 * KC * MT + 2 * KC * NB);     // FAST: per-slot byte delta to the next chunk's tap tile
    int64_t* s_db = s_da + MAX_FAST_SLOTS;                                           //       ... and to its activation tile

    // wave-uniform by construction; pinned to SGPRs (a value that comes out of a vector-memory load counts as divergent for the compiler,
    // and every counter, pointer and branch derived from it would live in VGPRs / go through exec masks)
    const int s_beg = __builtin_amdgcn_readfirstlane(p.pix_ptr[o]);
    const int n_slots = __builtin_amdgcn_readfirstlane(p.pix_ptr[o + 1]) - s_beg;
    const int cpk = p.cin_pad / KC;
    const int n_chunks = n_slots * cpk;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    f32x4 ra[AL], rb[BL];

    auto gload = [&](int q) {
        const int s = s_beg + q / cpk;
        const int ci0 = (q % cpk) * KC;
        const int tap = p.slot_tap[s];
        const int in = p.slot_in[s];
        const float coef = (FAST || p.unit_coef) ? 1.0f : p.slot_coef[s];
        const float* abase = p.tapsT + ((int64_t)tap * p.cin_pad + ci0) * p.cout_pad + m0;
#pragma unroll
        for (int i = 0; i < AL; i++) {
            const int f = tid + i * 256;
            if (A4 % 256 == 0 || f < A4) {
                const int r = f / (MT / 4);
                const int c4 = f % (MT / 4);
                ra[i] = *reinterpret_cast<const f32x4*>(abase + (int64_t)r * p.cout_pad + c4 * 4);
            }
        }
        if constexpr (FAST) {
            const float* xbase = p.X + ((int64_t)ci0 * p.HiWi + in) * p.ldx + b0;
#pragma unroll
            for (int i = 0; i < BL; i++) {
                const int f = tid + i * 256;
                if (B4 % 256 == 0 || f < B4) {
                    const int r = f / (NB / 4);
                    const int c4 = f % (NB / 4);
                    rb[i] = *reinterpret_cast<const f32x4*>(xbase + (int64_t)r * p.HiWi * p.ldx + c4 * 4);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < BL; i++) {
            const int f = tid + i * 256;
            if (B4 % 256 == 0 || f < B4) {
                const int r = f / (NB / 4);
                const int c4 = f % (NB / 4);
                const int ci = ci0 + r;
                const int b = b0 + c4 * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ci < p.Cin) {
                    const float* src = p.X + ((int64_t)ci * p.HiWi + in) * p.ldx + b;
                    if (p.vec_ok) {
                        if (b < p.n_vecs) v = *reinterpret_cast<const f32x4*>(src);
                    } else {
                        if (b + 0 < p.n_vecs) v.x = src[0];
                        if (b + 1 < p.n_vecs) v.y = src[1];
                        if (b + 2 < p.n_vecs) v.z = src[2];
                        if (b + 3 < p.n_vecs) v.w = src[3];
                    }
                    if (!p.unit_coef) v = v * coef;
                }
                rb[i] = v;
            }
        }
    };
    // FAST path: running pointers.  Within a slot consecutive chunks advance KC input channels (constant strides); the
    // slot's (tap, input pixel) are re-read only when the slot changes, so the steady state has no scalar loads,
    // divisions or 64-bit multiplies.  Per-thread element offsets are computed once.
    uint32_t a_off[AL], b_off[BL];      // element offsets inside one chunk (< 2^31: checked by the launcher)
    bool b_pad[BL];                     // this thread's activation row is a zero-padding row (only possible when KC < 16)
#pragma unroll
    for (int i = 0; i < AL; i++) {
        const int f = tid + i * 256;
        a_off[i] = (uint32_t)((f / (MT / 4)) * p.cout_pad + (f % (MT / 4)) * 4);
    }
#pragma unroll
    for (int i = 0; i < BL; i++) {
        const int f = tid + i * 256;
        // rows ci >= Cin exist only when Cin < KC (one zero-padded chunk per slot).  Their loads are pointed at a valid row (row 0
        // of the slot) so the address stays in range, but what they return is DISCARDED: lstore writes zeros for them, so a
        // NaN / Inf in that live activation row cannot leak into the product (0 x NaN) -- the reference has no such entries.
        const int r = f / (NB / 4);
        b_off[i] = (uint32_t)((int64_t)(r < p.Cin ? r : 0) * p.HiWi * p.ldx + (f % (NB / 4)) * 4);
        b_pad[i] = (KC < 16) && (r >= p.Cin);
    }
    // K order: input-channel chunk OUTER, slot INNER.  Workgroups of neighbouring output pixels (which share most of
    // their input pixels, at different slot positions) then touch the same activation tile within a few chunk-times,
    // so the second use hits the XCD's L2 instead of going back to the fabric.
    const int64_t a_step = (int64_t)KC * p.cout_pad;
    const int64_t b_step = (int64_t)KC * p.HiWi * p.ldx;
    int f_slot = 0;
    // FAST: every thread walks its own tile pointers.  The LDS table holds, per slot, the BYTE DELTA from this slot's tiles to
    // the next chunk's (next slot of the pixel, or slot 0 of the next channel chunk after the last one), so advancing is one
    // 64-bit add per load and the chunk loop carries no slot-offset / base-pointer arithmetic at all.
    const char* pa[AL];
    const char* pb[BL];
    int64_t da = 0, db = 0;
    if constexpr (FAST && !SPTR) {
        if (tid < n_slots) {
            const int nxt = (tid + 1 < n_slots) ? tid + 1 : 0;
            const int64_t wrap_a = (tid + 1 < n_slots) ? 0 : a_step, wrap_b = (tid + 1 < n_slots) ? 0 : b_step;
            s_da[tid] = 4 * ((int64_t)(p.slot_tap[s_beg + nxt] - p.slot_tap[s_beg + tid]) * p.cin_pad * p.cout_pad + wrap_a);
            s_db[tid] = 4 * ((int64_t)(p.slot_in[s_beg + nxt] - p.slot_in[s_beg + tid]) * p.ldx + wrap_b);
        }
        if (n_slots > 0) {
            const char* a0 = reinterpret_cast<const char*>(p.tapsT + m0 + (int64_t)p.slot_tap[s_beg] * p.cin_pad * p.cout_pad);
            const char* b0p = reinterpret_cast<const char*>(p.X + b0 + (int64_t)p.slot_in[s_beg] * p.ldx);
#pragma unroll
            for (int i = 0; i < AL; i++) pa[i] = a0 + 4 * (int64_t)a_off[i];
#pragma unroll
            for (int i = 0; i < BL; i++) pb[i] = b0p + 4 * (int64_t)b_off[i];
        }
        __syncthreads();
        if (n_slots > 0) {
            da = s_da[0];
            db = s_db[0];
        }
    }
    auto gload_fast = [&]() {
#pragma unroll
        for (int i = 0; i < AL; i++)
            if (A4 % 256 == 0 || tid + i * 256 < A4) ra[i] = *reinterpret_cast<const f32x4*>(pa[i]);
#pragma unroll
        for (int i = 0; i < BL; i++)
            if (B4 % 256 == 0 || tid + i * 256 < B4) rb[i] = *reinterpret_cast<const f32x4*>(pb[i]);
#pragma unroll
        for (int i = 0; i < AL; i++) pa[i] += da;
#pragma unroll
        for (int i = 0; i < BL; i++) pb[i] += db;
        f_slot = __builtin_amdgcn_readfirstlane((f_slot + 1 == n_slots) ? 0 : f_slot + 1);   // scalar ALU, not 4 VALU ops
        da = s_da[f_slot];      // for the NEXT call: the LDS read has a whole chunk to complete
        db = s_db[f_slot];
    };
    // ---- SPTR state.  One load instruction of the workgroup covers 1024/MT rows of the A tile (1024/NB of the B tile); load i of a
    // thread reads (row i*ARL + tid / row_lanes, four columns): a per-lane offset that is a constant of the launch plus a wave-uniform
    // pointer per load, which advances by the slot table's byte deltas (lane s of every wavefront keeps slot s's deltas in registers,
    // read back with v_readlane).
    uint32_t a_voff = 0, b_voff = 0;
    int dtab_a_lo = 0, dtab_a_hi = 0, dtab_b_lo = 0, dtab_b_hi = 0;
    int g_slot = 0;                                   // the loader's position inside its current GROUP of <= 64 slots (lane s of the tables = slot g_base + s)
    int g_base = 0, g_n = 0, g_row = 0;               // first slot / size of that group, channel chunk the loader is in
    float cf_pend = 1.0f;                             // GROUPS: coefficient of the chunk most recently requested (= the one the next LDS store writes)
    int st_slot = 0;                                  // one group: slot of the chunk the next LDS store writes (non-unit coefficients)
    float ctab = 1.0f;                                // lane s: coefficient of slot g_base + s
    const int n_slots_u = __builtin_amdgcn_readfirstlane(n_slots);
    auto uni64 = [](const char* q) {                 // readfirstlane on an already-scalar value is free; it keeps loop-carried pointers in SGPR pairs
        const uint64_t v = reinterpret_cast<uint64_t>(q);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    constexpr int ARL = 1024 / MT, BRL = 1024 / NB;
    const char* sa[AL];
    const char* sb[BL];
    // Slot GROUPS (round 5: fill-in-aware loaders).  A pixel with more than 64 slots -- keys whose inverse fills a block in: a doubly-stochastic key leaves
    // 500 .. 5 400 slots per output pixel of keyed VGG-16 -- is walked group by group, slot GROUP outermost: for each group of <= 64 slots, channel chunk
    // outer / slot inner as before (the order of a re-ordered f32 sum is free under the matrix-core contract), the tables of the next group loaded when the
    // loader has requested the group's last chunk.  One group (<= 64 slots: permutation, Givens, gain keys) is the round-2 walk, instruction for instruction.
    auto sptr_group = [&](const int base) {           // lane s: byte deltas from slot base + s to the next chunk's tiles, and its coefficient
        g_base = base;
        g_n = (n_slots_u - base) < 64 ? (n_slots_u - base) : 64;
        g_slot = 0;
        g_row = 0;
        if (lane < g_n) {
            const int nxt = (lane + 1 < g_n) ? lane + 1 : 0;
            const int64_t wrap_a = (lane + 1 < g_n) ? 0 : a_step, wrap_b = (lane + 1 < g_n) ? 0 : b_step;
            const int64_t d_a = 4 * ((int64_t)(p.slot_tap[s_beg + base + nxt] - p.slot_tap[s_beg + base + lane]) * p.cin_pad * p.cout_pad + wrap_a);
            const int64_t d_b = 4 * ((int64_t)(p.slot_in[s_beg + base + nxt] - p.slot_in[s_beg + base + lane]) * p.ldx + wrap_b);
            dtab_a_lo = (int)(uint32_t)d_a;
            dtab_a_hi = (int)(d_a >> 32);
            dtab_b_lo = (int)(uint32_t)d_b;
            dtab_b_hi = (int)(d_b >> 32);
            if (!p.unit_coef) ctab = p.slot_coef[s_beg + base + lane];
        }
        const int64_t tap0 = __builtin_amdgcn_readfirstlane(p.slot_tap[s_beg + base]);
        const int64_t in0 = __builtin_amdgcn_readfirstlane(p.slot_in[s_beg + base]);
#pragma unroll
        for (int i = 0; i < AL; i++) sa[i] = uni64(reinterpret_cast<const char*>(p.tapsT + m0 + (tap0 * p.cin_pad + (int64_t)i * ARL) * p.cout_pad));
#pragma unroll
        for (int i = 0; i < BL; i++) sb[i] = uni64(reinterpret_cast<const char*>(p.X + b0 + ((int64_t)i * BRL * p.HiWi + in0) * p.ldx));
    };
    if constexpr (SPTR) {
        static_assert(A4 % 256 == 0 && B4 % 256 == 0, "whole load instructions");
        a_voff = 4u * (uint32_t)((tid / (MT / 4)) * p.cout_pad + (tid % (MT / 4)) * 4);
        b_voff = 4u * (uint32_t)((int64_t)(tid / (NB / 4)) * p.HiWi * p.ldx + (tid % (NB / 4)) * 4);      // < 2^31: checked by the launcher
        if (n_slots > 0) sptr_group(0);
    }
    bool g_switch = false;                            // the loader has requested its group's last chunk: the next call starts with the next group's tables
    auto sptr_load = [&]() {                          // SPTR: the next chunk's tiles -> ra / rb, saddr form; then the scalar pointer walk
        if constexpr (SPTR) {
            // (the tables are replaced HERE, before this chunk's loads are issued and after the previous chunk's have been consumed -- never with tile loads
            // in flight: vector code between an asm-issued load and its wait invites the compiler to copy a destination register that has not landed yet,
            // tests/test_isa_lint.py.  One memory round trip, once per 64 * cpk chunks.)
            if constexpr (GROUPS) {
                if (g_switch) {
                    sptr_group(g_base + g_n < n_slots_u ? g_base + g_n : 0);
                    g_switch = false;
                }
            }
            if (!KN_ABL(p, 5)) {
#pragma unroll
                for (int i = 0; i < AL; i++) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(ra[i]) : "v"(a_voff), "s"(reinterpret_cast<uint64_t>(sa[i])));
            }
            if (!KN_ABL(p, 6)) {
#pragma unroll
                for (int i = 0; i < BL; i++) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(rb[i]) : "v"(b_voff), "s"(reinterpret_cast<uint64_t>(sb[i])));
            }
            if (KN_ABL(p, 4)) return;
            if constexpr (GROUPS) {
                if (!p.unit_coef) cf_pend = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ctab), g_slot));
            }
            const int64_t da = ((int64_t)__builtin_amdgcn_readlane(dtab_a_hi, g_slot) << 32) | (uint32_t)__builtin_amdgcn_readlane(dtab_a_lo, g_slot);
            const int64_t db = ((int64_t)__builtin_amdgcn_readlane(dtab_b_hi, g_slot) << 32) | (uint32_t)__builtin_amdgcn_readlane(dtab_b_lo, g_slot);
            const bool wrap = (g_slot + 1 == g_n);
            g_slot = wrap ? 0 : g_slot + 1;
            if constexpr (GROUPS) {
                g_row = wrap ? g_row + 1 : g_row;
                g_switch = wrap && g_row == cpk;                        // (wave-uniform)
            }
#pragma unroll
            for (int i = 0; i < AL; i++) sa[i] = uni64(sa[i] + da);
#pragma unroll
            for (int i = 0; i < BL; i++) sb[i] = uni64(sb[i] + db);
        }
    };
    auto sptr_landed = [&]() {                        // the asm loads are invisible to the compiler's vmcnt bookkeeping: explicit wait
        if constexpr (SPTR) {
            if constexpr (AL == 2 && BL == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rb[0]), "+v"(rb[1]));
            else if constexpr (AL == 1 && BL == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]));
            else if constexpr (AL == 1 && BL == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(rb[0]));
            else static_assert(!SPTR, "tile shape without a wait statement");
        }
    };
    auto LOAD = [&](int q) {
        if constexpr (SPTR) sptr_load();
        else if constexpr (FAST) gload_fast();     // chunks are requested strictly in order 0,1,2,...
        else gload(q);
    };
    // LDS tile rows are stored with their columns permuted: column c = w*(T*32) + t*32 + l (wavefront w, its sub-tile t, lane l)
    // lives at t*(W*32) + w*32 + l.  A wavefront's T sub-tile fragments are then W*32 dwords apart and consecutive k-steps a
    // whole row apart -- both multiples of 64 dwords when W = 2 -- so every fragment read of a chunk is ONE base register plus
    // immediate offsets (ds_read2st64_b32) instead of a v_add per read: 16 fewer VALU instructions per chunk and wavefront
    // (+2.3 % on the 128x128 tile, same-call A/B).
    uint32_t a_lds[AL], b_lds[BL];
#pragma unroll
    for (int i = 0; i < AL; i++) {
        const int f = tid + i * 256;
        const int c = (f % (MT / 4)) * 4;
        a_lds[i] = (uint32_t)((f / (MT / 4)) * MT + ((c / 32) % TM) * (WM * 32) + (c / (TM * 32)) * 32 + (c % 32));
    }
#pragma unroll
    for (int i = 0; i < BL; i++) {
        const int f = tid + i * 256;
        const int c = (f % (NB / 4)) * 4;
        b_lds[i] = (uint32_t)((f / (NB / 4)) * NB + ((c / 32) % TN) * (WN * 32) + (c / (TN * 32)) * 32 + (c % 32));
    }
    auto lstore = [&](int buf) {
        sptr_landed();
        if (KN_ABL(p, 1)) return;                   // (the wait stays: loads in flight own their registers)
        if constexpr (SPTR) {
            // float keys whose entries carry a coefficient (photometric gains: a_out[o] / a_in[i] per (output, input) pixel pair): the
            // activation tile of the chunk is scaled by its slot's coefficient on the way to LDS, as the generic loader does per element
            if (!p.unit_coef) {
                // GROUPS: the coefficient recorded when the chunk was requested (exactly one chunk waits in registers at a time; the tables may have moved on
                // to the next group since).  One group: read from the table at the store's own cursor, as in round 2 (kept literally: written with the
                // recorded value the compiler turns this branch into eight selects per chunk of every unit-coefficient launch: -1.5 % on the 128 x 128 layers).
                float cf;
                if constexpr (GROUPS) {
                    cf = cf_pend;
                } else {
                    cf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ctab), st_slot));
                    st_slot = (st_slot + 1 == n_slots_u) ? 0 : st_slot + 1;
                }
#pragma unroll
                for (int i = 0; i < BL; i++) rb[i] = rb[i] * cf;
            }
        }
        float* a = As + buf * KC * MT;
        float* b = Bs + buf * KC * NB;
#pragma unroll
        for (int i = 0; i < AL; i++) {
            const int f = tid + i * 256;
            if (A4 % 256 == 0 || f < A4) *reinterpret_cast<f32x4*>(a + a_lds[i]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BL; i++) {
            const int f = tid + i * 256;
            if (B4 % 256 == 0 || f < B4) {
                if constexpr (FAST && KC < 16) {
                    if (b_pad[i]) rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                *reinterpret_cast<f32x4*>(b + b_lds[i]) = rb[i];
            }
        }
    };

    // Software pipeline over K chunks (3 stages): chunk q is consumed from LDS buffer q&1 while chunk q+1 sits in registers
    // (global loads in flight) and is written to the other LDS buffer in the MIDDLE of chunk q's MFMAs, immediately
    // followed by the issue of chunk q+2's global loads.  The only work between two chunks' MFMAs is one s_barrier and
    // the first fragment read, so the co-resident wavefronts of a SIMD do not all leave the matrix pipe idle together.
    // (LDS WAR: the last readers of buffer (q+1)&1 are chunk q-1's fragment reads, all retired before the barrier that
    // ended chunk q-1.)
    if (n_chunks > 0) {
        LOAD(0);
        lstore(0);
    }
    if (n_chunks > 1) LOAD(1);
    // Bias column x homogeneous coordinate = one more rank-1 term of the product: staged as a 2-row chunk (row 1 zero) in its
    // own LDS area and accumulated by ONE MFMA k-step after the chunk loop -- instead of a scalar load, 8 multiplies and 8 adds
    // per stored row segment in the epilogue.
    float* bias_a = reinterpret_cast<float*>(s_db + MAX_FAST_SLOTS);   // [2][MT], columns permuted like the tiles
    float* bias_b = bias_a + 2 * MT;                                   // [2][NB]
    if (p.lastcol) {
        const float* xlast = p.X + p.last_in_row * p.ldx;
        for (int t = tid; t < MT + NB; t += 256) {
            if (t < MT) {
                const int c = t, m = m0 + c;
                const int pc = ((c / 32) % TM) * (WM * 32) + (c / (TM * 32)) * 32 + (c % 32);
                bias_a[pc] = (m < p.Cout) ? p.lastcol[(int64_t)m * p.HoWo + o] : 0.0f;
                bias_a[MT + pc] = 0.0f;
            } else {
                const int c = t - MT, n = b0 + c;
                const int pc = ((c / 32) % TN) * (WN * 32) + (c / (TN * 32)) * 32 + (c % 32);
                bias_b[pc] = (n < p.n_vecs) ? xlast[n] : 0.0f;
                bias_b[NB + pc] = 0.0f;
            }
        }
    }
    __syncthreads();
    const int arow = lane >> 5;
    const int acol = wm * 32 + (lane & 31);       // permuted tile columns (see lstore)
    const int bcol = wn * 32 + (lane & 31);
    constexpr int AS = WM * 32, BS = WN * 32;     // distance of a wavefront's sub-tile fragments
    // One chunk; the LDS buffer index is a compile-time constant (the loop below is unrolled by two), so tile addresses are
    // one per-thread base register plus immediates for both buffers.
    // per-thread fragment base of each buffer in its own register: every read of a chunk is then base + a small immediate and
    // pairs of sub-tile fragments merge into ds_read2st64_b32
    const float* afrag[2] = {As + arow * MT + acol, As + KC * MT + arow * MT + acol};
    const float* bfrag[2] = {Bs + arow * NB + bcol, Bs + KC * NB + arow * NB + bcol};
    auto chunk = [&](const int q, auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        const float* a = afrag[buf];
        const float* b = bfrag[buf];
        // fragments of k-step kk+2 are read from LDS while the MFMAs of k-step kk execute (register double buffer)
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; i++) af[0][i] = a[i * AS];
#pragma unroll
        for (int j = 0; j < TN; j++) bf[0][j] = b[j * BS];
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
            const int cur = (kk >> 1) & 1;
            if (kk == LS_AT && q + 1 < n_chunks) lstore(buf ^ 1);
            if (kk == GL_AT && q + 2 < n_chunks && !KN_ABL(p, 2)) LOAD(q + 2);
            if (kk + 2 < KC) {
#pragma unroll
                for (int i = 0; i < TM; i++) af[cur ^ 1][i] = a[(kk + 2) * MT + i * AS];
#pragma unroll
                for (int j = 0; j < TN; j++) bf[cur ^ 1][j] = b[(kk + 2) * NB + j * BS];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the next step's LDS reads ahead of this step's MFMAs
            __builtin_amdgcn_s_setprio(1);       // matrix burst wins arbitration over the co-resident waves' VALU/LDS issue (+1 %, A-B-A-B)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!KN_ABL(p, 0)) __syncthreads();
    };
    {
        int q = 0;
        for (; q + 1 < n_chunks; q += 2) {
            chunk(q, std::integral_constant<int, 0>{});
            chunk(q + 1, std::integral_constant<int, 1>{});
        }
        if (q < n_chunks) chunk(q, std::integral_constant<int, 0>{});
    }

    // bias term LAST, like the reference's row order (its rows stayed in their own LDS area since the prologue)
    if (p.lastcol) {
        float ab[TM], bb[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) ab[i] = bias_a[arow * MT + acol + i * AS];
#pragma unroll
        for (int j = 0; j < TN; j++) bb[j] = bias_b[arow * NB + bcol + j * BS];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[i], bb[j], acc[i][j], 0, 0, 0);
    }
    // ---- epilogue: ReLU, store (the bias term is already in the accumulators) ------------------------------------
    const int m_first = m0 + wm * (TM * 32);
    if (p.wide_store && b0 + NB <= p.n_vecs) {
        constexpr int COLS = TN * 32;
        constexpr int LPR = COLS / 4;
        float* stage = lds + wave * (8 * COLS);       // [8 rows][COLS]  (all tile reads are behind the last barrier)
        float* yp = p.Y + ((int64_t)(m_first + lane / LPR) * p.HoWo + o) * p.ldy + (b0 + wn * COLS + (lane % LPR) * 4);
        const int64_t row_step_bytes = (int64_t)(64 / LPR) * p.HoWo * p.ldy * 4;
        kn_store_tile<TM, TN>(acc, stage, lane, yp, row_step_bytes, m_first, p.Cout, m_first + TM * 32 <= p.Cout, p.relu, p.absmax);
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = b0 + wn * (TN * 32) + j * 32 + (lane & 31);
        if (n >= p.n_vecs) continue;
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m_first + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.Cout) {
                    float v = acc[i][j][r];
                    if (p.relu) v = (v < 0.0f) ? 0.0f : v;
                    p.Y[((int64_t)m * p.HoWo + o) * p.ldy + n] = v;
                }
            }
        }
    }
}


// Work items (output pixel, batch tile, Cout tile) are dealt to the 8 XCDs in contiguous chunks (blockIdx & 7 labels the XCD), Cout tile
// fastest.  TAIL: the items of a chunk beyond `tail_main` -- the last, partial round of resident workgroups -- are computed as four
// quarter tiles (MT/2 x NB/2) by four workgroups instead of one: a lone 128x128 workgroup on an otherwise idle CU runs at ~60 % of
// the matrix pipe for a full tile time while most CUs wait, the quarter tiles finish in about a third of that.  Same K order and MFMA
// shape per output element, so the result is bit-identical to the unsplit launch.
template <int MT, int NB, int KC, int WM, int WN, int FAST, bool TAIL>       // FAST: 0 generic loaders, 1 straight-line loaders with per-thread pointers, 2 with wave-uniform pointers (see convtaps_mfma_tile)
__global__ __launch_bounds__(256, 2) void convtaps_mfma_kernel(ConvArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * KC * MT + 2 * KC * NB + 4 * MAX_FAST_SLOTS + 2 * (MT + NB)];
    const int64_t n_items = (int64_t)p.n_pix * p.n_bt * p.n_mt;
    const int64_t chunk = (n_items + 7) >> 3;
    const int64_t xl = blockIdx.x & 7;
    const int64_t q = blockIdx.x >> 3;
    int64_t item = xl * chunk + q;
    int quad = -1;
    if (TAIL && q >= p.tail_main) {
        const int64_t t = q - p.tail_main;
        item = xl * chunk + p.tail_main + (t >> 2);
        quad = (int)(t & 3);
    }
    if (item >= ((xl + 1) * chunk < n_items ? (xl + 1) * chunk : n_items)) return;
    int mt, pi, bt;
    decode_conv_item(p, item, mt, pi, bt);
    const int o = __builtin_amdgcn_readfirstlane(p.pix_order[pi]);
#ifdef KN_ABLATION
    if (p.stamps && threadIdx.x == 0) {
        p.stamps[4 * (int64_t)blockIdx.x + 0] = (int64_t)__builtin_amdgcn_s_memrealtime();
        p.stamps[4 * (int64_t)blockIdx.x + 2] = (int64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        p.stamps[4 * (int64_t)blockIdx.x + 3] = quad;
    }
#endif
    bool done = false;
    if constexpr (TAIL) {
        if (quad >= 0) {
            convtaps_mfma_tile<MT / 2, NB / 2, KC, WM, WN, FAST>(p, o, mt * MT + (quad & 1) * (MT / 2), bt * NB + (quad >> 1) * (NB / 2), lds);
            done = true;
        }
    }
    if (!done) convtaps_mfma_tile<MT, NB, KC, WM, WN, FAST>(p, o, mt * MT, bt * NB, lds);
#ifdef KN_ABLATION
    if (p.stamps && threadIdx.x == 0) p.stamps[4 * (int64_t)blockIdx.x + 1] = (int64_t)__builtin_amdgcn_s_memrealtime();
#endif
}

// ---- f32 products on the bf16 matrix pipe: three-way split, six of nine cross products (KN_FLAG_BF16X3) ---------------------------------
// An f32 value is the sum of three bf16 values to within 2^-27 of itself (round-to-nearest at each step, residuals exact):
// x = xh + xm + xl, a = ah + am + al with |xm| <= 2^-9 |x|, |xl| <= 2^-18 |x|.  Their product needs nine bf16 x bf16 terms; the three
// smallest (am*xl, al*xm, al*xl: <= 2^-26 of the product together, a quarter of one f32 rounding, of either sign) are dropped, the other six
// are each exact in f32 and are accumulated in f32 by v_mfma_f32_32x32x16_bf16, which runs at 16x the rate of the f32-input MFMA: 6/16 of
// the matrix time of convtaps_mfma_kernel for a result that differs from it by ~1 ulp per term (measured: 2-3x the f32 kernel's own
// distance from the order-preserving result).  This is NOT bit-exact with anything and is only ever selected by the float-key contract
// (KeyedLayer._calibrate: measured against the order-preserving kernel on the layer's own input, 1e-5 * max(1, |y|) with 4x headroom) when
// the caller opted in, or asked for explicitly.
// Tile MT x NB (128 x 128, or 64 x 256 for 64-channel layers), four wavefronts of 64 x 64, K chunk = 16 channels of one slot = ONE MFMA
// k-step.  Taps are split once at create time into three bf16 planes laid out [plane][tap][channel chunk][cout][16 k] so that a chunk's
// MT x 16 tile is one contiguous piece per plane, already in its LDS image (32-byte rows, the two halves swapped on rows with bit 3 set:
// conflict-free ds_read_b128 fragments).  Activations arrive as 16-byte row pieces (k-major, the same two or four loads per thread as the
// f32 kernel: every register-destination load costs the matrix pipe ~29 cycles, eight dword loads per chunk cost a quarter of the launch),
// are split in registers (v_cvt_pk_bf16_f32) and stored k-major too; the MFMA's k-contiguous B fragments come out of LDS through the
// hardware transpose read ds_read_b64_tr_b16 (image (b) of the programming guide: 256-byte rows, 16-byte chunks XORed with the row bits).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int MT, int NB>
struct Bf16x3Lds {
    static constexpr int A_PLANE = MT * 32;                         // bytes: MT rows x 16 bf16
    static constexpr int B_PLANE = ((NB + 127) / 128) * 4096;       // bytes: per 128 columns an image of 16 rows x 256 bytes
    static constexpr int STAGE = 3 * A_PLANE + 3 * B_PLANE;
    static constexpr int BYTES = 2 * STAGE + 2 * MAX_FAST_SLOTS * 8 + 2 * (MT + NB) * 4;
};

template <int MT, int NB, int WM, int WN, bool COEF>
__device__ __forceinline__ void convtaps_bf16x3_tile(const ConvArgs& p, const int o, const int m0, const int b0, char* lds) {
    constexpr int TM = MT / (WM * 32), TN = NB / (WN * 32);
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && MT == WM * TM * 32 && NB == WN * TN * 32, "four wavefronts of (TM x 32) x (TN x 32)");
    constexpr int A_PLANE = Bf16x3Lds<MT, NB>::A_PLANE;
    constexpr int B_PLANE = Bf16x3Lds<MT, NB>::B_PLANE;
    constexpr int STAGE = Bf16x3Lds<MT, NB>::STAGE;
    constexpr int A16 = A_PLANE / 16;                // 16-byte pieces of one tap plane (threads tid < A16 load them)
    constexpr int TPR = NB / 4;                      // threads per activation row
    constexpr int RPP = 256 / TPR;                   // rows per load pass
    constexpr int BL = 16 / RPP;                     // load passes (dwordx4 per thread) per chunk
    static_assert(RPP >= 1 && BL >= 1 && BL <= 4, "activation tile 16 x NB as whole 16-byte passes");
    int64_t* s_da = reinterpret_cast<int64_t*>(lds + 2 * STAGE);
    int64_t* s_db = s_da + MAX_FAST_SLOTS;
    float* bias_a = reinterpret_cast<float*>(s_db + MAX_FAST_SLOTS);      // [2][MT]
    float* bias_b = bias_a + 2 * MT;                                      // [2][NB]

    const int s_beg = __builtin_amdgcn_readfirstlane(p.pix_ptr[o]);
    const int n_slots = __builtin_amdgcn_readfirstlane(p.pix_ptr[o + 1]) - s_beg;
    const int cpk = p.cin_pad / 16;
    const int n_chunks = n_slots * cpk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // per-slot byte deltas to the next chunk's tiles (channel chunk OUTER, slot INNER, like convtaps_mfma_tile)
    const int64_t a_chunk = (int64_t)p.cout_pad * 32;                    // one channel chunk of one tap, one plane
    const int64_t a_tap = (int64_t)cpk * a_chunk;
    const int64_t b_chunk = (int64_t)16 * p.HiWi * p.ldx * 4;
    if (tid < n_slots) {
        const int nxt = (tid + 1 < n_slots) ? tid + 1 : 0;
        const bool wrap = tid + 1 >= n_slots;
        s_da[tid] = (int64_t)(p.slot_tap[s_beg + nxt] - p.slot_tap[s_beg + tid]) * a_tap + (wrap ? a_chunk : 0);
        s_db[tid] = (int64_t)(p.slot_in[s_beg + nxt] - p.slot_in[s_beg + tid]) * p.ldx * 4 + (wrap ? b_chunk : 0);
    }
    // this thread's pieces: A = 16 bytes at tile offset tid * 16 of each plane (tid < A16); B = channel row r0 (+ RPP per pass), 4 batch columns
    const int c4 = tid % TPR, r0 = tid / TPR;
    const char* pa = nullptr;
    const char* pb = nullptr;
    if (n_slots > 0) {
        pa = reinterpret_cast<const char*>(p.tapsB) + (int64_t)p.slot_tap[s_beg] * a_tap + (int64_t)m0 * 32 + (tid < A16 ? tid : 0) * 16;
        pb = reinterpret_cast<const char*>(p.X + b0 + 4 * c4 + ((int64_t)r0 * p.HiWi + p.slot_in[s_beg]) * p.ldx);
    }
    const int64_t plane_b = p.tapsB_plane;                               // bytes between tap planes
    const int64_t pass_b = (int64_t)RPP * p.HiWi * p.ldx * 4;            // bytes between a thread's activation rows
    // LDS image of an activation plane: per 128 columns 16 rows of 256 bytes, 16-byte chunk ch of row k at 16 * (ch ^ (((k & 3) << 2) | ((k >> 2) & 3)))
    auto b_img = [](const int k, const int n) {                          // byte offset of element (k, n), n a multiple of 4
        const int nn = n & 127, ch = nn >> 3;
        return (n >> 7) * 4096 + 256 * k + 16 * (ch ^ (((k & 3) << 2) | ((k >> 2) & 3))) + 8 * ((nn >> 2) & 1);
    };
    int b_lds[BL];
#pragma unroll
    for (int i = 0; i < BL; i++) b_lds[i] = 3 * A_PLANE + b_img(r0 + i * RPP, 4 * c4);
    __syncthreads();
    int64_t da = 0, db = 0;
    int f_slot = 0, st_slot = 0;
    if (n_slots > 0) {
        da = s_da[0];
        db = s_db[0];
    }
    u32x4 ra[3];
    f32x4 rb[BL];
    auto gload = [&]() {
        if (tid < A16) {
#pragma unroll
            for (int pl = 0; pl < 3; pl++) ra[pl] = *reinterpret_cast<const u32x4*>(pa + pl * plane_b);
        }
#pragma unroll
        for (int i = 0; i < BL; i++) rb[i] = *reinterpret_cast<const f32x4*>(pb + i * pass_b);
        pa += da;
        pb += db;
        f_slot = __builtin_amdgcn_readfirstlane((f_slot + 1 == n_slots) ? 0 : f_slot + 1);
        da = s_da[f_slot];
        db = s_db[f_slot];
    };
    // registers -> LDS in pieces that sit between the MFMA groups of a chunk: the three tap planes (plain copies), then one activation row piece
    // per call: round-to-nearest split, two elements at a time (v_cvt_pk_bf16_f32 packs the pair): h = bf16(x), r1 = x - h (exact), m = bf16(r1),
    // r2 = r1 - m (exact), l = bf16(r2): x = h + m + l up to 2^-27 |x|, every part signed and unbiased.  (A truncation split is exact but
    // one-sided: the dropped cross terms then all carry the sign of the product and add up.)
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    float cf = 1.0f;
    auto store_taps = [&](const int buf) {
        char* A = lds + buf * STAGE;
        if (tid < A16) {
#pragma unroll
            for (int pl = 0; pl < 3; pl++) *reinterpret_cast<u32x4*>(A + pl * A_PLANE + tid * 16) = ra[pl];
        }
        if constexpr (COEF) {
            cf = p.slot_coef[s_beg + st_slot];
            st_slot = (st_slot + 1 == n_slots) ? 0 : st_slot + 1;
        }
    };
    auto split_row = [&](const int i, const int buf) {         // (plain scalar arithmetic: packed f32 VALU beside MFMAs is slow)
        u32x2 sh, sm, sl;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            float x0 = rb[i][2 * e], x1 = rb[i][2 * e + 1];
            if constexpr (COEF) {
                x0 = x0 * cf;
                x1 = x1 * cf;
            }
            const unsigned int hp = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2v{x0, x1}, bf16x2));
            const float t0 = x0 - __builtin_bit_cast(float, hp << 16), t1 = x1 - __builtin_bit_cast(float, hp & 0xffff0000u);
            const unsigned int mp = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2v{t0, t1}, bf16x2));
            const float q0 = t0 - __builtin_bit_cast(float, mp << 16), q1 = t1 - __builtin_bit_cast(float, mp & 0xffff0000u);
            sh[e] = hp;
            sm[e] = mp;
            sl[e] = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2v{q0, q1}, bf16x2));
        }
        char* B = lds + buf * STAGE + b_lds[i];
        *reinterpret_cast<u32x2*>(B + 0 * B_PLANE) = sh;
        *reinterpret_cast<u32x2*>(B + 1 * B_PLANE) = sm;
        *reinterpret_cast<u32x2*>(B + 2 * B_PLANE) = sl;
    };
    if (n_chunks > 0) {
        gload();
        store_taps(0);
#pragma unroll
        for (int i = 0; i < BL; i++) split_row(i, 0);
    }
    if (n_chunks > 1) gload();
    if (p.lastcol) {
        const float* xlast = p.X + p.last_in_row * p.ldx;
        for (int t = tid; t < MT + NB; t += 256) {
            if (t < MT) {
                const int m = m0 + t;
                bias_a[t] = (m < p.Cout) ? p.lastcol[(int64_t)m * p.HoWo + o] : 0.0f;
                bias_a[MT + t] = 0.0f;
            } else {
                const int n = b0 + (t - MT);
                bias_b[t - MT] = (n < p.n_vecs) ? xlast[n] : 0.0f;
                bias_b[NB + (t - MT)] = 0.0f;
            }
        }
    }
    __syncthreads();
    // fragment addresses.  A: row = wave tile origin + 32 * sub-tile + (lane & 31), k half = lane >> 5 (swapped on rows with bit 3 set).
    // B (transpose read, per 16-lane group: lane 4 q + pp supplies row q, columns 4 pp .. 4 pp + 3 of a 4 x 16 block and receives column
    // lane & 15 of its four rows): block rows 8 * (lane >> 5) + 4 * s + q, block columns tile origin + 16 * ((lane >> 4) & 1) ..
    const int lr = lane & 31, lh = lane >> 5;
    int a_off[TM], b_off[TN][2];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int r = wm * (TM * 32) + i * 32 + lr;
        a_off[i] = r * 32 + ((lh ^ ((r >> 3) & 1)) * 16);
    }
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
        for (int sb = 0; sb < 2; sb++)
            b_off[j][sb] = 3 * A_PLANE + b_img(8 * lh + 4 * sb + ((lane & 15) >> 2), wn * (TN * 32) + j * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto chunk_fn = [&](const int q, const int buf) {
        const char* base = lds + buf * STAGE;
        bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
#pragma unroll
            for (int i = 0; i < TM; i++) af[pl][i] = *reinterpret_cast<const bf16x8*>(base + pl * A_PLANE + a_off[i]);
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + pl * B_PLANE + b_off[j][0]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + pl * B_PLANE + b_off[j][1]));
                bf[pl][j] = __builtin_bit_cast(bf16x8, s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w});
            }
        }
        // six products, smallest first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); the next chunk's registers -> LDS and the loads of the chunk after
        // it sit between the groups
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        const bool more = q + 1 < n_chunks;                   // wave-uniform
#pragma unroll
        for (int t = 0; t < 6; t++) {
            if (more) {
                if (t == 0 && !KN_ABL(p, 11)) store_taps(buf ^ 1);
                if (t >= 1 && t - 1 < BL && !KN_ABL(p, 8)) split_row(t - 1, buf ^ 1);
            }
            if (t == 5 && q + 2 < n_chunks && !KN_ABL(p, 9)) gload();
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]][i], bf[PB[t]][j], acc[i][j], 0, 0, 0);
        }
        if (!KN_ABL(p, 10)) __syncthreads();
    };
    for (int q = 0; q < n_chunks; q++) chunk_fn(q, q & 1);

    // bias column x homogeneous coordinate: one exact f32 MFMA k-step, last (as in convtaps_mfma_tile)
    if (p.lastcol) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_a[lh * MT + wm * (TM * 32) + i * 32 + lr], bias_b[lh * NB + wn * (TN * 32) + j * 32 + lr], acc[i][j], 0, 0, 0);
    }
    const int m_first = m0 + wm * (TM * 32);
    constexpr int COLS = TN * 32, LPR = COLS / 4;
    float* stage = reinterpret_cast<float*>(lds) + wave * (8 * COLS);
    float* yp = p.Y + ((int64_t)(m_first + lane / LPR) * p.HoWo + o) * p.ldy + (b0 + wn * COLS + (lane % LPR) * 4);
    kn_store_tile<TM, TN>(acc, stage, lane, yp, (int64_t)(64 / LPR) * p.HoWo * p.ldy * 4, m_first, p.Cout, m_first + TM * 32 <= p.Cout, p.relu, p.absmax);
}

// Work items as in convtaps_mfma_kernel (contiguous chunks per XCD, Cout tile fastest).  TAIL: the items of a chunk beyond `tail_main` -- the last,
// partial round of resident workgroups -- run as four quarter tiles each (conv5_x of VGG-16: 196 items per XCD = 2.04 rounds of 96).
template <int MT, int NB, int WM, int WN, bool COEF, bool TAIL>
__global__ __launch_bounds__(256, 2) void convtaps_bf16x3_kernel(ConvArgs p) {
    __shared__ __attribute__((aligned(16))) char lds[Bf16x3Lds<MT, NB>::BYTES];
    const int64_t n_items = (int64_t)p.n_pix * p.n_bt * p.n_mt;
    const int64_t chunk = (n_items + 7) >> 3;
    const int64_t xl = blockIdx.x & 7;
    const int64_t q = blockIdx.x >> 3;
    int64_t item = xl * chunk + q;
    int quad = -1;
    if (TAIL && q >= p.tail_main) {
        const int64_t t = q - p.tail_main;
        item = xl * chunk + p.tail_main + (t >> 2);
        quad = (int)(t & 3);
    }
    if (item >= ((xl + 1) * chunk < n_items ? (xl + 1) * chunk : n_items)) return;
    int mt, pi, bt;
    decode_conv_item(p, item, mt, pi, bt);
    const int o = __builtin_amdgcn_readfirstlane(p.pix_order[pi]);
    if constexpr (TAIL) {
        if (quad >= 0) {
            convtaps_bf16x3_tile<MT / 2, NB / 2, WM, WN, COEF>(p, o, mt * MT + (quad & 1) * (MT / 2), bt * NB + (quad >> 1) * (NB / 2), lds);
            return;
        }
    }
    convtaps_bf16x3_tile<MT, NB, WM, WN, COEF>(p, o, mt * MT, bt * NB, lds);
}

// ---- one-shot small-K path -----------------------------------------------------------------------------------------------
// First layer of an image network (VGG conv1_1: Cin = 3, 9 taps): the whole contraction of one output pixel is
// K = slots*Cin + 1 (bias via the homogeneous row) <= 28 rows, while the output tile is 64 x 256 floats -- the layer is
// bound by its 3.3 GB of output writes, not by MFMA.  So no chunk loop: gather all K activation rows and tap rows into
// LDS once, one barrier, K/2 MFMA steps per wavefront, wide store.  Four workgroups per CU overlap each other's phases.
constexpr int SMALLK_MAX = 28;

__global__ __launch_bounds__(256, 4) void convtaps_smallk_kernel(ConvArgs p) {
    constexpr int MT = 64, NB = 256, TM = 2, TN = 2, KM = SMALLK_MAX;
    __shared__ __attribute__((aligned(16))) float lds[KM * MT + KM * NB];
    float* As = lds;              // [KM][64]
    float* Bs = lds + KM * MT;    // [KM][256]

    const int64_t n_items = (int64_t)p.n_pix * p.n_bt * p.n_mt;
    const int64_t chunk = (n_items + 7) >> 3;
    const int64_t xl = blockIdx.x & 7;
    const int64_t item = xl * chunk + (blockIdx.x >> 3);
    if (item >= ((xl + 1) * chunk < n_items ? (xl + 1) * chunk : n_items)) return;
    int mt, pi, bt;
    decode_conv_item(p, item, mt, pi, bt);
    const int o = p.pix_order[pi];
    const int m0 = mt * MT;
    const int b0 = bt * NB;
    const int s_beg = p.pix_ptr[o];
    const int n_slots = p.pix_ptr[o + 1] - s_beg;
    const int k_conv = n_slots * p.Cin;                 // rows of the contraction proper
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // activation rows: wavefront w gathers rows w, w+4, ... (one 1 KiB row per instruction)
    f32x4 bv[KM / 4];
#pragma unroll
    for (int r = 0; r < KM / 4; r++) {
        const int k = wave + 4 * r;
        bv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k < k_conv) {
            const int sl = k / p.Cin, ci = k - sl * p.Cin;
            const int64_t xrow = (int64_t)ci * p.HiWi + p.slot_in[s_beg + sl];
            bv[r] = *reinterpret_cast<const f32x4*>(p.X + xrow * p.ldx + b0 + lane * 4);
        } else if (k == k_conv && p.lastcol) {
            bv[r] = *reinterpret_cast<const f32x4*>(p.X + p.last_in_row * p.ldx + b0 + lane * 4);
        }
    }
    // tap rows (scaled by the slot coefficient) and the bias row
    constexpr int AV = (KM * MT / 4 + 255) / 256;
    f32x4 av[AV];
#pragma unroll
    for (int r = 0; r < AV; r++) {
        const int idx = tid + 256 * r;
        const int k = idx >> 4, m4 = (idx & 15) * 4;
        av[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (k < k_conv) {
            const int sl = k / p.Cin, ci = k - sl * p.Cin;
            const float cf = p.slot_coef[s_beg + sl];
            av[r] = *reinterpret_cast<const f32x4*>(p.tapsT + ((int64_t)p.slot_tap[s_beg + sl] * p.cin_pad + ci) * p.cout_pad + m0 + m4);
            av[r] = av[r] * cf;
        } else if (k == k_conv && p.lastcol) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int m = m0 + m4 + e;
                av[r][e] = (m < p.Cout) ? p.lastcol[(int64_t)m * p.HoWo + o] : 0.0f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < KM / 4; r++) *reinterpret_cast<f32x4*>(Bs + (wave + 4 * r) * NB + lane * 4) = bv[r];
#pragma unroll
    for (int r = 0; r < AV; r++) {
        const int idx = tid + 256 * r;
        if (idx < KM * MT / 4) *reinterpret_cast<f32x4*>(As + idx * 4) = av[r];
    }
    __syncthreads();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
    const int arow = lane >> 5;
    const int acol = lane & 31;
    const int bcol = wave * (TN * 32) + (lane & 31);
    const int k_steps = (k_conv + (p.lastcol ? 1 : 0) + 1) >> 1;
    for (int st = 0; st < k_steps; st++) {
        float af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) af[i] = As[(2 * st + arow) * MT + acol + i * 32];
#pragma unroll
        for (int j = 0; j < TN; j++) bf[j] = Bs[(2 * st + arow) * NB + bcol + j * 32];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();   // tiles are dead: reuse them as per-wavefront transposition slices

    constexpr int COLS = TN * 32;
    constexpr int LPR = COLS / 4;
    float* stage = lds + wave * (8 * COLS);
    float* yp = p.Y + ((int64_t)(m0 + lane / LPR) * p.HoWo + o) * p.ldy + (b0 + wave * COLS + (lane % LPR) * 4);
    kn_store_tile<TM, TN>(acc, stage, lane, yp, (int64_t)(64 / LPR) * p.HoWo * p.ldy * 4, m0, p.Cout, m0 + TM * 32 <= p.Cout, p.relu, p.absmax);
}

// ---- small-K path, persistent and software-pipelined ------------------------------------------------------------------------
// The one-shot kernel above runs gather -> MFMA -> store once per workgroup; a workgroup's LDS and wave slots stay allocated until
// its last store is acknowledged, so the three phases of a CU's resident workgroups largely ADD (measured: 0.22 ms of gathers,
// 0.31 ms of matrix pipe and 0.5 ms of stores take 0.9 ms together).  Here a workgroup is PERSISTENT: it walks its XCD's share
// of the output pixels, and while the 16 row-segment stores of pixel t drain, the activation rows of pixel t+1 (requested before
// those stores were issued, so the in-order vmcnt wait for them leaves the stores in flight) are written to the other LDS buffer
// and multiplied.  Per-pixel control data comes from a descriptor table built once at create time in PROCESSING order (one
// coalesced record per pixel, fetched two pixels ahead): X row of every contraction row, LDS offset of its tap row, coefficient,
// output pixel, bias values.  The tap matrix of the layer (9 x 3 x 64 floats for VGG conv1_1) stays resident in LDS for the
// workgroup's lifetime, so the A operand is never re-staged: fragments are read straight from the table through the descriptor's
// row offsets.
constexpr int SK_DESC_HDR = 96;       // dwords: [0..27] X row (-1 = none), [31] output pixel, [32..59] tap-row LDS offset, [64..91] coefficient
constexpr int SK_TAB_MAX = 61;        // tap-table rows that fit the LDS budget (+1 zero row, +2 bias rows = 16 KiB)

__global__ __launch_bounds__(256, 2) void convtaps_smallk_pipe_kernel(ConvArgs p) {
    constexpr int MT = 64, NB = 256, TM = 2, TN = 2, KM = SMALLK_MAX, KR = KM / 4;
    __shared__ __attribute__((aligned(16))) float lds[2 * KM * NB + (SK_TAB_MAX + 3) * MT];
    float* Tab = lds + 2 * KM * NB;                               // [tab_rows][64] taps, then a zero row, then two bias rows (double buffered)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int zero_off = p.sk_tab_rows * MT;                      // LDS element offset of the zero row; bias rows follow

    const int64_t n_items = (int64_t)p.n_pix * p.n_bt;            // n_mt == 1 on this path
    const int64_t chunk = (n_items + 7) >> 3;
    const int64_t xl = blockIdx.x & 7;
    const int64_t step = gridDim.x >> 3;
    const int64_t first = xl * chunk + (blockIdx.x >> 3);
    const int64_t end = (xl + 1) * chunk < n_items ? (xl + 1) * chunk : n_items;
    if (first >= end) return;
    const int64_t last = first + ((end - 1 - first) / step) * step;     // last item of this workgroup

    for (int i = tid; i < (p.sk_tab_rows + 3) * MT / 4; i += 256) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < p.sk_tab_rows * MT / 4) v = reinterpret_cast<const f32x4*>(p.tapsT)[i];
        reinterpret_cast<f32x4*>(Tab)[i] = v;
    }

    // per-item control data, one dword per lane and section (every wave keeps its own copy)
    struct Ctl {
        int x, a;
        float c, bias;
        int b0;
    };
    auto fetch_ctl = [&](int64_t item) -> Ctl {
        item = item <= last ? item : last;                        // past the end: re-read the last record (results unused)
        const int pi = (int)(item % p.n_pix);
        const int bt = (int)(item / p.n_pix);
        const int32_t* d = p.sk_desc + (int64_t)pi * p.sk_stride;
        Ctl c;
        c.x = d[lane];                                            // lanes 0..27: X rows; lane 31: output pixel; lanes 32..59: tap-row offsets
        c.a = c.x;
        c.c = reinterpret_cast<const float*>(d)[64 + (lane & 31)];
        c.bias = reinterpret_cast<const float*>(d)[SK_DESC_HDR + lane];
        c.b0 = bt * NB;
        return c;
    };
    f32x4 g[KR];
    auto gather = [&](const Ctl& c) {
        const float* xb = p.X + c.b0 + lane * 4;
#pragma unroll
        for (int r = 0; r < KR; r++) {
            const int xr = __builtin_amdgcn_readlane(c.x, wave + 4 * r);
            g[r] = *reinterpret_cast<const f32x4*>(xb + (int64_t)(xr < 0 ? 0 : xr) * p.ldx);
        }
    };
    auto stage_in = [&](const Ctl& c, const int buf) {            // registers -> LDS: contraction rows (scaled), bias row
        float* B = lds + buf * KM * NB;
#pragma unroll
        for (int r = 0; r < KR; r++) {
            const int k = wave + 4 * r;
            const int xr = __builtin_amdgcn_readlane(c.x, k);
            const float cf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c.c), k));
            f32x4 v = g[r];
            if (!p.unit_coef) v = v * cf;
            if (xr < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};            // padding rows are stored as zeros whatever the dummy load returned
            *reinterpret_cast<f32x4*>(B + k * NB + lane * 4) = v;
        }
        if (wave == 0) Tab[zero_off + (1 + buf) * MT + lane] = c.bias;
    };

    // The descriptor registers are carried around the loop.  hipcc's waitcnt pass merges the loop-entry state (their loads are the
    // youngest requests) with the back-edge state (16 younger stores) and would wait vmcnt(2) at their first use in EVERY iteration,
    // draining the stores.  An empty asm that "reads" them pins that wait to a point where the count is exact on its own path: in
    // the prologue, and behind the stores at the bottom of the loop body (vmcnt(16)); later uses see plain registers.
    auto settle = [](Ctl& c) { asm volatile("" : "+v"(c.x), "+v"(c.a), "+v"(c.c), "+v"(c.bias)); };
    Ctl cur = fetch_ctl(first);
    settle(cur);
    gather(cur);
    Ctl nxt = fetch_ctl(first + step);
    __syncthreads();                                              // tap table complete
    stage_in(cur, 0);
    settle(nxt);
    __syncthreads();

    const int arow = lane >> 5;
    const int acol = lane & 31;
    const int bcol = wave * (TN * 32) + (lane & 31);
    constexpr int COLS = TN * 32, LPR = COLS / 4;
    int buf = 0;
    float am_carry = 0.0f;                                        // kn_spmm_screen: max |y| over this wavefront's pixels, committed once after the loop
    for (int64_t it = first; it <= last; it += step) {
        gather(nxt);                                              // rows of the NEXT pixel: issued before this pixel's stores
        Ctl nn = fetch_ctl(it + 2 * step);
        kn_order();                                               // keep these requests ahead of the matrix phase (the scheduler would sink them to their use)
        // ---- MFMA phase of the current pixel --------------------------------------------------------------------------------
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
        const float* B = lds + buf * KM * NB;
        // tap-row offset of lane's k (even k in lanes 0..31, odd k in lanes 32..63); the bias marker selects this buffer's bias row
        int a_adj = cur.a;
        a_adj = (a_adj == zero_off + MT) ? a_adj + buf * MT : a_adj;
#pragma unroll
        for (int st = 0; st < KM / 2; st++) {
            const int a0 = __builtin_amdgcn_readlane(a_adj, 32 + 2 * st);
            const int a1 = __builtin_amdgcn_readlane(a_adj, 32 + 2 * st + 1);
            const float* ar = Tab + (arow ? a1 : a0) + acol;
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) af[i] = ar[i * 32];
#pragma unroll
            for (int j = 0; j < TN; j++) bf[j] = B[(2 * st + arow) * NB + bcol + j * 32];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                          // every wave is done with this buffer's rows: reuse it as store staging
        // ---- epilogue of the current pixel: 16 row-segment stores per wave, left in flight ----------------------------------
        {
            const int o = __builtin_amdgcn_readlane(cur.x, 31);
            float* stage = lds + buf * KM * NB + wave * (8 * COLS);
            float* yp = p.Y + ((int64_t)(lane / LPR) * p.HoWo + o) * p.ldy + (cur.b0 + wave * COLS + (lane % LPR) * 4);
            kn_store_tile<TM, TN, true>(acc, stage, lane, yp, (int64_t)(64 / LPR) * p.HoWo * p.ldy * 4, 0, p.Cout, true, p.relu, p.absmax, p.absmax ? &am_carry : nullptr);   // Cout == 64 on this path
        }
        // ---- next pixel's rows -> the other buffer (their loads are older than the stores above: the wait leaves those in flight) ----
        stage_in(nxt, buf ^ 1);
        cur = nxt;
        nxt = nn;
        settle(nxt);                                              // wait for the record fetched at the top: 16 younger stores stay in flight
        __syncthreads();
        buf ^= 1;
    }
    if (p.absmax) kn_wave_absmax_commit(am_carry, p.absmax, lane);       // kn_spmm_screen: ONE commit per wavefront, after its last pixel
}

// ---- order-preserving path on the factored operator (KN_FLAG_EXACT) ------------------------------------------------
// The reference applies a Conv2dTiledMatrix as the canonical CSR of its expansion (scipy csr_matrix((v,(r,c))) sorts each
// row by column), so output row (co, o) accumulates, in f32 with separate multiply and add, over
//     ci = 0..Cin-1 (ascending), and inside one ci over the pixel's slots by ascending input pixel,   then the bias column.
// The slot lists are already sorted by input pixel and tapsT[tap][ci][co] keeps the 8 values of a column for 8
// consecutive output channels contiguous, i.e. the factored operator IS a pattern-grouped CSR (kn_csr.hip) whose values
// and column indices can be generated on the fly.  One wavefront = one output pixel x 8 output channels x 64*VEC batch
// columns; values arrive as wave-uniform scalar loads.  No expansion is materialised, so a bit-exact forward of the
// tiled key-nets is possible at any size (VGG-16: 15 G non-zeros would be 120 GB as CSR).
#pragma clang fp contract(off)
template <int VEC>
__global__ __launch_bounds__(256) void convtaps_exact_kernel(ConvArgs p, int n_cob, int64_t n_rb) {
    constexpr int RBX = 8;
    const int64_t n_ct = (p.n_vecs + 64 * VEC - 1) / (64 * VEC);
    const int64_t n_items = n_ct * n_rb;
    const int64_t chunk = (n_items + 7) >> 3;
    const int64_t xl = blockIdx.x & 7;
    const int64_t item = xl * chunk + (blockIdx.x >> 3);
    if (item >= ((xl + 1) * chunk < n_items ? (xl + 1) * chunk : n_items)) return;
    const int64_t ct = item / n_rb;
    const int64_t rb = item - ct * n_rb;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t w = rb * 4 + wave;                       // (pixel index in processing order, channel bundle)
    if (w >= (int64_t)p.n_pix * n_cob) return;
    const int o = p.pix_order[w / n_cob];
    const int co0 = (int)(w % n_cob) * RBX;
    const int s_beg = p.pix_ptr[o];
    const int n_slots = p.pix_ptr[o + 1] - s_beg;
    const int64_t c = ct * (64 * VEC) + (int64_t)lane * VEC;
    const bool active = c < p.n_vecs;
    const float* xc = p.X + (active ? c : 0);

    float acc[RBX][VEC];
#pragma unroll
    for (int r = 0; r < RBX; r++)
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[r][v] = 0.0f;
    // VEC == 1 (narrow batches; operators the pipelined kernel does not take: more than 64 slots per pixel, or several taps on one (output, input) pixel
    // pair -- a doubly-stochastic key has both): one batch column per lane, so the PACKED instructions pair two output channels instead of two columns --
    // v_pk_mul_f32 (a_r, a_r+1) x (x, x), v_pk_add_f32 into the pair's running sums: the same separate IEEE multiply and add per element at half the
    // vector instructions (round 5; the reference's VGG-16 with doubly-stochastic keys runs four layers here by contract).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[RBX / 2];
#pragma unroll
    for (int r = 0; r < RBX / 2; r++) acc2[r] = f32x2{0.0f, 0.0f};

    for (int ci = 0; ci < p.Cin; ci++) {
        const float* xrow = xc + (int64_t)ci * p.HiWi * p.ldx;
        int s = 0;
        while (s < n_slots) {
            const int in = p.slot_in[s_beg + s];
            // value of the expansion's entry (row, col): entries of several taps that hit the same (output, input) pixel pair
            // are ONE stored non-zero, their f32 sum in entry order (scipy sums duplicates; kn_export_csr does the same)
            float ar[RBX];
            {
                const float* a = p.tapsT + ((int64_t)p.slot_tap[s_beg + s] * p.cin_pad + ci) * p.cout_pad + co0;
                const float coef = p.unit_coef ? 1.0f : p.slot_coef[s_beg + s];
#pragma unroll
                for (int r = 0; r < RBX; r++) ar[r] = p.unit_coef ? a[r] : (coef == 1.0f ? a[r] : coef * a[r]);
            }
            s++;
            while (s < n_slots && p.slot_in[s_beg + s] == in) {
                const float* a = p.tapsT + ((int64_t)p.slot_tap[s_beg + s] * p.cin_pad + ci) * p.cout_pad + co0;
                const float coef = p.unit_coef ? 1.0f : p.slot_coef[s_beg + s];
#pragma unroll
                for (int r = 0; r < RBX; r++) {
                    const float t = p.unit_coef ? a[r] : (coef == 1.0f ? a[r] : coef * a[r]);
                    ar[r] = ar[r] + t;
                }
                s++;
            }
            float xv[VEC];
            if constexpr (VEC == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(xrow + (int64_t)in * p.ldx);
                xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            } else {
#pragma unroll
                for (int v = 0; v < VEC; v++) xv[v] = xrow[(int64_t)in * p.ldx + v];
            }
            if constexpr (VEC == 1) {
                const f32x2 xx = {xv[0], xv[0]};
#pragma unroll
                for (int r = 0; r < RBX / 2; r++) {
                    const f32x2 pr = f32x2{ar[2 * r], ar[2 * r + 1]} * xx;
                    acc2[r] = acc2[r] + pr;
                }
            } else {
#pragma unroll
                for (int r = 0; r < RBX; r++) {
#pragma unroll
                    for (int v = 0; v < VEC; v++) {
                        const float pr = ar[r] * xv[v];
                        acc[r][v] = acc[r][v] + pr;
                    }
                }
            }
        }
    }
    if constexpr (VEC == 1) {
#pragma unroll
        for (int r = 0; r < RBX; r++) acc[r][0] = acc2[r / 2][r % 2];
    }
    if (!active) return;
    const float* xlast = p.lastcol ? (p.X + p.last_in_row * p.ldx + c) : nullptr;
#pragma unroll
    for (int r = 0; r < RBX; r++) {
        const int m = co0 + r;
        if (m >= p.Cout) continue;
        const int64_t row = (int64_t)m * p.HoWo + o;
        if (xlast) {
            const float lc = p.lastcol[row];
            if (lc != 0.0f) {                              // the bias entry exists in the reference's row only when it is stored
#pragma unroll
                for (int v = 0; v < VEC; v++) {
                    const float bp = lc * xlast[v];
                    acc[r][v] = acc[r][v] + bp;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            float t = acc[r][v];
            if (p.relu) t = (t < 0.0f) ? 0.0f : t;
            p.Y[row * p.ldy + c + v] = t;
        }
    }
}

// ---- KN_FLAG_EXACT on FILLED-IN operators (round 5) ------------------------------------------------------------------------------------
// A float key whose inverse is dense inside its blocks (the reference's doubly-stochastic keys, test/test_keynet.py:116-129) fills the keyed conv in:
// 500 - 5 400 slots per output pixel instead of 9, and one (output pixel, input pixel) pair is hit by several taps -- ONE stored non-zero of the
// reference's matrix, whose value is the f32 sum of its terms fl(coef * tap) in entry order.  convtaps_exact_kernel above forms those stored values on
// every lane (the value of an entry does not depend on the batch column: 64 lanes repeat the same multiplies and adds) and waits for every load; here
//   * lane l forms the stored value of output channel co0 + (l & 31) only: one 128-byte load of tapsT[tap][ci][co0 ..], one v_mul_f32 by the slot's
//     coefficient, one v_add_f32 per further term of the column -- each value is formed once per 64 batch columns instead of 64 times;
//   * the 32 x 64 rounded products of a stored column come from ONE matrix instruction with a zero accumulator (v_mfma_f32_32x32x1_2b_f32: bit-identical
//     to v_mul_f32, kn_csr_mfma.hip) and 16 packed adds put them on the running sums -- separate rounding of product and sum, the expansion's column order
//     (input channel outer, the pixel's input pixels ascending inner): the reference's arithmetic, bit for bit;
//   * the walk is scalar: the pixel's slot list is a list of 16-byte records {input pixel, tap offset, coefficient, first | last slot of its column}
//     (convtaps_fill_records_kernel, padded to a multiple of 8 with records that change nothing) fetched four at a time by s_load_dwordx4, one batch
//     ahead; the operand loads of a slot (activation row segment + value row) run PF = 8 slots ahead in a register ring with counted vmcnt waits.
// One wavefront = one output pixel x 32 output channels x 64 batch columns; no LDS, no barriers.
// WIDE (64 output channels per wavefront): lane l forms the value of channel co0 + l -- 64 DIFFERENT values per slot for the same four vector instructions and the
// same scalar bookkeeping -- and behind a column's last term v_permlane32_swap hands the matrix pipe channels 0-31 in both halves of one register and channels 32-63 in
// both halves of another: two matrix instructions, twice 16 packed adds.  The slot bookkeeping (what separates this kernel from the roof) is paid once per 64 channels.
// TREG (operators with at most 16 taps: every conv window up to 4 x 4): the lane's value row lives in REGISTERS -- 16 VGPRs hold tapsT[0 .. ntaps)[ci][co0 + (l & 31)],
// reloaded when the walk enters the next input channel -- and a slot picks its tap by the scalar index mode (s_set_gpr_idx_on): no value-row load per slot, half the
// vector-memory instructions (the CU's one address unit serves four SIMDs).
// TILES = 2 (round 6; batches that are whole 128-column tiles): one wavefront = one output pixel x 32 (64) output channels x 128 batch columns.  Lane l loads its two adjacent
// columns c0 + 2 l, c0 + 2 l + 1 of a slot's activation row with ONE 8-byte load; the stored value of a column -- formed once, as before -- multiplies both column tiles: two (four)
// matrix instructions and twice the packed adds per stored column for the SAME slot bookkeeping, which is what separates this kernel from the no-FMA roof.
// Two result blocks alternate, so a block's adds sit behind the next block's matrix instruction instead of waiting for their own.
struct FillRec {
    int32_t in, tapoff;     // tapoff: offset of the tap's plane in tapsT (floats), or the tap index (TREG)
    float coef;
    int32_t flags;          // bit 0: first slot of a stored column, bit 1: last
};

__global__ __launch_bounds__(256) void convtaps_fill_records_kernel(const int32_t* __restrict__ pix_ptr, const int32_t* __restrict__ fill_ptr, const int32_t* __restrict__ slot_in,
                                                                    const int32_t* __restrict__ slot_tap, const float* __restrict__ slot_coef, int unit_coef, int tap_stride /* 1: tap index (TREG) */,
                                                                    int HoWo, FillRec* __restrict__ rec) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= HoWo) return;
    const int lane = threadIdx.x & 63;
    const int s_beg = pix_ptr[o], n = pix_ptr[o + 1] - s_beg;
    const int r_beg = fill_ptr[o], n_pad = fill_ptr[o + 1] - r_beg;
    for (int k = lane; k < n_pad; k += 64) {
        FillRec r = {0, 0, 0.0f, 0};                       // padding: a slot that is neither first nor last adds 0 * tap to a value nobody reads
        if (k < n) {
            const int s = s_beg + k;
            r.in = slot_in[s];
            r.tapoff = slot_tap[s] * tap_stride;
            r.coef = unit_coef ? 1.0f : slot_coef[s];
            r.flags = ((k == 0 || slot_in[s - 1] != r.in) ? 1 : 0) | ((k == n - 1 || slot_in[s + 1] != r.in) ? 2 : 0);
        }
        rec[r_beg + k] = r;
    }
}

#pragma clang fp contract(off)
template <int NT, bool WIDE, int TILES = 1> // TILES: 64-column tiles per wavefront (2: see above); taps held in registers: 16, or 0 = one value-row load per slot (a 9-, 10- or 12-register tap vector measured 119 - 122 VGPRs against 99 with 16: not instantiated)
__global__ __launch_bounds__(256, 2) void convtaps_exact_fill_kernel(ConvArgs p, const int32_t* __restrict__ fill_ptr, const FillRec* __restrict__ rec, int n_cc, int n_ct,
                                                                     int64_t n_wg) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x32 __attribute__((ext_vector_type(32)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    constexpr bool TREG = NT > 0;
    typedef float f32xNT __attribute__((ext_vector_type(NT > 0 ? NT : 1)));
    static_assert(TILES == 1 || TILES == 2, "one or two 64-column tiles per wavefront");
    typedef std::conditional_t<TILES == 2, f32x2, float> xrow_t;    // a lane's activations of one slot: column c0 + lane, or columns c0 + 2 lane and c0 + 2 lane + 1
    constexpr int PF = 8;                                  // slots in flight per wavefront (ring of operand registers) = one unrolled loop body
    constexpr int LPS = TREG ? 1 : 2;                      // vector loads per slot
    // workgroup -> XCD x = blockIdx & 7 owns a contiguous range of the work (pixels in processing order: the 196 pixels of a key block read the same input pixels)
    const int64_t chunk = (n_wg + 7) >> 3;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (wg >= n_wg || (blockIdx.x >> 3) >= chunk) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t wi = wg * 4 + wave;                      // (pixel, channel block, column tile), column tile fastest
    const int per_pix = n_cc * n_ct;
    if (wi >= (int64_t)p.n_pix * per_pix) return;          // (wave-uniform; no barriers in this kernel)
    const int pi = (int)(wi / per_pix);
    const int rem = (int)(wi - (int64_t)pi * per_pix);
    const int o = __builtin_amdgcn_readfirstlane(p.pix_order[pi]);
    constexpr int CH = WIDE ? 64 : 32;                     // output channels per wavefront
    constexpr int NH = WIDE ? 2 : 1;                       // 32-channel halves
    const int co0 = __builtin_amdgcn_readfirstlane((rem / n_ct) * CH);
    const int64_t c0 = (int64_t)(rem % n_ct) * (64 * TILES);
    const int64_t c = c0 + TILES * lane;
    const bool active = c < p.n_vecs;                      // (TILES == 2: the launcher guarantees n_vecs % 128 == 0 -- every lane is active)
    const int r_beg = __builtin_amdgcn_readfirstlane(fill_ptr[o]);
    const int n_pad = __builtin_amdgcn_readfirstlane(fill_ptr[o + 1]) - r_beg;         // multiple of 8

    f32x2 acc[NH][TILES][16];
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
        for (int t = 0; t < TILES; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[h][t][q] = f32x2{0.0f, 0.0f};

    if (n_pad > 0) {
        auto uni = [](const uint64_t v) {
            return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
        };
        const uint64_t rbase = uni(reinterpret_cast<uint64_t>(rec + r_beg));
        const uint64_t xbase = uni(reinterpret_cast<uint64_t>(p.X));
        const uint64_t abase = uni(reinterpret_cast<uint64_t>(p.tapsT + co0));
        const uint32_t b_off = 4u * (uint32_t)(active ? c : c0);                 // lane's byte offset inside an activation row (inactive lanes: a valid address, result unused)
        const uint32_t a_off = 4u * (uint32_t)(WIDE ? lane : (lane & 31));       // lane's byte offset inside the 32 (64) values of a value row
        const uint32_t ldx_b = 4u * (uint32_t)p.ldx;                             // (HiWi * ldx * 4 < 2^32, HiWi and 4 * ldx < 2^24: checked by the launcher)
        uint32_t ldx_v = ldx_b;                                                  // the same in a vector register: the row offset is ONE v_mad_u32_u24 (a scalar multiply + a vector add cost an issue slot of the scalar unit more)
        asm volatile("" : "+v"(ldx_v));
        const uint64_t ci_step = (uint64_t)(uint32_t)p.HiWi * ldx_b;             // bytes between the planes of two input channels
        const int n_bat = n_pad >> 2;                      // record batches (4 slots) per input channel; even
        // fetch cursor (wave-uniform): batch index inside the pixel's list, input channel (as activation-row and value-row offsets)
        uint32_t bq_off = 0;                               // byte offset (inside the pixel's list) of the batch whose records are loaded next
        const uint32_t list_bytes = 16u * (uint32_t)n_pad;
        int ci_left = p.Cin - 1;                           // input channels behind the one the fetch cursor is in
        int bf = 0;                                        // batch the vector fetches read next
        uint64_t x_ci = xbase;                             // activation plane of the fetch cursor's input channel
        uint32_t ci_a = 0;                                 // ci * cout_pad of the fetch cursor
        i32x16 R0, R1;                                     // two record batches (four 16-byte records each)
        auto load_batch = [&](i32x16& R) {                 // ONE s_load of the four records of the next batch (scalar-offset form), advance (past the pixel's last batch: its first again)
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(R) : "s"(rbase), "s"(bq_off));
            bq_off = (bq_off + 64u == list_bytes) ? 0u : bq_off + 64u;
        };
        auto batch_landed = [&](i32x16& R) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(R)); };
        auto rec_of = [](const i32x16& R, const int k) { return i32x4{R[4 * k], R[4 * k + 1], R[4 * k + 2], R[4 * k + 3]}; };
        float xa[PF], cfr[PF];
        xrow_t xb[PF];
        int flr[PF], tpr[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) {
            xa[q] = 0.0f;
            xb[q] = xrow_t{};
            cfr[q] = 0.0f;
            flr[q] = tpr[q] = 0;
        }
        // the vector loads of one slot into ring position q (activation row segment; value row unless TREG); its coefficient, flags and tap ride along in scalar registers
        auto fetch = [&](xrow_t& rb, float& ra, float& cf, int& fl, int& tp, const i32x4 r) {
            const uint32_t xoff = __umul24((uint32_t)r.x, ldx_v) + b_off;         // v_mad_u32_u24: row offset + lane offset in one vector instruction, nothing on the scalar unit
            if constexpr (TILES == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(rb) : "v"(xoff), "s"(x_ci));
            else asm volatile("global_load_dword %0, %1, %2" : "=&v"(rb) : "v"(xoff), "s"(x_ci));
            if constexpr (!TREG) {
                const uint64_t aaddr = abase + 4ull * (uint64_t)(ci_a + (uint32_t)r.y);
                asm volatile("global_load_dword %0, %1, %2" : "=&v"(ra) : "v"(a_off), "s"(aaddr));
            }
            tp = r.y;
            const int cbits = r.z;                         // (through a scalar: __builtin_bit_cast on a vector ELEMENT reads element 0 with this compiler)
            cf = __builtin_bit_cast(float, cbits);
            fl = r.w;
        };
        auto batch_done = [&]() {                          // the fetch cursor leaves a batch: next batch, next input channel behind the pixel's last one (past the end: the last channel again)
            if (++bf == n_bat) {
                bf = 0;
                if (ci_left > 0) {
                    ci_left--;
                    x_ci += ci_step;
                    ci_a += (uint32_t)p.cout_pad;
                }
            }
        };
        auto landed = [&](xrow_t& rb, float& ra) {           // (one wait per slot: a wait that names two ring positions made the compiler copy one of them ahead of it -- a register still owned by its load)
            if constexpr (TREG) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rb) : "n"(LPS * (PF - 1)));
            else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rb), "+v"(ra) : "n"(LPS * (PF - 1)));
        };
        // prologue: slots 0 .. 7 in flight, the records of slots 8 .. 11 on their way
        load_batch(R0);
        batch_landed(R0);
        load_batch(R1);
        fetch(xb[0], xa[0], cfr[0], flr[0], tpr[0], rec_of(R0, 0));
        fetch(xb[1], xa[1], cfr[1], flr[1], tpr[1], rec_of(R0, 1));
        fetch(xb[2], xa[2], cfr[2], flr[2], tpr[2], rec_of(R0, 2));
        fetch(xb[3], xa[3], cfr[3], flr[3], tpr[3], rec_of(R0, 3));
        batch_done();
        batch_landed(R1);
        load_batch(R0);
        fetch(xb[4], xa[4], cfr[4], flr[4], tpr[4], rec_of(R1, 0));
        fetch(xb[5], xa[5], cfr[5], flr[5], tpr[5], rec_of(R1, 1));
        fetch(xb[6], xa[6], cfr[6], flr[6], tpr[6], rec_of(R1, 2));
        fetch(xb[7], xa[7], cfr[7], flr[7], tpr[7], rec_of(R1, 3));
        batch_done();
        batch_landed(R0);                                  // (no scalar load is in flight across the loop's back edge: the compiler may copy a loop-carried register tuple there)

        f32x32 d;                                          // the products of the last finished column, not yet on the running sums
#pragma unroll
        for (int q = 0; q < 32; q++) d[q] = 0.0f;
        f32x32 d2;                                         // TILES == 2: the second result block (two matrix instructions in flight: a block's adds sit behind the NEXT block's matrix instruction)
#pragma unroll
        for (int q = 0; q < 32; q++) d2[q] = 0.0f;
        f32x32 zero;
#pragma unroll
        for (int q = 0; q < 32; q++) zero[q] = 0.0f;
        float arun = 0.0f;                                 // stored value of the column being formed (this lane's output channel)
        // one slot: its term joins the column's value; behind the column's last term the PREVIOUS column's products go onto the running sums (16 packed adds) and this
        // column's products are formed (one matrix instruction, zero accumulator) -- the adds of column k sit behind the matrix instruction of column k by at least one slot
        f32xNT At;                                         // TREG: this lane's value row, one register per tap, for the input channel the walk is in
#pragma unroll
        for (int q = 0; q < (NT > 0 ? NT : 1); q++) At[q] = 0.0f;
        const int nb8 = n_pad >> 3;                        // loop bodies per input channel
        int body_left = 0;                                 // bodies until the walk enters the next input channel
        const float* a_ci = p.tapsT + co0 + (WIDE ? lane : (lane & 31));   // TREG: tapsT[0][ci][this lane's output channel] of the channel the walk enters next
        auto consume = [&](xrow_t& rb, float& ra, const float cf, const int fl, const int tp) {
            landed(rb, ra);
            const float av = TREG ? At[tp] : ra;           // (TREG: scalar index mode, no memory access)
            const float t = cf * av;                       // fl(coef * tap): the term as the reference stores it (coef == 1: the tap itself)
            arun = arun + t;                               // (the column's first term joins +0.0: the same value bit for bit but for the sign of a zero, which no sum that starts at +0.0 can show)
            if (fl & 2) {
                auto add_from = [&](f32x2 (&a)[16], const f32x32& dd) {      // the products in dd onto 32 x 64 running sums
                    a[0] = a[0] + f32x2{dd[0], dd[1]};     // (compiler-visible: the hazard recognizer spaces this first reader of the matrix instruction's result; the rest follow it)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 1; q < 16; q++) {
                        const f32x2 p2 = {dd[2 * q], dd[2 * q + 1]};
                        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[q]) : "v"(p2));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto add_d = [&](f32x2 (&a)[16]) { add_from(a, d); };
                if constexpr (TILES == 2) {
                    // two column tiles: the column's stored value (formed ONCE) multiplies both -- 2 (4) matrix instructions per stored column for the same slot bookkeeping.
                    // Result blocks alternate between d and d2: a block's 16 adds are issued behind the matrix instruction of the NEXT block, the last block's stay pending.
                    float a_lo = arun, a_hi = arun;
                    if constexpr (WIDE) {
                        const unsigned ab = __builtin_bit_cast(unsigned, arun);
                        const auto sw = __builtin_amdgcn_permlane32_swap(ab, ab, false, false);
                        const unsigned lo_b = sw[0], hi_b = sw[1];
                        a_lo = __builtin_bit_cast(float, lo_b);
                        a_hi = __builtin_bit_cast(float, hi_b);
                    }
                    const float rb0 = rb.x, rb1 = rb.y;
                    add_from(acc[NH - 1][1], d2);          // pending: the previous column's last block
                    d = __builtin_amdgcn_mfma_f32_32x32x1f32(a_lo, rb0, zero, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f32_32x32x1f32(a_lo, rb1, zero, 0, 0, 0);
                    add_from(acc[0][0], d);
                    if constexpr (WIDE) {
                        d = __builtin_amdgcn_mfma_f32_32x32x1f32(a_hi, rb0, zero, 0, 0, 0);
                        add_from(acc[0][1], d2);
                        d2 = __builtin_amdgcn_mfma_f32_32x32x1f32(a_hi, rb1, zero, 0, 0, 0);
                        add_from(acc[1][0], d);
                    }
                } else {
                add_d(acc[NH - 1][0]);                     // pending: the previous column's (last) block
                if constexpr (WIDE) {
                    const unsigned ab = __builtin_bit_cast(unsigned, arun);
                    const auto sw = __builtin_amdgcn_permlane32_swap(ab, ab, false, false);     // [0]: lanes 0-31 of arun in both halves, [1]: lanes 32-63 in both halves
                    const unsigned lo_b = sw[0], hi_b = sw[1];
                    d = __builtin_amdgcn_mfma_f32_32x32x1f32(__builtin_bit_cast(float, lo_b), rb, zero, 0, 0, 0);
                    add_d(acc[0][0]);                      // channels 0-31: behind their own matrix instruction (the compiler spaces it; other wavefronts fill the gap)
                    d = __builtin_amdgcn_mfma_f32_32x32x1f32(__builtin_bit_cast(float, hi_b), rb, zero, 0, 0, 0);
                } else {
                    d = __builtin_amdgcn_mfma_f32_32x32x1f32(arun, rb, zero, 0, 0, 0);
                }
                }
                arun = 0.0f;
            }
        };
        const int n_it = (n_pad >> 3) * p.Cin;
        for (int it = 0; it < n_it; it++) {
            if constexpr (TREG) {
                if (body_left == 0) {                      // the walk enters an input channel: its value row into the registers (the compiler's own loads and waits)
                    body_left = nb8;
                    const int tap_stride = p.cin_pad * p.cout_pad;
#pragma unroll
                    for (int q = 0; q < NT; q++)
                        if (q < p.ntaps) At[q] = a_ci[(int64_t)q * tap_stride];
                    a_ci += p.cout_pad;
                    asm volatile("" : "+v"(At));           // the compiler's wait for these loads stays inside this block (a loop-carried pending load would put vmcnt(0) in front of every body)
                }
                body_left--;
            }
            // slots 0 .. 3 of this body; their ring positions are refilled from batch R0 (landed: loaded half a body ago), R1's next batch leaves now
            load_batch(R1);
            consume(xb[0], xa[0], cfr[0], flr[0], tpr[0]);
            fetch(xb[0], xa[0], cfr[0], flr[0], tpr[0], rec_of(R0, 0));
            consume(xb[1], xa[1], cfr[1], flr[1], tpr[1]);
            fetch(xb[1], xa[1], cfr[1], flr[1], tpr[1], rec_of(R0, 1));
            consume(xb[2], xa[2], cfr[2], flr[2], tpr[2]);
            fetch(xb[2], xa[2], cfr[2], flr[2], tpr[2], rec_of(R0, 2));
            consume(xb[3], xa[3], cfr[3], flr[3], tpr[3]);
            fetch(xb[3], xa[3], cfr[3], flr[3], tpr[3], rec_of(R0, 3));
            batch_done();
            batch_landed(R1);
            load_batch(R0);
            consume(xb[4], xa[4], cfr[4], flr[4], tpr[4]);
            fetch(xb[4], xa[4], cfr[4], flr[4], tpr[4], rec_of(R1, 0));
            consume(xb[5], xa[5], cfr[5], flr[5], tpr[5]);
            fetch(xb[5], xa[5], cfr[5], flr[5], tpr[5], rec_of(R1, 1));
            consume(xb[6], xa[6], cfr[6], flr[6], tpr[6]);
            fetch(xb[6], xa[6], cfr[6], flr[6], tpr[6], rec_of(R1, 2));
            consume(xb[7], xa[7], cfr[7], flr[7], tpr[7]);
            fetch(xb[7], xa[7], cfr[7], flr[7], tpr[7], rec_of(R1, 3));
            batch_done();
            batch_landed(R0);
        }
        // the products still pending; everything in flight lands (re-loads of valid rows, never used)
        if constexpr (TILES == 2) {
#pragma unroll
            for (int q = 0; q < 16; q++) acc[NH - 1][1][q] = acc[NH - 1][1][q] + f32x2{d2[2 * q], d2[2 * q + 1]};
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) acc[NH - 1][0][q] = acc[NH - 1][0][q] + f32x2{d[2 * q], d[2 * q + 1]};
        }
        asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
        for (int q = 0; q < PF; q++) asm volatile("" : "+v"(xb[q]), "+v"(xa[q]));
    }
    // epilogue: bias column last (separate multiply and add, skipped where the stored entry is absent), ReLU, store.
    // D layout: register 16 * blk + r of lane l = (channel 8 * (r / 4) + 4 * (l / 32) + r % 4, B lane 32 * blk + l % 32); B lane j carries column c0 + j (one tile) or
    // columns c0 + 2 j + t of tile t (two tiles: a lane's two columns are adjacent in memory).
    const int half = lane >> 5;
    const int64_t colo = c0 + TILES * (lane & 31);
    float xl[2][TILES];
#pragma unroll
    for (int blk = 0; blk < 2; blk++)
#pragma unroll
        for (int t = 0; t < TILES; t++) {
            xl[blk][t] = 0.0f;
            if (p.lastcol && colo + TILES * 32 * blk + t < p.n_vecs) xl[blk][t] = p.X[p.last_in_row * p.ldx + colo + TILES * 32 * blk + t];
        }
#pragma unroll
    for (int h = 0; h < NH; h++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = co0 + 32 * h + 8 * (r / 4) + 4 * half + (r % 4);
            if (m < p.Cout) {
                const int64_t row = (int64_t)m * p.HoWo + o;
                const float lc = p.lastcol ? p.lastcol[row] : 0.0f;
#pragma unroll
                for (int blk = 0; blk < 2; blk++) {
                    float vv[TILES];
#pragma unroll
                    for (int t = 0; t < TILES; t++) {
                        float v = acc[h][t][(16 * blk + r) / 2][(16 * blk + r) % 2];
                        if (lc != 0.0f) {
                            const float bp = lc * xl[blk][t];
                            v = v + bp;
                        }
                        if (p.relu) v = (v < 0.0f) ? 0.0f : v;
                        vv[t] = v;
                    }
                    const int64_t cc = colo + TILES * 32 * blk;
                    if constexpr (TILES == 2) {
                        *reinterpret_cast<f32x2*>(p.Y + row * p.ldy + cc) = f32x2{vv[0], vv[1]};       // (n_vecs % 128 == 0, ldy even, Y 8-byte aligned: the launcher's conditions)
                    } else {
                        if (cc < p.n_vecs) p.Y[row * p.ldy + cc] = vv[0];
                    }
                }
            }
        }
    }
}

// Software-pipelined instantiation of the order-preserving path for the common operator shape: unit coefficients
// (identity / permutation keys) and no (output, input) pixel pair hit twice, so the contraction is a plain double loop
// "ci ascending, slot ascending" with ONE stored value per step.  The slot table lives in two VGPRs (lane s = slot s,
// read back with v_readlane), step q+1's activation row (16 B per lane) and its RBX tap values (one s_load) are issued
// before step q's 2*RBX packed multiplies / adds, so neither latency is exposed -- the generic kernel above waits
// vmcnt(0) on every step.  RBX = 16 output channels per wavefront halves the activation gathers per MAC.
// VEC = batch columns per lane: 4 (a wavefront covers 256 columns, 16-byte row loads) or 2 (128 columns, 8-byte loads: the half-batch windows of the
// overlapped forward at 256 images, and batches that fill 128-column tiles better than 256-column ones).  Same instruction count per MAC either way
// (one packed multiply and one packed add per output channel and column pair); what VEC = 2 halves is the work per wavefront -- finer balance.
template <int RBX, bool COEF = false, int XD = 2, int VEC = 4>       // XD = activation rows in flight per wavefront (2, or 4 for wide batches: see the launcher)
__global__ __launch_bounds__(256) void convtaps_exact_pipe_kernel(ConvArgs p, int n_cob, int64_t n_rb) {
    static_assert(VEC == 4 || VEC == 2, "two or four batch columns per lane");
    constexpr int NP = VEC / 2;                                           // column pairs per lane
    typedef float xrow_t __attribute__((ext_vector_type(VEC)));
    const int64_t n_ct = (p.n_vecs + 64 * VEC - 1) / (64 * VEC);
    const int64_t n_items = n_ct * n_rb;
    const int64_t chunk = (n_items + 7) >> 3;
    const int64_t xl = blockIdx.x & 7;
    const int64_t item = xl * chunk + (blockIdx.x >> 3);
    if (item >= ((xl + 1) * chunk < n_items ? (xl + 1) * chunk : n_items)) return;
    const int64_t ct = item / n_rb;
    const int64_t rb = item - ct * n_rb;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t w = rb * 4 + wave;
    if (w >= (int64_t)p.n_pix * n_cob) return;
    // wave-uniform by construction; pinned to SGPRs here so that the whole step-advance logic below stays on the scalar ALU (a value
    // that comes out of a vector-memory load counts as divergent for the compiler, and would drag the loop counters into VGPRs)
    // w -> (channel-bundle group, pixel, bundle inside the group), bundle fastest: the bundles of a pixel that run side by side share its gathered
    // activation rows in L2.  G = p.tail_main bundles per group (0 = all n_cob: one group).  With G = n_cob / 8 every XCD's contiguous share of the
    // items is ONE group over all pixels, so its slice of the taps (1/8 of 9.4 MB on the 512-channel layers) stays L2-resident for the scalar loads.
    const int G = p.tail_main > 0 ? p.tail_main : n_cob;
    const int64_t per_group = (int64_t)p.n_pix * G;
    const int cg = (int)(w / per_group);
    const int64_t rem = w - (int64_t)cg * per_group;
    const int o = __builtin_amdgcn_readfirstlane(p.pix_order[rem / G]);
    const int co0 = __builtin_amdgcn_readfirstlane((cg * G + (int)(rem % G)) * RBX);
    const int s_beg = __builtin_amdgcn_readfirstlane(p.pix_ptr[o]);
    const int n_slots = __builtin_amdgcn_readfirstlane(p.pix_ptr[o + 1]) - s_beg;
    const int64_t c = ct * (64 * VEC) + (int64_t)lane * VEC;
    const bool active = c < p.n_vecs;
    const uint32_t lane_off_bytes = 4u * (uint32_t)(active ? c : 0);      // byte offset: (uniform base) + zext(VGPR) selects the saddr load form

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc[RBX][NP];                                                   // [output channel][columns 0-1 | 2-3]
#pragma unroll
    for (int r = 0; r < RBX; r++)
#pragma unroll
        for (int h = 0; h < NP; h++) acc[r][h] = f32x2{0.f, 0.f};

    if (n_slots > 0) {
        // lane s: element offsets of slot s (32-bit: the launcher checks the ranges)
        int my_xoff = 0, my_aoff = 0;
        float my_coef = 1.0f;                                     // COEF: lane s = coefficient of slot s
        if (lane < n_slots) {
            my_xoff = p.slot_in[s_beg + lane] * (int)p.ldx;
            my_aoff = p.slot_tap[s_beg + lane] * (p.cin_pad * p.cout_pad);
            if constexpr (COEF) my_coef = p.slot_coef[s_beg + lane];
        }
        const int ch_x = __builtin_amdgcn_readfirstlane(p.HiWi * (int)p.ldx);      // one input channel of X
        const float* a_base = p.tapsT + co0;
        const int n_q = n_slots * p.Cin;
        // two wave-uniform cursors over the steps (slot inner, channel outer): the activation rows may run further ahead than the tap values
        int s = 0, ci_x = 0, q_next = 0;                         // cursor of the activation-row requests
        int s_a = 0, ci_a = 0, q_a = 0;                          // cursor of the tap-value requests
        // Operand fetch of one step, written as inline asm so that both addresses stay scalar: the activation row is a saddr-form vector
        // load (wave-uniform 64-bit base in SGPRs + the lane's constant byte offset: no per-step vector address arithmetic), the step's
        // RBX tap values one s_load into an SGPR tuple that the packed multiplies read directly.  The compiler's waitcnt bookkeeping does
        // not see these loads; the waits are written out below.  Each fetch advances its cursor, branch-free (selects on wave-uniform
        // values, all on the scalar ALU); past the end the last step's operands are fetched again, so the waits stay counted ones.
        typedef float taps_t __attribute__((ext_vector_type(RBX)));
        auto fetch_x = [&](xrow_t& xr) {
            const int xo = __builtin_amdgcn_readlane(my_xoff, s) + ci_x;
            const uint64_t xaddr = reinterpret_cast<uint64_t>(p.X + xo);
            if constexpr (VEC == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(xr) : "v"(lane_off_bytes), "s"(xaddr));
            else asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(xr) : "v"(lane_off_bytes), "s"(xaddr));
            q_next++;
            const bool more = q_next < n_q;
            const bool wrap = (s + 1 == n_slots);
            s = more ? (wrap ? 0 : s + 1) : s;
            ci_x = __builtin_amdgcn_readfirstlane(ci_x + ((more && wrap) ? ch_x : 0));
        };
        auto fetch_a = [&](taps_t& ar, float& cf) {
            const int ao = __builtin_amdgcn_readlane(my_aoff, s_a) + ci_a;
            const uint64_t aaddr = reinterpret_cast<uint64_t>(a_base + ao);
            if constexpr (RBX == 16) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(ar) : "s"(aaddr));
            else asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(ar) : "s"(aaddr));
            if constexpr (COEF) cf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_coef), s_a));
            q_a++;
            const bool more = q_a < n_q;
            const bool wrap = (s_a + 1 == n_slots);
            s_a = more ? (wrap ? 0 : s_a + 1) : s_a;
            ci_a = ci_a + ((more && wrap) ? p.cout_pad : 0);
        };
        auto advance = [&]() {};                                 // (the fetches advance their own cursors)
        // the "+" operands make the multiplies that follow depend on the wait
        auto taps_landed = [&](taps_t& ar) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ar)); };
        auto row_landed = [&](xrow_t& xr, auto younger) {                 // vector loads return in order
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xr) : "n"(decltype(younger)::value));
        };
        // acc[r] += x * av[r] (separate IEEE multiply and add).  The packed multiply reads an aligned SGPR pair and broadcasts its low
        // or high half with op_sel, so one pair serves two output channels.  The instruction order is pinned (volatile asm): the four
        // multiplies of channel pair k are followed by the four adds of pair k-1, so no add waits on the multiply right in front of it
        // (left to itself the scheduler serialises "mul, add, mul, add" through one temporary in the second half of the loop body).
        // mul4: the four products of ONE channel pair (VEC = 4: two column pairs x two channels) or of TWO channel pairs (VEC = 2: one column pair x
        // four channels); add4 adds them to their running sums.  `r` is the first of the 2 (VEC = 4) or 4 (VEC = 2) channels a call covers.
        auto mul4 = [&](const f32x2& xlo, const f32x2& xhi, const f32x2& a2, const f32x2& b2, f32x2 (&pr)[4]) {
            if constexpr (VEC == 4) {
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr[0]) : "v"(xlo), "s"(a2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr[1]) : "v"(xhi), "s"(a2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[2]) : "v"(xlo), "s"(a2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[3]) : "v"(xhi), "s"(a2));
            } else {
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr[0]) : "v"(xlo), "s"(a2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[1]) : "v"(xlo), "s"(a2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr[2]) : "v"(xlo), "s"(b2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[3]) : "v"(xlo), "s"(b2));
            }
        };
        auto add4 = [&](int r, const f32x2 (&pr)[4]) {
            if constexpr (VEC == 4) {
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r][0]) : "v"(pr[0]));
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r][NP - 1]) : "v"(pr[1]));
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r + 1][0]) : "v"(pr[2]));
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r + 1][NP - 1]) : "v"(pr[3]));
            } else {
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r][0]) : "v"(pr[0]));
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r + 1][0]) : "v"(pr[1]));
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r + 2][0]) : "v"(pr[2]));
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[r + 3][0]) : "v"(pr[3]));
            }
        };
        // COEF (float keys whose entries carry a coefficient): the stored non-zero of the reference is fl(coef * tap) -- one more packed
        // multiply per channel pair, tap pair (SGPR) x the step's coefficient (broadcast from a VGPR pair's low half), and the products
        // are then formed from that VGPR pair.
        auto mul4v = [&](const f32x2& xlo, const f32x2& xhi, const f32x2& a2, const f32x2& b2, f32x2 (&pr)[4]) {
            if constexpr (VEC == 4) {
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr[0]) : "v"(xlo), "v"(a2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr[1]) : "v"(xhi), "v"(a2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[2]) : "v"(xlo), "v"(a2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[3]) : "v"(xhi), "v"(a2));
            } else {
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr[0]) : "v"(xlo), "v"(a2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[1]) : "v"(xlo), "v"(a2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(pr[2]) : "v"(xlo), "v"(b2));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(pr[3]) : "v"(xlo), "v"(b2));
            }
        };
        auto mac = [&](const xrow_t& xv, const taps_t& av, const float cf) {
            const f32x2 xlo = {xv[0], xv[1]}, xhi = {xv[VEC - 2], xv[VEC - 1]};
            f32x2 pa[4], pb[4];
            constexpr int CH = (VEC == 4) ? 2 : 4;                       // channels per mul4 / add4
            if constexpr (COEF) {
                const f32x2 cf2 = {cf, cf};
                f32x2 sa, sb, sc, sd;
#pragma unroll
                for (int r = 0; r < RBX; r += 2 * CH) {
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(sa) : "s"(f32x2{av[r], av[r + 1]}), "v"(cf2));
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(sb) : "s"(f32x2{av[r + 2], av[r + 3]}), "v"(cf2));
                    if constexpr (VEC == 2) {
                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(sc) : "s"(f32x2{av[r + 4], av[r + 5]}), "v"(cf2));
                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(sd) : "s"(f32x2{av[r + 6], av[r + 7]}), "v"(cf2));
                    }
                    if (r > 0) add4(r - CH, pb);
                    if constexpr (VEC == 4) {
                        mul4v(xlo, xhi, sa, sa, pa);
                        mul4v(xlo, xhi, sb, sb, pb);
                    } else {
                        mul4v(xlo, xhi, sa, sb, pa);
                        mul4v(xlo, xhi, sc, sd, pb);
                    }
                    add4(r, pa);
                }
                add4(RBX - CH, pb);
                return;
            }
#pragma unroll
            for (int r = 0; r < RBX; r += 2 * CH) {
                if constexpr (VEC == 4) mul4(xlo, xhi, f32x2{av[r], av[r + 1]}, f32x2{av[r], av[r + 1]}, pa);
                else mul4(xlo, xhi, f32x2{av[r], av[r + 1]}, f32x2{av[r + 2], av[r + 3]}, pa);
                if (r > 0) add4(r - CH, pb);
                if constexpr (VEC == 4) mul4(xlo, xhi, f32x2{av[r + 2], av[r + 3]}, f32x2{av[r + 2], av[r + 3]}, pb);
                else mul4(xlo, xhi, f32x2{av[r + 4], av[r + 5]}, f32x2{av[r + 6], av[r + 7]}, pb);
                add4(r, pa);
            }
            add4(RBX - CH, pb);
        };
        // Two steps per trip, operands double-buffered.  Step q+1's tap load is issued right after step q's taps have landed (scalar
        // loads return out of order, so lgkmcnt can only be waited to zero) and its activation row right after that; both have step q's
        // 2*RBX packed multiplies / adds to land.  kn_order() keeps the compiler from moving the arithmetic across the fetches.
        if constexpr (XD == 4) {
            // FOUR activation rows in flight (wide batches: a gathered row of a [D, 4096] block misses L2 -- the grouped CSR pipeline's finding),
            // tap values one step ahead as before.  One step: this step's taps have landed -> request the next step's -> wait for this step's
            // row (three younger rows stay in flight) -> arithmetic -> request the row four steps ahead into the register just released.
            xrow_t x0, x1, x2, x3;
            taps_t a0, a1;
            float c0 = 1.0f, c1 = 1.0f;
            fetch_x(x0);
            fetch_x(x1);
            fetch_x(x2);
            fetch_x(x3);
            fetch_a(a0, c0);
            auto step4 = [&](const int q, xrow_t& xq, taps_t& aq, float& cq, taps_t& an, float& cn) {
                taps_landed(aq);
                fetch_a(an, cn);
                row_landed(xq, std::integral_constant<int, 3>());
                kn_order();
                if (q < n_q) mac(xq, aq, cq);
                kn_order();
                fetch_x(xq);
            };
            for (int q = 0; q < n_q; q += 4) {
                step4(q, x0, a0, c0, a1, c1);
                step4(q + 1, x1, a1, c1, a0, c0);
                step4(q + 2, x2, a0, c0, a1, c1);
                step4(q + 3, x3, a1, c1, a0, c0);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(a0));
        } else {
        xrow_t x0, x1;
        taps_t a0, a1;
        float c0 = 1.0f, c1 = 1.0f;
        fetch_x(x0);
        fetch_a(a0, c0);
        advance();
        int q = 0;
        for (; q + 1 < n_q; q += 2) {
            taps_landed(a0);
            fetch_x(x1);
            fetch_a(a1, c1);
            advance();
            row_landed(x0, std::integral_constant<int, 1>());
            kn_order();
            mac(x0, a0, c0);
            kn_order();
            taps_landed(a1);
            fetch_x(x0);
            fetch_a(a0, c0);
            advance();
            row_landed(x1, std::integral_constant<int, 1>());
            kn_order();
            mac(x1, a1, c1);
            kn_order();
        }
        taps_landed(a0);
        row_landed(x0, std::integral_constant<int, 0>());
        if (q < n_q) mac(x0, a0, c0);
        }
    }
    if (!active) return;
    const float* xlast = p.lastcol ? (p.X + p.last_in_row * p.ldx + c) : nullptr;
    xrow_t xl4;
#pragma unroll
    for (int v = 0; v < VEC; v++) xl4[v] = 0.f;
    if (xlast) xl4 = *reinterpret_cast<const xrow_t*>(xlast);
#pragma unroll
    for (int r = 0; r < RBX; r++) {
        const int m = co0 + r;
        if (m >= p.Cout) continue;
        const int64_t row = (int64_t)m * p.HoWo + o;
        xrow_t t;
#pragma unroll
        for (int v = 0; v < VEC; v++) t[v] = acc[r][v / 2][v % 2];
        if (xlast) {
            const float lc = p.lastcol[row];
            if (lc != 0.0f) {
                const xrow_t bp = xl4 * lc;
                t = t + bp;
            }
        }
        if (p.relu) {
#pragma unroll
            for (int v = 0; v < VEC; v++) t[v] = (t[v] < 0.0f) ? 0.0f : t[v];
        }
        __builtin_nontemporal_store(t, reinterpret_cast<xrow_t*>(p.Y + row * p.ldy + c));      // the output is not re-read by this launch (conv1_1: 1.33 -> 1.11 ms)
    }
}

// (KN_FLAG_EXACT with the multiplies on the matrix pipe -- convtaps_exact_mfma_kernel, round 4 -- was bit-exact but measured 4-8 % SLOWER than the pipeline
// above on every keyed VGG-16 layer: the f32 matrix instruction runs on the vector ALU's own FP32 lanes, so the two do not overlap.  Removed in round 5;
// profiles/HISTORY.md has the numbers.)

// kn_convtaps_drop_zero_entries: the reference's UNTILED keyed conv CSR has no entry where a tap value is exactly 0, the order-preserving kernels
// above add fl(0 * x) there -- the same bits while x is finite (+-0 added to a sum that is never -0), a NaN the reference does not have when it
// is not.  One thread per (zero entry z = (tap, co, ci), output pixel o, batch column b): if the pixel has a slot with that tap and the activation
// it would have read is not finite, output (co, o, b) is recomputed in the reference's own sequence -- every stored entry except the zero-valued
// ones, channel outer, slots by ascending input pixel inner, bias last -- with separate multiply and add.  Nothing to do otherwise (the usual case).
__global__ __launch_bounds__(256) void convtaps_zero_guard_kernel(ConvArgs p, const int32_t* __restrict__ zero_ent, int64_t n_zero) {
    const int64_t n_ct = (p.n_vecs + 255) / 256;
    const int64_t blk = blockIdx.x;
    const int64_t zo = blk / n_ct;                                   // (zero entry, pixel) pair
    const int64_t b = (blk - zo * n_ct) * 256 + threadIdx.x;
    if (zo >= n_zero * p.HoWo || b >= p.n_vecs) return;
    const int z = (int)(zo / p.HoWo), o = (int)(zo % p.HoWo);
    const int t = zero_ent[3 * z], co = zero_ent[3 * z + 1], ci = zero_ent[3 * z + 2];
    const int s0 = p.pix_ptr[o], s1 = p.pix_ptr[o + 1];
    bool bad = false;
    for (int s = s0; s < s1; s++)
        if (p.slot_tap[s] == t) {
            const float xv = p.X[((int64_t)ci * p.HiWi + p.slot_in[s]) * p.ldx + b];
            bad = bad || !(__builtin_fabsf(xv) <= 3.4028234663852886e38f);
        }
    if (!bad) return;
    float acc = 0.0f;
    for (int c2 = 0; c2 < p.Cin; c2++)
        for (int s = s0; s < s1; s++) {
            float a = p.tapsT[((int64_t)p.slot_tap[s] * p.cin_pad + c2) * p.cout_pad + co];
            if (!p.unit_coef) a = p.slot_coef[s] * a;               // the reference's stored value: fl(coef * tap)
            if (a == 0.0f) continue;                                 // absent from the reference's row
            const float pr = a * p.X[((int64_t)c2 * p.HiWi + p.slot_in[s]) * p.ldx + b];
            acc = acc + pr;
        }
    const int64_t row = (int64_t)co * p.HoWo + o;
    if (p.lastcol) {
        const float lc = p.lastcol[row];
        if (lc != 0.0f) {
            const float bp = p.X[p.last_in_row * p.ldx + b] * lc;
            acc = acc + bp;
        }
    }
    if (p.relu) acc = (acc < 0.0f) ? 0.0f : acc;
    p.Y[row * p.ldy + b] = acc;
}

// homogeneous row of the output:  Y[last, b] = lastcol[last] * X[last, b]
__global__ __launch_bounds__(256) void conv_lastrow_kernel(const float* __restrict__ lastcol, int64_t out_last, const float* __restrict__ xlast,
                                                           float* __restrict__ ylast, int64_t n_vecs, int relu, float* absmax) {
    const float w = lastcol[out_last];
    float am = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vecs; i += (int64_t)gridDim.x * blockDim.x) {
        float v = w * xlast[i];
        if (relu) v = (v < 0.0f) ? 0.0f : v;
        ylast[i] = v;
        am = fmaxf(am, fabsf(v));
    }
    if (absmax) kn_wave_absmax_commit(am, absmax, threadIdx.x & 63);     // the tile epilogues cover every other row (kn_spmm_screen)
}

void convtaps_free(ConvTapsDev& c) {
    void* ptrs[] = {c.tapsT, c.pix_ptr, c.slot_in, c.slot_tap, c.slot_coef, c.pix_order, c.lastcol, c.sk_desc, c.tapsB, c.zero_ent, c.ex_ptr, c.ex_tab, c.ex_order, c.fill_ptr, c.fill_rec};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    c = ConvTapsDev();
}

// resident workgroups per XCD for a kernel (occupancy x CUs of one XCD), cached per instantiation
template <typename K>
static int64_t xcd_slots(K kernel) {
    int occ = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, 0) != hipSuccess || occ <= 0)
        return 0;
    return (int64_t)occ * (prop.multiProcessorCount / 8);
}

// Dynamic LDS that caps a kernel at `want` workgroups per CU (0 = no cap): the smallest allocation of which want + 1 copies no
// longer fit the CU's 160 KiB.  The kernel never touches it.
static unsigned lds_pad_for_occupancy(size_t static_lds, int want) {
    if (want <= 0) return 0;
    const size_t total = 160 * 1024;
    const size_t need = total / (size_t)(want + 1) + 256;           // per-workgroup footprint that admits only `want`
    if (need <= static_lds) return 0;
    const size_t pad = need - static_lds;
    return (static_lds + pad <= 64 * 1024) ? (unsigned)pad : 0;     // beyond 64 KiB per workgroup needs a function attribute: not used
}

template <int MT, int NB, int KC, int WM, int WN>
static void launch_conv(ConvArgs a, const Tuning& tune, hipStream_t s) {
    const int64_t items = (int64_t)a.n_pix * a.n_bt * a.n_mt;
    const int64_t chunk = (items + 7) / 8;
    constexpr size_t static_lds = sizeof(float) * (2 * KC * MT + 2 * KC * NB + 4 * MAX_FAST_SLOTS + 2 * (MT + NB));
    // Workgroups per CU.  All of a launch's workgroups take the same time, so a launch proceeds in "rounds" of the resident set; with
    // only a few rounds (VGG conv5_x: 196 items per XCD = 1.53 rounds of 128) a third of the work would fall into the last, partial
    // round.  Three workgroups per CU run as fast as four on the long layers (-1.5 %), and 196 items = 2.04 rounds of 96: when a
    // launch has fewer than four rounds, the resident set whose partial round is the smaller fraction wins (measured on conv5_x:
    // 122.9 -> 125.4 TFLOP/s, same-process A/B).  Tuning::occ forces a cap (diagnostic build).
    int occ_cap = tune.occ;
    if (occ_cap == 0 && MT == 128 && NB == 128 && KC == 16) {
        const double r4 = (double)chunk / 128.0, r3 = (double)chunk / 96.0;
        const double f4 = r4 - (double)(int64_t)r4, f3 = r3 - (double)(int64_t)r3;
        if (r4 < 4.0 && f4 > 0.0 && f3 < f4) occ_cap = 3;
    }
    const unsigned pad = lds_pad_for_occupancy(static_lds, occ_cap);
    auto D = [&](const char* loader, bool tail) {        // evaluated only by kn_spmm_plan
        char b[200];
        snprintf(b, sizeof(b), "convtaps_mfma_kernel<MT=%d,NB=%d,KC=%d> loader=%s%s%s occ_cap=%d", MT, NB, KC, loader, a.unit_coef ? "" : "+coef",
                 tail ? " tail_split" : "", occ_cap);
        return std::string(b);
    };
    const bool fast = a.vec_ok && a.unit_coef && (a.Cin % KC == 0 || a.Cin < KC) && (a.n_vecs % NB == 0) && ((int64_t)KC * a.HiWi * a.ldx < (int64_t)1 << 31) &&
                      a.max_slots <= MAX_FAST_SLOTS && (int64_t)a.HiWi * a.ldx < (int64_t)1 << 31 && (int64_t)a.ntaps * a.cin_pad * a.cout_pad < (int64_t)1 << 31;
    // scalar-pointer loaders (MODE 2): 16-row chunks of whole channels, a thread's offsets inside one chunk in 31 bits.  Tuning::no_sptr
    // (KN_NO_SPTR=1 when the operator is created) keeps the other loaders for the parity tests' side-by-side.
    bool sptr = false;
    const bool fast_shape = a.vec_ok && (a.n_vecs % NB == 0) && ((int64_t)KC * a.HiWi * a.ldx < (int64_t)1 << 31) &&      // (any number of slots per pixel: slot groups)
                            (int64_t)a.HiWi * a.ldx < (int64_t)1 << 31 && (int64_t)a.ntaps * a.cin_pad * a.cout_pad < (int64_t)1 << 31;
    if constexpr (KC == 16) sptr = fast_shape && a.Cin % 16 == 0 && 4 * ((int64_t)(1024 / NB) * a.HiWi * a.ldx + NB) < (int64_t)1 << 31 && !tune.no_sptr;
    a.tail_main = (int32_t)chunk;
    if constexpr (MT == 128 && NB == 128 && KC == 16) {
        if ((fast || sptr) && a.wide_store && !tune.no_tail_split && a.max_slots <= MAX_FAST_SLOTS) {
            // resident workgroups per XCD of the instantiation that is actually launched (the two loader modes may differ in registers)
            static const int64_t slots_free_1 = xcd_slots(convtaps_mfma_kernel<MT, NB, KC, WM, WN, 1, true>);
            static const int64_t slots_free_2 = xcd_slots(convtaps_mfma_kernel<MT, NB, KC, WM, WN, 2, true>);
            const int64_t slots_free = sptr ? slots_free_2 : slots_free_1;
            const int64_t slots = (pad > 0 && slots_free > 0) ? std::min<int64_t>(slots_free, (int64_t)occ_cap * 32) : slots_free;
            const int64_t rem = slots > 0 ? chunk % slots : 0;
            if (rem > 0) {   // the last, partial round of resident workgroups (measured: pays even when it fills half the machine)
                a.tail_main = (int32_t)(chunk - rem);
                const int64_t grid = 8 * ((int64_t)a.tail_main + 4 * rem);
#ifdef KN_ABLATION
                // DIAGNOSTIC BUILD ONLY (tools/ablate_conv.sh): per-workgroup time stamps of this launch, appended to a file.  Synchronises the
                // stream and uses one process-wide device buffer: not thread-safe, not capturable, never compiled into the product library.
                if (const char* path = getenv("KN_STAMPS")) {
                    static int64_t* dbuf = nullptr;
                    static int seq = 0;
                    const size_t cap = (size_t)1 << 22;
                    if (!dbuf && hipMalloc((void**)&dbuf, cap * 4 * sizeof(int64_t)) != hipSuccess) dbuf = nullptr;
                    if (dbuf && (size_t)grid <= cap) {
                        (void)hipMemsetAsync(dbuf, 0, (size_t)grid * 4 * sizeof(int64_t), s);
                        a.stamps = dbuf;
                        if (sptr) hipLaunchKernelGGL((convtaps_mfma_kernel<MT, NB, KC, WM, WN, 2, true>), dim3((unsigned)grid), dim3(256), pad, s, a);
                        else hipLaunchKernelGGL((convtaps_mfma_kernel<MT, NB, KC, WM, WN, 1, true>), dim3((unsigned)grid), dim3(256), pad, s, a);
                        (void)hipStreamSynchronize(s);
                        std::vector<int64_t> h((size_t)grid * 4);
                        (void)hipMemcpy(h.data(), dbuf, h.size() * sizeof(int64_t), hipMemcpyDeviceToHost);
                        if (FILE* f = fopen(path, "ab")) {
                            const int64_t hdr[4] = {(int64_t)0x7374616d70, (int64_t)seq++, grid, ((int64_t)a.n_pix << 32) | (int64_t)(a.cin_pad * 1000 + a.n_mt)};
                            fwrite(hdr, sizeof(int64_t), 4, f);
                            fwrite(h.data(), sizeof(int64_t), h.size(), f);
                            fclose(f);
                        }
                        return;
                    }
                }
#endif
                if (sptr) KN_LAUNCH(D("sptr(wave-uniform pointers)", true), (convtaps_mfma_kernel<MT, NB, KC, WM, WN, 2, true>), dim3((unsigned)grid), dim3(256), pad, s, a);
                else KN_LAUNCH(D("fast(per-thread pointers)", true), (convtaps_mfma_kernel<MT, NB, KC, WM, WN, 1, true>), dim3((unsigned)grid), dim3(256), pad, s, a);
                return;
            }
        }
    }
    const int64_t grid = 8 * chunk;
    if constexpr (KC == 16) {
        if (sptr && a.max_slots > MAX_FAST_SLOTS) {      // pixels with more than 64 slots: the slot-group instantiation (no tail split: such launches are long)
            KN_LAUNCH(D("sptr(wave-uniform pointers, slot groups)", false), (convtaps_mfma_kernel<MT, NB, KC, WM, WN, 3, false>), dim3((unsigned)grid), dim3(256), pad, s, a);
            return;
        }
        if (sptr) {
            KN_LAUNCH(D("sptr(wave-uniform pointers)", false), (convtaps_mfma_kernel<MT, NB, KC, WM, WN, 2, false>), dim3((unsigned)grid), dim3(256), pad, s, a);
            return;
        }
    }
    if (fast) KN_LAUNCH(D("fast(per-thread pointers)", false), (convtaps_mfma_kernel<MT, NB, KC, WM, WN, 1, false>), dim3((unsigned)grid), dim3(256), pad, s, a);
    else KN_LAUNCH(D("generic", false), (convtaps_mfma_kernel<MT, NB, KC, WM, WN, 0, false>), dim3((unsigned)grid), dim3(256), pad, s, a);
}

// Filled-in operators under KN_FLAG_EXACT: does this operator take convtaps_exact_fill_kernel?  (fill_ptr was laid out at create: convtaps_create_impl.)
bool convtaps_fill_ok(const ConvTapsDev& A) { return A.fill_ptr != nullptr && A.fill_n > 0 && !A.tune.no_fill_exact; }

// ... and its record lists, built on the device from the slot lists at the first kn_spmm that needs them (16 bytes per slot: VGG-16 under doubly-stochastic
// keys 0.4 - 4 GB per layer that runs in the reference's order)
int convtaps_build_fill(ConvTapsDev& A, hipStream_t s) {
    if (A.fill_rec || !convtaps_fill_ok(A)) return KN_OK;
    int32_t* d = nullptr;
    KN_HIP(hipMalloc(reinterpret_cast<void**>(&d), (size_t)A.fill_n * sizeof(FillRec)));
    const int HoWo = (int)(A.Hout * A.Wout);
    hipLaunchKernelGGL(convtaps_fill_records_kernel, dim3((unsigned)((HoWo + 3) / 4)), dim3(256), 0, s, A.pix_ptr, A.fill_ptr, A.slot_in, A.slot_tap, A.slot_coef, A.unit_coef ? 1 : 0,
                       A.ntaps <= 16 ? 1 : (int)(A.cin_pad * A.cout_pad), HoWo, reinterpret_cast<FillRec*>(d));
    if (hipGetLastError() != hipSuccess) {
        (void)hipFree(d);
        return fail(KN_ERR_HIP, "convtaps_fill_records_kernel launch failed");
    }
    // The records are published only when they EXIST: a second thread / stream that sees the pointer launches convtaps_exact_fill_kernel on its own stream with no
    // ordering against stream s (round-5 advisor finding).  One host wait per operator lifetime; the caller holds Handle::lazy_mu.
    if (hipStreamSynchronize(s) != hipSuccess) {
        (void)hipFree(d);
        return fail(KN_ERR_HIP, "convtaps_fill_records_kernel failed");
    }
    __atomic_store_n(&A.fill_rec, d, __ATOMIC_RELEASE);
    return KN_OK;
}

// Can this operator / operand take convtaps_bf16x3_kernel?  (The planes must exist: convtaps_build_bf16 at first use, kn_api.hip.)
bool convtaps_bf16x3_ok(const ConvTapsDev& A, const float* x, int64_t ldx, int64_t n_vecs, const float* y, int64_t ldy) {
    const bool wide = A.Cout > 64;
    const bool planes = A.tapsB != nullptr || plan_sink() != nullptr;      // kn_spmm_plan allocates nothing: the first real kn_spmm builds the planes
    return planes && A.Cin % 16 == 0 && A.cin_pad == A.Cin && (wide ? (A.cout_pad % 128 == 0) : (A.cout_pad == 64)) && n_vecs > 0 && n_vecs % 128 == 0 &&
           ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)x) % 16 == 0 && ((uintptr_t)y) % 16 == 0 && A.max_slots <= MAX_FAST_SLOTS;
}

// Three bf16 planes of the taps, laid out as the kernel's LDS image: [plane][tap][channel chunk of 16][cout_pad][16 k], the two 8-k halves of
// a row swapped on rows (cout) with bit 3 set.  Round-to-nearest-even split at each step: w = h + m + l to 2^-27 |w|, the residuals r1, r2 exact.
int convtaps_build_bf16(ConvTapsDev& A, const std::vector<float>& taps /* [ntaps][Cout][Cin] */) {
    if (A.tapsB) return KN_OK;
    if (A.Cin % 16 != 0 || A.cin_pad != A.Cin || A.cout_pad % 64 != 0) return KN_OK;      // not eligible: nothing to build
    const int64_t cpk = A.cin_pad / 16;
    const size_t plane = (size_t)(A.ntaps * cpk * A.cout_pad * 16);
    std::vector<uint16_t> hb(3 * plane, 0);
    for (int64_t t = 0; t < A.ntaps; t++)
        for (int64_t co = 0; co < A.Cout; co++)
            for (int64_t ci = 0; ci < A.Cin; ci++) {
                const float w = taps[(size_t)((t * A.Cout + co) * A.Cin + ci)];
                auto rne = [](float v) -> uint32_t {                       // f32 -> bf16 bits in the upper half, round to nearest even
                    uint32_t b;
                    std::memcpy(&b, &v, 4);
                    if ((b & 0x7f800000u) == 0x7f800000u) return b & 0xffff0000u;   // inf / nan unchanged (taps are finite)
                    b += 0x7fffu + ((b >> 16) & 1u);
                    return b & 0xffff0000u;
                };
                auto asf = [](uint32_t b) {
                    float f;
                    std::memcpy(&f, &b, 4);
                    return f;
                };
                const uint32_t h = rne(w);
                const float r1 = w - asf(h);
                const uint32_t m = rne(r1);
                const float r2 = r1 - asf(m);
                const uint32_t l = rne(r2);
                const int64_t k = ci % 16, half = k / 8;
                const int64_t pos = ((t * cpk + ci / 16) * A.cout_pad + co) * 16 + ((half ^ ((co >> 3) & 1)) * 8) + (k % 8);
                hb[(size_t)pos] = (uint16_t)(h >> 16);
                hb[plane + (size_t)pos] = (uint16_t)(m >> 16);
                hb[2 * plane + (size_t)pos] = (uint16_t)(l >> 16);
            }
    uint16_t* d = nullptr;
    int rc = upload(&d, hb.data(), hb.size());
    if (rc) return rc;
    A.tapsB_plane = (int64_t)plane * 2;
    __atomic_store_n(&A.tapsB, d, __ATOMIC_RELEASE);         // upload() is a synchronous copy: the planes are in HBM before the pointer is visible
    return KN_OK;
}

int convtaps_spmm(const ConvTapsDev& A, int64_t rows, int64_t cols, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy,
                  uint32_t flags, hipStream_t s, float* absmax, bool* absmax_fused) {
    (void)rows;
    (void)cols;
    ConvArgs a;
    a.absmax = nullptr;
    if (absmax_fused) *absmax_fused = false;
    a.tapsT = A.tapsT;
    a.pix_ptr = A.pix_ptr;
    a.slot_in = A.slot_in;
    a.slot_tap = A.slot_tap;
    a.slot_coef = A.slot_coef;
    a.pix_order = A.pix_order;
    a.lastcol = A.has_last ? A.lastcol : nullptr;
    a.X = x;
    a.Y = y;
    a.ldx = ldx;
    a.ldy = ldy;
    a.cin_pad = (int32_t)A.cin_pad;
    a.cout_pad = (int32_t)A.cout_pad;
    a.Cin = (int32_t)A.Cin;
    a.Cout = (int32_t)A.Cout;
    a.HiWi = (int32_t)(A.Hin * A.Win);
    a.HoWo = (int32_t)(A.Hout * A.Wout);
    a.n_vecs = (int32_t)n_vecs;
    a.relu = (flags & KN_FLAG_RELU) ? 1 : 0;
    a.unit_coef = A.unit_coef ? 1 : 0;
    a.vec_ok = (n_vecs % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x) % 16 == 0) ? 1 : 0;
    a.n_pix = a.HoWo;
    a.ntaps = (int32_t)A.ntaps;
    a.last_in_row = A.Cin * A.Hin * A.Win;

    a.stamps = nullptr;
#ifdef KN_ABLATION
    a.abl = A.tune.abl;
#endif
    a.sk_desc = A.sk_desc;
    a.sk_stride = (int32_t)A.sk_stride;
    a.sk_tab_rows = (int32_t)A.sk_tab_rows;
    a.tail_main = 0;
    if (flags & KN_FLAG_EXACT) {
        const bool v4 = a.vec_ok && n_vecs >= 256;
        const int pipe_mode = A.tune.exact_pipe;
        const bool pipe_shape = pipe_mode > 0 && !A.has_dups && A.max_slots <= 64 && (a.last_in_row + 1) * ldx < ((int64_t)1 << 31) &&
                                (int64_t)A.ntaps * A.cin_pad * A.cout_pad < ((int64_t)1 << 31);
        const bool pipe4 = pipe_shape && v4 && ldy % 4 == 0 && ((uintptr_t)y) % 16 == 0;
        // two batch columns per lane (128-column tiles): a half-batch window of the overlapped forward at 256 images, or a batch that fills 128-column
        // tiles better than 256-column ones (384 images).  Tuning::exact_vec = 2 | 4 forces either where both apply (diagnostic build).
        const bool pipe2_ok = pipe_shape && n_vecs >= 128 && n_vecs % 2 == 0 && ldx % 2 == 0 && ldy % 2 == 0 && ((uintptr_t)x) % 8 == 0 && ((uintptr_t)y) % 8 == 0;
        const bool pipe2 = pipe2_ok && (A.tune.exact_vec == 2 || (A.tune.exact_vec != 4 && (!pipe4 || (n_vecs + 127) / 128 * 128 < (n_vecs + 255) / 256 * 256)));
        const bool pipe = pipe4 || pipe2;
        const int64_t n_ct = pipe2 ? (n_vecs + 127) / 128 : (v4 ? (n_vecs + 255) / 256 : (n_vecs + 63) / 64);
        // 16 output channels per wavefront when that still leaves every SIMD several wavefronts, else 8
        const int rbx = (pipe && pipe_mode >= 16 && A.Cout % 16 == 0 && (int64_t)a.n_pix * (A.Cout / 16) * n_ct >= 4096) ? 16 : 8;
        const int n_cob = (int)((A.Cout + rbx - 1) / rbx);
        const int64_t n_rb = ((int64_t)a.n_pix * n_cob + 3) / 4;
        // Channel-bundle groups (convtaps_exact_pipe_kernel): with g groups every XCD works on 1/g (g = 8) or 2/g of the output channels for all
        // pixels, so its share of the taps stays in its 4 MiB L2 for the scalar tap loads -- 9.4 MB of taps on the 512-channel layers of VGG-16.
        // Same-process A/B, exact mode, ms at 1 / 4 / 8 groups: conv3_2 (2.4 MB of taps) 13.63 / 13.49 / 14.60, conv4_1 6.99 / 6.75 / 6.84,
        // conv4_2 14.05 / 13.54 / 13.48, conv4_3 14.01 / 13.34 / 13.53, conv5_1 4.26 / 3.82 / 3.82, conv5_2 4.24 / 3.84 / 4.10; layers with small tap
        // matrices lose 5 % (conv1_2, conv2_x: the bundles of a pixel no longer share its gathered rows in one XCD).  Rule: 4 groups when the taps exceed half
        // of the L2 (2 MB: VGG-16 conv3_x and up; AllConvNet's 192-channel layers, 1.3 MB at 16 column tiles per layer, lose 10 % when grouped).  Tuning::exact_cob_groups overrides (diagnostic build).
        a.tail_main = 0;
        {
            const int64_t tap_bytes = 4 * A.ntaps * A.cin_pad * A.cout_pad;
            int g = tap_bytes > (2 << 20) ? 4 : 1;
            if (A.tune.exact_cob_groups > 0) g = A.tune.exact_cob_groups;
            while (g > 1 && n_cob % g != 0) g >>= 1;
            if (g > 1) a.tail_main = n_cob / g;
        }
        const int64_t grid = ((n_ct * n_rb + 7) / 8) * 8;
        // a factored stand-in of an untiled CSR that carries the stored-column table (kn_convtaps_drop_zero_entries) on a wide batch: the matrix-pipe
        // grouped kernel reads its values from the tap table (kn_csr_mfma.hip, TAPS).  Tuning::no_exact_table keeps the conv pipeline instead.
        const bool table = A.ex_tab != nullptr && n_vecs >= 128 && !A.tune.no_exact_table;
        if (table) {
            int rc = convtaps_exact_table_spmm(A, x, ldx, n_vecs, y, ldy, a.relu, s);
            if (rc) return rc;
        }
        // four activation rows in flight when the batch spans several 256-column tiles (the rows of a [D, 4096] block are L2 misses; at one tile --
        // VGG-16 at 256 images -- three rows in flight measured 2-3 % slower than two).  Tuning::exact_xd = 2 | 4 overrides (diagnostic build).
        // filled-in operators (more than 64 slots per pixel, or several slots on one pixel pair): convtaps_exact_fill_kernel (its records exist from the first kn_spmm on)
        const bool fill = !pipe && !table && convtaps_fill_ok(A) && (A.fill_rec != nullptr || plan_sink() != nullptr) && (int64_t)a.HiWi * ldx * 4 < ((int64_t)1 << 32) &&
                          a.HiWi < (1 << 24) && ldx * 4 < ((int64_t)1 << 24);
        bool xd4 = n_ct >= 4;
        if (A.tune.exact_xd > 0) xd4 = A.tune.exact_xd == 4;
        if (table) {
            // (launched above)
        } else
        if (pipe2 && rbx == 16 && A.unit_coef) KN_LAUNCH("convtaps_exact_pipe_kernel<16,128-column tiles>", (convtaps_exact_pipe_kernel<16, false, 2, 2>), dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else if (pipe2 && rbx == 16) KN_LAUNCH("convtaps_exact_pipe_kernel<16,coef,128-column tiles>", (convtaps_exact_pipe_kernel<16, true, 2, 2>), dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else if (pipe2 && A.unit_coef) KN_LAUNCH("convtaps_exact_pipe_kernel<8,128-column tiles>", (convtaps_exact_pipe_kernel<8, false, 2, 2>), dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else if (pipe2) KN_LAUNCH("convtaps_exact_pipe_kernel<8,coef,128-column tiles>", (convtaps_exact_pipe_kernel<8, true, 2, 2>), dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else if (pipe && rbx == 16 && A.unit_coef && xd4) KN_LAUNCH("convtaps_exact_pipe_kernel<16,rows in flight=4>", (convtaps_exact_pipe_kernel<16, false, 4>), dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else if (pipe && rbx == 16 && xd4) KN_LAUNCH("convtaps_exact_pipe_kernel<16,coef,rows in flight=4>", (convtaps_exact_pipe_kernel<16, true, 4>), dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else if (pipe && rbx == 16 && A.unit_coef) KN_LAUNCH("convtaps_exact_pipe_kernel<16>", (convtaps_exact_pipe_kernel<16>), dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else if (pipe && rbx == 16) KN_LAUNCH("convtaps_exact_pipe_kernel<16,coef>", (convtaps_exact_pipe_kernel<16, true>), dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else if (pipe && A.unit_coef) KN_LAUNCH("convtaps_exact_pipe_kernel<8>", (convtaps_exact_pipe_kernel<8>), dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else if (pipe) KN_LAUNCH("convtaps_exact_pipe_kernel<8,coef>", (convtaps_exact_pipe_kernel<8, true>), dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else if (fill) {
            // 64 output channels per wavefront (the slot bookkeeping once per 64 channels) when that still leaves every SIMD its three wavefronts, else 32;
            // two 64-column tiles per wavefront (the bookkeeping once per 128 columns) on batches of whole 128-column tiles when that still leaves every SIMD its two
            // wavefronts (the two-tile forms hold 128 + 64 result registers: two wavefronts per SIMD).  Tuning::no_fill_tiles2 keeps one tile (parity tests' side-by-side).
            const bool t2_ok = A.ntaps <= 16 && n_vecs % 128 == 0 && ldx % 2 == 0 && ldy % 2 == 0 && ((uintptr_t)x) % 8 == 0 && ((uintptr_t)y) % 8 == 0 && !A.tune.no_fill_tiles2;
            const int64_t n_ct2 = n_vecs / 128;
            const bool wide1 = A.ntaps <= 16 && A.Cout > 32 && (int64_t)a.n_pix * ((A.Cout + 63) / 64) * ((n_vecs + 63) / 64) >= 3 * 1024;
            const bool wide2 = t2_ok && A.Cout > 32 && (int64_t)a.n_pix * ((A.Cout + 63) / 64) * n_ct2 >= 2 * 1024;
            const bool narrow2 = t2_ok && !wide2 && !wide1 && (int64_t)a.n_pix * ((A.Cout + 31) / 32) * n_ct2 >= 2 * 1024;      // (64 channels x one tile is the same work per wavefront: kept where it qualifies)
            bool tiles2 = wide2 || narrow2;
            bool wide_sel = tiles2 ? wide2 : wide1;
            if (A.tune.fill_form > 0 && A.ntaps <= 16) {             // (diagnostic build: force a form where the operands allow it -- tools/fill_bench.py)
                const bool want2 = A.tune.fill_form >= 3 && t2_ok;
                tiles2 = want2;
                wide_sel = (A.tune.fill_form == 2 || A.tune.fill_form == 4) && A.Cout > 32;
            }
            const int n_ctf = tiles2 ? (int)n_ct2 : (int)((n_vecs + 63) / 64);
            const bool wide = wide_sel;
            const int n_cc = (int)((A.Cout + (wide ? 63 : 31)) / (wide ? 64 : 32));
            const int64_t n_wg = ((int64_t)a.n_pix * n_cc * n_ctf + 3) / 4;
            KN_REQUIRE(n_wg + 8 < ((int64_t)1 << 31), KN_ERR_UNSUPPORTED, "grid too large for the filled-in order-preserving kernel");
            const std::string d = std::string("convtaps_exact_fill_kernel") + (A.ntaps <= 16 ? (wide ? "<taps in registers, 64 channels per wavefront" : "<taps in registers") : "") +
                                  (A.ntaps <= 16 ? (tiles2 ? ", two column tiles per wavefront>" : ">") : "") +
                                  " (stored values formed per lane, products on the matrix pipe, " + std::to_string(A.fill_n) + " slot records)";
            const dim3 gridf((unsigned)(((n_wg + 7) / 8) * 8));
            const FillRec* rec = reinterpret_cast<const FillRec*>(A.fill_rec);
            if (tiles2 && wide) KN_LAUNCH(d, (convtaps_exact_fill_kernel<16, true, 2>), gridf, dim3(256), 0, s, a, A.fill_ptr, rec, n_cc, n_ctf, n_wg);
            else if (tiles2) KN_LAUNCH(d, (convtaps_exact_fill_kernel<16, false, 2>), gridf, dim3(256), 0, s, a, A.fill_ptr, rec, n_cc, n_ctf, n_wg);
            else if (wide) KN_LAUNCH(d, (convtaps_exact_fill_kernel<16, true>), gridf, dim3(256), 0, s, a, A.fill_ptr, rec, n_cc, n_ctf, n_wg);
            else if (A.ntaps <= 16) KN_LAUNCH(d, (convtaps_exact_fill_kernel<16, false>), gridf, dim3(256), 0, s, a, A.fill_ptr, rec, n_cc, n_ctf, n_wg);
            else KN_LAUNCH(d, (convtaps_exact_fill_kernel<0, false>), gridf, dim3(256), 0, s, a, A.fill_ptr, rec, n_cc, n_ctf, n_wg);
        }
        else if (v4) KN_LAUNCH("convtaps_exact_kernel<vec=4>", convtaps_exact_kernel<4>, dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        else KN_LAUNCH("convtaps_exact_kernel<vec=1>", convtaps_exact_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, a, n_cob, n_rb);
        if (A.has_last) {
            const int64_t out_last = A.Cout * A.Hout * A.Wout;
            KN_LAUNCH("conv_lastrow_kernel", conv_lastrow_kernel, dim3((unsigned)std::min<int64_t>((n_vecs + 255) / 256, 256)), dim3(256), 0, s, A.lastcol, out_last,
                               x + a.last_in_row * ldx, y + out_last * ldy, n_vecs, a.relu, a.absmax);
        }
        if (A.n_zero > 0) {                                  // kn_convtaps_drop_zero_entries: behind the main kernel, on its stream
            const int64_t gz = A.n_zero * a.HoWo * ((n_vecs + 255) / 256);
            KN_REQUIRE(gz < ((int64_t)1 << 31), KN_ERR_UNSUPPORTED, "too many zero-valued tap entries for the guard launch");
            KN_LAUNCH("convtaps_zero_guard_kernel<" + std::to_string(A.n_zero) + " zero tap entries>", convtaps_zero_guard_kernel, dim3((unsigned)gz), dim3(256), 0, s, a, A.zero_ent, A.n_zero);
        }
        KN_HIP(hipGetLastError());
        return KN_OK;
    }
    a.wide_store = (a.vec_ok && ldy % 4 == 0 && ((uintptr_t)y) % 16 == 0) ? 1 : 0;
    // every matrix-core kernel below streams its tiles out through kn_store_tile when the stores are wide and the batch fills whole tiles of the
    // kernel that will run (128 columns for the 128 x 128 and bf16x3 tiles -- so also the half-batch windows of the overlapped forward at 256 images --,
    // 256 for the 64 x 256 and small-K tiles): then max |Y| rides in the epilogues (tiles + conv_lastrow_kernel for the homogeneous row)
    const int64_t nb_tile = (((flags & KN_FLAG_BF16X3) && convtaps_bf16x3_ok(A, x, ldx, n_vecs, y, ldy)) || (A.cout_pad % 128 == 0 && A.Cout > 64)) ? 128 : 256;
    if (absmax && a.wide_store && n_vecs % nb_tile == 0) {
        a.absmax = absmax;
        if (absmax_fused) *absmax_fused = true;
    }
    a.max_slots = A.max_slots;
    a.ntaps = (int32_t)A.ntaps;
    a.last_in_row = A.Cin * A.Hin * A.Win;
    a.tapsB = A.tapsB;
    a.tapsB_plane = A.tapsB_plane;
    if ((flags & KN_FLAG_BF16X3) && convtaps_bf16x3_ok(A, x, ldx, n_vecs, y, ldy)) {
        const bool wide = A.Cout > 64;                          // 128 x 128 tiles; 64-channel layers take 64 x 128 (four wavefronts of 32 x 64)
        a.n_mt = (int32_t)(A.cout_pad / (wide ? 128 : 64));
        a.n_bt = (int32_t)(n_vecs / 128);
        const int64_t items = (int64_t)a.n_pix * a.n_bt * a.n_mt;
        const int64_t chunk = (items + 7) / 8;
        a.tail_main = (int32_t)chunk;
        const std::string d = std::string("convtaps_bf16x3_kernel<") + (wide ? "128x128" : "64x128") + ", 3-way bf16 split, 6 products>" + (A.unit_coef ? "" : "+coef");
        if (wide) {
            // last partial round of resident workgroups as quarter tiles (same split as launch_conv)
            static const int64_t slots0 = xcd_slots(convtaps_bf16x3_kernel<128, 128, 2, 2, false, true>);
            static const int64_t slots1 = xcd_slots(convtaps_bf16x3_kernel<128, 128, 2, 2, true, true>);
            const int64_t slots = A.unit_coef ? slots0 : slots1;
            const int64_t rem = slots > 0 ? chunk % slots : 0;
            if (rem > 0 && !A.tune.no_tail_split) {
                a.tail_main = (int32_t)(chunk - rem);
                const int64_t grid = 8 * ((int64_t)a.tail_main + 4 * rem);
                if (A.unit_coef) KN_LAUNCH(d + " tail_split", (convtaps_bf16x3_kernel<128, 128, 2, 2, false, true>), dim3((unsigned)grid), dim3(256), 0, s, a);
                else KN_LAUNCH(d + " tail_split", (convtaps_bf16x3_kernel<128, 128, 2, 2, true, true>), dim3((unsigned)grid), dim3(256), 0, s, a);
            } else {
                if (A.unit_coef) KN_LAUNCH(d, (convtaps_bf16x3_kernel<128, 128, 2, 2, false, false>), dim3((unsigned)(8 * chunk)), dim3(256), 0, s, a);
                else KN_LAUNCH(d, (convtaps_bf16x3_kernel<128, 128, 2, 2, true, false>), dim3((unsigned)(8 * chunk)), dim3(256), 0, s, a);
            }
        } else {
            if (A.unit_coef) KN_LAUNCH(d, (convtaps_bf16x3_kernel<64, 128, 2, 2, false, false>), dim3((unsigned)(8 * chunk)), dim3(256), 0, s, a);
            else KN_LAUNCH(d, (convtaps_bf16x3_kernel<64, 128, 2, 2, true, false>), dim3((unsigned)(8 * chunk)), dim3(256), 0, s, a);
        }
        if (A.has_last) {
            const int64_t out_last = A.Cout * A.Hout * A.Wout;
            KN_LAUNCH("conv_lastrow_kernel", conv_lastrow_kernel, dim3((unsigned)std::min<int64_t>((n_vecs + 255) / 256, 256)), dim3(256), 0, s, A.lastcol, out_last,
                      x + a.last_in_row * ldx, y + out_last * ldy, n_vecs, a.relu, a.absmax);
        }
        KN_HIP(hipGetLastError());
        return KN_OK;
    }
    const bool big_m = A.cout_pad % 128 == 0 && A.Cout > 64;
    const bool k16 = A.cin_pad % 16 == 0;
    if (!A.tune.no_smallk && (int64_t)A.max_slots * A.Cin + (A.has_last ? 1 : 0) <= SMALLK_MAX && a.wide_store && n_vecs % 256 == 0 && A.cout_pad % 64 == 0) {
        a.n_mt = (int32_t)(A.cout_pad / 64);
        a.n_bt = (int32_t)(n_vecs / 256);
        const int64_t items = (int64_t)a.n_pix * a.n_bt * a.n_mt;
        const bool no_pipe = A.tune.no_smallk_pipe != 0;      // (the one-shot kernel, for the parity tests' side-by-side)
        if (a.sk_desc && a.n_mt == 1 && !no_pipe) {
            static const int64_t slots = xcd_slots(convtaps_smallk_pipe_kernel);
            const int64_t per_xcd = std::min<int64_t>(std::max<int64_t>(slots, 32), (items + 7) / 8);
            KN_LAUNCH("convtaps_smallk_pipe_kernel", convtaps_smallk_pipe_kernel, dim3((unsigned)(8 * per_xcd)), dim3(256), 0, s, a);
        } else {
            KN_LAUNCH("convtaps_smallk_kernel", convtaps_smallk_kernel, dim3((unsigned)(((items + 7) / 8) * 8)), dim3(256), 0, s, a);
        }
    } else if (big_m) {
        a.n_mt = (int32_t)(A.cout_pad / 128);
        a.n_bt = (int32_t)((n_vecs + 127) / 128);
        if (k16) launch_conv<128, 128, 16, 2, 2>(a, A.tune, s);
        else launch_conv<128, 128, 4, 2, 2>(a, A.tune, s);
    } else {
        a.n_mt = (int32_t)(A.cout_pad / 64);
        a.n_bt = (int32_t)((n_vecs + 255) / 256);
        if (k16) launch_conv<64, 256, 16, 1, 4>(a, A.tune, s);
        else launch_conv<64, 256, 4, 1, 4>(a, A.tune, s);
    }
    if (A.has_last) {
        const int64_t out_last = A.Cout * A.Hout * A.Wout;
        KN_LAUNCH("conv_lastrow_kernel", conv_lastrow_kernel, dim3((unsigned)std::min<int64_t>((n_vecs + 255) / 256, 256)), dim3(256), 0, s, A.lastcol, out_last,
                           x + a.last_in_row * ldx, y + out_last * ldy, n_vecs, a.relu, a.absmax);
    }
    KN_HIP(hipGetLastError());
    return KN_OK;
}

}  // namespace kn
