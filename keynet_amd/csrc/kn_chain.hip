// kn_chain.hip -- a WHOLE keyed network as one launch: every operator of an untiled key-net (stored-order CSR: the permutation
// key-nets of BASELINE configs[0]-[1]) applied back to back with the activations resident in LDS.
//
// Replaces the nn.Sequential walk of KeyedModel.forward (keynet/system.py:130-133) for key-nets small enough that a launch per layer is
// what bounds them: LeNet_AvgPool at 1024 images is 7 launches of ~20 us each for 88 MB of algorithmic traffic (11 us at the HBM
// roof).  Batch columns are independent (scipy's csr_matvecs never mixes them), so a workgroup owns BT = 4 columns and walks all
// operators; the feature-major activation block of its columns ([features][4] f32 = one 16-byte LDS word per feature) ping-pongs
// between two LDS buffers (LeNet: 4705 + 1177 features = 92 KB of the CU's 160 KB), the operators stream from L2 (2.6 MB).
//
// Arithmetic: one LANE owns one output row for the workgroup's four batch columns and accumulates it strictly serially over the row's
// STORED non-zeros, f32 multiply then f32 add (no contraction) -- the rounding sequence of scipy, bit for bit, like kn_csr.hip.
//
// Operator layout (built once by chain_create from the CSR the reference holds):
//   * rows of a layer are dealt to lanes sorted by (length descending, column pattern, row): the 64 rows of a slice (= one wavefront)
//     then have (almost always) equal lengths, and rows sharing one column sequence (the Cout rows of a conv output pixel, every row of
//     a keyed nn.Linear) sit in adjacent lanes, where their LDS reads of one activation are a broadcast;
//   * values: sliced ELL in quads, [slice][k / 4][lane][k % 4] -- a lane's next four values are ONE 16-byte load, a wavefront's load is
//     one contiguous 1 KiB piece;
//   * columns: stored as the LDS byte offset of the feature (no address arithmetic in the walk); rows sharing a pattern read ONE copy of
//     it (per-lane base, quad stride 1: the lanes of a group hit the same address), slices of unrelated rows (pooling) store them like
//     the values;
//   * a row shorter than its slice's longest row is padded with (zero feature, 0.0f), see chain_rows: no predicate anywhere;
//   * layers whose rows share column patterns (conv, Linear) keep their pool of patterns in LDS during their walk (chain_rows_cl, chain_rows_thin).
//
// Measured (LeNet_AvgPool, 1024 images, tools/chain_bench.py, tools/chain_stamps.sh): 44 us per forward against 138 us for seven launches; per
// layer (timestamps at the barriers, median of 256 workgroups) conv1 7.8, pool1 3.3, conv2 12.1, pool2 2.3, fc1 8-12, fc2 2.6, fc3 1.9 us.  The
// floor of this formulation is vector-instruction issue, not memory: a SIMD issues one vector instruction per ~4.5 cycles whatever its kind, and a
// stored non-zero costs 5.5 of them per 64 rows and four batch columns (without the FMA: two packed multiplies, two packed adds, one ds_read_b128,
// half a load) -- ~17 us per launch for LeNet's 323 k non-zeros.  What was tried on the way is in DESIGN.md 5.
#include "kn_internal.h"
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <array>
#include <cstring>
#include <numeric>
#include <type_traits>
#include <unordered_map>

#pragma clang fp contract(off)

namespace kn {

static constexpr int CHAIN_BT = 4;            // batch columns per workgroup
static constexpr int CHAIN_THREADS = 1024;    // 16 wavefronts
static constexpr int CHAIN_MAX_LAYERS = 12;
static constexpr size_t CHAIN_LDS_BYTES = 160 * 1024;
static constexpr size_t CHAIN_THIN_POOL_BYTES = 12 * 1024;    // column pool of a thin layer that may be staged in LDS (two copies)
static constexpr int CHAIN_OVERREAD_QUADS = 16;                // >= ring depth + next-slice quads: how far the walk may request past the end of an array

struct ChainLayerArg {
    const float* vals;          // quads: [slice][q][lane][row of the lane (rpl)][4]
    const int32_t* cols;        // pool of column quads; an entry is the LDS BYTE offset of the feature in this layer's input buffer
    const int32_t* lane_meta;   // [n_slices * 64][4]: output row (-1 = empty slot), index of the lane's first column QUAD in `cols`, second output row (rpl = 2) or -1, 0
    const int32_t* slice_info;  // [n_slices][4]: quads, column quad stride, first value quad / (64 * rpl), 0
    int32_t n_slices, n_rows, relu;
    int32_t cols_quads;         // size of the layer's column pool in quads when the pool is staged in LDS before the walk:
                                //   > 0 a THIN layer (chain_rows_thin, two copies), < 0 a layer of shared patterns on all wavefronts
                                //   (chain_rows_cl, one copy); 0 = columns from memory (chain_rows)
    int32_t rpl;                // output rows per lane: 1, or 2 (chain_rows_cl: two rows of one column pattern share every activation read)
    int32_t stage_off;          // float4 index of the LDS area the pool is staged in
    int32_t early;              // 1 = the pool is staged while the PREVIOUS layer runs (layer 0: with the input), 0 = at the start of this layer
    int32_t vstride;            // general / pattern walks: distance between a lane's consecutive value quads, in units of 16 * rpl bytes (64 = every lane its own copy; the number of
                                // lanes per pixel when lanes that carry the SAME value sequence share one copy: see chain_build_layer)
    int32_t seq_len;            // > 0: a SEQUENTIAL thin layer (chain_rows_thin_seq) -- the stored entries of the one column pattern all of these rows share; the
    int32_t seq_base;           //      layer's input buffer (LDS byte offset seq_base) is laid out in that pattern's order by the layer before it
};

struct ChainArgs {
    ChainLayerArg L[CHAIN_MAX_LAYERS];
    ChainLayerArg LX[CHAIN_MAX_LAYERS];                            // the rows of a sequential thin layer that do NOT share its main pattern (a keyed Linear's homogeneous row): general walk, own wavefronts
    const float* X;
    float* Y;
    int64_t ldx, ldy;
    int32_t n_layers, n_vecs, n_in, n_out, buf1_off, zero_off;     // float4 indices: start of the second activation buffer; the always-zero feature
#ifdef KN_ABLATION
    unsigned long long* stamps;                                    // [workgroup][16] 100 MHz timestamps at the phase boundaries (tools/chain_stamps.sh), or null
    unsigned long long* wstamps;                                   // [workgroup][12 layers][16 wavefronts][40] inside the walks: 0-7 per walk, 8 + 4 * slice + k per slice
#endif
};
#ifdef KN_ABLATION
#define CHAIN_STAMP(k)                                                                           \
    do {                                                                                         \
        if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); \
    } while (0)
// inside a walk: `drain` = wait for everything requested so far first (what the stamp then shows is the LATENCY of those requests; the walk is perturbed)
#define CHAIN_WSTAMP(k, drain)                                                                   \
    do {                                                                                         \
        if (ws) {                                                                                \
            if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               \
            if (lane == 0) ws[(k)] = wall_clock64();                                             \
        }                                                                                        \
    } while (0)
#else
#define CHAIN_STAMP(k) do { } while (0)
#define CHAIN_WSTAMP(k, drain) do { } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// Operator words are requested through BUFFER resources: (array base in four scalar registers) + (the lane's own 32-bit byte offset) + (a wave-uniform scalar offset: slice /
// running quad) + immediate.  The plain pointer form `base + uniform + lane offset` is compiled to a 64-bit vector add per request (v_lshl_add_u64 + global_load ... off) --
// two of the ~36 vector-ALU instructions of a conv quad, one of 12 of a thin quad.  A request past an array's end is inside its over-read padding, as before.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_res(const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000); }
template <typename T>
__device__ __forceinline__ T chain_ldq(const __amdgpu_buffer_rsrc_t res, const uint32_t lane_off, const uint32_t uniform_off) {
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(res, lane_off, uniform_off, 0));
}
// Slice records are read with SCALAR loads (the address is wave-uniform): a vector load + readfirstlane makes the wavefront wait for the load where
// the readfirstlane stands -- at the START of every slice, for records it needs two slices later (44.3 against 45.5 us per LeNet launch).
typedef const int32_t __attribute__((address_space(4))) * chain_const_i32;

// (x0, x1) * v for the v in the LOW / HIGH half of an aligned register pair: the packed multiply picks the half with op_sel, so the four values of
// a loaded quad need no move into a pair of their own (the compiler spends a v_mov on .y / .w: two of ~24 vector instructions per quad, and the walk
// is bound by vector-instruction issue).  The same IEEE f32 multiply either way.
typedef float chain_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ chain_f32x2 chain_mul_lo(const chain_f32x2 x, const chain_f32x2 vpair) {
    chain_f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(vpair), "v"(x));
    return r;
}
__device__ __forceinline__ chain_f32x2 chain_mul_hi(const chain_f32x2 x, const chain_f32x2 vpair) {
    chain_f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(vpair), "v"(x));
    return r;
}

// The activation buffers are addressed as OFFSETS into this one LDS array, never through pointers: a pointer picked at run time
// (buf[l & 1]) loses its address space, and the compiler then reads LDS with flat_load -- slow, and counted on vmcnt AND lgkmcnt, so
// every wait for an activation would also drain the operator words requested ahead (measured: 420 cycles per non-zero instead of ~15).
extern __shared__ __attribute__((aligned(16))) float chain_lds_dyn[];
// ST = the whole 160 KiB as ONE statically sized array at LDS address 0: every LDS address in the walk is then the loaded byte offset itself.
// With the dynamically sized array the compiler adds the array's (link-time, zero) base to each of them: four `v_add 0` per quad, and a lone
// wavefront pays 5-6 cycles of issue per instruction.  Used when the key-net needs more than half of the LDS anyway (one workgroup per CU).
template <bool ST>
__device__ __forceinline__ float* chain_lds_base() {
    if constexpr (ST) {
        __shared__ __attribute__((aligned(16))) float chain_lds_all[CHAIN_LDS_BYTES / 4];
        return chain_lds_all;
    } else {
        return chain_lds_dyn;
    }
}
#define chain_lds (chain_lds_base<ST>())

// What a wavefront requests of a layer's operator BEFORE the barrier that ends the previous layer (the operator words do not depend on the activations):
// the lane records of its first two slices and the first ring of value (and, for the general walk, column) quads of the first one.  The walk starts
// with these in registers instead of with two dependent memory round trips (records, then quads: ~1.5 us per layer when exposed -- seven layers).
#ifndef KN_CHAIN_DV
#define KN_CHAIN_DV 6           // value quads in flight per wavefront of the thin walk (a multiple of 6.  LeNet forward, same box: 6 -> 37.6 us, 12 -> 38.6 us with a few
                                // spilled address registers, 18 spills the ring itself)
#endif
// wavefronts per slice of a sequential thin layer: two, one per pair of batch columns (as in chain_rows_thin).  Measured and dropped, round 6: ONE wavefront per slice for all four
// columns -- the slice's value stream loaded once instead of twice, but 21 instead of 11 vector instructions per quad on a wavefront that is alone on its SIMD: fc1 13.3 against
// 10.4 us, the LeNet forward 41.8 against 38.4 us (profiles/r06_lenet_chain_breakdown.txt).  The walk below is written for either.
static constexpr int CHAIN_SEQ_WPS = 2;
#ifndef KN_CHAIN_SEQ_DV
#define KN_CHAIN_SEQ_DV 6        // value quads in flight per wavefront of the sequential thin walk (a multiple of 3; 9 and 12 measured: no faster -- the walk does not wait for its values)
#endif
#ifndef KN_CHAIN_SEQ_PRE
#define KN_CHAIN_SEQ_PRE 4       // the wavefronts that will walk a sequential thin layer request its first KN_CHAIN_SEQ_PRE value quads BEFORE the barrier that ends the layer in front of
                                 // it (0 = none; all six do not fit the 128 registers of 16 wavefronts per CU: two of them spill, with a wait for the data in front of the barrier)
#endif
static constexpr int CHAIN_D = 4, CHAIN_NP = 2;       // ring depth / next-slice quads requested early (pattern walk, one row per lane)
static constexpr int CHAIN_D_ROWS = 2;                // ... of the general walk (what takes it are short rows of unrelated patterns -- keyed pooling: 9 entries = 3 quads -- and it
                                                      // holds a column quad per value quad: the register budget of 16 wavefronts per CU is 128)
struct ChainMeta {
    int row, row1, nq;
    uint32_t cstride_b;                    // bytes between a row's consecutive column quads (16, or 16 * 64 for per-lane columns)
    uint32_t coff, voff;                   // byte offsets of the lane's quad 0 in L.cols (or in the staged pool) / L.vals
};
struct ChainPre {
    ChainMeta m0, m1;
    f32x4 v[2 * CHAIN_D];
    i32x4 c[CHAIN_D];
};

// lane + slice record of slice s for this lane; R = rows per lane of the layer's layout, POOL = columns are read from the pool staged in LDS at
// float4 index pool4 (then coff is an LDS byte offset), else from memory
template <int R, bool POOL>
__device__ __forceinline__ ChainMeta chain_load_meta(const ChainLayerArg& L, int s, const int lane, const int pool4) {
    constexpr int RPS = 64;
    // (Measured and dropped, round 6: every workgroup walking a layer's slices in its own rotated order, so that the CUs of an XCD do not ask the L2 for the same lines at the
    // same time -- 37.6-38.1 against 37.0 us per LeNet forward on the same box; tools/micro/l2_read_rate.hip shows why: workgroups in lockstep read an L2-resident array at
    // 26.5 TB/s, staggered ones at 28.4.)
    s = s < L.n_slices ? s : L.n_slices - 1;                    // past the end: the last slice again (unused)
    const chain_const_i32 si = (chain_const_i32)(uintptr_t)(L.slice_info + 4 * s);
    const i32x4 lm = *reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(L.lane_meta) + 16u * (uint32_t)(s * RPS + lane));      // (uniform base + 32-bit offset: saddr form)
    ChainMeta m;
    m.row = lm.x;
    if constexpr (R == 2) m.row1 = lm.z;      // (one row per lane: never read)
    m.nq = si[0];
    m.cstride_b = 16u * (uint32_t)si[1];
    m.voff = 16u * (uint32_t)R * (uint32_t)lm.w;                  // (the lane's own value base: lanes whose rows carry one value sequence -- a conv channel at every interior pixel -- share a copy)
    m.coff = 16u * (uint32_t)(POOL ? (pool4 + lm.y) : lm.y);
    return m;
}

// One output row per lane, all four batch columns: acc[j] = acc[j] + v * x[j] over the row's stored non-zeros, serial in k.
// A wavefront walks its slices s = wave, wave + 16, ...; per slice the operator words arrive through a register ring of D quads per lane
// (a thin layer -- a 121-row Linear is two slices -- runs on few wavefronts: the L2 latency of its serial walk has to be covered inside the
// wavefront), and the slice after this one is requested early: its lane records two slices ahead, its first NP quads one slice ahead (the
// rows of a conv or pooling layer are 10-55 entries long: a slice is mostly latency unless the next one is already on its way).
// Rows shorter than their slice's longest row are padded with (ZERO feature, 0.0f): the zero feature is an LDS word that always holds
// +0.0, so a padded step adds +0.0 * +0.0 = +0.0 to a sum that started at +0.0 and therefore can never be -0.0 -- the sum is unchanged,
// bit for bit, and no NaN / Inf of a live activation can leak through a padded entry.  So there is no predicate anywhere in the walk.
// Column indices are stored as LDS BYTE offsets of the layer's input buffer (16 * column + buffer base): an activation read is one
// ds_read_b128 at the loaded value, no address arithmetic.
template <bool ST, int PART>          // PART: 1 = the lane records, 2 = the first ring, 3 = both
__device__ __forceinline__ void chain_rows_pre(const ChainLayerArg& L, const int wave, const int lane, ChainPre& pre) {
    constexpr int NW = CHAIN_THREADS / 64, D = CHAIN_D_ROWS;
    if (wave >= L.n_slices) return;
    if (PART & 1) {
        pre.m0 = chain_load_meta<1, false>(L, wave, lane, 0);
        pre.m1 = chain_load_meta<1, false>(L, wave + NW, lane, 0);
    }
    if (!(PART & 2)) return;
    const __amdgpu_buffer_rsrc_t cres = chain_res(L.cols), vres = chain_res(L.vals);
#pragma unroll
    for (int i = 0; i < D; i++) {
        pre.c[i] = chain_ldq<i32x4>(cres, pre.m0.coff, (uint32_t)i * pre.m0.cstride_b);
        pre.v[i] = chain_ldq<f32x4>(vres, pre.m0.voff, (uint32_t)i * (16u * (uint32_t)L.vstride));
    }
}

template <bool ST>
__device__ __forceinline__ void chain_rows(const ChainLayerArg& L, const int out_off, const int wave, const int lane, const ChainPre& pre, unsigned long long* const ws) {
    constexpr int NW = CHAIN_THREADS / 64, D = CHAIN_D_ROWS, NP = CHAIN_NP;      // (64 rows per slice = one wavefront)
    static_assert(NP <= D, "the next slice's early quads become the head of its ring");
    const int n_slices = L.n_slices;
    if (wave >= n_slices) return;
    // Operator words: the slice's offset and the running quad offset are scalar, the lane's own offset is a constant of the slice (chain_ldq) -- no 64-bit vector
    // address arithmetic and no clamp: a request past a slice's last quad reads the next slice's words (or the arrays' zero padding) and is never used.
    const __amdgpu_buffer_rsrc_t cres = chain_res(L.cols), vres = chain_res(L.vals);
    const uint32_t vstride_b = 16u * (uint32_t)L.vstride;          // bytes between a lane's consecutive value quads (wave-uniform)
    auto fetch = [&](const ChainMeta& m, const int q, i32x4& c, f32x4& v) {
        c = chain_ldq<i32x4>(cres, m.coff, (uint32_t)q * m.cstride_b);
        v = chain_ldq<f32x4>(vres, m.voff, (uint32_t)q * vstride_b);
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    CHAIN_WSTAMP(0, false);
    ChainMeta m0 = pre.m0, m1 = pre.m1;
    i32x4 c[D], cn[NP];
    f32x4 v[D], vn[NP];
#pragma unroll
    for (int i = 0; i < D; i++) {
        c[i] = pre.c[i];
        v[i] = pre.v[i];
    }
    CHAIN_WSTAMP(2, true);                                          // first ring landed
    for (int s = wave; s < n_slices; s += NW) {
#pragma unroll
        for (int i = 0; i < NP; i++) fetch(m1, i, cn[i], vn[i]);
        const ChainMeta m2 = chain_load_meta<1, false>(L, s + 2 * NW, lane, 0);
        __builtin_amdgcn_sched_barrier(0);                        // (requested now, not where they are first used)
        f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        // activations of a quad are read from LDS one quad AHEAD of their use (x double buffer): with two wavefronts on a CU (a Linear) the
        // ~100 cycles of ds_read latency per quad would otherwise sit in the serial chain of every row
        auto xread = [&](const i32x4& cq, f32x4 (&x)[4]) {
            x[0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(chain_lds) + cq.x);
            x[1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(chain_lds) + cq.y);
            x[2] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(chain_lds) + cq.z);
            x[3] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(chain_lds) + cq.w);
        };
        auto macs = [&](const f32x4 (&x)[4], const f32x4& vq) {
            const f32x2 vp[2] = {f32x2{vq.x, vq.y}, f32x2{vq.z, vq.w}};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const f32x2 x01 = {x[e].x, x[e].y}, x23 = {x[e].z, x[e].w};
                const f32x2 p01 = (e & 1) ? chain_mul_hi(x01, vp[e >> 1]) : chain_mul_lo(x01, vp[e >> 1]);
                const f32x2 p23 = (e & 1) ? chain_mul_hi(x23, vp[e >> 1]) : chain_mul_lo(x23, vp[e >> 1]);
                a01 = a01 + p01;
                a23 = a23 + p23;
            }
        };
        const int nq = m0.nq;
        f32x4 xa[4], xb[4];
        xread(c[0], xa);                                           // (a slice without entries reads the padded quad 0: unused)
        int q = 0;
        for (; q + D <= nq; q += D) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                // slot i holds quad q + i and its activations are in flight / landed; request the activations of the next quad (slot i + 1, or
                // slot 0 of the next trip -- refilled D - 1 quads ago), then the arithmetic of this quad, then refill slot i
                f32x4 (&xc)[4] = (i & 1) ? xb : xa;
                f32x4 (&xn)[4] = (i & 1) ? xa : xb;
                xread(c[(i + 1) % D], xn);
                macs(xc, v[i]);
                fetch(m0, q + D + i, c[i], v[i]);
                // pin the request HERE: left alone the scheduler sinks it towards its use D quads later (shorter live ranges), which is
                // exactly the latency cover the ring exists for (seen in the ISA: vmcnt(1) right behind the load)
                __builtin_amdgcn_sched_barrier(0);
            }
            static_assert(D % 2 == 0, "the x double buffer alternates per quad: an even ring keeps slot 0 in xa");
        }
#pragma unroll
        for (int i = 0; i < D - 1; i++) {
            if (q + i < nq) {
                f32x4 (&xc)[4] = (i & 1) ? xb : xa;
                f32x4 (&xn)[4] = (i & 1) ? xa : xb;
                xread(c[i + 1], xn);
                macs(xc, v[i]);
            }
        }
        if (m0.row >= 0) {
            f32x4 t = {a01.x, a01.y, a23.x, a23.y};
            if (L.relu) {                                          // torch relu: NaN stays NaN
                t.x = (t.x < 0.0f) ? 0.0f : t.x;
                t.y = (t.y < 0.0f) ? 0.0f : t.y;
                t.z = (t.z < 0.0f) ? 0.0f : t.z;
                t.w = (t.w < 0.0f) ? 0.0f : t.w;
            }
            *reinterpret_cast<f32x4*>(&chain_lds[out_off + 4 * m0.row]) = t;
        }
        if (s == wave) CHAIN_WSTAMP(3, false);                      // first slice done
        // the next slice becomes the current one: its early quads are the head of the ring, the rest is requested now
        m0 = m1;
        m1 = m2;
#pragma unroll
        for (int i = 0; i < NP; i++) {
            c[i] = cn[i];
            v[i] = vn[i];
        }
#pragma unroll
        for (int i = NP; i < D; i++) fetch(m0, i, c[i], v[i]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same walk with the COLUMN quads read from LDS (layers whose rows share column patterns -- the Cout rows of a conv output pixel -- and
// whose pool of patterns fits beside the activations: LeNet's conv1 38 KB, conv2 41 KB).  Why: a wave-wide 16-byte load costs the CU's one
// texture addresser 16 cycles whatever the number of distinct addresses, so the column quad (the same 16 bytes for all lanes of a pattern)
// cost as much as the value quad -- 32 addresser cycles per wavefront and quad, 128 per SIMD with four SIMDs sharing it, against 64 cycles
// of arithmetic: the conv layers ran at the addresser's rate (conv2: 157 k non-zeros x 8 B / 64 B per clock = 8.2 us of its 15.6).  Staged
// once per launch and layer by all sixteen wavefronts (one pass through the addresser instead of one per lane), a pattern's quad is a
// broadcast ds_read_b128; the activation reads run one quad behind the column reads and one ahead of the arithmetic, as in the thin walk.
//
// R = 2 (round 5): a lane owns TWO output rows of one column pattern (two output channels of a conv pixel).  With one row per lane the walk is bound
// by the CU's LDS read port, not by arithmetic: every stored non-zero costs its lane one 16-byte activation read, i.e. 5 wave-wide ds_read_b128
// (8 clocks each of the one 128 B / clock port) per quad and wavefront = 40 clocks, against 16 clocks of the CU's four SIMDs for the quad's 16 packed
// multiplies / adds -- LeNet conv2: 50 slices x 13 quads x 40 clocks = 10.8 us of the 12.0 it took.  Two rows per lane share every activation read:
// the same LDS clocks now carry twice the arithmetic (conv2 12.0 -> see profiles/r05_lenet_chain_breakdown.txt).  Each row still sums its own stored
// sequence serially, multiply then add: same bits.
template <bool ST, int R, int PART>
__device__ __forceinline__ void chain_rows_cl_pre(const ChainLayerArg& L, const int wave, const int lane, ChainPre& pre) {
    constexpr int NW = CHAIN_THREADS / 64, D = (R == 2) ? 2 : CHAIN_D;      // (R = 2: a quad is 32 bytes per lane -- the same bytes in flight with half the ring)
    if (wave >= L.n_slices) return;
    if (PART & 1) {
        pre.m0 = chain_load_meta<R, true>(L, wave, lane, L.stage_off);
        pre.m1 = chain_load_meta<R, true>(L, wave + NW, lane, L.stage_off);
    }
    if (!(PART & 2)) return;
    const __amdgpu_buffer_rsrc_t vres = chain_res(L.vals);
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int r = 0; r < R; r++) pre.v[i * R + r] = chain_ldq<f32x4>(vres, pre.m0.voff + 16u * r, (uint32_t)i * (16u * R * (uint32_t)L.vstride));
}

template <bool ST, int R>
__device__ __forceinline__ void chain_rows_cl(const ChainLayerArg& L, const int out_off, const int wave, const int lane, const ChainPre& pre, unsigned long long* const ws) {
    constexpr int NW = CHAIN_THREADS / 64, D = (R == 2) ? 2 : CHAIN_D, NP = (R == 2) ? 1 : CHAIN_NP;
    static_assert(NP <= D && D % 2 == 0, "ring handover / x double buffer");
    const int n_slices = L.n_slices;
    if (wave >= n_slices) return;
    const __amdgpu_buffer_rsrc_t vres = chain_res(L.vals);
    const uint32_t vstride_b = 16u * R * (uint32_t)L.vstride;      // bytes between a lane's consecutive value quads (wave-uniform)
    auto ldv = [&](const ChainMeta& m, const int q, const int r) { return chain_ldq<f32x4>(vres, m.voff + 16u * r, (uint32_t)q * vstride_b); };
    auto ldc = [&](const ChainMeta& m, const int q) { return *reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(chain_lds) + m.coff + 16u * (uint32_t)q); };     // (the pool is padded: quads past a row's end are readable)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    ChainMeta m0 = pre.m0, m1 = pre.m1;
    f32x4 v[D][R], vn[NP][R];
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int r = 0; r < R; r++) v[i][r] = pre.v[i * R + r];
    int si = 0;                                                   // (slice counter: timestamps of the diagnostic build only)
    for (int s = wave; s < n_slices; s += NW, si++) {
        if (si < 8) CHAIN_WSTAMP(8 + 4 * si + 0, false);
#pragma unroll
        for (int i = 0; i < NP; i++)
#pragma unroll
            for (int r = 0; r < R; r++) vn[i][r] = ldv(m1, i, r);
        const ChainMeta m2 = chain_load_meta<R, true>(L, s + 2 * NW, lane, L.stage_off);
        __builtin_amdgcn_sched_barrier(0);                        // (requested now, not where they are first used)
        f32x2 a01[R], a23[R];
#pragma unroll
        for (int r = 0; r < R; r++) a01[r] = a23[r] = f32x2{0.f, 0.f};
        auto xread = [&](const i32x4& cq, f32x4 (&x)[4]) {
            x[0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(chain_lds) + cq.x);
            x[1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(chain_lds) + cq.y);
            x[2] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(chain_lds) + cq.z);
            x[3] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(chain_lds) + cq.w);
        };
        auto macs = [&](const f32x4 (&x)[4], const f32x4 (&vq)[R]) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const f32x2 x01 = {x[e].x, x[e].y}, x23 = {x[e].z, x[e].w};
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const f32x2 vp = (e >> 1) ? f32x2{vq[r].z, vq[r].w} : f32x2{vq[r].x, vq[r].y};
                    const f32x2 p01 = (e & 1) ? chain_mul_hi(x01, vp) : chain_mul_lo(x01, vp);
                    const f32x2 p23 = (e & 1) ? chain_mul_hi(x23, vp) : chain_mul_lo(x23, vp);
                    a01[r] = a01[r] + p01;
                    a23[r] = a23[r] + p23;
                }
            }
        };
        const int nq = m0.nq;
        i32x4 c1 = ldc(m0, 1);                                    // columns two quads ahead, activations one quad ahead
        f32x4 xa[4], xb[4];
        {
            const i32x4 c0 = ldc(m0, 0);
            xread(c0, xa);                                         // (a slice without entries reads the padded quad 0: unused)
        }
        int q = 0;
#ifdef KN_ABLATION
        if (ws && si < 8) {                                        // when the slice's first activations have landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            CHAIN_WSTAMP(8 + 4 * si + 1, false);
        }
#endif
        for (; q + D <= nq; q += D) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                f32x4 (&xc)[4] = (i & 1) ? xb : xa;
                f32x4 (&xn)[4] = (i & 1) ? xa : xb;
                const i32x4 c2 = ldc(m0, q + i + 2);
                xread(c1, xn);
                __builtin_amdgcn_sched_barrier(0);                 // next quad's LDS reads in flight under this quad's arithmetic
                macs(xc, v[i]);
#pragma unroll
                for (int r = 0; r < R; r++) v[i][r] = ldv(m0, q + D + i, r);
                __builtin_amdgcn_sched_barrier(0);                 // the value request stays HERE (see chain_rows)
                c1 = c2;
            }
        }
#pragma unroll
        for (int i = 0; i < D - 1; i++) {
            if (q + i < nq) {
                f32x4 (&xc)[4] = (i & 1) ? xb : xa;
                f32x4 (&xn)[4] = (i & 1) ? xa : xb;
                const i32x4 c2 = ldc(m0, q + i + 2);
                xread(c1, xn);
                __builtin_amdgcn_sched_barrier(0);
                macs(xc, v[i]);
                c1 = c2;
            }
        }
        if (si < 8) CHAIN_WSTAMP(8 + 4 * si + 2, false);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int row = r ? m0.row1 : m0.row;
            if (row >= 0) {
                f32x4 t = {a01[r].x, a01[r].y, a23[r].x, a23[r].y};
                if (L.relu) {                                      // torch relu: NaN stays NaN
                    t.x = (t.x < 0.0f) ? 0.0f : t.x;
                    t.y = (t.y < 0.0f) ? 0.0f : t.y;
                    t.z = (t.z < 0.0f) ? 0.0f : t.z;
                    t.w = (t.w < 0.0f) ? 0.0f : t.w;
                }
                *reinterpret_cast<f32x4*>(&chain_lds[out_off + 4 * row]) = t;
            }
        }
        m0 = m1;
        m1 = m2;
        // the next slice's lane record (requested a whole slice ago) is consumed HERE, before the value requests below: left to the compiler its first use
        // is the column read that opens the next slice, and the wait it places there (the counter is in-order, its count across the loop edge
        // conservative) also covers the value quads requested a few instructions earlier -- a full memory latency at every slice start
        asm volatile("" : "+v"(m0.coff), "+v"(m0.row));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NP; i++)
#pragma unroll
            for (int r = 0; r < R; r++) v[i][r] = vn[i][r];
#pragma unroll
        for (int i = NP; i < D; i++)
#pragma unroll
            for (int r = 0; r < R; r++) v[i][r] = ldv(m0, i, r);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// A THIN layer (a keyed nn.Linear: 121 rows = two slices, 785 columns) is a serial walk that few wavefronts can work on, and a wavefront alone on
// its SIMD pays ~5.5 cycles per VECTOR instruction whatever it is (tools/micro/dep_chain.hip: one quad of the general walk -- 16 packed
// multiplies / adds, 4 ds_read_b128, 2 loads, a few moves -- = 170-190 cycles; scalar instructions are free: trimming them changed nothing).
// So the walk is made of fewer vector instructions per wavefront: each slice is walked by TWO wavefronts, one per pair of batch columns
// (8 packed instructions + 4 ds_read_b64 per quad instead of 16 + 4 ds_read_b128), and the column quads come from LDS, where all sixteen
// wavefronts first stage the layer's column pool (one or two shared patterns: a few KB) -- twice, the second copy with + 8 bytes on every
// entry, so that a wavefront's activation address is the staged entry itself (no per-read address add) and only the VALUE quads are loaded from
// memory (the two wavefronts of a slice load the same values: with the columns also from memory that would double the texture addresser's
// work, which is shared by the CU -- the reason an earlier two-/four-lanes-per-row variant was slower).
// (the sequential thin walk's value requests: chain_rows_thin_seq below)
static_assert(CHAIN_SEQ_WPS == 2, "the sequential walk is written for one pair of batch columns per wavefront");
// value quad (trip base + i) of the lane: scalar offset = the trip's byte offset (+ 4096 for i >= 4), immediate = 1024 * (i & 3)
__device__ __forceinline__ f32x4 chain_seq_ldq(const __amdgpu_buffer_rsrc_t vres, const uint32_t voff, const uint32_t trip_b, const int i) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vres, voff + 1024u * (uint32_t)(i & 3), trip_b + 4096u * (uint32_t)(i >> 2), 0));
}

template <int DV, bool ST, int PART>
__device__ __forceinline__ void chain_rows_thin_pre(const ChainLayerArg& L, const int wave, const int lane, ChainPre& pre) {
    constexpr int RPS = 64;
    const int wps = L.seq_len > 0 ? CHAIN_SEQ_WPS : 2;                            // wavefronts per slice
    if (wave >= wps * L.n_slices || !(PART & 1)) return;
    const int half = wave >= L.n_slices ? 1 : 0, sl = wave - half * L.n_slices;
    const i32x4 lm = *reinterpret_cast<const i32x4*>(reinterpret_cast<const char*>(L.lane_meta) + 16u * (uint32_t)(sl * RPS + lane));
    pre.m0.row = lm.x;
    if (L.seq_len > 0) {
        // every slice of a sequential layer holds ceil(seq_len / 4) quads (chain_build_layer checks it): the lane's value base needs no slice record, so the walk's first value
        // requests do not wait for a memory round trip (the lane record is first used when the row is stored)
        pre.m0.voff = 16u * ((uint32_t)(sl * ((L.seq_len + 3) >> 2)) * RPS + (uint32_t)lane);
        if (KN_CHAIN_SEQ_PRE && (PART & 2)) {
            const __amdgpu_buffer_rsrc_t vres = chain_res(L.vals);
#pragma unroll
            for (int i = 0; i < KN_CHAIN_SEQ_PRE; i++) pre.v[i] = chain_seq_ldq(vres, pre.m0.voff, 0u, i);
        }
        return;
    }
    const i32x4 info = *reinterpret_cast<const i32x4*>(L.slice_info + 4 * sl);
    pre.m0.nq = __builtin_amdgcn_readfirstlane(info.x);
    pre.m0.voff = 16u * ((uint32_t)__builtin_amdgcn_readfirstlane(info.z) * RPS + (uint32_t)lane);
    pre.m0.coff = (uint32_t)lm.y;
    // (the value ring itself is requested by the walk: DV quads per lane held across the barrier next to the other walk kinds' words do not fit the
    // 128 registers a wavefront has at 16 wavefronts per CU -- the compiler spills them)
}

template <int DV, bool ST>
__device__ __forceinline__ void chain_rows_thin(const ChainLayerArg& L, const int out_off, const int wave, const int lane, const ChainPre& pre, unsigned long long* const ws) {
    constexpr int RPS = 64;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int pool = 4 * L.cols_quads;                                            // entries of one copy
    const int cols_off4 = L.stage_off;
    if (wave >= 2 * L.n_slices) return;
    const int half = wave >= L.n_slices ? 1 : 0;
    const int nq = pre.m0.nq;
    const __amdgpu_buffer_rsrc_t vres = chain_res(L.vals);
    const uint32_t voff = pre.m0.voff;
    const int cbase = 4 * cols_off4 + half * pool + 4 * (int)pre.m0.coff;         // int index of the row's column quad 0 in the wavefront's copy (+ 4 q)
    auto ldc = [&](const int q) { return *reinterpret_cast<const i32x4*>(&reinterpret_cast<const int*>(chain_lds)[cbase + 4 * q]); };      // (the pool is padded: requests past the row's end are readable)
    auto ldv = [&](const int q) { return chain_ldq<f32x4>(vres, voff, (uint32_t)q * (16u * RPS)); };
    auto xread = [&](const i32x4& cq, f32x2 (&x)[4]) {
        x[0] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(chain_lds) + cq.x);
        x[1] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(chain_lds) + cq.y);
        x[2] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(chain_lds) + cq.z);
        x[3] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(chain_lds) + cq.w);
    };
    f32x2 acc = {0.f, 0.f};
    auto macs = [&](const f32x2 (&x)[4], const f32x4& vq) {
        const f32x2 vp[2] = {f32x2{vq.x, vq.y}, f32x2{vq.z, vq.w}};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const f32x2 p = (e & 1) ? chain_mul_hi(x[e], vp[e >> 1]) : chain_mul_lo(x[e], vp[e >> 1]);
            acc = acc + p;
        }
    };
    f32x4 v[DV];
#pragma unroll
    for (int i = 0; i < DV; i++) v[i] = ldv(i);
    __builtin_amdgcn_sched_barrier(0);
    CHAIN_WSTAMP(2, true);                                                       // records + first ring landed
    // Software pipeline of one wavefront alone on its SIMD: the running sum is a chain of dependent packed adds (~14 clocks each, 4 per quad: the floor of
    // this walk is ~56 clocks per quad), and nothing else may sit in that chain.  Column quads are read from LDS THREE quads ahead, the activations they
    // address TWO quads ahead (an LDS round trip is 75-100 clocks -- more than one quad's arithmetic: one quad ahead, as in rounds 3-4, left ~40 clocks
    // of every quad exposed), values DV quads ahead.
    f32x2 x[3][4];
    i32x4 cq[2];                                   // cq[k & 1] = column quad k + 2 (read while quad k runs; consumed by the activation request of quad k + 1's turn)
    {
        const i32x4 c0 = ldc(0), c1 = ldc(1);
        cq[0] = ldc(2);
        xread(c0, x[0]);
        xread(c1, x[1]);
    }
    static_assert(DV % 6 == 0, "the activation buffers rotate with period 3, the column quads with period 2: an unrolled trip must be a multiple of both");
    // (Measured and dropped, round 5: the quad's instructions in a hand-written order -- four products, then the four dependent adds with the quad's LDS reads
    // and value request between them -- changes nothing: 36.9-37.3 against 36.5 us per LeNet forward.  A wavefront ALONE on its SIMD pays ~9 clocks per vector
    // instruction whatever stands between them (tools/micro/dep_chain.hip: 20 instructions of a quad = 190 clocks), so fc1 costs 197 quads x 15 instructions.)
    auto quad = [&](const int i, const int k, const bool refill) {          // i: ring slot (static), k: quad index
        cq[(i + 1) & 1] = ldc(k + 3);
        xread(cq[i & 1], x[(i + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);                 // the next quads' LDS reads in flight under this quad's arithmetic
        macs(x[i % 3], v[i]);
        if (refill) v[i] = ldv(k + DV);
        __builtin_amdgcn_sched_barrier(0);                 // the value request stays HERE (see chain_rows)
    };
    int q = 0;
    for (; q + DV <= nq; q += DV) {
#pragma unroll
        for (int i = 0; i < DV; i++) quad(i, q + i, true);
    }
#pragma unroll
    for (int i = 0; i < DV - 1; i++) {
        if (q + i < nq) quad(i, q + i, false);
    }
    if (pre.m0.row >= 0) {
        f32x2 t = acc;
        if (L.relu) {                                          // torch relu: NaN stays NaN
            t.x = (t.x < 0.0f) ? 0.0f : t.x;
            t.y = (t.y < 0.0f) ? 0.0f : t.y;
        }
        *reinterpret_cast<f32x2*>(&chain_lds[out_off + 4 * pre.m0.row + 2 * half]) = t;
    }
}

// A thin layer walked SEQUENTIALLY (round 6).  All rows of a keyed nn.Linear (but the homogeneous one) share ONE stored column sequence P.  When the layer BEFORE it writes its
// output row f to LDS position pos(f) with pos(P_k) = k -- its lane records carry positions instead of rows, nothing else changes -- this layer's k-th activation sits at
// seq_base + 16 k: no column quads, no staged pool, and a quad's four activations are two ds_read2_b64 at constant offsets (the compiler merges the constant-offset reads).
// Two wavefronts per slice, one per pair of batch columns.  Same values, same order, same separate multiply / add per row: bit-identical.  The row's last 1-3 entries
// (len % 4) are a tail of their own: a padded quad would multiply whatever lies behind the pattern by 0.0f (NaN if it is not finite).  Rows that do not share P run on other
// wavefronts through the general walk (ChainArgs::LX).
//
// What the walk costs is the serial chain of adds: a packed add issues ~15 clocks after the add it depends on, ONE other instruction between them is free, every further one
// adds its ~4 clocks of issue, and every 64-lane dword a memory instruction returns into registers another ~4 (tools/micro/seq_walk_sched.hip).  So the arithmetic of quad k
// is ONE block of eight instructions in which the four adds of quad k - 1's products alternate with the four multiplies of quad k,
//     acc += p0;  p0 = v0 * x0;   acc += p1;  p1 = v1 * x1;   acc += p2;  p2 = v2 * x2;   acc += p3;  p3 = v3 * x3
// (the same multiplies and adds in the same order per row -- products rounded, then added one by one -- but no add directly behind the add it depends on, where gfx950 wants a
// wait state: the compiler's `s_nop 0`), and the value quads come through a buffer resource (per-lane 32-bit offset + scalar offset + immediate: no 64-bit vector add per
// quad).  12 issued vector / memory instructions per quad instead of 18; fc1 of the LeNet 10.6 -> 8.5 us.  Dealing the two LDS reads into the gaps between the adds was
// measured too: slower (9.8 us), as the micro-benchmark says.
template <int DV, bool ST>
__device__ __forceinline__ void chain_rows_thin_seq(const ChainLayerArg& L, const int out_off, const int wave, const int lane, const ChainPre& pre, unsigned long long* const ws) {
    constexpr int RPS = 64;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    static_assert(16u * RPS == 1024u, "a slice's quads are 1 KiB apart");
    static_assert(DV % 3 == 0 && KN_CHAIN_SEQ_PRE <= DV && KN_CHAIN_SEQ_PRE <= 2 * CHAIN_D, "the activation buffers rotate with period 3; part of the ring arrives in ChainPre::v");
    if (wave >= 2 * L.n_slices) return;
    const int half = wave >= L.n_slices ? 1 : 0;
    const int nq = L.seq_len >> 2, rem = L.seq_len & 3;                           // full quads, tail entries (wave-uniform)
    const uint32_t voff = pre.m0.voff;
    const __amdgpu_buffer_rsrc_t vres = chain_res(L.vals);
    uint32_t xtrip = (uint32_t)L.seq_base + 8u * (uint32_t)half;                  // LDS byte address of the trip's first activation word (+ 64 per quad, + 16 per entry)
    auto xread = [&](const uint32_t a, f32x2 (&x)[4]) {
#pragma unroll
        for (int e = 0; e < 4; e++) x[e] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(chain_lds) + a + 16u * (uint32_t)e);
    };
    f32x4 v[DV];
#pragma unroll
    for (int i = 0; i < DV; i++) v[i] = i < KN_CHAIN_SEQ_PRE ? pre.v[i] : chain_seq_ldq(vres, voff, 0u, i);   // (the first KN_CHAIN_SEQ_PRE: requested before the barrier that ended the previous layer)
    __builtin_amdgcn_sched_barrier(0);
    CHAIN_WSTAMP(2, true);
    f32x2 x[3][4];                                                                // activations two quads ahead of the arithmetic, values DV quads ahead
    xread(xtrip, x[0]);
    xread(xtrip + 64u, x[1]);
    f32x2 acc = {0.f, 0.f}, p0 = {0.f, 0.f}, p1 = {0.f, 0.f}, p2 = {0.f, 0.f}, p3 = {0.f, 0.f};      // (the first block adds +0 four times to +0: still +0)
    uint32_t trip_b = 0;                                                          // byte offset of the trip's first value quad
    auto quad = [&](const int i, const bool refill) {
        xread(xtrip + 64u * (uint32_t)(i + 2), x[(i + 2) % 3]);                   // quad k + 2 (reads past the pattern's end land inside LDS and are never used)
        __builtin_amdgcn_sched_barrier(0);
        const f32x2 vlo = {v[i].x, v[i].y}, vhi = {v[i].z, v[i].w};
        asm volatile("v_pk_add_f32 %0, %0, %1\n\t"
                     "v_pk_mul_f32 %1, %9, %5 op_sel_hi:[0,1]\n\t"
                     "v_pk_add_f32 %0, %0, %2\n\t"
                     "v_pk_mul_f32 %2, %9, %6 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
                     "v_pk_add_f32 %0, %0, %3\n\t"
                     "v_pk_mul_f32 %3, %10, %7 op_sel_hi:[0,1]\n\t"
                     "v_pk_add_f32 %0, %0, %4\n\t"
                     "v_pk_mul_f32 %4, %10, %8 op_sel:[1,0] op_sel_hi:[1,1]"
                     : "+v"(acc), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)
                     : "v"(x[i % 3][0]), "v"(x[i % 3][1]), "v"(x[i % 3][2]), "v"(x[i % 3][3]), "v"(vlo), "v"(vhi));
        if (refill) v[i] = chain_seq_ldq(vres, voff, trip_b + 1024u * DV, i);
        __builtin_amdgcn_sched_barrier(0);
    };
    int q = 0;
    for (; q + DV <= nq; q += DV) {
#pragma unroll
        for (int i = 0; i < DV; i++) quad(i, true);
        xtrip += 64u * DV;
        trip_b += 1024u * DV;
    }
    int done = 0;                                                                 // quads of the last, partial trip
#pragma unroll
    for (int i = 0; i < DV - 1; i++) {
        if (q + i < nq) {
            quad(i, false);
            done = i + 1;
        }
    }
    acc = acc + p0;                                                               // the last full quad's products
    acc = acc + p1;
    acc = acc + p2;
    acc = acc + p3;
    if (rem) {                                                                    // the pattern's last 1-3 entries: quad nq, entries e < rem only
        // (slot `done` of the ring holds quad nq's values: requested DV quads ago, or by the prologue; its activations were read two quads ago into x[done % 3] -- all three
        // buffers are indexed statically below)
#pragma unroll
        for (int i = 0; i < DV; i++) {
            if (i == done) {
                const f32x2 vp[2] = {f32x2{v[i].x, v[i].y}, f32x2{v[i].z, v[i].w}};
#pragma unroll
                for (int e = 0; e < 3; e++)
                    if (e < rem) acc = acc + ((e & 1) ? chain_mul_hi(x[i % 3][e], vp[e >> 1]) : chain_mul_lo(x[i % 3][e], vp[e >> 1]));      // the products, then their adds (separate rounding)
            }
        }
    }
    if (pre.m0.row >= 0) {
        f32x2 t = acc;
        if (L.relu) {                                          // torch relu: NaN stays NaN
            t.x = (t.x < 0.0f) ? 0.0f : t.x;
            t.y = (t.y < 0.0f) ? 0.0f : t.y;
        }
        *reinterpret_cast<f32x2*>(&chain_lds[out_off + 4 * pre.m0.row + 2 * half]) = t;
    }
}

// Column pool of a layer into its LDS staging area: a pattern layer's quads as they are, a thin layer's twice (the second copy with + 8 bytes on every
// entry: the address of the wavefront's column pair).  `q` = a quad already loaded from L.cols + 4 * i.
template <bool ST>
__device__ __forceinline__ void chain_stage_quad(const ChainLayerArg& L, const int i, const i32x4 q) {
    *reinterpret_cast<i32x4*>(&chain_lds[4 * (L.stage_off + i)]) = q;
    if (L.cols_quads > 0) *reinterpret_cast<i32x4*>(&chain_lds[4 * (L.stage_off + L.cols_quads + i)]) = q + 8;
}
template <bool ST>
__device__ __forceinline__ void chain_stage_now(const ChainLayerArg& L, const int tid) {
    const int n = L.cols_quads > 0 ? L.cols_quads : -L.cols_quads;
    for (int i = tid; i < n; i += CHAIN_THREADS) chain_stage_quad<ST>(L, i, *reinterpret_cast<const i32x4*>(L.cols + 4 * i));
}

template <bool ST, int PART>
__device__ __forceinline__ void chain_pre(const ChainLayerArg& L, const int wave, const int lane, ChainPre& pre) {
    if (L.cols_quads > 0 || L.seq_len > 0) chain_rows_thin_pre<KN_CHAIN_DV, ST, PART>(L, wave, lane, pre);
    else if (L.cols_quads < 0 && L.rpl == 2) chain_rows_cl_pre<ST, 2, PART>(L, wave, lane, pre);
    else if (L.cols_quads < 0) chain_rows_cl_pre<ST, 1, PART>(L, wave, lane, pre);
    else chain_rows_pre<ST, PART>(L, wave, lane, pre);
}

template <bool ST>
__global__ __launch_bounds__(CHAIN_THREADS) void chain_kernel(ChainArgs a) {
    const int boff[2] = {0, 4 * a.buf1_off};       // float offsets of the two activation buffers
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // column group of this workgroup: the eight groups that share one 128-byte line of a feature row go to ONE XCD (blockIdx % 8 labels
    // the XCD), i.e. XCD x owns the contiguous range [x * chunk, (x + 1) * chunk) -- dealt round robin, every input line would be fetched
    // into all eight L2s (measured on a 4705-feature input: +27 us)
    const int64_t n_grp = (a.n_vecs + CHAIN_BT - 1) / CHAIN_BT;
    const int64_t chunk = (((n_grp + 7) >> 3) + 7) & ~(int64_t)7;
    const int64_t grp = (int64_t)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= chunk || grp >= n_grp) return;
    const int64_t c0 = grp * CHAIN_BT;
    CHAIN_STAMP(0);
    const bool full = (c0 + CHAIN_BT <= a.n_vecs) && (a.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
    for (int f = tid; f < a.n_in; f += CHAIN_THREADS) {
        const float* src = a.X + (int64_t)f * a.ldx + c0;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (full) {
            v = *reinterpret_cast<const f32x4*>(src);
        } else {
            if (c0 + 0 < a.n_vecs) v.x = src[0];
            if (c0 + 1 < a.n_vecs) v.y = src[1];
            if (c0 + 2 < a.n_vecs) v.z = src[2];
            if (c0 + 3 < a.n_vecs) v.w = src[3];
        }
        *reinterpret_cast<f32x4*>(&chain_lds[4 * f]) = v;
    }
    if (tid == 0) *reinterpret_cast<f32x4*>(&chain_lds[4 * a.zero_off]) = f32x4{0.f, 0.f, 0.f, 0.f};      // what padded operator entries read
    // the first layer's column pool and first operator words travel with the input (nothing here depends on it)
    if (a.L[0].cols_quads != 0 && a.L[0].early) chain_stage_now<ST>(a.L[0], tid);
#ifndef KN_CHAIN_PRE
#define KN_CHAIN_PRE 0          // 1 = a layer's first operator words are requested before the barrier that ends the previous layer.  Built and measured (round 5):
                                // the words held across the barrier push the kernel over its 128 registers (16 wavefronts per CU), the spills cost more than the
                                // hidden round trips gain: 42.4 against 39.0 us per LeNet forward.  Kept as a build switch for a device with more registers per wavefront.
#endif
    ChainPre pre;
    if (KN_CHAIN_PRE) chain_pre<ST, (KN_CHAIN_PRE == 2 ? 1 : 3)>(a.L[0], wave, lane, pre);
    __syncthreads();
    CHAIN_STAMP(1);
    for (int l = 0; l < a.n_layers; l++) {
        const ChainLayerArg& L = a.L[l];
#ifdef KN_ABLATION
        unsigned long long* const ws = a.wstamps ? a.wstamps + (((size_t)blockIdx.x * CHAIN_MAX_LAYERS + l) * 16 + wave) * 40 : nullptr;
#else
        constexpr unsigned long long* ws = nullptr;
#endif
        const int out_off = (l & 1) ? boff[0] : boff[1];         // (the input buffer's base is folded into the stored column offsets)
        // The NEXT layer's column pool is staged by every wavefront right behind its own walk of THIS layer (into an area this layer does not read:
        // chain_create places it; the barrier that ends this layer publishes it): the wavefronts finish their walks a few microseconds apart, so all but
        // the last one hide the round trip, and the layer no longer starts with a stage + barrier of its own.
        const bool stage_next = (l + 1 < a.n_layers) && a.L[l + 1].cols_quads != 0 && a.L[l + 1].early;
        if (L.cols_quads != 0 && !L.early) {                     // a pool that found no free area while the previous layer ran: staged here, as before
            chain_stage_now<ST>(L, tid);
            __syncthreads();
        }
        // A sequential thin layer has two parts: wavefronts [0, CHAIN_SEQ_WPS n_slices) walk its main-pattern rows, the ones behind them its rows of other patterns (a keyed Linear's
        // homogeneous row) as a general-walk layer of their own (a.LX[l]).  One dispatch below serves both: (W, w) = the layer record and wavefront index this wavefront works on.
        const bool extra = L.seq_len > 0 && wave >= CHAIN_SEQ_WPS * L.n_slices;
        const ChainLayerArg& W = extra ? a.LX[l] : L;
        const int w = extra ? wave - CHAIN_SEQ_WPS * L.n_slices : wave;
        const bool seq_main = KN_CHAIN_SEQ_PRE && L.seq_len > 0 && !extra && l > 0;      // (its lane record and first value quads were requested before the barrier: below)
        if (KN_CHAIN_PRE == 0 && !seq_main) chain_pre<ST, 3>(W, w, lane, pre);
        if (KN_CHAIN_PRE == 2) chain_pre<ST, 2>(W, w, lane, pre);               // (the lane records crossed the barrier; the ring is requested here)
        if (W.seq_len > 0) chain_rows_thin_seq<KN_CHAIN_SEQ_DV, ST>(W, out_off, w, lane, pre, ws);
        else if (W.cols_quads > 0) chain_rows_thin<KN_CHAIN_DV, ST>(W, out_off, w, lane, pre, ws);
        else if (W.cols_quads < 0 && W.rpl == 2) chain_rows_cl<ST, 2>(W, out_off, w, lane, pre, ws);
        else if (W.cols_quads < 0) chain_rows_cl<ST, 1>(W, out_off, w, lane, pre, ws);
        else chain_rows<ST>(W, out_off, w, lane, pre, ws);
        CHAIN_WSTAMP(6, false);
        if (stage_next) chain_stage_now<ST>(a.L[l + 1], tid);
        // the next layer's first operator words: on their way across the barrier.  (A fresh object per layer: what a wavefront without a slice, or a walk
        // kind that needs fewer words, leaves unwritten is then undefined rather than the previous layer's values kept alive through the walk.)
        ChainPre nxt;
        if (KN_CHAIN_PRE && l + 1 < a.n_layers) chain_pre<ST, (KN_CHAIN_PRE == 2 ? 1 : 3)>(a.L[l + 1], wave, lane, nxt);
        // a sequential thin layer next: its value quads depend on nothing this layer computes, and the few wavefronts that will walk it hold nothing else now
        if (KN_CHAIN_SEQ_PRE && KN_CHAIN_PRE == 0 && l + 1 < a.n_layers && a.L[l + 1].seq_len > 0) chain_rows_thin_pre<KN_CHAIN_DV, ST, 3>(a.L[l + 1], wave, lane, pre);
        __syncthreads();
        if (KN_CHAIN_PRE) pre = nxt;
        CHAIN_WSTAMP(7, false);
        CHAIN_STAMP(2 + l);
    }
    const int res_off = (a.n_layers & 1) ? boff[1] : boff[0];
    for (int f = tid; f < a.n_out; f += CHAIN_THREADS) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&chain_lds[res_off + 4 * f]);
        float* dst = a.Y + (int64_t)f * a.ldy + c0;
        if (c0 + 0 < a.n_vecs) dst[0] = v.x;
        if (c0 + 1 < a.n_vecs) dst[1] = v.y;
        if (c0 + 2 < a.n_vecs) dst[2] = v.z;
        if (c0 + 3 < a.n_vecs) dst[3] = v.w;
    }
    CHAIN_STAMP(2 + a.n_layers);
}

// ---- host -----------------------------------------------------------------------------------------------------------------------
struct ChainDev {
    std::vector<void*> allocs;
    ChainArgs args;
    size_t lds_bytes = 0;
    size_t stream_bytes = 0;     // operator words ONE workgroup requests from L2 per forward (values, columns read from memory or staged, lane and slice records): the kernel's real
                                 // traffic is n_workgroups x this -- for LeNet 256 x 2.3 MB against an L2 that delivers 28 TB/s (tools/micro/l2_read_rate.hip)
};

void chain_free(ChainDev* c) {
    if (!c) return;
    for (void* p : c->allocs)
        if (p) (void)hipFree(p);
    delete c;
}

template <typename T>
static int chain_upload(ChainDev* c, const T** dst, const std::vector<T>& h) {
    T* d = nullptr;
    int rc = upload(&d, h.data(), h.size());
    if (rc) return rc;
    c->allocs.push_back(d);
    *dst = d;
    return KN_OK;
}

// column patterns of a layer: rows with an identical stored column sequence share one id (pat[r]); pat_rep[id] = a row that carries it
static void chain_patterns(int64_t rows, const std::vector<int32_t>& ip, const std::vector<int32_t>& ix, std::vector<int32_t>& pat, std::vector<int32_t>& pat_rep) {
    pat.assign((size_t)rows, -1);
    pat_rep.clear();
    std::unordered_map<uint64_t, std::vector<int32_t>> buckets;
    for (int64_t r = 0; r < rows; r++) {
        const int32_t s = ip[(size_t)r], e = ip[(size_t)r + 1];
        uint64_t h = 1469598103934665603ull ^ (uint64_t)(e - s);
        for (int32_t k = s; k < e; k++) {
            h ^= (uint64_t)(uint32_t)ix[(size_t)k];
            h *= 1099511628211ull;
        }
        auto& cand = buckets[h];
        int32_t found = -1;
        for (int32_t g : cand) {
            const int32_t rs = ip[(size_t)pat_rep[(size_t)g]], re = ip[(size_t)pat_rep[(size_t)g] + 1];
            if (re - rs == e - s && (e == s || std::memcmp(ix.data() + rs, ix.data() + s, sizeof(int32_t) * (size_t)(e - s)) == 0)) {
                found = g;
                break;
            }
        }
        if (found < 0) {
            found = (int32_t)pat_rep.size();
            pat_rep.push_back((int32_t)r);
            cand.push_back(found);
        }
        pat[(size_t)r] = found;
    }
}

// Does a layer take the SEQUENTIAL thin walk (chain_rows_thin_seq)?  A keyed nn.Linear behind another layer: (nearly) all rows share ONE stored column sequence P of distinct
// columns, few enough rows for two wavefronts per slice, a walk long enough to be bound by one wavefront's instruction issue.  pos = where the layer BEFORE it must put each of
// its output rows: pos[P_k] = k, the features P does not name behind them.
struct ChainSeqPlan {
    bool on = false;
    int32_t len = 0;
    std::vector<int32_t> pos, main_rows, other_rows;
};
static ChainSeqPlan chain_plan_seq(int64_t l, int64_t rows, int64_t cols, const std::vector<int32_t>& ip, const std::vector<int32_t>& ix, const Tuning& tune) {
    ChainSeqPlan P;
    if (l == 0 || rows == 0 || tune.chain_no_seq) return P;              // (layer 0 reads the caller's input: staged as it is)
    std::vector<int32_t> pat, pat_rep;
    chain_patterns(rows, ip, ix, pat, pat_rep);
    std::vector<int64_t> cnt(pat_rep.size(), 0);
    for (int64_t r = 0; r < rows; r++) cnt[(size_t)pat[(size_t)r]]++;
    const int32_t g = (int32_t)(std::max_element(cnt.begin(), cnt.end()) - cnt.begin());
    const int32_t rs = ip[(size_t)pat_rep[(size_t)g]], len = ip[(size_t)pat_rep[(size_t)g] + 1] - rs;
    const int64_t n_main = cnt[(size_t)g], n_other = rows - n_main;
    const int64_t ns = (n_main + 63) / 64;
    if (len < 64 || n_other > 64 || CHAIN_SEQ_WPS * ns + (n_other > 0 ? 1 : 0) > CHAIN_THREADS / 64 || ns > 8) return P;
    std::vector<int32_t> pos((size_t)cols, -1);
    for (int32_t k = 0; k < len; k++) {
        const int32_t col = ix[(size_t)(rs + k)];
        if (col < 0 || col >= cols || pos[(size_t)col] >= 0) return P;   // a column named twice (non-canonical rows may): no sequential layout
        pos[(size_t)col] = k;
    }
    int32_t nxt = len;
    for (int64_t col = 0; col < cols; col++)
        if (pos[(size_t)col] < 0) pos[(size_t)col] = nxt++;
    P.on = true;
    P.len = len;
    P.pos = std::move(pos);
    for (int64_t r = 0; r < rows; r++) (pat[(size_t)r] == g ? P.main_rows : P.other_rows).push_back((int32_t)r);
    return P;
}

// one layer: CSR (host copy, stored order) -> the sliced layout above.  in_base / zero_byte: LDS byte offsets of the layer's input buffer
// and of the always-zero feature.  in_pos / out_pos (or null): where the layer's input features / output rows sit inside their LDS buffers when the layer itself / the layer
// behind it is walked sequentially (chain_plan_seq).
static int chain_build_layer(ChainDev* c, ChainLayerArg& L, ChainLayerArg& LX, int64_t rows, int64_t cols, const std::vector<int32_t>& ip, const std::vector<int32_t>& ix,
                             const std::vector<float>& dt, int relu, int32_t in_base, int32_t zero_byte, const Tuning& tune, const size_t room_quads,
                             const ChainSeqPlan& plan, const std::vector<int32_t>* out_pos) {
    constexpr int RPS = 64;                      // room_quads: LDS left beside the activations, in 16-byte quads (what a staged column pool may take)
    (void)cols;
    const std::vector<int32_t>* in_pos = plan.on ? &plan.pos : nullptr;
    auto off = [&](int32_t col) { return in_base + 16 * (in_pos ? (*in_pos)[(size_t)col] : col); };
    std::vector<int32_t> pat, pat_rep;
    chain_patterns(rows, ip, ix, pat, pat_rep);
    struct Built {
        std::vector<float> vals;
        std::vector<int32_t> colpool, lane_meta, info;
        int64_t n_slices = 0;
        size_t pool_quads = 0;
        bool thin = false, all_shared = false;
        int longest = 0;
        int vstride = 64;                 // see ChainLayerArg::vstride
        size_t val_line_bytes = 0;        // 64-byte lines the wavefronts' value requests of one forward touch (what they ask the L2 for)
    };
    // R output rows per lane (R = 2: two rows of ONE column pattern; a pattern with an odd number of rows leaves one lane half empty)
    // `share` (general / pattern walks): lanes whose rows carry the SAME value sequence read ONE copy of it.  A keyed conv stores one weight sequence per (output channel, border
    // class) -- LeNet conv2: 145 distinct sequences for 3 137 rows, its pooling layers 4 for 1 177 -- while the walk used to stream a private copy per lane: 1.0 of the 2.3 MB a
    // workgroup asks the L2 for per forward, in a kernel whose 256 workgroups together run at 60-70 % of the L2's measured read rate (profiles/r06_lenet_chain_breakdown.txt).
    // Layout: the lanes of one column pattern (the channels of a pixel, `U` lanes) keep their order; patterns whose lanes carry identical sequences share a block
    // [quad][lane of the pattern][row of the lane][4] -- a quad's request of such a pattern is U * 16 * R contiguous bytes, the same for every pixel of the class.  A lane's base
    // goes into its record (lane_meta.w), the quad stride U into the layer record.  Reads past a block's last quad (a shorter row in a slice of longer ones) hit the next block's
    // finite values and multiply the always-zero feature: +-0.0 onto a sum that started at +0.0, as with the 0.0f padding before.
    auto build = [&](const int R, const std::vector<int32_t>* subset = nullptr, const bool share = false) {
        Built B;
        std::vector<int32_t> rows_sorted((size_t)rows);
        std::iota(rows_sorted.begin(), rows_sorted.end(), 0);
        if (subset) rows_sorted = *subset;                       // (a sequential thin layer: its main-pattern rows and the others are laid out separately)
        std::stable_sort(rows_sorted.begin(), rows_sorted.end(), [&](int32_t x, int32_t y) {
            const int32_t lx = ip[(size_t)x + 1] - ip[(size_t)x], ly = ip[(size_t)y + 1] - ip[(size_t)y];
            if (lx != ly) return lx > ly;
            if (pat[(size_t)x] != pat[(size_t)y]) return pat[(size_t)x] < pat[(size_t)y];
            // share: the rows of one pattern (the channels of a pixel) in a canonical order -- by their value sequences -- so that two pixels whose channels carry the same
            // sequences form the same block whatever order the output key gave their rows (which lane of a pattern takes which of its rows changes no result)
            return share && lx > 0 && std::memcmp(dt.data() + ip[(size_t)x], dt.data() + ip[(size_t)y], sizeof(float) * (size_t)lx) < 0;
        });
        // units = what a lane owns: R consecutive rows of the sorted order when they share a pattern (equal pattern => equal length), else one row
        std::vector<std::array<int32_t, 2>> units;
        for (size_t i = 0; i < rows_sorted.size();) {
            std::array<int32_t, 2> u = {rows_sorted[i], -1};
            if (R == 2 && i + 1 < rows_sorted.size() && pat[(size_t)rows_sorted[i + 1]] == pat[(size_t)rows_sorted[i]]) {
                u[1] = rows_sorted[i + 1];
                i += 2;
            } else {
                i += 1;
            }
            units.push_back(u);
        }
        const int64_t n_units = (int64_t)units.size();
        const int64_t n_slices = (n_units + RPS - 1) / RPS;
        B.n_slices = n_slices;
        auto len_of = [&](int32_t r) { return ip[(size_t)r + 1] - ip[(size_t)r]; };
        B.lane_meta.assign((size_t)(n_slices * RPS) * 4, 0);
        B.info.assign((size_t)(n_slices * 4), 0);
        for (size_t o = 0; o < (size_t)(n_slices * RPS); o++) B.lane_meta[4 * o] = B.lane_meta[4 * o + 2] = -1;
        // pass 1: per slice its longest row and whether its lanes share column patterns; per shared pattern the quads it must be readable for
        // (every lane of a slice walks up to the slice's LONGEST row)
        std::vector<int> s_max((size_t)n_slices, 0);
        std::vector<char> s_shared((size_t)n_slices, 0);
        std::vector<int> pat_quads(pat_rep.size(), 0);
        for (int64_t s = 0; s < n_slices; s++) {
            int mx = 0, distinct = 0;
            int32_t last_pat = -2;
            const int real = (int)std::min<int64_t>(RPS, n_units - s * RPS);
            for (int i = 0; i < real; i++) {
                const std::array<int32_t, 2>& u = units[(size_t)(s * RPS + i)];
                mx = std::max(mx, len_of(u[0]));
                if (pat[(size_t)u[0]] != last_pat) distinct++;
                last_pat = pat[(size_t)u[0]];
                B.lane_meta[4 * (size_t)(s * RPS + i)] = u[0];
                B.lane_meta[4 * (size_t)(s * RPS + i) + 2] = u[1];
            }
            s_max[(size_t)s] = mx;
            s_shared[(size_t)s] = (distinct * 2 <= real || R == 2) ? 1 : 0;     // most lanes share a pattern with a neighbour: one copy per pattern
        }
        // a layer that is shared almost everywhere (a conv layer and its odd last slice: the homogeneous row) stores every slice that way -- one layout
        // per layer lets its whole pool be staged in LDS (chain_rows_cl); the few unrelated rows then read scattered instead of lane-adjacent quads
        {
            int64_t units_shared = 0;
            for (int64_t s = 0; s < n_slices; s++)
                if (s_shared[(size_t)s]) units_shared += std::min<int64_t>(RPS, n_units - s * RPS);
            if (units_shared * 10 >= n_units * 9)
                for (int64_t s = 0; s < n_slices; s++) s_shared[(size_t)s] = 1;
        }
        for (int64_t s = 0; s < n_slices; s++) {
            if (!s_shared[(size_t)s]) continue;
            const int real = (int)std::min<int64_t>(RPS, n_units - s * RPS);
            for (int i = 0; i < real; i++) {
                const int32_t p = pat[(size_t)units[(size_t)(s * RPS + i)][0]];
                pat_quads[(size_t)p] = std::max(pat_quads[(size_t)p], (s_max[(size_t)s] + 3) / 4);
            }
        }
        // pass 2: storage.  Padding = (zero feature, 0.0f).
        B.colpool.assign(4, zero_byte);
        std::vector<int64_t> pat_cq(pat_rep.size(), -1);         // column quad offset of a pattern stored once
        int64_t vq = 0;                                          // running value-quad offset, in units of RPS * R quads
        int U = 1;                                               // share: lanes per column pattern (the longest run of units of one pattern)
        if (share) {
            std::map<std::string, int64_t> blocks;               // value sequences of a pattern's lanes -> base of their block, in units of 16 * R bytes
            for (size_t a = 0; a < units.size();) {
                size_t b = a + 1;
                while (b < units.size() && pat[(size_t)units[b][0]] == pat[(size_t)units[a][0]]) b++;
                U = std::max(U, (int)(b - a));
                a = b;
            }
            B.vstride = U;
            int64_t next_base = 0;
            for (size_t a = 0; a < units.size();) {
                size_t b = a + 1;
                while (b < units.size() && pat[(size_t)units[b][0]] == pat[(size_t)units[a][0]]) b++;
                std::string key;
                int len = 0;
                for (size_t u = a; u < b; u++)
                    for (int rr = 0; rr < R; rr++) {
                        const int32_t r = units[u][(size_t)rr];
                        const int l = r < 0 ? 0 : len_of(r);
                        len = std::max(len, l);
                        key.append(reinterpret_cast<const char*>(&l), sizeof(l));
                        if (l > 0) key.append(reinterpret_cast<const char*>(dt.data() + ip[(size_t)r]), sizeof(float) * (size_t)l);
                    }
                auto it = blocks.find(key);
                if (it == blocks.end()) {
                    const int nqc = std::max((len + 3) / 4, 1);
                    it = blocks.emplace(key, next_base).first;
                    const size_t f0 = (size_t)next_base * 4 * (size_t)R;
                    B.vals.resize(f0 + (size_t)nqc * (size_t)U * (size_t)R * 4, 0.0f);
                    for (size_t u = a; u < b; u++)
                        for (int rr = 0; rr < R; rr++) {
                            const int32_t r = units[u][(size_t)rr];
                            if (r < 0) continue;
                            const int32_t rs = ip[(size_t)r];
                            const int l = len_of(r);
                            for (int k = 0; k < l; k++) B.vals[f0 + ((((size_t)(k >> 2) * (size_t)U + (u - a)) * (size_t)R + (size_t)rr) * 4) + (size_t)(k & 3)] = dt[(size_t)(rs + k)];
                        }
                    next_base += (int64_t)nqc * U;
                }
                for (size_t u = a; u < b; u++) B.lane_meta[4 * u + 3] = (int32_t)(it->second + (int64_t)(u - a));
                a = b;
            }
        }
        for (int64_t s = 0; s < n_slices; s++) {
            const int nq = (s_max[(size_t)s] + 3) / 4;
            const int real = (int)std::min<int64_t>(RPS, n_units - s * RPS);
            if (!share) {
                const size_t v0 = B.vals.size();
                B.vals.resize(v0 + (size_t)nq * RPS * R * 4, 0.0f);
                for (int i = 0; i < real; i++)
                    for (int rr = 0; rr < R; rr++) {
                        const int32_t r = units[(size_t)(s * RPS + i)][(size_t)rr];
                        if (r < 0) continue;
                        const int32_t rs = ip[(size_t)r];
                        const int len = len_of(r);
                        for (int k = 0; k < len; k++) B.vals[v0 + ((((size_t)(k >> 2) * RPS + (size_t)i) * R + (size_t)rr) * 4) + (size_t)(k & 3)] = dt[(size_t)(rs + k)];
                    }
                for (int i = 0; i < RPS; i++) B.lane_meta[4 * (size_t)(s * RPS + i) + 3] = (int32_t)(vq * RPS + i);      // (every lane its own copy: base = slice's first quad * 64 + lane, stride 64)
            }
            {
                // 64-byte lines one wavefront's requests of this slice touch, quad by quad (what it asks the L2 for; identical addresses and neighbours inside a line are one request)
                std::vector<int64_t> lines;
                for (int q = 0; q < nq; q++) {
                    lines.clear();
                    for (int i = 0; i < real; i++) {
                        const int64_t byte0 = (int64_t)16 * R * ((int64_t)B.lane_meta[4 * (size_t)(s * RPS + i) + 3] + (int64_t)q * B.vstride);
                        for (int64_t l = byte0 / 64; l <= (byte0 + 16 * R - 1) / 64; l++) lines.push_back(l);
                    }
                    std::sort(lines.begin(), lines.end());
                    B.val_line_bytes += 64 * (size_t)(std::unique(lines.begin(), lines.end()) - lines.begin());
                }
            }
            int cstride = 1;
            if (s_shared[(size_t)s]) {
                for (int i = 0; i < RPS; i++) {
                    const size_t o = (size_t)(s * RPS + i);
                    if (i >= real) {                             // empty slots of the last slice read along with a real row's pattern (their values are 0)
                        B.lane_meta[4 * o + 1] = B.lane_meta[4 * (size_t)(s * RPS) + 1];
                        continue;
                    }
                    const int32_t r = units[o][0];
                    const int32_t p = pat[(size_t)r];
                    if (pat_cq[(size_t)p] < 0) {
                        pat_cq[(size_t)p] = (int64_t)B.colpool.size() / 4;
                        const int32_t rs = ip[(size_t)r];
                        const int len = len_of(r);
                        const size_t c0 = B.colpool.size();
                        B.colpool.resize(c0 + (size_t)std::max(pat_quads[(size_t)p], 1) * 4, zero_byte);
                        for (int k = 0; k < len; k++) B.colpool[c0 + (size_t)k] = off(ix[(size_t)(rs + k)]);
                    }
                    B.lane_meta[4 * o + 1] = (int32_t)pat_cq[(size_t)p];
                }
            } else {
                cstride = RPS;
                const size_t c0 = B.colpool.size();
                B.colpool.resize(c0 + (size_t)std::max(nq, 1) * RPS * 4, zero_byte);
                for (int i = 0; i < real; i++) {
                    const int32_t r = units[(size_t)(s * RPS + i)][0];
                    const int32_t rs = ip[(size_t)r];
                    const int len = len_of(r);
                    for (int k = 0; k < len; k++) B.colpool[c0 + ((size_t)(k >> 2) * RPS + (size_t)i) * 4 + (size_t)(k & 3)] = off(ix[(size_t)(rs + k)]);
                }
                for (int i = 0; i < RPS; i++) B.lane_meta[4 * (size_t)(s * RPS + i) + 1] = (int32_t)(c0 / 4 + (size_t)i);
            }
            B.info[(size_t)(4 * s + 0)] = nq;
            B.info[(size_t)(4 * s + 1)] = cstride;
            B.info[(size_t)(4 * s + 2)] = (int32_t)vq;
            vq += nq;
        }
        B.vals.resize(B.vals.size() + (size_t)CHAIN_OVERREAD_QUADS * (size_t)std::max(B.vstride, RPS) * R * 4, 0.0f);     // requests run past a slice's (and the array's) last quad: readable, never used
        if (share) {
            // ... and past a BLOCK's last quad: every lane of a slice walks the slice's longest row, a short row's lane (a keyed Linear's homogeneous row in a slice of
            // 2 000-entry rows) from its own small block on through whatever follows it, `vstride` lanes per quad.  The array must reach as far as the furthest such request
            // (found by the fuzzer's keyed-Linear layers: a first layer of 336 rows x 2 179 entries read 2.8 MB past the end).
            size_t reach = 0;                                                                     // in units of 16 * R bytes
            for (int64_t s = 0; s < n_slices; s++) {
                const size_t nq = (size_t)((s_max[(size_t)s] + 3) / 4) + (size_t)CHAIN_OVERREAD_QUADS;
                const int real = (int)std::min<int64_t>(RPS, n_units - s * RPS);
                for (int i = 0; i < real; i++) reach = std::max(reach, (size_t)B.lane_meta[4 * (size_t)(s * RPS + i) + 3] + nq * (size_t)B.vstride + 1);
            }
            if (B.vals.size() < reach * R * 4) B.vals.resize(reach * R * 4, 0.0f);
        }
        B.pool_quads = B.colpool.size() / 4 + 3;                                           // what is staged: the patterns + the three quads a walk requests ahead
        B.colpool.resize(B.colpool.size() + (size_t)CHAIN_OVERREAD_QUADS * RPS * 4, zero_byte);
        // thin layer: two wavefronts per slice fit the workgroup, every slice on shared patterns, a walk long enough to be bound by one wavefront's
        // instruction issue, a column pool of a few KB (staged twice)
        B.thin = n_slices >= 1 && 2 * n_slices <= CHAIN_THREADS / 64 && B.pool_quads * 16 <= CHAIN_THIN_POOL_BYTES;
        B.all_shared = n_slices >= 1;
        for (int64_t s = 0; s < n_slices; s++) {
            B.thin = B.thin && s_shared[(size_t)s];
            B.all_shared = B.all_shared && s_shared[(size_t)s];
            B.longest = std::max(B.longest, s_max[(size_t)s]);
        }
        return B;
    };
    auto place = [&](Built& X) {                                  // lane records carry the LDS position of an output row (= the row itself unless the next layer is sequential)
        if (!out_pos) return;
        for (size_t o = 0; o + 3 < X.lane_meta.size(); o += 4)
            for (int w : {0, 2})
                if (X.lane_meta[o + (size_t)w] >= 0) X.lane_meta[o + (size_t)w] = (*out_pos)[(size_t)X.lane_meta[o + (size_t)w]];
    };
    auto upload_built = [&](ChainLayerArg& T, const Built& X, const int value_readers = 1, const bool cols_read = true) -> int {
        // what a workgroup's wavefronts ask the L2 for per forward: the 64-byte lines of their value requests (a thin layer: once per wavefront of the slice), the column pool
        // (once: staged, or walked from memory), a 16-byte lane record per lane, the slice records.  The over-read padding behind the arrays is not counted.
        const size_t pad_c = (size_t)CHAIN_OVERREAD_QUADS * 64 * 4;
        c->stream_bytes += X.val_line_bytes * (size_t)value_readers + 4 * ((cols_read && X.colpool.size() > pad_c ? X.colpool.size() - pad_c : 0) + X.lane_meta.size() + X.info.size());
        T.vstride = X.vstride;
        int rc;
        if ((rc = chain_upload(c, &T.vals, X.vals)) || (rc = chain_upload(c, &T.cols, X.colpool)) || (rc = chain_upload(c, &T.lane_meta, X.lane_meta)) ||
            (rc = chain_upload(c, &T.slice_info, X.info)))
            return rc;
        return KN_OK;
    };
    std::memset(&LX, 0, sizeof(LX));
    if (plan.on) {
        // sequential thin layer: the rows of the main pattern in slices of their own (values only: their columns are implicit), the others as a general-walk layer
        Built Bm = build(1, &plan.main_rows);
        place(Bm);
        L.n_slices = (int32_t)Bm.n_slices;
        L.n_rows = (int32_t)rows;
        L.relu = relu;
        L.cols_quads = 0;
        L.rpl = 1;
        L.stage_off = 0;
        L.early = 0;
        L.seq_len = plan.len;
        L.seq_base = in_base;
        for (int64_t sl = 0; sl < Bm.n_slices; sl++)               // (chain_rows_thin_pre computes a slice's value base instead of loading its record)
            if (Bm.info[(size_t)(4 * sl + 2)] != (int32_t)(sl * ((plan.len + 3) / 4)) || Bm.vstride != RPS) return KN_ERR_UNSUPPORTED;      // (cannot happen: equal rows, private copies)
        int rc = upload_built(L, Bm, CHAIN_SEQ_WPS, false);       // (columns implicit: nothing read)
        if (rc) return rc;
        if (!plan.other_rows.empty()) {
            Built Bx = build(1, &plan.other_rows, true);
            place(Bx);
            LX.n_slices = (int32_t)Bx.n_slices;
            LX.n_rows = (int32_t)plan.other_rows.size();
            LX.relu = relu;
            LX.rpl = 1;
            if ((rc = upload_built(LX, Bx))) return rc;
        }
        return KN_OK;
    }
    Built B = build(1);
    // (chain_create drops either choice when the staging area does not fit beside the activations)
    int32_t cols_quads = (B.thin && B.longest >= 64) ? (int32_t)B.pool_quads : (B.all_shared && !tune.chain_no_cl) ? -(int32_t)B.pool_quads : 0;
    if ((cols_quads > 0 ? 2 * (size_t)cols_quads : (size_t)(-(int64_t)cols_quads)) > room_quads) cols_quads = 0;      // no room for the pool: columns from memory (general walk)
    int rpl = 1;
    if (cols_quads < 0 && !tune.chain_no_rpl2) {
        // Two rows per lane for a pattern layer whose patterns mostly hold two rows or more (the output channels of a conv pixel): halves the LDS reads
        // per multiply-add (see chain_rows_cl).  Not when it would leave fewer slices than wavefronts of work worth having (a thin layer stays thin).
        int64_t paired = 0;
        {
            std::vector<int32_t> cnt(pat_rep.size(), 0);
            for (int64_t r = 0; r < rows; r++) cnt[(size_t)pat[(size_t)r]]++;
            for (int32_t n : cnt) paired += n / 2 * 2;
        }
        if (paired * 10 >= rows * 9 && rows >= 2 * RPS * 8) {
            Built B2 = build(2);
            if (B2.pool_quads <= room_quads) {
                B = std::move(B2);
                cols_quads = -(int32_t)B.pool_quads;
                rpl = 2;
            }
        }
    }
    if (cols_quads <= 0 && !tune.chain_no_share) B = build(rpl, nullptr, true);      // general / pattern walks: one copy per distinct value sequence (same rows, same lanes, same columns)
    place(B);
    L.n_slices = (int32_t)B.n_slices;
    L.n_rows = (int32_t)rows;
    L.relu = relu;
    L.cols_quads = cols_quads;
    L.rpl = rpl;
    L.stage_off = 0;
    L.early = 0;
    L.seq_len = 0;
    L.seq_base = 0;
    return upload_built(L, B, cols_quads > 0 ? 2 : 1);
}

int chain_create(int64_t n_ops, kn_operator* const* ops, const uint32_t* flags, ChainDev** out, int64_t* rows_out, int64_t* cols_out, int64_t* nnz_out) {
    *out = nullptr;
    KN_REQUIRE(n_ops >= 1 && n_ops <= CHAIN_MAX_LAYERS, KN_ERR_UNSUPPORTED, "a chain holds 1..12 operators");
    size_t feat[2] = {0, 0};      // features held by the even / odd activation buffer
    int64_t nnz = 0;
    for (int64_t l = 0; l < n_ops; l++) {
        KN_REQUIRE(ops[l] != nullptr && ops[l]->kind == KIND_CSR, KN_ERR_UNSUPPORTED, "chain operators must be CSR handles (kn_csr_create / kn_tiled_create)");
        KN_REQUIRE(l == 0 || ops[l]->cols == ops[l - 1]->rows, KN_ERR_SHAPE, "operator shapes do not chain");
        KN_REQUIRE(ops[l]->device == ops[0]->device, KN_ERR_INVALID, "operators live on different devices");
        feat[l & 1] = std::max(feat[l & 1], (size_t)ops[l]->cols);
        feat[(l & 1) ^ 1] = std::max(feat[(l & 1) ^ 1], (size_t)ops[l]->rows);
        nnz += ops[l]->csr.nnz;
    }
    const size_t lds = (feat[0] + feat[1] + 1) * CHAIN_BT * sizeof(float);      // two activation buffers + the always-zero feature
    KN_REQUIRE(lds <= CHAIN_LDS_BYTES, KN_ERR_UNSUPPORTED, "activations of four batch columns do not fit the CU's 160 KiB of LDS");
    {
        // the kernel is written for gfx950's 160 KiB of LDS per workgroup: ask the device rather than assume (the caller falls back to one
        // launch per layer on KN_ERR_UNSUPPORTED)
        int dev = 0, max_lds = 0;
        KN_HIP(hipGetDevice(&dev));
        KN_HIP(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
        KN_REQUIRE((size_t)max_lds >= CHAIN_LDS_BYTES, KN_ERR_UNSUPPORTED, "this device offers less than 160 KiB of LDS per workgroup: the whole-net kernel does not apply");
    }
    std::unique_ptr<ChainDev, void (*)(ChainDev*)> c(new ChainDev(), chain_free);
    std::memset(&c->args, 0, sizeof(ChainArgs));
    // pass 1: the operators on the host, and which layers are walked sequentially (that decides the LDS order of the layer before them)
    struct HostCsr {
        std::vector<int32_t> ip, ix;
        std::vector<float> dt;
    };
    std::vector<HostCsr> H((size_t)n_ops);
    std::vector<ChainSeqPlan> plans((size_t)n_ops);
    for (int64_t l = 0; l < n_ops; l++) {
        const CsrDev& A = ops[l]->csr;
        HostCsr& h = H[(size_t)l];
        h.ip.resize((size_t)A.rows + 1);
        h.ix.resize((size_t)A.nnz);
        h.dt.resize((size_t)A.nnz);
        KN_HIP(hipMemcpy(h.ip.data(), A.indptr, sizeof(int32_t) * h.ip.size(), hipMemcpyDeviceToHost));
        if (A.nnz > 0) {
            KN_HIP(hipMemcpy(h.ix.data(), A.indices, sizeof(int32_t) * h.ix.size(), hipMemcpyDeviceToHost));
            KN_HIP(hipMemcpy(h.dt.data(), A.data, sizeof(float) * h.dt.size(), hipMemcpyDeviceToHost));
        }
        plans[(size_t)l] = chain_plan_seq(l, A.rows, A.cols, h.ip, h.ix, A.tune);
    }
    // pass 2: the layouts
    for (int64_t l = 0; l < n_ops; l++) {
        const CsrDev& A = ops[l]->csr;
        const HostCsr& h = H[(size_t)l];
        const std::vector<int32_t>* out_pos = (l + 1 < n_ops && plans[(size_t)l + 1].on) ? &plans[(size_t)l + 1].pos : nullptr;
        int rc = chain_build_layer(c.get(), c->args.L[l], c->args.LX[l], A.rows, A.cols, h.ip, h.ix, h.dt, (flags && (flags[l] & KN_FLAG_RELU)) ? 1 : 0,
                                   (int32_t)((l & 1) ? 16 * feat[0] : 0), (int32_t)(16 * (feat[0] + feat[1])), A.tune, (CHAIN_LDS_BYTES - lds) / 16, plans[(size_t)l], out_pos);
        if (rc) return rc;
    }
    c->args.n_layers = (int32_t)n_ops;
    c->args.n_in = (int32_t)ops[0]->cols;
    c->args.n_out = (int32_t)ops[n_ops - 1]->rows;
    c->args.buf1_off = (int32_t)feat[0];
    c->args.zero_off = (int32_t)(feat[0] + feat[1]);
    // Staging areas of the column pools, behind the activations.  A layer whose pool does not fit beside the activations reads its columns from memory
    // (general walk).  A pool is staged EARLY -- while the previous layer runs (layer 0: with the input) -- when it can have an area that layer does not
    // read: the base area when the previous layer has no pool, else right behind the previous layer's area when that still fits.
    const size_t base4 = feat[0] + feat[1] + 1;                   // float4 index
    const size_t limit4 = CHAIN_LDS_BYTES / 16;
    size_t stage_end4 = base4;
    size_t prev_end4 = base4;                                      // end of the area the previous layer reads (base4: none)
    for (int64_t l = 0; l < n_ops; l++) {
        ChainLayerArg& L = c->args.L[l];
        const size_t need = L.cols_quads > 0 ? 2 * (size_t)L.cols_quads : (size_t)(-(int64_t)L.cols_quads);
        if (need == 0) {
            prev_end4 = base4;
            continue;
        }
        KN_REQUIRE(base4 + need <= limit4, KN_ERR_INVALID, "internal: a column pool larger than the room chain_build_layer was given");
        const bool small = !ops[l]->csr.tune.chain_no_early;      // (name kept: every pool qualifies for early staging)
        if (small && prev_end4 + need <= limit4) {
            L.stage_off = (int32_t)prev_end4;
            L.early = 1;
        } else {
            L.stage_off = (int32_t)base4;
            L.early = (small && prev_end4 == base4) ? 1 : 0;
        }
        prev_end4 = (size_t)L.stage_off + need;
        stage_end4 = std::max(stage_end4, prev_end4);
    }
    const size_t stage_quads = stage_end4 - base4;
    c->lds_bytes = lds + stage_quads * 16;
    KN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS_BYTES));
    *rows_out = ops[n_ops - 1]->rows;
    *cols_out = ops[0]->cols;
    *nnz_out = nnz;
    *out = c.release();
    return KN_OK;
}

int chain_forward(const ChainDev* c, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, hipStream_t s) {
    ChainArgs a = c->args;
    a.X = x;
    a.Y = y;
    a.ldx = ldx;
    a.ldy = ldy;
    a.n_vecs = (int32_t)n_vecs;
    const int64_t n_grp = (n_vecs + CHAIN_BT - 1) / CHAIN_BT;
    const int64_t grid = 8 * ((((n_grp + 7) >> 3) + 7) & ~(int64_t)7);      // 8 XCD lanes x a chunk rounded to whole 128-byte lines (idle workgroups return at once)
#ifdef KN_ABLATION
    // diagnostic build only: KN_CHAIN_STAMPS=<file> makes every launch synchronous and dumps the timestamps of the LAST launch
    const char* stamp_path = getenv("KN_CHAIN_STAMPS");
    a.stamps = a.wstamps = nullptr;
    const size_t n_stamps = (size_t)grid * 16 + (getenv("KN_CHAIN_WSTAMPS") ? (size_t)grid * CHAIN_MAX_LAYERS * 16 * 40 : 0);
    if (stamp_path && !plan_sink()) {
        KN_HIP(hipMalloc((void**)&a.stamps, n_stamps * sizeof(unsigned long long)));
        KN_HIP(hipMemsetAsync(a.stamps, 0, n_stamps * sizeof(unsigned long long), s));
        if (n_stamps > (size_t)grid * 16) a.wstamps = a.stamps + (size_t)grid * 16;
    }
#endif
    int n_thin = 0, n_cl = 0, n_rpl2 = 0, n_early = 0, n_seq = 0;
    for (int l = 0; l < a.n_layers; l++) {
        n_seq += a.L[l].seq_len > 0 ? 1 : 0;
        n_thin += (a.L[l].cols_quads > 0 || a.L[l].seq_len > 0) ? 1 : 0;
        n_cl += a.L[l].cols_quads < 0 ? 1 : 0;
        n_rpl2 += a.L[l].rpl == 2 ? 1 : 0;
        n_early += (a.L[l].cols_quads != 0 && a.L[l].early) ? 1 : 0;
    }
    const std::string d = "chain_kernel<" + std::to_string(a.n_layers) + " operators (" + std::to_string(n_thin) + " on the thin walk -- " + std::to_string(n_seq) + " of them sequentially, " + std::to_string(n_cl) +
                          " with column patterns in LDS), " + std::to_string(n_rpl2) + " with two rows per lane, " + std::to_string(n_early) +
                          " column pools staged a layer early, 4 batch columns per workgroup, " + std::to_string(c->lds_bytes) + " B LDS, " + std::to_string(c->stream_bytes) +
                          " B of operator words per workgroup from L2>";
    if (2 * c->lds_bytes > CHAIN_LDS_BYTES) KN_LAUNCH(d, chain_kernel<true>, dim3((unsigned)grid), dim3(CHAIN_THREADS), 0, s, a);       // one workgroup per CU either way
    else KN_LAUNCH(d, chain_kernel<false>, dim3((unsigned)grid), dim3(CHAIN_THREADS), c->lds_bytes, s, a);
    KN_HIP(hipGetLastError());
#ifdef KN_ABLATION
    if (a.stamps) {
        std::vector<unsigned long long> h(n_stamps);
        KN_HIP(hipStreamSynchronize(s));
        KN_HIP(hipMemcpy(h.data(), a.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        KN_HIP(hipFree(a.stamps));
        if (FILE* f = fopen(stamp_path, "wb")) {
            const unsigned long long hdr[2] = {(unsigned long long)grid, (unsigned long long)(a.wstamps ? 1 : 0)};
            fwrite(hdr, sizeof(unsigned long long), 2, f);
            fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
            fclose(f);
        }
    }
#endif
    return KN_OK;
}

}  // namespace kn
