// kn_internal.h -- private to libkeynet_hip.so (gfx950 only; no CUDA / multi-backend paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <mutex>
#include <map>
#include <new>
#include <stdexcept>
#include <memory>
#include <type_traits>
#include "../../include/keynet_hip.h"

#ifdef KN_HOST_PACK_ONLY
// DIAGNOSTIC BUILD ONLY (tests/test_host_sanitize.py; never the product library): the HOST side of this library -- validation of
// caller-sized arrays, the std::vector index work that packs the reference's containers into the device formats, export, destroy --
// compiled with -fsanitize=address,undefined and run on a box WITHOUT a GPU.  "Device" memory is host heap here, so every create path
// runs to completion under the sanitizers (the packing loops, the copies INTO the packed buffers and the frees included); every compute
// entry point returns KN_ERR_NODEVICE before touching anything.
#include <cstdlib>
#include <cstring>
namespace kn {
namespace hostonly {
inline hipError_t Malloc(void** p, size_t n) {
    *p = std::malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t Free(void* p) {
    std::free(p);
    return hipSuccess;
}
inline hipError_t Memcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    std::memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t Memset(void* d, int v, size_t n) {
    std::memset(d, v, n);
    return hipSuccess;
}
inline hipError_t GetDeviceCount(int* n) {
    *n = 1;
    return hipSuccess;
}
inline hipError_t GetDevice(int* d) {
    *d = 0;
    return hipSuccess;
}
inline hipError_t DeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
    *v = 160 * 1024;
    return hipSuccess;
}
inline hipError_t Ok() { return hipSuccess; }
}  // namespace hostonly
}  // namespace kn
#define hipMalloc(p, n) kn::hostonly::Malloc((void**)(p), (n))
#define hipFree(p) kn::hostonly::Free((void*)(p))
#define hipMemcpy(d, s, n, k) kn::hostonly::Memcpy((void*)(d), (const void*)(s), (n), (k))
#define hipMemset(d, v, n) kn::hostonly::Memset((void*)(d), (v), (n))
#define hipGetDeviceCount(n) kn::hostonly::GetDeviceCount(n)
#define hipGetDevice(d) kn::hostonly::GetDevice(d)
#define hipDeviceGetAttribute(v, a, d) kn::hostonly::DeviceGetAttribute((v), (a), (d))
#define hipFuncSetAttribute(f, a, v) kn::hostonly::Ok()
#define hipGetLastError() kn::hostonly::Ok()
#define KN_HOST_ONLY_GUARD() return kn::fail(KN_ERR_NODEVICE, "KN_HOST_PACK_ONLY diagnostic build: no compute entry point runs")
#else
#define KN_HOST_ONLY_GUARD() do { } while (0)
#endif

namespace kn {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define KN_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            return kn::fail(KN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
        }                                                                                              \
    } while (0)

#define KN_REQUIRE(cond, code, msg)                                                                    \
    do {                                                                                               \
        if (!(cond)) return kn::fail((code), std::string(msg) + " [" #cond "]");                       \
    } while (0)

// Every extern "C" body runs inside guarded(): the create functions do std::vector work on caller-sized inputs, and no C++ exception
// may unwind through the C ABI into ctypes / cgo / JNI (include/keynet_hip.h: "no C++ exception crosses the boundary").
template <typename F>
static inline int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try { return fail(KN_ERR_NOMEM, "host allocation failed (std::bad_alloc)"); } catch (...) { return KN_ERR_NOMEM; }
    } catch (const std::length_error& e) {
        try { return fail(KN_ERR_NOMEM, std::string("host allocation of an impossible size (std::length_error: ") + e.what() + ")"); } catch (...) { return KN_ERR_NOMEM; }
    } catch (const std::exception& e) {
        try { return fail(KN_ERR_INVALID, std::string("C++ exception: ") + e.what()); } catch (...) { return KN_ERR_INVALID; }
    } catch (...) {
        try { return fail(KN_ERR_INVALID, "unknown C++ exception"); } catch (...) { return KN_ERR_INVALID; }
    }
}

// kn_spmm_plan: the dispatch logic runs exactly as in kn_spmm, but every launch site describes itself into the sink instead of launching.
struct PlanSink {
    std::string text;
    int launches = 0;
};
PlanSink*& plan_sink();   // thread-local; null outside kn_spmm_plan

#define KN_LAUNCH(desc, kernel, grid, block, lds, stream, ...)                                                   \
    do {                                                                                                         \
        if (kn::PlanSink* _ps = kn::plan_sink()) {                                                               \
            if (_ps->launches) _ps->text += "; ";                                                                \
            _ps->text += (desc);                                                                                 \
            _ps->text += " grid=" + std::to_string((unsigned long long)(grid).x);                                \
            _ps->launches++;                                                                                     \
        } else {                                                                                                 \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                   \
        }                                                                                                        \
    } while (0)

// Options of an operator handle.  The environment is read ONCE, when a handle is created (tuning_from_env, kn_api.hip: the library's only
// getenv), the values are recorded in the handle and kn_spmm_plan prints the ones that differ from the defaults: a handle is immutable after
// create, kn_spmm's dispatch depends on the handle and the call's arguments only.  The first block are the switches the parity tests use to put
// two formulations of one product side by side; the second block are measured tuning constants that only the diagnostic build (-DKN_ABLATION,
// tools/ablate_conv.sh) reads from the environment -- in the product library they are the defaults below.
struct Tuning {
    int no_sptr = 0;          // KN_NO_SPTR=1         conv-taps matrix-core kernel: per-thread pointers / generic loader instead of wave-uniform pointers
    int no_smallk_pipe = 0;   // KN_NO_SMALLK_PIPE=1  first-layer operators: the one-shot small-K kernel instead of the persistent pipeline
    int no_group_pipe = 0;    // KN_NO_GROUP_PIPE=1   CSR pattern groups: the plain grouped kernel instead of the software-pipelined one
    int no_big_groups = 0;    // KN_NO_BIG_GROUPS=1   a keyed Linear's rows as ordinary 16-row bundles instead of the LDS-staged workgroup kernel
    int no_exact_table = 0;   // KN_NO_EXACT_TABLE=1  factored untiled conv under KN_FLAG_EXACT: the conv pipeline instead of the stored-column table kernel
    int group_mfma = -1;      // KN_GROUP_MFMA=0|1    CSR pattern groups with their products on the matrix pipe: never / always (-1: the dispatch rule)
    int big_mfma16 = -1;      // KN_BIG_MFMA16=0|1    a keyed Linear's 16-row chunks on the matrix pipe: never / always (-1: the dispatch rule)
    int mf_nrb = 1;           // KN_MF_NRB=1|2|3      32-row blocks per chunk of the matrix-pipe grouped kernel
    int table_nrb = 0;        // KN_TABLE_NRB=1|2|3   32-channel blocks per workgroup of the table kernel (0: the rule)
    int no_fill_exact = 0;    // KN_NO_FILL_EXACT=1   filled-in conv operators (> 64 slots per pixel, or several slots on one (output, input) pixel pair) under KN_FLAG_EXACT: the generic kernel instead of convtaps_exact_fill_kernel
    int no_fill_tiles2 = 0;   // KN_NO_FILL_TILES2=1  ... on batches of whole 128-column tiles: one 64-column tile per wavefront instead of two
    // ---- diagnostic build only ----
    int occ = 0;              // KN_OCC               workgroups per CU cap of the 128 x 128 conv-taps launch (0: the rule)
    int no_tail_split = 0;    // KN_NO_TAIL_SPLIT
    int no_smallk = 0;        // KN_NO_SMALLK
    int exact_pipe = 16;      // KN_EXACT_PIPE        0 = plain exact conv kernel, 8 / 16 = channels per wavefront of the pipeline
    int exact_cob_groups = 0; // KN_EXACT_COB_GROUPS  (0: the rule)
    int exact_xd = 0;         // KN_EXACT_XD          2 | 4 activation rows in flight (0: the rule)
    int exact_vec = 0;        // KN_EXACT_VEC         2 | 4 batch columns per lane of the exact conv pipeline (0: the rule)
    int mf_pf = 8;            // KN_MF_PF             operand columns in flight of the matrix-pipe grouped kernel
    int table_window = 0;     // KN_TABLE_WINDOW      (0: the rule)
    int table_strip = -1;     // KN_TABLE_STRIP       (-1: best of the candidates, 0: keep the ball order)
    int conv_ball = 64;       // KN_CONV_BALL         output pixels per breadth-first ball of a conv-taps operator's processing order
    int no_patch = 0;         // KN_NO_PATCH
    int no_row_order = 0;     // KN_NO_ROW_ORDER
    int chain_no_cl = 0;      // KN_CHAIN_NO_CL
    int chain_no_rpl2 = 0;    // KN_CHAIN_NO_RPL2     whole-net kernel: one output row per lane in the pattern walk
    int chain_no_early = 0;   // KN_CHAIN_NO_EARLY    whole-net kernel: column pools staged at the start of their own layer
    int chain_no_share = 0;   // KN_CHAIN_NO_SHARE    whole-net kernel: every lane streams its own copy of its row's values (no shared value blocks)
    int chain_no_seq = 0;     // KN_CHAIN_NO_SEQ      whole-net kernel: thin layers read their columns from the staged pool instead of walking a re-ordered input sequentially
    int fill_form = 0;        // KN_FILL_FORM         filled-in order-preserving kernel: 0 = the dispatch rule, 1 / 2 = 32 / 64 channels x one column tile, 3 / 4 = 32 / 64 channels x two tiles
    int abl = 0;              // KN_ABL               kernel ablation mask (kn_conv.hip, KN_ABLATION code paths)
    std::string describe() const;   // "" when everything is at its default, else " opts{name=value,...}"
};
Tuning tuning_from_env();

enum Kind { KIND_CSR = 0, KIND_CONVTAPS = 1, KIND_DENSE = 2, KIND_CHAIN = 3, KIND_CSR64 = 4 };
struct ChainDev;   // kn_chain.hip

// Order-preserving CSR resident in HBM.
struct CsrDev {
    Tuning tune;                 // recorded at create
    int64_t rows = 0, cols = 0, nnz = 0;
    int32_t* indptr = nullptr;   // [rows+1]
    int32_t* indices = nullptr;  // [nnz]  stored order
    float* data = nullptr;       // [nnz]
    double* data64 = nullptr;    // [nnz] values of a float64 operator (KIND_CSR64: kn_csr_create_f64); `data` and the group lists are unused then
    // pattern groups (rows sharing one column sequence, e.g. the Cout rows of one conv output pixel, or all rows of a
    // dense Linear): see kn_csr.hip.  Rows not in any group are listed in `loose_rows`.
    int64_t n_groups = 0;
    int32_t* grp_colptr = nullptr;   // [n_groups+1] into grp_cols
    int32_t* grp_cols = nullptr;     // shared column sequence of each group
    int32_t* grp_rowptr = nullptr;   // [n_groups+1] into grp_rows
    int32_t* grp_rows = nullptr;     // member rows
    int64_t* grp_valptr = nullptr;   // [n_groups+1] into grp_vals (layout [col j][member r], r padded to RB)
    float* grp_vals = nullptr;
    int32_t* work_grp = nullptr;     // work items: (group, first member) pairs of RB rows
    int32_t* work_r0 = nullptr;
    int64_t n_work = 0;
    // big groups (a keyed nn.Linear: thousands of member rows over thousands of shared columns) run in the LDS-staged kernel:
    // work items = (group, first member) pairs of 32 rows; their members are NOT in work_grp/work_r0
    int32_t* big_grp = nullptr;
    int32_t* big_r0 = nullptr;
    int64_t n_big = 0;
    // loose rows with thousands of non-zeros (a row of a keyed nn.Linear whose pattern lost an entry to an exact zero): one wave per
    // (row, 64 batch columns) with a deep gather queue, as extra workgroups of the big-group launch
    int32_t* long_rows = nullptr;
    int64_t n_long = 0;
    int32_t* loose_rows = nullptr;
    int64_t n_loose = 0;
    int64_t grouped_nnz = 0;
    // PATCHED group members: rows whose stored column sequence is a group's sequence minus a few entries (a keyed conv row that lost a weight to an
    // exact zero) ride in the group with 0.0f at the missing positions; csr_patch_guard_kernel recomputes them in the reference's own sequence for
    // the batch columns whose activation at a missing position is not finite (see kn_csr.hip)
    // matrix-pipe products (kn_csr_mfma.hip): pattern groups with >= MF_MIN_MEMBERS members cut into chunks of 32 * (k + 1) member rows, k = 0..2
    // (list k holds the chunks of k + 1 row blocks); ws_* = the 16-row bundles of the REMAINING (small) groups for the vector-ALU kernels
    int32_t* mf_grp[3] = {nullptr, nullptr, nullptr};
    int32_t* mf_r0[3] = {nullptr, nullptr, nullptr};
    int64_t n_mf[3] = {0, 0, 0};
    int64_t mf_rows = 0;             // member rows covered by the mf_* lists
    int64_t mf_nnz = 0;              // their stored entries (mf_nnz / mf_rows = mean stored columns per row: the dispatch rule of csr_spmm_groups)
    int32_t* mf16_grp = nullptr;     // big pattern groups (a keyed Linear) in chunks of 16 member rows: csr_group_mfma16_kernel on narrow batches
    int32_t* mf16_r0 = nullptr;
    int64_t n_mf16 = 0;
    int32_t* ws_grp = nullptr;
    int32_t* ws_r0 = nullptr;
    int64_t n_ws = 0;
    int32_t* patch_rows = nullptr;   // [n_patch]
    int32_t* patch_ptr = nullptr;    // [n_patch+1] into patch_cols
    int32_t* patch_cols = nullptr;   // the missing column indices
    int64_t n_patch = 0;
};

// Factored conv operator  W = sum_e coef_e * taps[tap_e] (x) E[out_e,in_e] + lastcol + e_last.
struct ConvTapsDev {
    Tuning tune;                        // recorded at create
    int64_t Cin = 0, Hin = 0, Win = 0, Cout = 0, Hout = 0, Wout = 0;
    int64_t ntaps = 0;
    int64_t cin_pad = 0, cout_pad = 0;  // padded dims of tapsT
    float* tapsT = nullptr;             // [ntaps][cin_pad][cout_pad]  (co contiguous; zero padded)
    int64_t nslots = 0;                 // compute entries (zero taps dropped)
    int32_t* pix_ptr = nullptr;         // [HoWo+1]
    int32_t* slot_in = nullptr;         // [nslots] input pixel
    int32_t* slot_tap = nullptr;        // [nslots]
    float* slot_coef = nullptr;         // [nslots]
    int32_t* pix_order = nullptr;       // [HoWo] processing order of output pixels (locality)
    float* lastcol = nullptr;           // [Cout*HoWo+1] or null
    bool has_last = false;
    bool unit_coef = true;
    bool has_dups = false;              // some (output pixel, input pixel) pair is hit by more than one slot
    int max_slots = 0;
    // small-K pipeline (first layer of an image net: kn_conv.hip, convtaps_smallk_pipe_kernel): per-pixel descriptor records in
    // processing order, or null when the operator is not eligible
    int32_t* sk_desc = nullptr;
    int64_t sk_stride = 0, sk_tab_rows = 0;
    // kn_convtaps_drop_zero_entries: zero-valued tap entries (tap, co, ci) of taps that are not zero altogether -- absent from the reference's
    // untiled CSR; convtaps_zero_guard_kernel re-walks the affected output rows for batch columns with a non-finite activation there
    int32_t* zero_ent = nullptr;        // [n_zero][3]
    int64_t n_zero = 0;
    // ... and the stored-column table of the expansion (kn_csr_mfma.hip, TAPS): pixel o's columns ex_ptr[o] .. ex_ptr[o + 1], each (activation row, value row of tapsT)
    int32_t* ex_ptr = nullptr;          // [HoWo + 1]
    int32_t* ex_tab = nullptr;          // [ex_ptr[HoWo]][2]
    int32_t* ex_order = nullptr;        // [HoWo] processing order of the pixels for the table kernel (strips, kn_convtaps_drop_zero_entries)
    // filled-in operators under KN_FLAG_EXACT (kn_conv.hip, convtaps_exact_fill_kernel): per-pixel record lists {input pixel, tap offset, coefficient, first / last slot of
    // a stored column}, each padded to a multiple of 8 records; fill_ptr at create, the records on the device at the first kn_spmm that asks for them
    int32_t* fill_ptr = nullptr;        // [HoWo + 1] record offsets, or null when the operator is not eligible
    int32_t* fill_rec = nullptr;        // [fill_n][4]
    int64_t fill_n = 0;
    // bf16x3 path (kn_conv.hip, convtaps_bf16x3_kernel): the taps as three bf16 planes, built at the first kn_spmm that asks for them
    uint16_t* tapsB = nullptr;
    int64_t tapsB_plane = 0;
};

}  // namespace kn

struct kn_operator {
    int kind = kn::KIND_CSR;
    int device = 0;
    int64_t rows = 0, cols = 0;
    int64_t nnz_stored = 0;     // what the reference's nnz() reports
    int64_t nnz_expanded = 0;   // nnz of tocsr()
    kn::CsrDev csr;
    kn::ConvTapsDev ct;
    // host description of a conv-taps operator (export / lazy exact CSR)
    std::vector<int32_t> h_ent_out, h_ent_in, h_ent_tap;
    std::vector<float> h_ent_coef, h_taps, h_lastcol;
    std::mutex lazy_mu;
    kn_operator* exact = nullptr;  // lazily expanded CSR twin (KN_FLAG_EXACT on a conv-taps operator)
    // dense (Linear) operator: split-K conv-taps sub-operator + ordered reduction
    kn_operator* dense_sub = nullptr;
    int64_t dense_splits = 0;
    float* dense_lastcol = nullptr;   // [rows] bias column incl. the homogeneous 1
    // split-K partial sums [(rows-1) * splits, vecs]: ONE workspace PER STREAM (launches on one stream are ordered, launches on
    // different streams get different buffers, so concurrent kn_spmm calls on one handle never share partial sums).
    struct DenseWs {
        float* ptr = nullptr;
        int64_t vecs = 0;
    };
    std::map<hipStream_t, DenseWs> dense_ws;
    // an outgrown buffer is retired with an event recorded on its stream at that moment and freed by a later call once the event has
    // completed (every launch that could read it was queued before the event); kn_destroy frees what is left
    struct Retired {
        float* ptr = nullptr;
        hipEvent_t done = nullptr;
    };
    std::vector<Retired> dense_ws_retired;
    // a whole key-net of CSR operators as one launch (kn_chain.hip)
    kn::ChainDev* chain = nullptr;
};

namespace kn {
// kernels (kn_csr.hip / kn_conv.hip / kn_elementwise.hip)
int csr_build_groups(kn_operator* h, const int32_t* indptr, const int32_t* indices, const float* data);
int csr_spmm_planes(const CsrDev& A, const float* x, int64_t ldx, int64_t x_stride, int64_t n_planes, int64_t n_vecs, float* y, int64_t ldy, int64_t y_stride, uint32_t flags, hipStream_t s);
int csr_spmm(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, uint32_t flags, hipStream_t s, float* absmax = nullptr,
             bool* absmax_fused = nullptr);
template <typename TOUT>
int csr_f64_spmm(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, TOUT* y, int64_t ldy, uint32_t flags, hipStream_t s);      // kn_csr_f64.hip
int csr_group_mfma_spmm(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, int relu, hipStream_t s);
int csr_group_mfma16_spmm(const CsrDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, int relu, hipStream_t s);
static constexpr int MF_MIN_MEMBERS = 24;   // a pattern group takes the matrix-pipe kernel when its members fill >= 3/4 of a 32-row block
// `absmax` (device float or null): when the launch takes a kernel whose epilogue can fold max |Y| into its stores, the slot is raised atomically
// and *absmax_fused is set; otherwise the caller runs absmax_pass over Y afterwards (kn_spmm_screen)
int convtaps_spmm(const ConvTapsDev& A, int64_t rows, int64_t cols, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy,
                  uint32_t flags, hipStream_t s, float* absmax = nullptr, bool* absmax_fused = nullptr);
int absmax_pass(const float* y, int64_t rows, int64_t ld, int64_t n_vecs, float* absmax, hipStream_t s);
struct MfTaps;
int convtaps_exact_table_spmm(const ConvTapsDev& A, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, int relu, hipStream_t s);

// Raise *slot (a non-negative f32 kept as its bit pattern: for such values unsigned order == float order) to the wavefront's max of `m`.
// The slot is READ first and the atomic issued only when it would raise it -- after the first few tiles of a launch almost never.  Both
// alternatives were measured on the keyed VGG-16 forward (tools/ab_rescreen.py): an unconditional fire-and-forget atomic per tile (atomics on
// ONE address serialise at ~50 ns each: +20 ms per forward) and a scalar GLC load as the filter (same-address scalar loads serialise even
// harder: +54 ms).  The vector load's wait (vmcnt) also drains the wavefront's own stores, which is free where a workgroup ends with its tile
// and costly in the persistent first-layer kernel: that one carries its maximum across its pixels and commits once (kn_store_tile's `carry`).
__device__ __forceinline__ void kn_wave_absmax_commit(float m, float* slot, int lane) {
    auto step = [&](auto ctrl, auto row_mask) {               // DPP row shifts / broadcasts: vector ALU only, no LDS permute
        const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, m), __builtin_bit_cast(int, m), decltype(ctrl)::value,
                                                                              decltype(row_mask)::value, 0xf, false));
        m = (o > m) ? o : m;
    };
    step(std::integral_constant<int, 0x111>(), std::integral_constant<int, 0xf>());     // row_shr:1   lane i <- max(i, i-1)       (lanes without a source keep their own value)
    step(std::integral_constant<int, 0x112>(), std::integral_constant<int, 0xf>());     // row_shr:2
    step(std::integral_constant<int, 0x114>(), std::integral_constant<int, 0xf>());     // row_shr:4
    step(std::integral_constant<int, 0x118>(), std::integral_constant<int, 0xf>());     // row_shr:8   lane 15 of each row of 16 = the row's max
    step(std::integral_constant<int, 0x142>(), std::integral_constant<int, 0xa>());     // row_bcast:15 into rows 1, 3
    step(std::integral_constant<int, 0x143>(), std::integral_constant<int, 0xc>());     // row_bcast:31 into rows 2, 3: lane 63 = the wavefront's max
    if (lane == 63) {
        const unsigned bits = __float_as_uint(m);
        unsigned* u = reinterpret_cast<unsigned*>(slot);
        if (bits > __atomic_load_n(u, __ATOMIC_RELAXED)) atomicMax(u, bits);
    }
}
int relu_inplace(float* y, int64_t rows, int64_t ld, int64_t n_vecs, hipStream_t s);
int dense_reduce(const float* z, int64_t ldz, int64_t outs, int64_t splits, const float* lastcol, const float* xlast, float* y, int64_t ldy, int64_t n_vecs,
                 int relu, hipStream_t s);
int affine_to_linear(const float* x, int64_t n, int64_t d, float* out, int64_t ldo, hipStream_t s);
int linear_to_affine(const float* y, int64_t ldy, int64_t n, int64_t d, float* out, float* maxdev, hipStream_t s);
// Processing order for rows of a sparse pattern (host): breadth-first balls of `patch` rows over "shares a column", seeded along
// a global breadth-first sweep.  Columns referenced by more than `max_degree` of the rows (a bias column) do not link rows.
std::vector<int32_t> locality_order(const std::vector<int32_t>& row_ids, const int32_t* indptr, const int32_t* indices, int64_t n_cols, int patch,
                                    int max_degree);
int chain_create(int64_t n_ops, kn_operator* const* ops, const uint32_t* flags, ChainDev** out, int64_t* rows_out, int64_t* cols_out, int64_t* nnz_out);
int chain_forward(const ChainDev* c, const float* x, int64_t ldx, int64_t n_vecs, float* y, int64_t ldy, hipStream_t s);
void chain_free(ChainDev* c);
int convtaps_build_bf16(ConvTapsDev& A, const std::vector<float>& taps);
int convtaps_build_fill(ConvTapsDev& A, hipStream_t s);
bool convtaps_fill_ok(const ConvTapsDev& A);
bool convtaps_bf16x3_ok(const ConvTapsDev& A, const float* x, int64_t ldx, int64_t n_vecs, const float* y, int64_t ldy);
void csr_free(CsrDev& c);
void convtaps_free(ConvTapsDev& c);

template <typename T>
inline int upload(T** dptr, const T* h, size_t n) {
    *dptr = nullptr;
    const size_t n_copy = n;   // an empty host array has nothing to read, even when its pointer is non-null
    if (n == 0) n = 1;         // keep device pointers non-null
    hipError_t e = hipMalloc((void**)dptr, n * sizeof(T));
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? KN_ERR_NOMEM : KN_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    if (n_copy == 0) {
        e = hipMemset(*dptr, 0, n * sizeof(T));
        if (e != hipSuccess) return fail(KN_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
    } else if (h) {
        e = hipMemcpy(*dptr, h, n_copy * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(KN_ERR_HIP, std::string("hipMemcpy H2D: ") + hipGetErrorString(e));
    }
    return KN_OK;
}
}  // namespace kn
