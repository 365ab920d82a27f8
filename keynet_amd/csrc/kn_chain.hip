// kn_chain.hip -- a WHOLE keyed network as one launch: every operator of an untiled key-net (stored-order CSR: the permutation
// key-nets of BASELINE configs[0]-[1]) applied back to back with the activations resident in LDS.
//
// Replaces the nn.Sequential walk of KeyedModel.forward (keynet/system.py:130-133) for key-nets small enough that a launch per layer is
// what bounds them: LeNet_AvgPool at 1024 images is 7 launches of ~20 us each for 88 MB of algorithmic traffic (11 us at the HBM
// roof).  Batch columns are independent (scipy's csr_matvecs never mixes them), so a workgroup owns BT = 4 columns and walks all
// operators; the feature-major activation block of its columns ([features][4] f32 = one 16-byte LDS word per feature) ping-pongs
// between two LDS buffers (LeNet: 4705 + 1177 features = 92 KB of the CU's 160 KB), the operators stream from L2 (2.6 MB).
//
// Arithmetic: one LANE (or 2 / 4 lanes: CPL below) owns one output row and accumulates it strictly serially over the row's STORED
// non-zeros, f32 multiply then f32 add (no contraction) -- the rounding sequence of scipy, bit for bit, like kn_csr.hip.
//
// Operator layout (built once by chain_create from the CSR the reference holds):
//   * rows of a layer are dealt to lanes sorted by (length descending, column pattern, row): the rows of a slice (RPS rows = one
//     wavefront) then have (almost always) equal lengths, and rows sharing one column sequence (the Cout rows of a conv output pixel,
//     every row of a keyed nn.Linear) sit in adjacent lanes;
//   * values: sliced ELL in quads, [slice][k / 4][row slot][k % 4] -- a lane's next four values are ONE 16-byte load, a wavefront's
//     load is one contiguous 1 KiB piece;
//   * columns: rows sharing a pattern read ONE copy of it (per-lane base, quad stride 1: the lanes of a group hit the same address,
//     which the memory pipeline serves as one request); slices of unrelated rows (pooling) store them like the values.
//   * thin layers (a 121-row Linear) spread one row over 2 or 4 lanes (CPL = 2 or 1 batch columns per lane) so that the serial walk
//     over a row's columns runs on more wavefronts; the lanes of a row read the same operator words.
#include "kn_internal.h"
#include <algorithm>
#include <cstring>
#include <numeric>
#include <unordered_map>

#pragma clang fp contract(off)

namespace kn {

static constexpr int CHAIN_BT = 4;            // batch columns per workgroup
static constexpr int CHAIN_THREADS = 1024;    // 16 wavefronts
static constexpr int CHAIN_MAX_LAYERS = 12;
static constexpr size_t CHAIN_LDS_BYTES = 160 * 1024;

struct ChainLayerArg {
    const float* vals;          // quads: [slice][q][row slot][4]
    const int32_t* cols;        // pool of column quads
    const int32_t* lane_row;    // [n_slices * RPS] output row, -1 = empty slot
    const int32_t* lane_len;    // [n_slices * RPS]
    const int32_t* lane_cq;     // [n_slices * RPS] index of the row's first column QUAD in `cols` / 4
    const int32_t* slice_info;  // [n_slices][4]: min length, max length, column quad stride, first value quad / RPS  (i.e. the slice's quad offset in units of RPS quads)
    int32_t n_slices, n_rows, cpl, relu;
};

struct ChainArgs {
    ChainLayerArg L[CHAIN_MAX_LAYERS];
    const float* X;
    float* Y;
    int64_t ldx, ldy;
    int32_t n_layers, n_vecs, n_in, n_out, buf1_off;     // buf1_off: float4 index where the second activation buffer starts
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// one output row (or its CPL-column share) over the stored non-zeros [0, len): acc[j] = acc[j] + v * x[j], serial in k
template <int CPL>
__device__ __forceinline__ void chain_rows(const ChainLayerArg& L, const f32x4* __restrict__ in, f32x4* __restrict__ out, const int wave, const int lane) {
    constexpr int LPR = CHAIN_BT / CPL;        // lanes per row
    constexpr int RPS = 64 / LPR;              // rows per slice (wavefront)
    const int slot = lane / LPR;
    const int j0 = (lane % LPR) * CPL;
    const float* inf = reinterpret_cast<const float*>(in) + j0;
    for (int s = wave; s < L.n_slices; s += CHAIN_THREADS / 64) {
        const i32x4 info = *reinterpret_cast<const i32x4*>(L.slice_info + 4 * s);
        const int minlen = __builtin_amdgcn_readfirstlane(info.x), maxlen = __builtin_amdgcn_readfirstlane(info.y);
        const int cstride = __builtin_amdgcn_readfirstlane(info.z);
        const int64_t vq0 = (int64_t)__builtin_amdgcn_readfirstlane(info.w) * RPS;
        const int idx = s * RPS + slot;
        const int row = L.lane_row[idx];
        const int len = L.lane_len[idx];
        const f32x4* vp = reinterpret_cast<const f32x4*>(L.vals) + vq0 + slot;          // + q * RPS
        const i32x4* cp = reinterpret_cast<const i32x4*>(L.cols) + L.lane_cq[idx];      // + q * cstride
        float acc[CPL];
#pragma unroll
        for (int j = 0; j < CPL; j++) acc[j] = 0.0f;
        auto mac = [&](const int c, const float v) {
            float x[CPL];
            if constexpr (CPL == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(inf + 4 * c);
                x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
            } else if constexpr (CPL == 2) {
                const float2 t = *reinterpret_cast<const float2*>(inf + 4 * c);
                x[0] = t.x; x[1] = t.y;
            } else {
                x[0] = inf[4 * c];
            }
#pragma unroll
            for (int j = 0; j < CPL; j++) {
                const float p = v * x[j];
                acc[j] = acc[j] + p;
            }
        };
        // main part: whole quads that every row of the slice has (wave-uniform trip count, no predicate), two quads per trip with the
        // next trip's operator words requested before this trip's arithmetic
        const int nq = minlen >> 2;
        int q = 0;
        if (nq >= 2) {
            i32x4 c0 = cp[0], c1 = cp[cstride];
            f32x4 v0 = vp[0], v1 = vp[RPS];
            for (; q + 4 <= nq; q += 2) {
                const i32x4 c2 = cp[(int64_t)(q + 2) * cstride], c3 = cp[(int64_t)(q + 3) * cstride];
                const f32x4 v2 = vp[(int64_t)(q + 2) * RPS], v3 = vp[(int64_t)(q + 3) * RPS];
                mac(c0.x, v0.x); mac(c0.y, v0.y); mac(c0.z, v0.z); mac(c0.w, v0.w);
                mac(c1.x, v1.x); mac(c1.y, v1.y); mac(c1.z, v1.z); mac(c1.w, v1.w);
                c0 = c2; c1 = c3; v0 = v2; v1 = v3;
            }
            mac(c0.x, v0.x); mac(c0.y, v0.y); mac(c0.z, v0.z); mac(c0.w, v0.w);
            mac(c1.x, v1.x); mac(c1.y, v1.y); mac(c1.z, v1.z); mac(c1.w, v1.w);
            q += 2;
        }
        // the rest, element by element under the lane's own length (rows of a slice differ in length only at layer borders)
        for (int k = 4 * q; k < maxlen; k++) {
            if (k < len) {
                const int c = reinterpret_cast<const int32_t*>(cp + (int64_t)(k >> 2) * cstride)[k & 3];
                const float v = reinterpret_cast<const float*>(vp + (int64_t)(k >> 2) * RPS)[k & 3];
                mac(c, v);
            }
        }
        if (row >= 0) {
            float* o = reinterpret_cast<float*>(out + row) + j0;
#pragma unroll
            for (int j = 0; j < CPL; j++) o[j] = L.relu ? ((acc[j] < 0.0f) ? 0.0f : acc[j]) : acc[j];      // torch relu: NaN stays NaN
        }
    }
}

__global__ __launch_bounds__(CHAIN_THREADS) void chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float chain_lds[];
    f32x4* buf[2] = {reinterpret_cast<f32x4*>(chain_lds), reinterpret_cast<f32x4*>(chain_lds) + a.buf1_off};
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t c0 = (int64_t)blockIdx.x * CHAIN_BT;
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
        buf[0][f] = v;
    }
    __syncthreads();
    for (int l = 0; l < a.n_layers; l++) {
        const ChainLayerArg& L = a.L[l];
        const f32x4* in = buf[l & 1];
        f32x4* out = buf[(l & 1) ^ 1];
        if (L.cpl == 4) chain_rows<4>(L, in, out, wave, lane);
        else if (L.cpl == 2) chain_rows<2>(L, in, out, wave, lane);
        else chain_rows<1>(L, in, out, wave, lane);
        __syncthreads();
    }
    const f32x4* res = buf[a.n_layers & 1];
    for (int f = tid; f < a.n_out; f += CHAIN_THREADS) {
        const f32x4 v = res[f];
        float* dst = a.Y + (int64_t)f * a.ldy + c0;
        if (c0 + 0 < a.n_vecs) dst[0] = v.x;
        if (c0 + 1 < a.n_vecs) dst[1] = v.y;
        if (c0 + 2 < a.n_vecs) dst[2] = v.z;
        if (c0 + 3 < a.n_vecs) dst[3] = v.w;
    }
}

// ---- host -----------------------------------------------------------------------------------------------------------------------
struct ChainDev {
    std::vector<void*> allocs;
    ChainArgs args;
    size_t lds_bytes = 0;
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

// one layer: CSR (host copy, stored order) -> the sliced layout above
static int chain_build_layer(ChainDev* c, ChainLayerArg& L, int64_t rows, int64_t cols, const std::vector<int32_t>& ip, const std::vector<int32_t>& ix,
                             const std::vector<float>& dt, int relu) {
    // batch columns per lane: the widest form that still gives the workgroup's 16 wavefronts a slice each
    int cpl = 4;
    while (cpl > 1 && (rows * (CHAIN_BT / cpl) + 63) / 64 < CHAIN_THREADS / 64) cpl >>= 1;
    const int LPR = CHAIN_BT / cpl, RPS = 64 / LPR;
    // column patterns: rows with an identical stored column sequence share one copy
    std::vector<int32_t> pat((size_t)rows, -1);
    std::vector<int32_t> pat_rep;
    {
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
    std::vector<int32_t> order((size_t)rows);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        const int32_t lx = ip[(size_t)x + 1] - ip[(size_t)x], ly = ip[(size_t)y + 1] - ip[(size_t)y];
        if (lx != ly) return lx > ly;
        return pat[(size_t)x] < pat[(size_t)y];
    });
    const int64_t n_slices = (rows + RPS - 1) / RPS;
    std::vector<int32_t> lane_row((size_t)(n_slices * RPS), -1), lane_len((size_t)(n_slices * RPS), 0), lane_cq((size_t)(n_slices * RPS), 0), info((size_t)(n_slices * 4), 0);
    std::vector<float> vals;
    std::vector<int32_t> colpool(4, 0);                      // quad 0 = a harmless all-zero quad (empty slots point here)
    std::vector<int64_t> pat_cq(pat_rep.size(), -1);         // column quad offset of a pattern stored once
    int64_t vq = 0;                                          // running value-quad offset, in units of RPS quads
    for (int64_t s = 0; s < n_slices; s++) {
        int mn = INT32_MAX, mx = 0, distinct = 0;
        int32_t last_pat = -2;
        for (int i = 0; i < RPS; i++) {
            const int64_t o = s * RPS + i;
            if (o >= rows) break;
            const int32_t r = order[(size_t)o];
            const int len = ip[(size_t)r + 1] - ip[(size_t)r];
            mn = std::min(mn, len);
            mx = std::max(mx, len);
            if (pat[(size_t)r] != last_pat) distinct++;
            last_pat = pat[(size_t)r];
            lane_row[(size_t)o] = r;
            lane_len[(size_t)o] = len;
        }
        if (mn == INT32_MAX) mn = 0;
        const int nq = (mx + 3) / 4;
        const int real = (int)std::min<int64_t>(RPS, rows - s * RPS);
        const bool shared = distinct * 2 <= real;            // most lanes share a pattern with a neighbour: one copy per pattern
        // values: [q][slot][4], zero padded
        const size_t v0 = vals.size();
        vals.resize(v0 + (size_t)nq * RPS * 4, 0.0f);
        for (int i = 0; i < real; i++) {
            const int32_t r = order[(size_t)(s * RPS + i)];
            const int32_t rs = ip[(size_t)r];
            const int len = ip[(size_t)r + 1] - rs;
            for (int k = 0; k < len; k++) vals[v0 + ((size_t)(k >> 2) * RPS + (size_t)i) * 4 + (size_t)(k & 3)] = dt[(size_t)(rs + k)];
        }
        int cstride = 1;
        if (shared) {
            for (int i = 0; i < real; i++) {
                const int32_t r = order[(size_t)(s * RPS + i)];
                const int32_t p = pat[(size_t)r];
                if (pat_cq[(size_t)p] < 0) {
                    pat_cq[(size_t)p] = (int64_t)colpool.size() / 4;
                    const int32_t rs = ip[(size_t)r];
                    const int len = ip[(size_t)r + 1] - rs;
                    // padded to the slice's quad count + 2 (the main loop requests two quads ahead), with column 0 (a valid feature)
                    const size_t c0 = colpool.size();
                    colpool.resize(c0 + ((size_t)(len + 3) / 4 + 2) * 4, 0);
                    for (int k = 0; k < len; k++) colpool[c0 + (size_t)k] = ix[(size_t)(rs + k)];
                }
                lane_cq[(size_t)(s * RPS + i)] = (int32_t)pat_cq[(size_t)p];
            }
        } else {
            cstride = RPS;
            const size_t c0 = colpool.size();
            colpool.resize(c0 + (size_t)(nq + 2) * RPS * 4, 0);
            for (int i = 0; i < real; i++) {
                const int32_t r = order[(size_t)(s * RPS + i)];
                const int32_t rs = ip[(size_t)r];
                const int len = ip[(size_t)r + 1] - rs;
                for (int k = 0; k < len; k++) colpool[c0 + ((size_t)(k >> 2) * RPS + (size_t)i) * 4 + (size_t)(k & 3)] = ix[(size_t)(rs + k)];
                lane_cq[(size_t)(s * RPS + i)] = (int32_t)(c0 / 4 + (size_t)i);
            }
            for (int i = real; i < RPS; i++) lane_cq[(size_t)(s * RPS + i)] = (int32_t)(c0 / 4 + (size_t)i);
        }
        // a partial slice runs the unpredicated main part on its empty slots too: their operator words must be readable (they are: zero
        // values, column 0) and their lengths 0 keep them out of the predicated rest; min length counts real rows only
        info[(size_t)(4 * s + 0)] = mn;
        info[(size_t)(4 * s + 1)] = mx;
        info[(size_t)(4 * s + 2)] = cstride;
        info[(size_t)(4 * s + 3)] = (int32_t)vq;
        vq += nq;
    }
    vals.resize(vals.size() + (size_t)2 * RPS * 4, 0.0f);    // the main loop requests two quads ahead
    // shared patterns of a slice whose rows are SHORTER than the slice's maximum are read up to the maximum by the predicated rest only
    // through their own lengths; the unpredicated part stops at the slice minimum: in range by construction
    (void)cols;
    L.n_slices = (int32_t)n_slices;
    L.n_rows = (int32_t)rows;
    L.cpl = cpl;
    L.relu = relu;
    int rc;
    if ((rc = chain_upload(c, &L.vals, vals)) || (rc = chain_upload(c, &L.cols, colpool)) || (rc = chain_upload(c, &L.lane_row, lane_row)) ||
        (rc = chain_upload(c, &L.lane_len, lane_len)) || (rc = chain_upload(c, &L.lane_cq, lane_cq)) || (rc = chain_upload(c, &L.slice_info, info)))
        return rc;
    return KN_OK;
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
    const size_t lds = (feat[0] + feat[1]) * CHAIN_BT * sizeof(float);
    KN_REQUIRE(lds <= CHAIN_LDS_BYTES, KN_ERR_UNSUPPORTED, "activations of four batch columns do not fit the CU's 160 KiB of LDS");
    std::unique_ptr<ChainDev, void (*)(ChainDev*)> c(new ChainDev(), chain_free);
    std::memset(&c->args, 0, sizeof(ChainArgs));
    for (int64_t l = 0; l < n_ops; l++) {
        const CsrDev& A = ops[l]->csr;
        std::vector<int32_t> ip((size_t)A.rows + 1), ix((size_t)A.nnz);
        std::vector<float> dt((size_t)A.nnz);
        KN_HIP(hipMemcpy(ip.data(), A.indptr, sizeof(int32_t) * ip.size(), hipMemcpyDeviceToHost));
        if (A.nnz > 0) {
            KN_HIP(hipMemcpy(ix.data(), A.indices, sizeof(int32_t) * ix.size(), hipMemcpyDeviceToHost));
            KN_HIP(hipMemcpy(dt.data(), A.data, sizeof(float) * dt.size(), hipMemcpyDeviceToHost));
        }
        int rc = chain_build_layer(c.get(), c->args.L[l], A.rows, A.cols, ip, ix, dt, (flags && (flags[l] & KN_FLAG_RELU)) ? 1 : 0);
        if (rc) return rc;
    }
    c->args.n_layers = (int32_t)n_ops;
    c->args.n_in = (int32_t)ops[0]->cols;
    c->args.n_out = (int32_t)ops[n_ops - 1]->rows;
    c->args.buf1_off = (int32_t)feat[0];
    c->lds_bytes = lds;
    KN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS_BYTES));
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
    const int64_t grid = (n_vecs + CHAIN_BT - 1) / CHAIN_BT;
    KN_LAUNCH("chain_kernel<" + std::to_string(a.n_layers) + " operators, 4 batch columns per workgroup, " + std::to_string(c->lds_bytes) + " B LDS>", chain_kernel,
              dim3((unsigned)grid), dim3(CHAIN_THREADS), c->lds_bytes, s, a);
    KN_HIP(hipGetLastError());
    return KN_OK;
}

}  // namespace kn
