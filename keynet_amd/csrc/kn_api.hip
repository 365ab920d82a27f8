// kn_api.hip -- extern "C" surface of libkeynet_hip.so (see include/keynet_hip.h for the reference interface each
// entry point replaces).  Host-side conversion of the reference's operator containers into HBM-resident formats.
#include "kn_internal.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <numeric>
#include <unordered_map>

extern "C" int kn_destroy(kn_handle_t h);

namespace kn {

static thread_local std::string g_err;
static thread_local PlanSink* g_plan = nullptr;
PlanSink*& plan_sink() { return g_plan; }
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// The library's only look at the environment: called once per operator create, the result is recorded in the handle (kn_internal.h: Tuning).
Tuning tuning_from_env() {
    Tuning t;
    struct Knob {
        const char* name;
        int Tuning::*field;
    };
    static const Knob knobs[] = {
        {"KN_NO_SPTR", &Tuning::no_sptr}, {"KN_NO_SMALLK_PIPE", &Tuning::no_smallk_pipe}, {"KN_NO_GROUP_PIPE", &Tuning::no_group_pipe},
        {"KN_NO_BIG_GROUPS", &Tuning::no_big_groups}, {"KN_NO_EXACT_TABLE", &Tuning::no_exact_table}, {"KN_GROUP_MFMA", &Tuning::group_mfma},
        {"KN_BIG_MFMA16", &Tuning::big_mfma16}, {"KN_MF_NRB", &Tuning::mf_nrb}, {"KN_TABLE_NRB", &Tuning::table_nrb}, {"KN_NO_FILL_EXACT", &Tuning::no_fill_exact},
        {"KN_NO_FILL_TILES2", &Tuning::no_fill_tiles2},
#ifdef KN_ABLATION
        {"KN_OCC", &Tuning::occ}, {"KN_NO_TAIL_SPLIT", &Tuning::no_tail_split}, {"KN_NO_SMALLK", &Tuning::no_smallk}, {"KN_EXACT_PIPE", &Tuning::exact_pipe},
        {"KN_EXACT_COB_GROUPS", &Tuning::exact_cob_groups}, {"KN_EXACT_XD", &Tuning::exact_xd}, {"KN_EXACT_VEC", &Tuning::exact_vec}, {"KN_MF_PF", &Tuning::mf_pf},
        {"KN_TABLE_WINDOW", &Tuning::table_window}, {"KN_TABLE_STRIP", &Tuning::table_strip}, {"KN_NO_PATCH", &Tuning::no_patch}, {"KN_CONV_BALL", &Tuning::conv_ball},
        {"KN_NO_ROW_ORDER", &Tuning::no_row_order}, {"KN_CHAIN_NO_CL", &Tuning::chain_no_cl}, {"KN_CHAIN_NO_RPL2", &Tuning::chain_no_rpl2}, {"KN_CHAIN_NO_EARLY", &Tuning::chain_no_early}, {"KN_CHAIN_NO_SEQ", &Tuning::chain_no_seq}, {"KN_CHAIN_NO_SHARE", &Tuning::chain_no_share}, {"KN_FILL_FORM", &Tuning::fill_form}, {"KN_ABL", &Tuning::abl},
#endif
    };
    for (const Knob& k : knobs)
        if (const char* v = getenv(k.name)) t.*(k.field) = atoi(v);
    t.mf_nrb = std::max(1, std::min(3, t.mf_nrb));
    if (t.table_nrb < 0 || t.table_nrb > 3) t.table_nrb = 0;
    return t;
}

std::string Tuning::describe() const {
    static const Tuning d;
    std::string o;
    auto add = [&](const char* n, int v, int dv) {
        if (v != dv) o += (o.empty() ? "" : ",") + std::string(n) + "=" + std::to_string(v);
    };
    add("no_sptr", no_sptr, d.no_sptr); add("no_smallk_pipe", no_smallk_pipe, d.no_smallk_pipe); add("no_group_pipe", no_group_pipe, d.no_group_pipe);
    add("no_big_groups", no_big_groups, d.no_big_groups); add("no_exact_table", no_exact_table, d.no_exact_table); add("group_mfma", group_mfma, d.group_mfma);
    add("big_mfma16", big_mfma16, d.big_mfma16); add("mf_nrb", mf_nrb, d.mf_nrb); add("table_nrb", table_nrb, d.table_nrb); add("no_fill_exact", no_fill_exact, d.no_fill_exact);
    add("no_fill_tiles2", no_fill_tiles2, d.no_fill_tiles2);
    add("occ", occ, d.occ); add("no_tail_split", no_tail_split, d.no_tail_split); add("no_smallk", no_smallk, d.no_smallk); add("exact_pipe", exact_pipe, d.exact_pipe);
    add("exact_cob_groups", exact_cob_groups, d.exact_cob_groups); add("exact_xd", exact_xd, d.exact_xd); add("exact_vec", exact_vec, d.exact_vec); add("mf_pf", mf_pf, d.mf_pf);
    add("table_window", table_window, d.table_window); add("table_strip", table_strip, d.table_strip); add("no_patch", no_patch, d.no_patch); add("conv_ball", conv_ball, d.conv_ball);
    add("no_row_order", no_row_order, d.no_row_order); add("chain_no_cl", chain_no_cl, d.chain_no_cl); add("chain_no_rpl2", chain_no_rpl2, d.chain_no_rpl2); add("chain_no_early", chain_no_early, d.chain_no_early); add("chain_no_seq", chain_no_seq, d.chain_no_seq); add("chain_no_share", chain_no_share, d.chain_no_share); add("fill_form", fill_form, d.fill_form); add("abl", abl, d.abl);
    return o.empty() ? o : " opts{" + o + "}";
}

static int ensure_device(int* dev) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(KN_ERR_NODEVICE, "no HIP device visible (libkeynet_hip.so needs an MI355X / gfx950)");
    KN_HIP(hipGetDevice(dev));
    return KN_OK;
}

// canonical (row, col)-sorted CSR from COO, duplicates summed in sorted order (scipy csr_matrix((v,(r,c))) semantics)
static void coo_to_csr(int64_t rows, std::vector<int64_t>& r, std::vector<int64_t>& c, std::vector<float>& v, std::vector<int32_t>& indptr,
                       std::vector<int32_t>& indices, std::vector<float>& data) {
    const size_t n = r.size();
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return r[a] != r[b] ? r[a] < r[b] : c[a] < c[b]; });
    indptr.assign((size_t)rows + 1, 0);
    indices.clear();
    data.clear();
    indices.reserve(n);
    data.reserve(n);
    int64_t pr = -1, pc = -1;
    for (size_t k = 0; k < n; k++) {
        const size_t i = order[k];
        if (r[i] == pr && c[i] == pc) {
            data.back() = data.back() + v[i];
        } else {
            indices.push_back((int32_t)c[i]);
            data.push_back(v[i]);
            indptr[(size_t)r[i] + 1]++;
            pr = r[i];
            pc = c[i];
        }
    }
    for (int64_t i = 0; i < rows; i++) indptr[(size_t)i + 1] += indptr[(size_t)i];
}

struct OperatorDeleter {
    void operator()(kn_operator* h) const { (void)kn_destroy(h); }
};
typedef std::unique_ptr<kn_operator, OperatorDeleter> OperatorPtr;   // a create that fails or throws half way frees its device memory

template <typename TV>
static int csr_create_impl(int64_t rows, int64_t cols, int64_t nnz, const int32_t* indptr, const int32_t* indices, const TV* data,
                           kn_operator** out) {
    KN_REQUIRE(out != nullptr, KN_ERR_INVALID, "out handle is NULL");
    *out = nullptr;
    KN_REQUIRE(rows >= 0 && cols >= 0 && nnz >= 0, KN_ERR_INVALID, "negative size");
    KN_REQUIRE(rows < INT32_MAX && cols < INT32_MAX && nnz < INT32_MAX, KN_ERR_UNSUPPORTED, "int32 index range exceeded");
    KN_REQUIRE(indptr && (nnz == 0 || (indices && data)), KN_ERR_INVALID, "NULL CSR array");
    KN_REQUIRE(indptr[0] == 0 && indptr[rows] == nnz, KN_ERR_INVALID, "indptr does not span nnz");
    for (int64_t i = 0; i < rows; i++) KN_REQUIRE(indptr[i] <= indptr[i + 1], KN_ERR_INVALID, "indptr not monotone");
    for (int64_t k = 0; k < nnz; k++) KN_REQUIRE(indices[k] >= 0 && indices[k] < cols, KN_ERR_INVALID, "column index out of range");
    int dev = 0;
    int rc = ensure_device(&dev);
    if (rc) return rc;
    OperatorPtr h(new kn_operator());
    h->kind = std::is_same<TV, double>::value ? KIND_CSR64 : KIND_CSR;
    h->device = dev;
    h->rows = rows;
    h->cols = cols;
    h->nnz_stored = nnz;
    h->nnz_expanded = nnz;
    h->csr.tune = tuning_from_env();
    h->csr.rows = rows;
    h->csr.cols = cols;
    h->csr.nnz = nnz;
    if ((rc = upload(&h->csr.indptr, indptr, (size_t)rows + 1)) || (rc = upload(&h->csr.indices, indices, (size_t)nnz))) return rc;
    if constexpr (std::is_same<TV, double>::value) {
        // a float64 operator (kn_csr_create_f64): one row kernel in f64 arithmetic, no pattern groups (kn_csr_f64.hip)
        if ((rc = upload(&h->csr.data64, data, (size_t)nnz))) return rc;
    } else {
        if ((rc = upload(&h->csr.data, data, (size_t)nnz)) || (rc = csr_build_groups(h.get(), indptr, indices, data))) return rc;
    }
    *out = h.release();
    return KN_OK;
}

// expanded canonical CSR of a conv-taps operator (tocsr() of the equivalent Conv2dTiledMatrix)
static void convtaps_expand(const kn_operator* h, const std::vector<int64_t>& last_rows, const std::vector<float>& last_vals,
                            std::vector<int32_t>& indptr, std::vector<int32_t>& indices, std::vector<float>& data) {
    const ConvTapsDev& c = h->ct;
    const int64_t HoWo = c.Hout * c.Wout, HiWi = c.Hin * c.Win;
    std::vector<int64_t> r, cc;
    std::vector<float> v;
    const size_t n = h->h_ent_out.size() * (size_t)(c.Cout * c.Cin) + last_rows.size();
    r.reserve(n);
    cc.reserve(n);
    v.reserve(n);
    for (size_t e = 0; e < h->h_ent_out.size(); e++) {
        const float* T = h->h_taps.data() + (size_t)h->h_ent_tap[e] * (size_t)(c.Cout * c.Cin);
        const float coef = h->h_ent_coef[e];
        for (int64_t ic = 0; ic < c.Cout; ic++)
            for (int64_t jc = 0; jc < c.Cin; jc++) {
                r.push_back(h->h_ent_out[e] + ic * HoWo);
                cc.push_back(h->h_ent_in[e] + jc * HiWi);
                v.push_back(coef == 1.0f ? T[ic * c.Cin + jc] : coef * T[ic * c.Cin + jc]);
            }
    }
    for (size_t k = 0; k < last_rows.size(); k++) {
        r.push_back(last_rows[k]);
        cc.push_back(c.Cin * HiWi);
        v.push_back(last_vals[k]);
    }
    coo_to_csr(h->rows, r, cc, v, indptr, indices, data);
}

struct ConvBuild {
    int64_t inshape[3], outshape[3];
    std::vector<float> taps;   // [ntaps][Cout][Cin]
    std::vector<int32_t> ent_out, ent_in, ent_tap;
    std::vector<float> ent_coef;
    bool has_last = false;
    std::vector<int64_t> last_rows;   // explicit entries of the last column (row, value), may hold explicit zeros
    std::vector<float> last_vals;
    int64_t nnz_stored = 0;
};

static int convtaps_create_impl(ConvBuild& b, kn_operator** out) {
    *out = nullptr;
    const int64_t Cin = b.inshape[0], Hin = b.inshape[1], Win = b.inshape[2];
    const int64_t Cout = b.outshape[0], Hout = b.outshape[1], Wout = b.outshape[2];
    KN_REQUIRE(Cin > 0 && Hin > 0 && Win > 0 && Cout > 0 && Hout > 0 && Wout > 0, KN_ERR_INVALID, "non-positive shape");
    const int64_t HoWo = Hout * Wout, HiWi = Hin * Win;
    const int64_t rows = Cout * HoWo + (b.has_last ? 1 : 0), cols = Cin * HiWi + (b.has_last ? 1 : 0);
    KN_REQUIRE(rows < INT32_MAX && cols < INT32_MAX, KN_ERR_UNSUPPORTED, "int32 index range exceeded");
    const int64_t ntaps = (int64_t)(b.taps.size() / (size_t)(Cout * Cin));
    const size_t nent = b.ent_out.size();
    for (size_t e = 0; e < nent; e++) {
        KN_REQUIRE(b.ent_out[e] >= 0 && b.ent_out[e] < HoWo, KN_ERR_INVALID, "entry output pixel out of range");
        KN_REQUIRE(b.ent_in[e] >= 0 && b.ent_in[e] < HiWi, KN_ERR_INVALID, "entry input pixel out of range");
        KN_REQUIRE(b.ent_tap[e] >= 0 && b.ent_tap[e] < ntaps, KN_ERR_INVALID, "entry tap id out of range");
    }
    for (size_t k = 0; k < b.last_rows.size(); k++) KN_REQUIRE(b.last_rows[k] >= 0 && b.last_rows[k] < rows, KN_ERR_INVALID, "last-column row out of range");
    int dev = 0;
    int rc = ensure_device(&dev);
    if (rc) return rc;

    OperatorPtr h(new kn_operator());
    h->kind = KIND_CONVTAPS;
    h->device = dev;
    h->rows = rows;
    h->cols = cols;
    ConvTapsDev& c = h->ct;
    c.tune = tuning_from_env();
    c.Cin = Cin; c.Hin = Hin; c.Win = Win; c.Cout = Cout; c.Hout = Hout; c.Wout = Wout;
    c.ntaps = ntaps;
    const int64_t KC = Cin >= 16 ? 16 : 4;
    const int64_t MT = Cout > 64 ? 128 : 64;
    c.cin_pad = (Cin + KC - 1) / KC * KC;
    c.cout_pad = (Cout + MT - 1) / MT * MT;
    c.has_last = b.has_last;

    // transposed, zero-padded taps; zero taps contribute nothing and are dropped from the compute lists
    std::vector<float> tapsT((size_t)(std::max<int64_t>(ntaps, 1) * c.cin_pad * c.cout_pad), 0.0f);
    std::vector<char> tap_zero((size_t)std::max<int64_t>(ntaps, 1), 1);
    for (int64_t t = 0; t < ntaps; t++)
        for (int64_t co = 0; co < Cout; co++)
            for (int64_t ci = 0; ci < Cin; ci++) {
                const float w = b.taps[(size_t)((t * Cout + co) * Cin + ci)];
                tapsT[(size_t)((t * c.cin_pad + ci) * c.cout_pad + co)] = w;
                if (w != 0.0f) tap_zero[(size_t)t] = 0;
            }
    // slots grouped by output pixel, ascending input pixel inside a pixel
    std::vector<size_t> order;
    order.reserve(nent);
    for (size_t e = 0; e < nent; e++)
        if (!tap_zero[(size_t)b.ent_tap[e]] && b.ent_coef[e] != 0.0f) order.push_back(e);
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
        return b.ent_out[x] != b.ent_out[y] ? b.ent_out[x] < b.ent_out[y] : b.ent_in[x] < b.ent_in[y];
    });
    std::vector<int32_t> pix_ptr((size_t)HoWo + 1, 0), slot_in(order.size()), slot_tap(order.size()), pix_order((size_t)HoWo);
    std::vector<float> slot_coef(order.size());
    c.unit_coef = true;
    for (size_t k = 0; k < order.size(); k++) {
        const size_t e = order[k];
        pix_ptr[(size_t)b.ent_out[e] + 1]++;
        slot_in[k] = b.ent_in[e];
        slot_tap[k] = b.ent_tap[e];
        slot_coef[k] = b.ent_coef[e];
        if (b.ent_coef[e] != 1.0f) c.unit_coef = false;
        if (k > 0 && b.ent_out[order[k - 1]] == b.ent_out[e] && b.ent_in[order[k - 1]] == b.ent_in[e]) c.has_dups = true;
    }
    int mx = 0;
    for (int64_t o = 0; o < HoWo; o++) {
        mx = std::max(mx, pix_ptr[(size_t)o + 1]);
        pix_ptr[(size_t)o + 1] += pix_ptr[(size_t)o];
    }
    c.max_slots = mx;
    c.nslots = (int64_t)order.size();
    // record offsets of the filled-in order-preserving kernel (convtaps_exact_fill_kernel): every pixel's slot list padded to a multiple of 8 records
    std::vector<int32_t> fill_ptr;
    if ((c.has_dups || mx > 64) && !c.tune.no_fill_exact && ntaps * c.cin_pad * c.cout_pad < ((int64_t)1 << 29) && (int64_t)order.size() + 8 * HoWo < ((int64_t)1 << 31)) {
        fill_ptr.assign((size_t)HoWo + 1, 0);
        for (int64_t o = 0; o < HoWo; o++) fill_ptr[(size_t)o + 1] = fill_ptr[(size_t)o] + (pix_ptr[(size_t)o + 1] - pix_ptr[(size_t)o] + 7) / 8 * 8;
        c.fill_n = fill_ptr[(size_t)HoWo];
    }
    // Processing order of the output pixels (kn::locality_order): two pixels are neighbours when they share an input pixel;
    // balls of 64 are, for a keyed 3x3 conv, roughly 8x8 patches of the un-keyed image.  One patch = the workgroups resident
    // on one XCD at a time (32 CUs x 4 workgroups / 2 Cout tiles), so the gathered activation rows of a patch (a ~10x10
    // halo instead of 64 x 9 scattered pixels) are fetched into that XCD's L2 about once.  This recovers the locality a
    // permutation key destroys, without knowing the key: only the operator's own sparsity structure is used.
    {
        std::vector<int32_t> ids((size_t)HoWo);
        for (int64_t o = 0; o < HoWo; o++) ids[(size_t)o] = (int32_t)o;
        pix_order = locality_order(ids, pix_ptr.data(), slot_in.data(), HiWi, std::max(1, c.tune.conv_ball), 1 << 30);
    }
    std::vector<float> lastcol;
    if (b.has_last) {
        lastcol.assign((size_t)rows, 0.0f);
        for (size_t k = 0; k < b.last_rows.size(); k++) lastcol[(size_t)b.last_rows[k]] += b.last_vals[k];
    }
    h->h_ent_out = b.ent_out;
    h->h_ent_in = b.ent_in;
    h->h_ent_tap = b.ent_tap;
    h->h_ent_coef = b.ent_coef;
    h->h_taps = b.taps;
    // last column kept as explicit (row,value) pairs packed in h_lastcol as [row0,val0,row1,val1,...] bit patterns
    h->h_lastcol.resize(b.last_rows.size() * 2);
    for (size_t k = 0; k < b.last_rows.size(); k++) {
        const int32_t rr = (int32_t)b.last_rows[k];
        std::memcpy(&h->h_lastcol[2 * k], &rr, 4);
        h->h_lastcol[2 * k + 1] = b.last_vals[k];
    }
    // nnz of the expansion = what the reference's csr holds: one stored entry per (output pixel, input pixel) PAIR and channel pair -- several slots on one pair
    // (a filled-in operator) are one stored non-zero
    int64_t n_pairs = 0;
    if (order.size() == nent) {
        for (size_t k = 0; k < order.size(); k++)
            if (k == 0 || b.ent_out[order[k - 1]] != b.ent_out[order[k]] || b.ent_in[order[k - 1]] != b.ent_in[order[k]]) n_pairs++;
    } else {                                               // (entries of all-zero taps were left out of the slot lists but are stored by the reference)
        std::vector<int64_t> key(nent);
        for (size_t e = 0; e < nent; e++) key[e] = (int64_t)b.ent_out[e] * HiWi + b.ent_in[e];
        std::sort(key.begin(), key.end());
        n_pairs = (int64_t)(std::unique(key.begin(), key.end()) - key.begin());
    }
    h->nnz_expanded = n_pairs * Cout * Cin + (int64_t)b.last_rows.size();
    h->nnz_stored = b.nnz_stored > 0 ? b.nnz_stored : ntaps * Cout * Cin + (int64_t)nent + (int64_t)b.last_rows.size();

    // small-K pipeline descriptors (layout: kn_conv.hip SK_DESC_HDR): eligible when one pixel's whole contraction (slots x Cin + bias
    // row) fits 28 rows, one 64-wide Cout tile covers the layer and the tap matrix fits its LDS table
    std::vector<int32_t> sk_desc;
    if ((int64_t)mx * Cin + (b.has_last ? 1 : 0) <= 28 && Cout == 64 && c.cout_pad == 64 && ntaps * c.cin_pad <= 61 && cols < INT32_MAX - 1) {
        const int64_t stride = 96 + c.cout_pad;
        const int32_t zero_off = (int32_t)(ntaps * c.cin_pad * 64);
        sk_desc.assign((size_t)(HoWo * stride), 0);
        for (int64_t pi = 0; pi < HoWo; pi++) {
            const int32_t o = pix_order[(size_t)pi];
            int32_t* d = sk_desc.data() + (size_t)(pi * stride);
            float* df = reinterpret_cast<float*>(d);
            for (int k = 0; k < 32; k++) {
                d[k] = -1;
                d[32 + k] = zero_off;
                df[64 + k] = 0.0f;
            }
            int k = 0;
            for (int32_t sl = pix_ptr[(size_t)o]; sl < pix_ptr[(size_t)o + 1]; sl++)
                for (int64_t ci = 0; ci < Cin; ci++, k++) {
                    d[k] = (int32_t)(ci * HiWi + slot_in[(size_t)sl]);
                    d[32 + k] = (int32_t)((slot_tap[(size_t)sl] * c.cin_pad + ci) * 64);
                    df[64 + k] = slot_coef[(size_t)sl];
                }
            if (b.has_last) {
                d[k] = (int32_t)(Cin * HiWi);             // homogeneous coordinate of X against the bias row
                d[32 + k] = zero_off + 64;                 // marker: "this buffer's bias row"
                df[64 + k] = 1.0f;
                for (int64_t m = 0; m < Cout; m++) df[96 + m] = lastcol[(size_t)(m * HoWo + o)];
            }
            d[31] = o;
        }
        c.sk_stride = stride;
        c.sk_tab_rows = ntaps * c.cin_pad;
    }
    if ((!sk_desc.empty() && (rc = upload(&c.sk_desc, sk_desc.data(), sk_desc.size()))) || (rc = upload(&c.tapsT, tapsT.data(), tapsT.size())) || (rc = upload(&c.pix_ptr, pix_ptr.data(), pix_ptr.size())) ||
        (rc = upload(&c.slot_in, slot_in.data(), slot_in.size())) || (rc = upload(&c.slot_tap, slot_tap.data(), slot_tap.size())) ||
        (rc = upload(&c.slot_coef, slot_coef.data(), slot_coef.size())) || (rc = upload(&c.pix_order, pix_order.data(), pix_order.size())) ||
        (rc = upload(&c.lastcol, lastcol.data(), lastcol.size())) || (!fill_ptr.empty() && (rc = upload(&c.fill_ptr, fill_ptr.data(), fill_ptr.size()))))
        return rc;
    *out = h.release();
    return KN_OK;
}

static void last_pairs(const kn_operator* h, std::vector<int64_t>& rows, std::vector<float>& vals) {
    const size_t n = h->h_lastcol.size() / 2;
    rows.resize(n);
    vals.resize(n);
    for (size_t k = 0; k < n; k++) {
        int32_t rr;
        std::memcpy(&rr, &h->h_lastcol[2 * k], 4);
        rows[k] = rr;
        vals[k] = h->h_lastcol[2 * k + 1];
    }
}

}  // namespace kn

using namespace kn;

extern "C" {

int kn_abi_version(void) { return KN_ABI_VERSION; }

const char* kn_last_error(void) { return g_err.c_str(); }

int kn_device_info(int* n_devices, char* arch_buf, int64_t arch_buf_len) {
    return guarded([&]() -> int {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    if (n_devices) *n_devices = n;
    if (arch_buf && arch_buf_len > 0) {
        arch_buf[0] = 0;
        if (n > 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
                std::strncpy(arch_buf, prop.gcnArchName, (size_t)arch_buf_len - 1);
                arch_buf[arch_buf_len - 1] = 0;
            }
        }
    }
    return KN_OK;
    });
}

int kn_csr_create(int64_t rows, int64_t cols, int64_t nnz, const int32_t* indptr, const int32_t* indices, const float* data, kn_handle_t* out) {
    return guarded([&]() -> int {
    return csr_create_impl(rows, cols, nnz, indptr, indices, data, out);
    });
}

int kn_csr_create_f64(int64_t rows, int64_t cols, int64_t nnz, const int32_t* indptr, const int32_t* indices, const double* data, kn_handle_t* out) {
    return guarded([&]() -> int {
    return csr_create_impl(rows, cols, nnz, indptr, indices, data, out);
    });
}

int kn_dtype_bits(kn_handle_t h, int* bits) {
    return guarded([&]() -> int {
    KN_REQUIRE(h && bits, KN_ERR_INVALID, "NULL argument");
    *bits = h->kind == KIND_CSR64 ? 64 : 32;
    return KN_OK;
    });
}

int kn_tiled_create(int64_t rows, int64_t cols, int64_t nblocks, const int64_t* blocks, int64_t ntiles, const int64_t* tile_ptr,
                    const int32_t* tile_row, const int32_t* tile_col, const float* tile_val, kn_handle_t* out) {
    return guarded([&]() -> int {
    KN_REQUIRE(out != nullptr, KN_ERR_INVALID, "out handle is NULL");
    *out = nullptr;
    KN_REQUIRE(rows >= 0 && cols >= 0 && nblocks >= 0 && ntiles >= 0, KN_ERR_INVALID, "negative size");
    KN_REQUIRE((nblocks == 0 || blocks) && tile_ptr, KN_ERR_INVALID, "NULL tile array");
    std::vector<int64_t> r, c;
    std::vector<float> v;
    int64_t stored = tile_ptr[ntiles] - tile_ptr[0];
    for (int64_t b = 0; b < nblocks; b++) {
        const int64_t i = blocks[3 * b], j = blocks[3 * b + 1], k = blocks[3 * b + 2];
        KN_REQUIRE(k >= 0 && k < ntiles, KN_ERR_INVALID, "block references a missing tile");
        for (int64_t e = tile_ptr[k]; e < tile_ptr[k + 1]; e++) {
            const int64_t rr = i + tile_row[e], cc = j + tile_col[e];
            KN_REQUIRE(rr >= 0 && rr < rows && cc >= 0 && cc < cols, KN_ERR_INVALID, "tile entry outside the matrix");
            r.push_back(rr);
            c.push_back(cc);
            v.push_back(tile_val[e]);
        }
    }
    std::vector<int32_t> ip, ix;
    std::vector<float> dt;
    coo_to_csr(rows, r, c, v, ip, ix, dt);
    int rc = csr_create_impl(rows, cols, (int64_t)ix.size(), ip.data(), ix.data(), dt.data(), out);
    if (rc == KN_OK) (*out)->nnz_stored = stored;
    return rc;
    });
}

int kn_conv2dtiled_create(int64_t rows, int64_t cols, const int64_t inshape[3], const int64_t outshape[3], int64_t nblocks, const int64_t* blocks,
                          int64_t nent, const int64_t* tile_keys, const uint8_t* tile_isbias, const float* tile_chan, const float* tile_bias,
                          kn_handle_t* out) {
    return guarded([&]() -> int {
    KN_REQUIRE(out != nullptr, KN_ERR_INVALID, "out handle is NULL");
    *out = nullptr;
    KN_REQUIRE(rows >= 0 && cols >= 0 && nblocks >= 0 && nent >= 0, KN_ERR_INVALID, "negative size");
    KN_REQUIRE(inshape && outshape && (nblocks == 0 || blocks) && (nent == 0 || (tile_keys && tile_isbias)), KN_ERR_INVALID, "NULL argument");
    for (int k = 0; k < 3; k++)
        KN_REQUIRE(inshape[k] > 0 && outshape[k] > 0 && inshape[k] < INT32_MAX && outshape[k] < INT32_MAX, KN_ERR_INVALID, "inshape / outshape entries must be positive");
    KN_REQUIRE(rows < INT32_MAX && cols < INT32_MAX, KN_ERR_UNSUPPORTED, "int32 index range exceeded");
    const int64_t Cin = inshape[0], HiWi = inshape[1] * inshape[2], Cout = outshape[0], HoWo = outshape[1] * outshape[2];
    const bool has_last = (rows == Cout * HoWo + 1);
    KN_REQUIRE(rows == Cout * HoWo + (has_last ? 1 : 0) && cols == Cin * HiWi + (has_last ? 1 : 0), KN_ERR_SHAPE,
               "matrix shape does not match inshape/outshape (keynet/sparse.py:731-736)");
    ConvBuild b;
    for (int k = 0; k < 3; k++) {
        b.inshape[k] = inshape[k];
        b.outshape[k] = outshape[k];
    }
    b.has_last = has_last;
    // entries by tile id; channel matrices de-duplicated by content
    std::unordered_map<int64_t, std::vector<int64_t>> by_k;
    std::vector<int64_t> chan_idx((size_t)nent, -1);
    int64_t nc = 0, nb = 0;
    const size_t msz = (size_t)(Cout * Cin);
    std::map<std::string, int32_t> dedup;
    std::vector<int32_t> tap_of((size_t)nent, -1);
    std::vector<float> bias_of((size_t)nent, 0.0f);
    int64_t stored = 0;
    for (int64_t e = 0; e < nent; e++) {
        by_k[tile_keys[3 * e + 2]].push_back(e);
        if (tile_isbias[e]) {
            KN_REQUIRE(tile_bias != nullptr, KN_ERR_INVALID, "bias tiles without tile_bias");
            bias_of[(size_t)e] = tile_bias[nb++];
            stored += 1;
        } else {
            KN_REQUIRE(tile_chan != nullptr, KN_ERR_INVALID, "channel tiles without tile_chan");
            const float* m = tile_chan + (size_t)nc * msz;
            nc++;
            std::string key(reinterpret_cast<const char*>(m), msz * sizeof(float));
            auto it = dedup.find(key);
            if (it == dedup.end()) {
                const int32_t id = (int32_t)dedup.size();
                dedup.emplace(std::move(key), id);
                b.taps.insert(b.taps.end(), m, m + msz);
                tap_of[(size_t)e] = id;
            } else {
                tap_of[(size_t)e] = it->second;
            }
            stored += (int64_t)msz;
        }
    }
    b.nnz_stored = stored;
    for (int64_t bl = 0; bl < nblocks; bl++) {
        const int64_t i = blocks[3 * bl], j = blocks[3 * bl + 1], k = blocks[3 * bl + 2];
        auto it = by_k.find(k);
        if (it == by_k.end()) continue;
        for (int64_t e : it->second) {
            const int64_t itl = tile_keys[3 * e], jtl = tile_keys[3 * e + 1];
            if (tile_isbias[e]) {
                KN_REQUIRE(has_last && j + jtl == Cin * HiWi, KN_ERR_INVALID, "bias tile outside the last column (keynet/sparse.py:773)");
                KN_REQUIRE(i + itl >= 0 && i + itl < rows, KN_ERR_INVALID, "bias tile row out of range");
                b.last_rows.push_back(i + itl);
                b.last_vals.push_back(bias_of[(size_t)e]);
            } else {
                KN_REQUIRE(i + itl >= 0 && i + itl < HoWo && j + jtl >= 0 && j + jtl < HiWi, KN_ERR_INVALID,
                           "spatial tile entry outside the channel-(0,0) plane");
                b.ent_out.push_back((int32_t)(i + itl));
                b.ent_in.push_back((int32_t)(j + jtl));
                b.ent_tap.push_back(tap_of[(size_t)e]);
                b.ent_coef.push_back(1.0f);
            }
        }
    }
    return convtaps_create_impl(b, out);
    });
}

int kn_convtaps_create(const int64_t inshape[3], const int64_t outshape[3], int64_t ntaps, const float* taps, int64_t nent, const int32_t* ent_out,
                       const int32_t* ent_in, const int32_t* ent_tap, const float* ent_coef, const float* lastcol, kn_handle_t* out) {
    return guarded([&]() -> int {
    KN_REQUIRE(out != nullptr, KN_ERR_INVALID, "out handle is NULL");
    *out = nullptr;
    KN_REQUIRE(inshape && outshape && ntaps >= 0 && nent >= 0, KN_ERR_INVALID, "bad argument");
    KN_REQUIRE((ntaps == 0 || taps) && (nent == 0 || (ent_out && ent_in && ent_tap)), KN_ERR_INVALID, "NULL array");
    ConvBuild b;
    for (int k = 0; k < 3; k++) {
        b.inshape[k] = inshape[k];
        b.outshape[k] = outshape[k];
    }
    const int64_t Cout = outshape[0], Cin = inshape[0];
    for (int k = 0; k < 3; k++)
        KN_REQUIRE(inshape[k] > 0 && outshape[k] > 0 && inshape[k] < INT32_MAX && outshape[k] < INT32_MAX, KN_ERR_INVALID, "inshape / outshape entries must be positive");
    KN_REQUIRE(Cout * outshape[1] * outshape[2] < INT32_MAX && Cin * inshape[1] * inshape[2] < INT32_MAX, KN_ERR_UNSUPPORTED, "int32 index range exceeded");
    b.taps.assign(taps, taps + (size_t)(ntaps * Cout * Cin));
    b.ent_out.assign(ent_out, ent_out + nent);
    b.ent_in.assign(ent_in, ent_in + nent);
    b.ent_tap.assign(ent_tap, ent_tap + nent);
    if (ent_coef) b.ent_coef.assign(ent_coef, ent_coef + nent);
    else b.ent_coef.assign((size_t)nent, 1.0f);
    b.has_last = lastcol != nullptr;
    if (lastcol) {
        const int64_t rows = Cout * outshape[1] * outshape[2] + 1;
        for (int64_t r = 0; r < rows; r++)
            if (lastcol[r] != 0.0f) {
                b.last_rows.push_back(r);
                b.last_vals.push_back(lastcol[r]);
            }
    }
    return convtaps_create_impl(b, out);
    });
}

int kn_convtaps_drop_zero_entries(kn_handle_t h) {
    return guarded([&]() -> int {
    KN_REQUIRE(h != nullptr, KN_ERR_INVALID, "NULL handle");
    KN_REQUIRE(h->kind == KIND_CONVTAPS && h->dense_sub == nullptr, KN_ERR_UNSUPPORTED, "kn_convtaps_drop_zero_entries: not a conv-taps operator");
    std::lock_guard<std::mutex> g(h->lazy_mu);
    if (h->ct.zero_ent != nullptr) return KN_OK;
    const ConvTapsDev& c = h->ct;
    std::vector<int32_t> z;
    for (int64_t t = 0; t < c.ntaps; t++) {
        const float* T = h->h_taps.data() + (size_t)(t * c.Cout * c.Cin);
        bool any = false;
        for (int64_t k = 0; k < c.Cout * c.Cin && !any; k++) any = T[k] != 0.0f;
        if (!any) continue;                                   // a tap that is zero altogether has no slots at all (convtaps_create_impl)
        for (int64_t co = 0; co < c.Cout; co++)
            for (int64_t ci = 0; ci < c.Cin; ci++)
                if (T[co * c.Cin + ci] == 0.0f) {
                    z.push_back((int32_t)t);
                    z.push_back((int32_t)co);
                    z.push_back((int32_t)ci);
                }
    }
    int32_t* d = nullptr;
    int rc = upload(&d, z.data(), z.size());
    if (rc) return rc;
    h->ct.zero_ent = d;
    h->ct.n_zero = (int64_t)(z.size() / 3);
    // The stored-column table of the expansion for the matrix-pipe kernel (kn_csr_mfma.hip, TAPS): per output pixel, channel outer and the pixel's slots
    // by ascending input pixel inner -- the order of convtaps_create_impl's slot lists -- each column as (activation row, value row of tapsT).  Unit
    // coefficients only (a coefficient entry's stored value is fl(coef * tap): the conv pipeline handles those), no duplicate (pixel, pixel) pairs, and
    // at most 64 M columns (512 MB); otherwise the operator simply has no table and KN_FLAG_EXACT takes the conv pipeline.
    if (c.unit_coef && !c.has_dups && c.Cout % 32 == 0) {
        const int64_t HoWo = c.Hout * c.Wout, HiWi = c.Hin * c.Win;
        const size_t nent = h->h_ent_out.size();
        std::vector<char> tap_zero((size_t)std::max<int64_t>(c.ntaps, 1), 1);
        for (int64_t t = 0; t < c.ntaps; t++)
            for (int64_t k = 0; k < c.Cout * c.Cin; k++)
                if (h->h_taps[(size_t)(t * c.Cout * c.Cin + k)] != 0.0f) {
                    tap_zero[(size_t)t] = 0;
                    break;
                }
        std::vector<size_t> order;
        order.reserve(nent);
        for (size_t e = 0; e < nent; e++)
            if (!tap_zero[(size_t)h->h_ent_tap[e]] && h->h_ent_coef[e] != 0.0f) order.push_back(e);
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
            return h->h_ent_out[x] != h->h_ent_out[y] ? h->h_ent_out[x] < h->h_ent_out[y] : h->h_ent_in[x] < h->h_ent_in[y];
        });
        std::vector<int64_t> pp((size_t)HoWo + 1, 0);
        for (size_t k = 0; k < order.size(); k++) pp[(size_t)h->h_ent_out[order[k]] + 1]++;
        for (int64_t o = 0; o < HoWo; o++) pp[(size_t)o + 1] += pp[(size_t)o];
        const int64_t total = (int64_t)order.size() * c.Cin;
        if (total > 0 && total <= ((int64_t)64 << 20) && (c.Cin * HiWi + 1) < INT32_MAX && c.ntaps * c.cin_pad < INT32_MAX) {
            std::vector<int32_t> ex_ptr((size_t)HoWo + 1), ex_tab((size_t)(2 * total));
            for (int64_t o = 0; o <= HoWo; o++) ex_ptr[(size_t)o] = (int32_t)(pp[(size_t)o] * c.Cin);
            for (int64_t o = 0; o < HoWo; o++) {
                const int64_t s0 = pp[(size_t)o], ns = pp[(size_t)o + 1] - s0;
                int32_t* out = ex_tab.data() + (size_t)(2 * s0 * c.Cin);
                for (int64_t ci = 0; ci < c.Cin; ci++)
                    for (int64_t sl = 0; sl < ns; sl++) {
                        const size_t e = order[(size_t)(s0 + sl)];
                        *out++ = (int32_t)(ci * HiWi + h->h_ent_in[e]);
                        *out++ = (int32_t)(h->h_ent_tap[e] * c.cin_pad + ci);
                    }
            }
            // Processing order of the pixels for the table kernel.  Its workgroups (one output pixel x 32 * NRB channels x 256 batch columns) sweep the
            // input channels in the same order but start as earlier ones retire, so the ~128 resident on an XCD sit at staggered phases of the sweep:
            // an activation row is found in that XCD's L2 again only when the workgroup sharing it was dispatched a few places earlier (measured:
            // row-major / ball orders keep the horizontal reuse only, ~1/3 of the gathers miss).  Strips of `w` pixels swept row by row put the vertical
            // neighbour w places back: (w + 2) / w fetches per activation byte.  The candidates are scored on the operator's own structure (no
            // knowledge of the key): visits of an input pixel not seen within the last `window` output pixels count as fetches.  Measured on the AllConvNet
            // forward (FETCH_SIZE x 2 of the seven launches): ball order 15.7 GB, this choice 12.6 GB (conv2 takes w = 8, conv5 w = 2, each the best of
            // {2, 4, 8} when forced); the times do not move (the kernel is bound by vector-ALU issue).
            std::vector<int32_t> ex_order((size_t)HoWo);
            {
                auto fetches = [&](const std::vector<int32_t>& ord, int64_t window) {
                    std::vector<int64_t> seen((size_t)HiWi, -((int64_t)1 << 40));
                    int64_t n = 0;
                    for (int64_t k = 0; k < HoWo; k++) {
                        const int64_t o = ord[(size_t)k];
                        for (int64_t sl = pp[(size_t)o]; sl < pp[(size_t)o + 1]; sl++) {
                            const int64_t in = h->h_ent_in[order[(size_t)sl]];
                            if (k - seen[(size_t)in] > window) n++;
                            seen[(size_t)in] = k;
                        }
                    }
                    return n;
                };
                auto strips = [&](int64_t w) {
                    std::vector<int32_t> ord;
                    ord.reserve((size_t)HoWo);
                    for (int64_t x0 = 0; x0 < c.Wout; x0 += w)
                        for (int64_t y = 0; y < c.Hout; y++)
                            for (int64_t x = x0; x < std::min(x0 + w, c.Wout); x++) ord.push_back((int32_t)(y * c.Wout + x));
                    return ord;
                };
                const int nrb = c.Cout % 96 == 0 ? 3 : (c.Cout % 64 == 0 ? 2 : 1);       // (the window below was tuned with three row blocks per workgroup; with one -- exact_table_launch -- its choices
                                                                                     // still measured best: conv2 strips of 8 / 4 / 2 / 16 pixels 3.4 / 4.3 / 4.3 / 7.9 GB)
                const int64_t n_cc = c.Cout / (32 * nrb);
                // places a sharer may lie back: the rows gathered meanwhile by the resident workgroups (~1.5 KB per place and input channel at 256 columns) within half an L2 slice
                const int64_t env_window = c.tune.table_window, env_strip = c.tune.table_strip;      // (diagnostic build: A/B knobs; product: the rule)
                const int64_t window = env_window > 0 ? env_window : std::max<int64_t>(2, std::min<int64_t>(64, (2048 * 2) / (3 * c.Cin * n_cc)));
                std::vector<int32_t> best((size_t)HoWo);
                KN_HIP(hipMemcpy(best.data(), c.pix_order, (size_t)HoWo * sizeof(int32_t), hipMemcpyDeviceToHost));
                int64_t best_n = fetches(best, window);
                if (env_strip > 0) {
                    best = strips(env_strip);
                } else if (env_strip < 0) {
                    for (int64_t w = 2; w <= c.Wout; w *= 2) {
                        std::vector<int32_t> cand = strips(w);
                        const int64_t n = fetches(cand, window);
                        if (n < best_n) {
                            best_n = n;
                            best.swap(cand);
                        }
                    }
                }
                ex_order = best;
            }
            int32_t *dp = nullptr, *dt = nullptr, *dord = nullptr;
            if ((rc = upload(&dp, ex_ptr.data(), ex_ptr.size()))) return rc;
            if ((rc = upload(&dt, ex_tab.data(), ex_tab.size())) || (rc = upload(&dord, ex_order.data(), ex_order.size()))) {
                (void)hipFree(dp);
                if (dt) (void)hipFree(dt);
                return rc;
            }
            h->ct.ex_ptr = dp;
            h->ct.ex_tab = dt;
            h->ct.ex_order = dord;
        }
    }
    return KN_OK;
    });
}

int kn_dense_create(int64_t rows, int64_t cols, const float* W, kn_handle_t* out) {
    return guarded([&]() -> int {
    KN_REQUIRE(out != nullptr, KN_ERR_INVALID, "out handle is NULL");
    *out = nullptr;
    KN_REQUIRE(rows >= 2 && cols >= 2 && W != nullptr, KN_ERR_INVALID, "bad argument");
    const int64_t outs = rows - 1, ins = cols - 1;
    KN_REQUIRE(ins % 256 == 0, KN_ERR_UNSUPPORTED, "dense operator needs (cols-1) % 256 == 0");
    for (int64_t c = 0; c < ins; c++) KN_REQUIRE(W[(size_t)(outs * cols + c)] == 0.0f, KN_ERR_INVALID, "last row is not homogeneous (e_last)");
    // K slices: feature f = ci*S + s  <->  (input channel ci, pseudo-pixel s); S chosen so a launch has >= ~1000 workgroups
    int64_t S = 16;
    while (S > 1 && ins % (16 * S) != 0) S >>= 1;
    const int64_t cinp = ins / S;
    ConvBuild b;
    b.inshape[0] = cinp; b.inshape[1] = 1; b.inshape[2] = S;
    b.outshape[0] = outs; b.outshape[1] = 1; b.outshape[2] = S;
    b.has_last = false;
    b.taps.resize((size_t)(S * outs * cinp));
    for (int64_t s = 0; s < S; s++)
        for (int64_t co = 0; co < outs; co++) {
            const float* wr = W + (size_t)(co * cols);
            float* t = b.taps.data() + (size_t)((s * outs + co) * cinp);
            for (int64_t ci = 0; ci < cinp; ci++) t[ci] = wr[ci * S + s];
        }
    for (int64_t s = 0; s < S; s++) {
        b.ent_out.push_back((int32_t)s);
        b.ent_in.push_back((int32_t)s);
        b.ent_tap.push_back((int32_t)s);
        b.ent_coef.push_back(1.0f);
    }
    kn_operator* sub = nullptr;
    int rc = convtaps_create_impl(b, &sub);
    if (rc) return rc;
    OperatorPtr subp(sub);
    OperatorPtr h(new kn_operator());
    h->kind = KIND_DENSE;
    h->device = sub->device;
    h->rows = rows;
    h->cols = cols;
    h->dense_sub = subp.release();
    h->dense_splits = S;
    std::vector<float> lastcol((size_t)rows);
    int64_t nnz = 0;
    for (int64_t r = 0; r < rows; r++) {
        lastcol[(size_t)r] = W[(size_t)(r * cols + ins)];
        for (int64_t c = 0; c < cols; c++) nnz += (W[(size_t)(r * cols + c)] != 0.0f);
    }
    h->nnz_stored = nnz;
    h->nnz_expanded = nnz;
    if ((rc = upload(&h->dense_lastcol, lastcol.data(), lastcol.size()))) return rc;
    *out = h.release();
    return KN_OK;
    });
}

int kn_chain_create(int64_t n_ops, const kn_handle_t* ops, const uint32_t* flags, kn_handle_t* out) {
    return guarded([&]() -> int {
    KN_REQUIRE(out != nullptr, KN_ERR_INVALID, "out handle is NULL");
    *out = nullptr;
    KN_REQUIRE(ops != nullptr && n_ops >= 1, KN_ERR_INVALID, "no operators");
    int cur = -1;
    KN_HIP(hipGetDevice(&cur));
    KN_REQUIRE(ops[0] != nullptr && cur == ops[0]->device, KN_ERR_INVALID, "create the chain under the device its operators live on");
    OperatorPtr h(new kn_operator());
    h->kind = KIND_CHAIN;
    h->device = cur;
    int64_t nnz = 0;
    int rc = chain_create(n_ops, ops, flags, &h->chain, &h->rows, &h->cols, &nnz);
    if (rc) return rc;
    h->nnz_stored = nnz;
    h->nnz_expanded = nnz;
    *out = h.release();
    return KN_OK;
    });
}

int kn_destroy(kn_handle_t h) {
    return guarded([&]() -> int {
    if (!h) return KN_OK;
    if (h->chain) chain_free(h->chain);
    if (h->exact) kn_destroy(h->exact);
    if (h->dense_sub) kn_destroy(h->dense_sub);
    if (h->dense_lastcol) (void)hipFree(h->dense_lastcol);
    for (auto& kv : h->dense_ws)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    for (auto& r : h->dense_ws_retired) {
        if (r.done) (void)hipEventDestroy(r.done);
        if (r.ptr) (void)hipFree(r.ptr);
    }
    csr_free(h->csr);
    convtaps_free(h->ct);
    delete h;
    return KN_OK;
    });
}

int kn_nnz(kn_handle_t h, int64_t* nnz) {
    return guarded([&]() -> int {
    KN_REQUIRE(h && nnz, KN_ERR_INVALID, "NULL argument");
    *nnz = h->nnz_stored;
    return KN_OK;
    });
}

int kn_nnz_expanded(kn_handle_t h, int64_t* nnz) {
    return guarded([&]() -> int {
    KN_REQUIRE(h && nnz, KN_ERR_INVALID, "NULL argument");
    *nnz = h->nnz_expanded;
    return KN_OK;
    });
}

int kn_shape(kn_handle_t h, int64_t* rows, int64_t* cols) {
    return guarded([&]() -> int {
    KN_REQUIRE(h && rows && cols, KN_ERR_INVALID, "NULL argument");
    *rows = h->rows;
    *cols = h->cols;
    return KN_OK;
    });
}

int kn_export_csr(kn_handle_t h, int32_t* indptr, int32_t* indices, float* data) {
    return guarded([&]() -> int {
    KN_REQUIRE(h && indptr, KN_ERR_INVALID, "NULL argument");
    if (h->kind == KIND_CSR) {
        KN_HIP(hipMemcpy(indptr, h->csr.indptr, sizeof(int32_t) * (size_t)(h->rows + 1), hipMemcpyDeviceToHost));
        if (h->csr.nnz > 0) {
            KN_REQUIRE(indices && data, KN_ERR_INVALID, "NULL argument");
            KN_HIP(hipMemcpy(indices, h->csr.indices, sizeof(int32_t) * (size_t)h->csr.nnz, hipMemcpyDeviceToHost));
            KN_HIP(hipMemcpy(data, h->csr.data, sizeof(float) * (size_t)h->csr.nnz, hipMemcpyDeviceToHost));
        }
        return KN_OK;
    }
    KN_REQUIRE(h->kind != KIND_CSR64, KN_ERR_UNSUPPORTED, "kn_export_csr: a float64 operator is exported by kn_export_csr_f64");
    KN_REQUIRE(h->kind == KIND_CONVTAPS, KN_ERR_UNSUPPORTED, "kn_export_csr: dense operators and chains are exported by their creator (the host keeps the matrices)");
    std::vector<int32_t> ip, ix;
    std::vector<float> dt;
    std::vector<int64_t> lr;
    std::vector<float> lv;
    last_pairs(h, lr, lv);
    convtaps_expand(h, lr, lv, ip, ix, dt);
    KN_REQUIRE((int64_t)ix.size() == h->nnz_expanded, KN_ERR_UNSUPPORTED, "duplicate (row,col) entries in a conv-taps operator: expanded nnz differs");
    std::memcpy(indptr, ip.data(), sizeof(int32_t) * ip.size());
    if (!ix.empty()) {
        KN_REQUIRE(indices && data, KN_ERR_INVALID, "NULL argument");
        std::memcpy(indices, ix.data(), sizeof(int32_t) * ix.size());
        std::memcpy(data, dt.data(), sizeof(float) * dt.size());
    }
    return KN_OK;
    });
}

int kn_export_csr_f64(kn_handle_t h, int32_t* indptr, int32_t* indices, double* data) {
    return guarded([&]() -> int {
    KN_REQUIRE(h && indptr, KN_ERR_INVALID, "NULL argument");
    KN_REQUIRE(h->kind == KIND_CSR64, KN_ERR_UNSUPPORTED, "kn_export_csr_f64: not a float64 CSR operator (kn_csr_create_f64)");
    KN_HIP(hipMemcpy(indptr, h->csr.indptr, sizeof(int32_t) * (size_t)(h->rows + 1), hipMemcpyDeviceToHost));
    if (h->csr.nnz > 0) {
        KN_REQUIRE(indices && data, KN_ERR_INVALID, "NULL argument");
        KN_HIP(hipMemcpy(indices, h->csr.indices, sizeof(int32_t) * (size_t)h->csr.nnz, hipMemcpyDeviceToHost));
        KN_HIP(hipMemcpy(data, h->csr.data64, sizeof(double) * (size_t)h->csr.nnz, hipMemcpyDeviceToHost));
    }
    return KN_OK;
    });
}

// split-K partial-sum workspace of a dense operator for stream `s`, grown to `n_vecs` batch columns on demand.  The growing call is not
// capturable in a HIP graph: run one eager forward per stream and batch size first (KeyedModel.capture does), or kn_reserve_workspace.
static int dense_workspace(kn_handle_t h, hipStream_t s, int64_t n_vecs, float** out) {
    const int64_t outs = h->rows - 1, S = h->dense_splits;
    std::lock_guard<std::mutex> g(h->lazy_mu);
    kn_operator::DenseWs& w = h->dense_ws[s];
    if (w.vecs < n_vecs) {
        // reap retired buffers whose last possible reader has finished, then retire this stream's outgrown one behind an event
        for (size_t k = 0; k < h->dense_ws_retired.size();) {
            kn_operator::Retired& r = h->dense_ws_retired[k];
            if (r.done == nullptr || hipEventQuery(r.done) == hipSuccess) {
                if (r.done) (void)hipEventDestroy(r.done);
                (void)hipFree(r.ptr);
                h->dense_ws_retired.erase(h->dense_ws_retired.begin() + (long)k);
            } else {
                k++;
            }
        }
        (void)hipGetLastError();                            // hipEventQuery's hipErrorNotReady is not an error of this call
        if (w.ptr) {
            kn_operator::Retired r;
            r.ptr = w.ptr;                                  // launches already queued on `s` may still use it
            if (hipEventCreateWithFlags(&r.done, hipEventDisableTiming) != hipSuccess || hipEventRecord(r.done, s) != hipSuccess) {
                if (r.done) (void)hipEventDestroy(r.done);
                r.done = nullptr;
                (void)hipStreamSynchronize(s);              // no event: wait for the stream, then the buffer is free at the next reap
            }
            h->dense_ws_retired.push_back(r);
        }
        w.ptr = nullptr;
        w.vecs = 0;
        hipError_t e = hipMalloc((void**)&w.ptr, sizeof(float) * (size_t)(outs * S) * (size_t)n_vecs);
        if (e != hipSuccess) {
            w.ptr = nullptr;
            return fail(e == hipErrorOutOfMemory ? KN_ERR_NOMEM : KN_ERR_HIP, std::string("dense workspace hipMalloc: ") + hipGetErrorString(e) +
                        " (during a HIP-graph capture: run one eager kn_spmm on the capture stream first, or call kn_reserve_workspace)");
        }
        w.vecs = n_vecs;
    }
    *out = w.ptr;
    return KN_OK;
}

// kn_spmm / kn_spmm_screen.  `absmax` (device float or null): raised to max |Y| -- folded into the stores of the matrix-core kernels, one
// extra pass over Y behind every other kernel family.
static int spmm_impl(kn_handle_t h, const float* x_dev, int64_t ldx, int64_t n_vecs, float* y_dev, int64_t ldy, uint32_t flags, float* absmax, void* stream) {
    KN_REQUIRE(h != nullptr, KN_ERR_INVALID, "NULL handle");
    if (plan_sink() == nullptr) KN_HOST_ONLY_GUARD();
    KN_REQUIRE(n_vecs >= 0, KN_ERR_INVALID, "negative n_vecs");
    if (n_vecs == 0 || h->rows == 0) return KN_OK;
    KN_REQUIRE(x_dev && y_dev, KN_ERR_INVALID, "NULL activation pointer");
    KN_REQUIRE(ldx >= n_vecs && ldy >= n_vecs, KN_ERR_SHAPE, "leading dimension smaller than n_vecs");
    KN_REQUIRE(n_vecs < INT32_MAX, KN_ERR_UNSUPPORTED, "n_vecs too large");
    KN_REQUIRE(x_dev != y_dev, KN_ERR_INVALID, "x and y alias");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    {
        // the operator lives in ONE device's HBM: running it from another device would dereference foreign memory
        int cur = -1;
        KN_HIP(hipGetDevice(&cur));
        KN_REQUIRE(cur == h->device, KN_ERR_INVALID, "operator was created on another HIP device than the current one (create it under the device of x)");
    }
    int rc = KN_OK;
    bool fused = false;
    if (h->kind == KIND_CSR) {
        rc = csr_spmm(h->csr, x_dev, ldx, n_vecs, y_dev, ldy, flags, s, absmax, &fused);
    } else if (h->kind == KIND_CSR64) {
        rc = csr_f64_spmm<float>(h->csr, x_dev, ldx, n_vecs, y_dev, ldy, flags, s);      // the f64 result rounded to f32 once (what the next layer consumes)
    } else if (h->kind == KIND_CHAIN) {
        rc = chain_forward(h->chain, x_dev, ldx, n_vecs, y_dev, ldy, s);   // order-preserving by construction; ReLU flags were fixed at create
    } else if (h->kind == KIND_DENSE) {
        KN_REQUIRE(!(flags & KN_FLAG_EXACT), KN_ERR_UNSUPPORTED, "KN_FLAG_EXACT on a dense (MFMA) operator: create it with kn_csr_create instead");
        const int64_t outs = h->rows - 1, S = h->dense_splits;
        float* ws = nullptr;
        if (plan_sink() == nullptr) {
            rc = dense_workspace(h, s, n_vecs, &ws);
            if (rc) return rc;
        }
        rc = convtaps_spmm(h->dense_sub->ct, outs * S, h->cols - 1, x_dev, ldx, n_vecs, ws, n_vecs, 0, s);
        if (rc) return rc;
        rc = dense_reduce(ws, n_vecs, outs, S, h->dense_lastcol, x_dev + (h->cols - 1) * ldx, y_dev, ldy, n_vecs, (flags & KN_FLAG_RELU) ? 1 : 0, s);
    } else {
        // KN_FLAG_EXACT is honoured inside convtaps_spmm by the order-preserving kernel on the factored operator
        // lazily built side tables: double-checked under Handle::lazy_mu; both builders publish their pointer (release) only after the data is in HBM
        if ((flags & KN_FLAG_BF16X3) && !(flags & KN_FLAG_EXACT) && __atomic_load_n(&h->ct.tapsB, __ATOMIC_ACQUIRE) == nullptr && plan_sink() == nullptr) {
            std::lock_guard<std::mutex> g(h->lazy_mu);           // bf16 planes of the taps, once (not capturable: like any first use)
            rc = convtaps_build_bf16(h->ct, h->h_taps);
            if (rc) return rc;
        }
        if ((flags & KN_FLAG_EXACT) && __atomic_load_n(&h->ct.fill_rec, __ATOMIC_ACQUIRE) == nullptr && plan_sink() == nullptr && convtaps_fill_ok(h->ct)) {
            std::lock_guard<std::mutex> g(h->lazy_mu);           // record lists of the filled-in order-preserving kernel, once (built on the device on this stream, waited for, then published)
            rc = convtaps_build_fill(h->ct, s);
            if (rc) return rc;
        }
        rc = convtaps_spmm(h->ct, h->rows, h->cols, x_dev, ldx, n_vecs, y_dev, ldy, flags, s, absmax, &fused);
    }
    if (rc) return rc;
    if (absmax && !fused) return absmax_pass(y_dev, h->rows, ldy, n_vecs, absmax, s);
    return KN_OK;
}

int kn_spmm(kn_handle_t h, const float* x_dev, int64_t ldx, int64_t n_vecs, float* y_dev, int64_t ldy, uint32_t flags, void* stream) {
    return guarded([&]() -> int { return spmm_impl(h, x_dev, ldx, n_vecs, y_dev, ldy, flags, nullptr, stream); });
}

int kn_spmm_screen(kn_handle_t h, const float* x_dev, int64_t ldx, int64_t n_vecs, float* y_dev, int64_t ldy, uint32_t flags, float* y_absmax_dev, void* stream) {
    return guarded([&]() -> int { return spmm_impl(h, x_dev, ldx, n_vecs, y_dev, ldy, flags, y_absmax_dev, stream); });
}

int kn_spmm_planes(kn_handle_t h, const float* x_dev, int64_t ldx, int64_t x_plane_stride, int64_t n_planes, int64_t n_vecs, float* y_dev, int64_t ldy, int64_t y_plane_stride,
                   uint32_t flags, void* stream) {
    return guarded([&]() -> int {
    KN_REQUIRE(h != nullptr, KN_ERR_INVALID, "NULL handle");
    KN_REQUIRE(h->kind == KIND_CSR, KN_ERR_UNSUPPORTED, "kn_spmm_planes: a float32 CSR operator only (loop over kn_spmm for the others)");
    if (plan_sink() == nullptr) KN_HOST_ONLY_GUARD();
    KN_REQUIRE(n_vecs >= 0 && n_vecs < INT32_MAX && n_planes >= 0, KN_ERR_INVALID, "n_vecs / n_planes out of range");
    if (n_vecs == 0 || n_planes == 0 || h->rows == 0) return KN_OK;
    KN_REQUIRE(x_dev && y_dev, KN_ERR_INVALID, "NULL activation pointer");
    KN_REQUIRE(ldx >= n_vecs && ldy >= n_vecs, KN_ERR_SHAPE, "leading dimension smaller than n_vecs");
    KN_REQUIRE(n_planes == 1 || (x_plane_stride >= h->cols * ldx && y_plane_stride >= h->rows * ldy), KN_ERR_SHAPE, "planes overlap (stride smaller than one block)");
    KN_REQUIRE(x_dev != y_dev, KN_ERR_INVALID, "x and y alias");
    int cur = -1;
    KN_HIP(hipGetDevice(&cur));
    KN_REQUIRE(cur == h->device, KN_ERR_INVALID, "operator was created on another HIP device than the current one (create it under the device of x)");
    return csr_spmm_planes(h->csr, x_dev, ldx, x_plane_stride, n_planes, n_vecs, y_dev, ldy, y_plane_stride, flags, reinterpret_cast<hipStream_t>(stream));
    });
}

int kn_spmm_f64(kn_handle_t h, const float* x_dev, int64_t ldx, int64_t n_vecs, double* y_dev, int64_t ldy, uint32_t flags, void* stream) {
    return guarded([&]() -> int {
    KN_REQUIRE(h != nullptr, KN_ERR_INVALID, "NULL handle");
    KN_REQUIRE(h->kind == KIND_CSR64, KN_ERR_UNSUPPORTED, "kn_spmm_f64: not a float64 CSR operator (kn_csr_create_f64); float32 operators return float32 (kn_spmm)");
    if (plan_sink() == nullptr) KN_HOST_ONLY_GUARD();
    KN_REQUIRE(n_vecs >= 0 && n_vecs < INT32_MAX, KN_ERR_INVALID, "n_vecs out of range");
    if (n_vecs == 0 || h->rows == 0) return KN_OK;
    KN_REQUIRE(x_dev && y_dev, KN_ERR_INVALID, "NULL activation pointer");
    KN_REQUIRE(ldx >= n_vecs && ldy >= n_vecs, KN_ERR_SHAPE, "leading dimension smaller than n_vecs");
    KN_REQUIRE((const void*)x_dev != (const void*)y_dev, KN_ERR_INVALID, "x and y alias");
    int cur = -1;
    KN_HIP(hipGetDevice(&cur));
    KN_REQUIRE(cur == h->device, KN_ERR_INVALID, "operator was created on another HIP device than the current one (create it under the device of x)");
    return csr_f64_spmm<double>(h->csr, x_dev, ldx, n_vecs, y_dev, ldy, flags, reinterpret_cast<hipStream_t>(stream));
    });
}

int kn_absmax(const float* x_dev, int64_t rows, int64_t ld, int64_t n_vecs, float* absmax_dev, void* stream) {
    return guarded([&]() -> int {
    KN_HOST_ONLY_GUARD();
    KN_REQUIRE(rows >= 0 && n_vecs >= 0, KN_ERR_INVALID, "negative size");
    KN_REQUIRE(absmax_dev != nullptr && (x_dev || rows * n_vecs == 0), KN_ERR_INVALID, "NULL pointer");
    KN_REQUIRE(ld >= n_vecs, KN_ERR_SHAPE, "leading dimension smaller than n_vecs");
    return absmax_pass(x_dev, rows, ld, n_vecs, absmax_dev, reinterpret_cast<hipStream_t>(stream));
    });
}

int kn_release_side_tables(kn_handle_t h) {
    return guarded([&]() -> int {
    KN_REQUIRE(h != nullptr, KN_ERR_INVALID, "NULL handle");
    if (h->kind != KIND_CONVTAPS) return KN_OK;
    std::lock_guard<std::mutex> g(h->lazy_mu);
    int32_t* rec = __atomic_load_n(&h->ct.fill_rec, __ATOMIC_ACQUIRE);
    uint16_t* tb = __atomic_load_n(&h->ct.tapsB, __ATOMIC_ACQUIRE);
    if (!rec && !tb) return KN_OK;
    int cur = -1;
    KN_HIP(hipGetDevice(&cur));
    KN_REQUIRE(cur == h->device, KN_ERR_INVALID, "operator was created on another HIP device than the current one");
    KN_HIP(hipDeviceSynchronize());                               // a launch on ANY stream may still be reading the tables (rare call: a layer changed its contract)
    __atomic_store_n(&h->ct.fill_rec, (int32_t*)nullptr, __ATOMIC_RELEASE);
    __atomic_store_n(&h->ct.tapsB, (uint16_t*)nullptr, __ATOMIC_RELEASE);
    if (rec) (void)hipFree(rec);
    if (tb) (void)hipFree(tb);
    return KN_OK;
    });
}

int kn_reserve_workspace(kn_handle_t h, int64_t n_vecs, void* stream) {
    return guarded([&]() -> int {
    KN_REQUIRE(h != nullptr, KN_ERR_INVALID, "NULL handle");
    KN_REQUIRE(n_vecs >= 0 && n_vecs < INT32_MAX, KN_ERR_INVALID, "n_vecs out of range");
    if (h->kind != KIND_DENSE || n_vecs == 0) return KN_OK;      // only a dense (split-K) operator keeps per-call state (its partial sums)
    int cur = -1;
    KN_HIP(hipGetDevice(&cur));
    KN_REQUIRE(cur == h->device, KN_ERR_INVALID, "operator was created on another HIP device than the current one");
    float* ws = nullptr;
    return dense_workspace(h, reinterpret_cast<hipStream_t>(stream), n_vecs, &ws);
    });
}

int kn_spmm_plan(kn_handle_t h, int64_t n_vecs, int64_t ldx, int64_t ldy, uint32_t flags, char* buf, int64_t buf_len) {
    return guarded([&]() -> int {
    KN_REQUIRE(buf != nullptr && buf_len > 0, KN_ERR_INVALID, "NULL buffer");
    buf[0] = 0;
    PlanSink sink;
    plan_sink() = &sink;
    // 16-byte aligned stand-ins for the activation pointers: the dispatch logic looks at their alignment only, and nothing is launched
    const int rc = kn_spmm(h, reinterpret_cast<const float*>((uintptr_t)4096), ldx, n_vecs, reinterpret_cast<float*>((uintptr_t)8192), ldy, flags, nullptr);
    plan_sink() = nullptr;
    if (rc) return rc;
    if (h->kind == KIND_CSR || h->kind == KIND_CSR64) sink.text += h->csr.tune.describe();      // options recorded at create that differ from the defaults
    else if (h->kind == KIND_CONVTAPS) sink.text += h->ct.tune.describe();
    else if (h->kind == KIND_DENSE) sink.text += h->dense_sub->ct.tune.describe();
    std::strncpy(buf, sink.text.c_str(), (size_t)buf_len - 1);
    buf[buf_len - 1] = 0;
    return KN_OK;
    });
}

int kn_relu(float* y_dev, int64_t rows, int64_t ld, int64_t n_vecs, void* stream) {
    return guarded([&]() -> int {
    KN_HOST_ONLY_GUARD();
    KN_REQUIRE(y_dev || rows * n_vecs == 0, KN_ERR_INVALID, "NULL pointer");
    KN_REQUIRE(ld >= n_vecs, KN_ERR_SHAPE, "leading dimension smaller than n_vecs");
    return relu_inplace(y_dev, rows, ld, n_vecs, reinterpret_cast<hipStream_t>(stream));
    });
}

int kn_affine_to_linear(const float* x_dev, int64_t n, int64_t d, float* out_dev, int64_t ldo, void* stream) {
    return guarded([&]() -> int {
    KN_HOST_ONLY_GUARD();
    KN_REQUIRE(n >= 0 && d >= 0, KN_ERR_INVALID, "negative size");
    KN_REQUIRE((x_dev || n * d == 0) && (out_dev || n == 0), KN_ERR_INVALID, "NULL pointer");
    KN_REQUIRE(ldo >= n, KN_ERR_SHAPE, "leading dimension smaller than n");
    return affine_to_linear(x_dev, n, d, out_dev, ldo, reinterpret_cast<hipStream_t>(stream));
    });
}

int kn_linear_to_affine(const float* y_dev, int64_t ldy, int64_t n, int64_t d, float* out_dev, float* maxdev_dev, void* stream) {
    return guarded([&]() -> int {
    KN_HOST_ONLY_GUARD();
    KN_REQUIRE(n >= 0 && d >= 0, KN_ERR_INVALID, "negative size");
    KN_REQUIRE((y_dev || n == 0) && (out_dev || n * d == 0), KN_ERR_INVALID, "NULL pointer");
    KN_REQUIRE(ldy >= n, KN_ERR_SHAPE, "leading dimension smaller than n");
    return linear_to_affine(y_dev, ldy, n, d, out_dev, maxdev_dev, reinterpret_cast<hipStream_t>(stream));
    });
}

}  // extern "C"
