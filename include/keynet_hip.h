/*
 * keynet_hip.h -- C ABI of libkeynet_hip.so: the MI355X (gfx950) engine behind the keyed forward of visym/keynet.
 *
 * Drop-in boundary.  The reference has exactly one plug-in seam for the forward path: the object stored in
 * KeyedLayer.W (keynet/layer.py:81-82), selected by layergen(..., backend=) (keynet/system.py:303-314), whose
 * torchdot() is the hot loop (keynet/layer.py:92).  Each entry point below names the reference interface it
 * replaces.  Plain pointers and sizes only; no torch / numpy / scipy types cross this boundary.
 *
 * Conventions
 *   - every function returns KN_OK (0) or a KN_ERR_* code; kn_last_error() gives the message (thread-local);
 *     no C++ exception crosses the boundary.
 *   - "host" pointers are ordinary CPU memory and are COPIED during the call (caller keeps ownership);
 *     "dev" pointers are HIP device memory owned by the caller (PyTorch-ROCm allocations in the Python host).
 *   - activations are FEATURE-MAJOR: X is [n_features, n_vecs] f32, element (f, b) at x[f*ldx + b].  This is the
 *     layout the reference hands to scipy (x_affine.t() -> ravel, keynet/layer.py:92 + scipy _matmul_multivector).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All compute calls are asynchronous and
 *     stream-ordered; a handle may be used from several streams/threads concurrently (it is immutable after create).
 */
#ifndef KEYNET_HIP_H
#define KEYNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KN_ABI_VERSION 4

enum kn_status {
    KN_OK = 0,
    KN_ERR_INVALID = 1,    /* bad argument (NULL, negative size, index out of range): the reference's `assert`s */
    KN_ERR_SHAPE = 2,      /* non-conformal shapes: keynet/sparse.py:605 */
    KN_ERR_HIP = 3,        /* a HIP runtime call failed (message has the hipError string) */
    KN_ERR_NOMEM = 4,
    KN_ERR_NODEVICE = 5,   /* no gfx950 device visible */
    KN_ERR_UNSUPPORTED = 6
};

/* kn_spmm flags */
#define KN_FLAG_RELU   1u  /* fuse torch.nn.functional.relu over the whole output incl. the homogeneous row
                              (unkeyed nn.ReLU after a keyed layer: keynet/system.py:92; keyed ReLU: keynet/layer.py:93) */
#define KN_FLAG_EXACT  2u  /* demand the reference's accumulation order and mul-then-add rounding (bit-exact with
                              scipy csr_matvecs).  CSR operators always honour it; conv-tap operators switch from the
                              MFMA kernel to an order-preserving VALU kernel that walks the factored operator in the
                              expansion's column order (no CSR is materialised: works at VGG-16 scale).
                              What "bit-exact" means per operator: a CSR / tiled operator holds the reference's own stored
                              values, so the result IS scipy's.  A conv-taps operator keyed directly in factored form
                              (kn_convtaps_create) whose entries carry float coefficients holds the TERMS coef * tap of
                              every stored value; a pixel pair hit by several taps is one stored value = the f32 sum of
                              its terms in entry order.  The flag then gives the order-preserving product of THOSE
                              stored values -- bit-equal to scipy csr_matvecs on kn_export_csr of this handle.  The
                              reference formed the same values inside scipy's SpGEMM (keynet/layer.py:35) in SpGEMM's
                              order: they agree to ~2e-6 relative, not bit for bit, which is inside the reference's
                              1e-5 criterion for float keys and outside "bit-exact with the reference" */

#define KN_FLAG_BF16X3 4u  /* allow a conv-taps operator to form its f32 products on the bf16 matrix pipe: operands split exactly into three
                              bf16 parts, six of the nine cross products kept (the dropped ones are <= 2^-23 of a product), f32 accumulate.
                              ~1 ulp per term away from the f32-MFMA result; never bit-exact; ignored with KN_FLAG_EXACT and by operators /
                              operands that do not qualify (Cin % 16, Cout > 64, n_vecs % 128).  No reference counterpart: the Python host
                              sets it only for layers whose calibration measured the result inside the float-key tolerance. */

typedef struct kn_operator* kn_handle_t;   /* opaque keyed operator resident in HBM */

int         kn_abi_version(void);
const char* kn_last_error(void);
/* number of visible HIP devices and the gcnArchName of the current one (buf may be NULL) */
int         kn_device_info(int* n_devices, char* arch_buf, int64_t arch_buf_len);

/* ---- operators ---------------------------------------------------------------------------------------------- */

/* Replaces keynet.sparse.SparseMatrix(A) for a scipy CSR matrix (keynet/sparse.py:419-425).
 * indptr[rows+1], indices[nnz], data[nnz] are HOST arrays; the STORED order of each row is preserved on the device
 * (keyed matrices are SpGEMM outputs: unsorted, non-canonical; sorting them changes results -- SURVEY 8c). */
int kn_csr_create(int64_t rows, int64_t cols, int64_t nnz,
                  const int32_t* indptr, const int32_t* indices, const float* data,
                  kn_handle_t* out);

/* The same container when its scipy matrix is FLOAT64 (SparseMatrix keeps `A.dtype`, keynet/sparse.py:423; the public challenge key-net the
 * reference ships, demo/keynet_challenge_lenet_10AUG20.pkl + demo/challenge.ipynb cell 5, carries float64 conv / pool operators).  torchdot
 * (keynet/sparse.py:488-492) coerces x to float32 and hands both to scipy: numpy up-casts, csr_matvecs runs in float64 (x up-cast element by
 * element, f64 multiply then f64 add in stored order) and the layer returns a float64 block that the next layer's coercion rounds to f32.
 * kn_spmm on such a handle writes that block rounded to f32 ONCE (what the next layer consumes); kn_spmm_f64 writes the float64 block itself
 * (what the reference's torchdot returns).  Bit-exact with scipy either way; KN_FLAG_EXACT is implied, kn_chain_create refuses the handle. */
int kn_csr_create_f64(int64_t rows, int64_t cols, int64_t nnz,
                      const int32_t* indptr, const int32_t* indices, const double* data,
                      kn_handle_t* out);

/* Replaces keynet.sparse.TiledMatrix / DiagonalTiledMatrix (keynet/sparse.py:517-571, 657-687):
 * blocks[nblocks][3] = (row0, col0, k) as iterated by __iter__; tile k = COO entries tile_ptr[k]..tile_ptr[k+1] of
 * (tile_row, tile_col, tile_val) relative to the block origin.  Semantics = tosparse() (keynet/sparse.py:621-641). */
int kn_tiled_create(int64_t rows, int64_t cols,
                    int64_t nblocks, const int64_t* blocks,
                    int64_t ntiles, const int64_t* tile_ptr,
                    const int32_t* tile_row, const int32_t* tile_col, const float* tile_val,
                    kn_handle_t* out);

/* Replaces keynet.sparse.Conv2dTiledMatrix (keynet/sparse.py:690-776): spatial blocks x dense channel matrices.
 *   inshape/outshape = (C,H,W); rows = Cout*Hout*Wout (+1), cols = Cin*Hin*Win (+1)
 *   blocks[nblocks][3]   = (i, j, k) incl. the bias blocks (j == Cin*Hin*Win)             (sparse.py:753,773,776)
 *   tile_keys[nent][3]   = (it, jt, k) in dict order                                      (sparse.py:764)
 *   tile_isbias[nent]    = 1 for the 1x1 bias tiles                                       (sparse.py:767-772)
 *   tile_chan            = f32[n_chan][Cout][Cin] for the non-bias entries, in tile_keys order
 *   tile_bias            = f32[n_bias] for the bias entries, in tile_keys order
 * Expansion rule (sparse.py:802-812): W[i+it+ic*HoWo, j+jt+jc*HiWi] = tile[(it,jt,k)][ic,jc]. */
int kn_conv2dtiled_create(int64_t rows, int64_t cols,
                          const int64_t inshape[3], const int64_t outshape[3],
                          int64_t nblocks, const int64_t* blocks,
                          int64_t nent, const int64_t* tile_keys, const uint8_t* tile_isbias,
                          const float* tile_chan, const float* tile_bias,
                          kn_handle_t* out);

/* The same operator in factored form  W = sum_e coef_e * ( taps[tap_e] (x) E[out_e, in_e] )  + last column + e_last row,
 * i.e. what Conv2dTiledMatrix stores after de-duplicating its channel matrices.  Used by the direct-to-tiled keying
 * (never materialises the 15 G-nnz CSR of keyed VGG-16; keynet/layer.py:24-41 + sparse.py:720-776 collapsed).
 *   taps      f32[ntaps][Cout][Cin]
 *   ent_*     [nent]: output pixel (0..HoWo), input pixel (0..HiWi), tap id, coefficient
 *   lastcol   f32[Cout*HoWo + 1] = W[:, Cin*HiWi] (bias column incl. the homogeneous 1 at the end), or NULL => no
 *             homogeneous row/column at all (bias=False operators, rows == Cout*HoWo). */
int kn_convtaps_create(const int64_t inshape[3], const int64_t outshape[3],
                       int64_t ntaps, const float* taps,
                       int64_t nent, const int32_t* ent_out, const int32_t* ent_in, const int32_t* ent_tap, const float* ent_coef,
                       const float* lastcol,
                       kn_handle_t* out);

/* Declares that the ZERO-valued entries of a kn_convtaps_create operator's expansion are ABSENT from the reference operator it stands for.
 * A tiled reference operator keeps the zeros of its dense channel matrices as explicit entries (keynet/sparse.py:802-812), and 0 * Inf = NaN
 * reaches its output like any other entry; an UNTILED keyed conv layer is a scipy CSR (keynet/layer.py:24-41) from which the keying SpGEMM
 * has dropped every exact zero (a filter weight the reference's value round trip turned into 0.0).  When such a CSR is stored in ascending
 * column order -- identity / channel-replicated permutation keys on both sides -- its product in stored order IS the factored operator's
 * order-preserving product (KN_FLAG_EXACT), entry for entry, and the host may hand over the 0.3 MB of taps instead of the CSR's hundreds of
 * MB; after this call kn_spmm with KN_FLAG_EXACT additionally restores, for batch columns whose activation at such an absent position is not
 * finite, exactly what the reference's row (without that entry) computes.  KN_ERR_UNSUPPORTED on other operator kinds. */
int kn_convtaps_drop_zero_entries(kn_handle_t h);

/* A DENSE operator (keyed nn.Linear: keynet/layer.py:67-70 stores it as a sparse matrix whose every entry is present)
 * for callers that accept the float-key tolerance (1e-5) instead of scipy's exact accumulation order -- i.e. the tiled
 * key-nets, whose conv layers already run on the matrix cores.  W is the full keyed matrix, HOST, row-major
 * [rows][cols] INCLUDING the homogeneous row (e_last) and the bias column; rows-1 outputs, cols-1 inputs.
 * kn_spmm computes it as a split-K f32-MFMA GEMM (K slices = pseudo-pixels of the conv-taps kernel) followed by an
 * ordered reduction over the slices (deterministic: no atomics).  Returns KN_ERR_UNSUPPORTED when cols-1 is not a
 * multiple of 256 (use kn_csr_create then).  KN_FLAG_EXACT is refused on such a handle. */
int kn_dense_create(int64_t rows, int64_t cols, const float* W, kn_handle_t* out);

/* Replaces the nn.Sequential walk of KeyedModel.forward (keynet/system.py:130-133, `self._keynet.forward(x)`) for a key-net whose
 * operators are all CSR handles (kn_csr_create / kn_tiled_create: the permutation key-nets) and whose activations are small enough
 * to stay on chip: kn_spmm on the returned handle applies ops[0] ... ops[n_ops-1] back to back in ONE launch, the activations of
 * four batch columns resident in LDS, ReLU after operator l when flags[l] & KN_FLAG_RELU (flags may be NULL).  Same arithmetic as
 * n_ops kn_spmm calls with KN_FLAG_EXACT (stored order, f32 multiply then add): bit-identical results.  The handle keeps its own
 * copy of the operators (ops may be destroyed afterwards).  X is [ops[0].cols, n_vecs], Y is [ops[n_ops-1].rows, n_vecs].
 * KN_ERR_UNSUPPORTED when an operator is not a CSR handle, n_ops > 12, or (max features in + out of a layer) * 16 B > 160 KiB. */
int kn_chain_create(int64_t n_ops, const kn_handle_t* ops, const uint32_t* flags, kn_handle_t* out);

int kn_destroy(kn_handle_t h);

/* SparseMatrix.nnz / TiledMatrix.nnz / Conv2dTiledMatrix.nnz (keynet/sparse.py:494,649,778): stored parameters. */
int kn_nnz(kn_handle_t h, int64_t* nnz);
/* nnz of the expanded operator the reference applies (tocsr()), = the algorithmic MAC count per input vector */
int kn_nnz_expanded(kn_handle_t h, int64_t* nnz);
int kn_shape(kn_handle_t h, int64_t* rows, int64_t* cols);
/* SparseMatrix.dtype (keynet/sparse.py:423) as a width: 64 for a kn_csr_create_f64 operator, 32 for every other handle */
int kn_dtype_bits(kn_handle_t h, int* bits);

/* tocsr()/tocoo() (keynet/sparse.py:502-507, 643-647, 816-835): expanded operator into HOST buffers sized by
 * kn_nnz_expanded (indptr[rows+1]).  CSR operators come back in stored order; tiled ones canonical (sorted). */
int kn_export_csr(kn_handle_t h, int32_t* indptr, int32_t* indices, float* data);
/* the same for a kn_csr_create_f64 operator (stored order, float64 values); kn_export_csr refuses such a handle and vice versa */
int kn_export_csr_f64(kn_handle_t h, int32_t* indptr, int32_t* indices, double* data);

/* ---- the hot path -------------------------------------------------------------------------------------------- */

/* Replaces SparseMatrix.torchdot / TiledMatrix.torchdot (keynet/sparse.py:488-492, 603-612):
 *   Y[rows, n_vecs] = W . X[cols, n_vecs]      (+ ReLU when KN_FLAG_RELU)
 * x_dev/y_dev: device f32, feature-major, leading dimensions ldx/ldy >= n_vecs (floats).  x and y must not alias. */
int kn_spmm(kn_handle_t h, const float* x_dev, int64_t ldx, int64_t n_vecs,
            float* y_dev, int64_t ldy, uint32_t flags, void* stream);

/* SparseMatrix.torchdot of a FLOAT64 operator (kn_csr_create_f64) with the result dtype the reference returns: Y[rows, n_vecs] float64 =
 * W . (double)X, accumulated in float64 in stored order (+ ReLU when KN_FLAG_RELU: F.relu on the float64 block, keynet/layer.py:93).
 * x_dev float32 as everywhere (keynet/sparse.py:489-491), y_dev device float64 with leading dimension ldy (doubles).  KN_ERR_UNSUPPORTED on
 * any other handle: float32 operators return float32. */
int kn_spmm_f64(kn_handle_t h, const float* x_dev, int64_t ldx, int64_t n_vecs,
                double* y_dev, int64_t ldy, uint32_t flags, void* stream);

/* The same operator applied to n_planes independent activation blocks in ONE launch per kernel: block p is X_p = x_dev + p * x_plane_stride (floats), its result
 * Y_p = y_dev + p * y_plane_stride.  Bit-identical to n_planes kn_spmm calls (same kernels, grid dimension y = plane).  No reference counterpart as a call -- the reference
 * applies the fused operator (keynet/sparse.py:603-612); this is what the split application of a FILLED-IN conv needs (keynet_amd/sparse.py: the spatial matrix of all taps
 * applied to every input channel's plane, 64 .. 512 planes per layer -- one launch instead of one per plane).  float32 CSR handles whose rows are pattern groups / loose rows
 * only; KN_ERR_UNSUPPORTED otherwise (the caller loops over kn_spmm).  Always the stored order (KN_FLAG_EXACT is implied for CSR operators); KN_FLAG_RELU is honoured. */
int kn_spmm_planes(kn_handle_t h, const float* x_dev, int64_t ldx, int64_t x_plane_stride, int64_t n_planes, int64_t n_vecs,
                   float* y_dev, int64_t ldy, int64_t y_plane_stride, uint32_t flags, void* stream);

/* kn_spmm that additionally raises *y_absmax_dev (device f32, caller-initialised, e.g. to 0) to max |Y[r, b]| over the block it wrote,
 * stream-ordered.  No reference counterpart: the reference applies ONE arithmetic on every call (keynet/sparse.py:488-492), so its 1e-5
 * float-key agreement holds per call; a host that runs a layer on the matrix cores because a calibration batch showed the re-ordered sum
 * inside that tolerance must notice when a later batch is larger.  max |Y| of layer l is max |X| of layer l+1, and the matrix-core kernels
 * fold it into their store epilogue (no extra pass over a multi-GB activation block); behind the other kernel families the library runs
 * one reduction pass over Y.  NaN entries are ignored, +-Inf counts.  y_absmax_dev == NULL is kn_spmm. */
int kn_spmm_screen(kn_handle_t h, const float* x_dev, int64_t ldx, int64_t n_vecs,
                   float* y_dev, int64_t ldy, uint32_t flags, float* y_absmax_dev, void* stream);

/* max |X[r, b]| of an activation block (the input of the first layer: nothing produced it on the device), raised into *absmax_dev like
 * kn_spmm_screen does. */
int kn_absmax(const float* x_dev, int64_t rows, int64_t ld, int64_t n_vecs, float* absmax_dev, void* stream);

/* Drops the side tables a conv-taps operator builds at first use (the slot records of the filled-in order-preserving kernel: 16 bytes per slot, 0.4 - 4 GB per layer of the
 * reference's doubly-stochastic VGG-16; the bf16 planes of KN_FLAG_BF16X3).  They are rebuilt by the next kn_spmm that needs them.  No reference counterpart (memory management
 * of this library): the host calls it for a layer whose calibration settled on a contract that does not use them.  Waits for the device; not for the hot path. */
int kn_release_side_tables(kn_handle_t h);

/* Pre-size the per-stream state kn_spmm would otherwise allocate at first use on `stream` for `n_vecs` batch columns (only a
 * kn_dense_create handle has any: its split-K partial sums), so that the first kn_spmm on that stream can be captured into a HIP graph.
 * A no-op for every other operator kind. */
int kn_reserve_workspace(kn_handle_t h, int64_t n_vecs, void* stream);

/* Introspection: which kernels kn_spmm would launch for this operator, batch width, leading dimensions (16-byte aligned activations
 * assumed) and flags -- the SAME dispatch code runs, every launch site describes itself instead of launching.  No reference
 * counterpart (scipy has one kernel); it exists so that measurements can state which loader / tile shape produced them
 * ("choices evidenced").  Writes a NUL-terminated, ';'-separated list into buf.  Launches nothing and allocates nothing
 * (with KN_FLAG_BF16X3 it reports eligibility from the operator's shape; the bf16 planes are built by the first real kn_spmm). */
int kn_spmm_plan(kn_handle_t h, int64_t n_vecs, int64_t ldx, int64_t ldy, uint32_t flags, char* buf, int64_t buf_len);

/* unkeyed nn.ReLU on a whole activation block (keynet/system.py:92) when it could not be fused */
int kn_relu(float* y_dev, int64_t rows, int64_t ld, int64_t n_vecs, void* stream);

/* keynet.torch.affine_to_linear (keynet/torch.py:65-68) fused with the x.t() of keynet/layer.py:92:
 * x_dev [n, d] row-major images  ->  out_dev [d+1, n] feature-major with the ones row appended. */
int kn_affine_to_linear(const float* x_dev, int64_t n, int64_t d, float* out_dev, int64_t ldo, void* stream);

/* keynet.torch.linear_to_affine (keynet/torch.py:71-77): y_dev [d+1, n] feature-major -> out_dev [n, d] row-major;
 * *maxdev_dev (device float, may be NULL) receives max_b |y[d, b] - 1| so the host can raise ValueError when > 1e-3. */
int kn_linear_to_affine(const float* y_dev, int64_t ldy, int64_t n, int64_t d, float* out_dev, float* maxdev_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KEYNET_HIP_H */
