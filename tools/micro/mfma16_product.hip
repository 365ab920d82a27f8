// The same question as mfma_product.hip for  v_mfma_f32_16x16x1_4b_f32  (16 rows x 4 blocks of 16 columns: the 16-row form of the matrix-pipe grouped kernel),
// and its D layout: is it with a ZERO accumulator the IEEE-rounded product?   D[i][j] = 0 + a[i] * b[j]
// If yes, the reference's  y = y + (a * x)  (separate f32 multiply and f32 add: scipy csr_matvecs) can take its MULTIPLIES from the matrix
// pipe and keep only the ADD on the vector ALU -- same bits, half the vector instructions.  This program compares, bit for bit, the MFMA's
// outer products with v_mul_f32 on random bit patterns, random normal floats and a table of special values (signed zeros, denormals,
// products that underflow / overflow / round to the smallest normal, infinities, NaNs); NaNs compare equal as a class; a product of -0
// comes back as +0 from the MFMA (-0 + +0), which is reported separately (adding either zero to a running sum that is never -0 is the same).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/micro/mfma16_product.hip -o /tmp/mfma16_product && /tmp/mfma16_product
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

// one wavefront: a[16] (rows, the same for the four blocks), b[64] (four blocks of 16 columns: lane l supplies ITS OWN column l) -> prod[16][64]
__global__ void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ Pm, float* __restrict__ Pv, int n_sets) {
    const int lane = threadIdx.x & 63;
    for (int s = blockIdx.x; s < n_sets; s += gridDim.x) {
        const float a = A[s * 32 + (lane & 15)];
        const float b = B[s * 64 + lane];
        f32x16 c;
        for (int r = 0; r < 16; r++) c[r] = 0.0f;
        // 16x16x1, 4 blocks: block = lane / 16 for both operands
        f32x16 d = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0);
        // D layout per block blk (regs 4*blk .. 4*blk+3): reg r, lane l -> row i = 4*(l/16) + r%4, column j = 16*blk + l%16
        for (int blk = 0; blk < 4; blk++)
            for (int r = 0; r < 4; r++) {
                const int i = 4 * (lane / 16) + r;
                const int j = blk * 16 + (lane & 15);
                Pm[((size_t)s * 32 + i) * 64 + j] = d[4 * blk + r];
            }
        for (int i = 0; i < 16; i++) {
            const float ai = A[s * 32 + i];
            float p;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(ai), "v"(b));
            Pv[((size_t)s * 32 + i) * 64 + lane] = p;
        }
    }
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float fl(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int n_sets = 1 << 14;      // x 2048 products = 33.5 M products
    std::vector<float> A((size_t)n_sets * 32), B((size_t)n_sets * 64);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    const float specials[] = {0.0f, -0.0f, 1.0f, -1.0f, fl(0x00000001), fl(0x80000001), fl(0x007fffff), fl(0x00800000), fl(0x00800001), fl(0x7f7fffff), fl(0xff7fffff),
                              fl(0x7f800000), fl(0xff800000), fl(0x7fc00000), fl(0x7f800001), 1e-20f, -1e-20f, 1e-19f, 1e-30f, 1e30f, 3e38f, 1.5f, 0.5f, fl(0x3f800001), fl(0x3f7fffff),
                              fl(0x1e000000), fl(0x20000000), fl(0x1f800000), fl(0x1fffffff), 2e-38f, 1.1754944e-38f, 5e-39f};
    const int ns = sizeof(specials) / sizeof(float);
    for (int s = 0; s < n_sets; s++) {
        const int mode = s % 4;      // 0 random normals, 1 random bit patterns, 2 specials x specials, 3 tiny x tiny (underflow region)
        for (int i = 0; i < 32; i++) {
            float v;
            if (mode == 0) v = nd(rng);
            else if (mode == 1) v = fl((uint32_t)rng());
            else if (mode == 2) v = specials[(i + s / 4) % ns];
            else v = fl((uint32_t)(0x1d000000u + (rng() % 0x06000000u)) | ((rng() & 1) << 31));
            A[(size_t)s * 32 + i] = v;
        }
        for (int j = 0; j < 64; j++) {
            float v;
            if (mode == 0) v = nd(rng) * 100.f;
            else if (mode == 1) v = fl((uint32_t)rng());
            else if (mode == 2) v = specials[(j * 7 + s / 4) % ns];
            else v = fl((uint32_t)(0x1d000000u + (rng() % 0x06000000u)) | ((rng() & 1) << 31));
            B[(size_t)s * 64 + j] = v;
        }
    }
    float *dA, *dB, *dPm, *dPv;
    const size_t np = (size_t)n_sets * 2048;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dPm, np * 4); hipMalloc(&dPv, np * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, dA, dB, dPm, dPv, n_sets);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    std::vector<float> Pm(np), Pv(np);
    hipMemcpy(Pm.data(), dPm, np * 4, hipMemcpyDeviceToHost);
    hipMemcpy(Pv.data(), dPv, np * 4, hipMemcpyDeviceToHost);
    size_t same = 0, nan_both = 0, zero_sign = 0, diff = 0, denorm_out = 0, denorm_in = 0, host_mismatch = 0;
    int shown = 0;
    for (size_t s = 0; s < (size_t)n_sets; s++)
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 64; j++) {
                const size_t idx = (s * 32 + i) * 64 + j;
                const uint32_t m = bits(Pm[idx]), v = bits(Pv[idx]);
                const float a = A[s * 32 + i], b = B[s * 64 + j];
                volatile float hp = a * b;                        // host IEEE product (x86 SSE: no flush)
                const uint32_t h = bits(hp);
                const bool vn = (v & 0x7fffffffu) > 0x7f800000u, hn = (h & 0x7fffffffu) > 0x7f800000u;
                if (!(h == v || (vn && hn))) host_mismatch++;     // does v_mul_f32 itself agree with the host product (denormal mode)?
                if ((v & 0x7f800000u) == 0 && (v & 0x007fffffu)) denorm_out++;
                if (((bits(a) & 0x7f800000u) == 0 && (bits(a) & 0x007fffffu)) || ((bits(b) & 0x7f800000u) == 0 && (bits(b) & 0x007fffffu))) denorm_in++;
                if (m == v) { same++; continue; }
                const bool mn = (m & 0x7fffffffu) > 0x7f800000u;
                if (mn && vn) { nan_both++; continue; }
                if ((m | v) == 0x80000000u && (m & v) == 0) { zero_sign++; continue; }       // +0 vs -0
                diff++;
                if (shown < 24) {
                    printf("DIFF a=%08x (%g) b=%08x (%g): mfma=%08x (%g) v_mul=%08x (%g) host=%08x\n", bits(a), a, bits(b), b, m, Pm[idx], v, Pv[idx], h);
                    shown++;
                }
            }
    printf("products (16 rows of each set) %zu: identical %zu, both NaN %zu, zero sign only (mfma +0 / v_mul -0) %zu, DIFFERENT %zu\n", np / 2, same, nan_both, zero_sign, diff);
    printf("v_mul_f32 denormal results %zu, products with a denormal operand %zu, v_mul vs host IEEE mismatches %zu\n", denorm_out, denorm_in, host_mismatch);
    return diff ? 1 : 0;
}
