// Probe (dev tool, gfx950): lane maps of ds_read_b64_tr_b16 and v_mfma_f32_16x16x32_f16, and the accuracy of the
// 2-piece f16 split (3 products, f32 accumulate) against an f64 dot product.  Prints PASS/FAIL lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __fp16 half4v __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// each lane reads through its OWN element address addr[lane] (8-byte aligned: multiple of 4 halfs)
__global__ void tr_probe(const int* addr, float* out) {
    __shared__ __attribute__((aligned(16))) _Float16 s[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) s[i] = (_Float16)(float)i;
    __syncthreads();
    half4v a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) half4v*)(s + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)a[j];
}

// D = A * B with A[i][k], B[k][j] given in global memory (row-major 16x32, 32x16); lane map under test:
// A lane l elem e = A[l & 15][8 * (l >> 4) + e], B lane l elem e = B[8 * (l >> 4) + e][l & 15], D lane l reg r = D[4 * (l >> 4) + r][l & 15]
__global__ void mfma_probe(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    half8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + e];
        b[e] = (_Float16)B[(8 * (l >> 4) + e) * 16 + (l & 15)];
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

// split accuracy: one 16x16 tile, K = 32 real channels; three MFMAs (hi*hi, lo*hi, hi*lo) on operands scaled to 2^14
__global__ void split_probe(const float* A, const float* B, float sa, float sb, float* D) {
    const int l = threadIdx.x;
    half8 ah, al, bh, bl;
    for (int e = 0; e < 8; ++e) {
        const float av = A[(l & 15) * 32 + 8 * (l >> 4) + e] * sa, bv = B[(8 * (l >> 4) + e) * 16 + (l & 15)] * sb;
        ah[e] = (_Float16)av; al[e] = (_Float16)(av - (float)ah[e]);
        bh[e] = (_Float16)bv; bl[e] = (_Float16)(bv - (float)bh[e]);
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r] / (sa * sb);
}

int main() {
    int* daddr; float* dout;
    CK(hipMalloc(&daddr, 64 * sizeof(int))); CK(hipMalloc(&dout, 4096 * sizeof(float)));
    // (1) transpose read, natural image: lane m of a 16-lane group supplies chunk m of its group's 64-element block
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int> addr(64);
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, m = l & 15;
            // pass 0: [4][16] image per group, row stride 16;  pass 1: rows 40 halfs apart, groups 200 apart, chunks of a row reversed
            addr[l] = pass == 0 ? g * 64 + 4 * m : g * 200 + (m >> 2) * 40 + 4 * (3 - (m & 3));
        }
        CK(hipMemcpy(daddr, addr.data(), 64 * sizeof(int), hipMemcpyHostToDevice));
        tr_probe<<<1, 64>>>(daddr, dout);
        std::vector<float> o(256);
        CK(hipMemcpy(o.data(), dout, 256 * sizeof(float), hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                // hypothesis: lane i of a group gets, as element j, half (i & 3) of the chunk supplied by lane 4*j + (i >> 2)
                const int i = l & 15, src = (l & ~15) + 4 * j + (i >> 2);
                const float want = (float)(addr[src] + (i & 3));
                if (o[l * 4 + j] != want) { if (bad < 8) printf("  tr pass %d lane %d elem %d: got %g want %g\n", pass, l, j, o[l * 4 + j], want); ++bad; }
            }
        printf("%s tr_b16 lane map (pass %d): %d mismatches\n", bad ? "FAIL" : "PASS", pass, bad);
        if (bad) { printf("  raw lanes 0..19:\n"); for (int l = 0; l < 20; ++l) printf("   lane %2d: %g %g %g %g\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]); }
    }
    // (2) MFMA lane map with asymmetric integer operands (exact in f16 / f32)
    std::vector<float> A(512), B(512), D(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)((k * 5 + j * 13) % 7 - 3);
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
    mfma_probe<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double r = 0; for (int k = 0; k < 32; ++k) r += (double)A[i * 32 + k] * B[k * 16 + j];
        if (D[i * 16 + j] != (float)r) ++bad;
    }
    printf("%s mfma_f32_16x16x32_f16 lane map: %d mismatches\n", bad ? "FAIL" : "PASS", bad);
    // (3) split accuracy on random data of mixed magnitude
    srand(1);
    double worst = 0, worst32 = 0;
    for (int rep = 0; rep < 50; ++rep) {
        float ma = 0, mb = 0;
        for (int i = 0; i < 512; ++i) {
            A[i] = (float)((rand() / (double)RAND_MAX - 0.5) * (rep % 5 == 0 ? 1e-3 : 0.4)) * ((rand() & 7) == 0 ? 0.01f : 1.0f);
            B[i] = (float)((rand() / (double)RAND_MAX - 0.3) * (rep % 7 == 0 ? 300.0 : 3.0));
            ma = fmaxf(ma, fabsf(A[i])); mb = fmaxf(mb, fabsf(B[i]));
        }
        int ea, eb; frexpf(ma, &ea); frexpf(mb, &eb);          // m = f * 2^e, f in [0.5, 1): m * 2^(15 - e) in [2^14, 2^15)
        const float sa = ldexpf(1.0f, 15 - ea), sb = ldexpf(1.0f, 15 - eb);
        CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
        split_probe<<<1, 64>>>(dA, dB, sa, sb, dD);
        CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double r = 0, mag = 0; float r32 = 0;
            for (int k = 0; k < 32; ++k) { r += (double)A[i * 32 + k] * B[k * 16 + j]; mag += fabs((double)A[i * 32 + k] * B[k * 16 + j]); r32 = fmaf(A[i * 32 + k], B[k * 16 + j], r32); }
            worst = fmax(worst, fabs(D[i * 16 + j] - r) / mag);
            worst32 = fmax(worst32, fabs((double)r32 - r) / mag);
        }
    }
    printf("%s f16 split (3 products): worst |err| / sum|a||b| = %.3e   (f32 fmaf chain: %.3e)\n", worst < 4e-7 ? "PASS" : "FAIL", worst, worst32);
    return 0;
}
