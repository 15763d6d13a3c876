// Dev probe (gfx950): lane maps of v_mfma_f32_4x4x1_16B_f32 and of its A-block broadcast (CBSZ / ABID).
//   hypothesis, lane = 4 * blk + j:  D[lane][r] = A[4 * src(blk) + r] * B[lane]
//   CBSZ = 0: src(blk) = blk;   CBSZ = s: src(blk) = (blk & ~(2^s - 1)) + ABID  (one block of A serves 2^s blocks)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int CBSZ, int ABID>
__global__ void k(const float* A, const float* B, float* D) {
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[threadIdx.x], B[threadIdx.x], c, CBSZ, ABID, 0);
    for (int r = 0; r < 4; ++r) D[threadIdx.x * 4 + r] = c[r];
}
template <int CBSZ, int ABID> int check(const float* dA, const float* dB, float* dD, const float* A, const float* B) {
    k<CBSZ, ABID><<<1, 64>>>(dA, dB, dD);
    float D[256];
    CK(hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l >> 2, src = (blk & ~((1 << CBSZ) - 1)) + ABID;
            const float want = A[4 * src + r] * B[l];
            if (D[l * 4 + r] != want) { if (bad < 4) printf("   lane %d r %d: got %g want %g\n", l, r, D[l * 4 + r], want); ++bad; }
        }
    printf("%s 4x4x1 CBSZ=%d ABID=%d: %d mismatches\n", bad ? "FAIL" : "PASS", CBSZ, ABID, bad);
    return bad;
}
int main() {
    float A[64], B[64], *dA, *dB, *dD;
    for (int i = 0; i < 64; ++i) { A[i] = (float)(i + 1); B[i] = (float)(100 + 3 * i); }
    CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B, 256, hipMemcpyHostToDevice));
    check<0, 0>(dA, dB, dD, A, B);
    check<4, 0>(dA, dB, dD, A, B);
    check<4, 5>(dA, dB, dD, A, B);
    check<4, 15>(dA, dB, dD, A, B);
    check<2, 0>(dA, dB, dD, A, B);
    check<2, 3>(dA, dB, dD, A, B);
    check<1, 1>(dA, dB, dD, A, B);
    return 0;
}
