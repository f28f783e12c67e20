#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ unsigned pk(float lo, float hi) { const bf16x2_t v = {(__bf16)lo, (__bf16)hi}; return __builtin_bit_cast(unsigned, v); }
// A[32][4] (row i, channel c), B[4][32]; lane (i = l & 31, h = l >> 5) holds channels 2h, 2h+1
__global__ void k(const float *A, const float *B, float *D) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    const float a0 = A[i * 4 + 2 * h], a1 = A[i * 4 + 2 * h + 1];
    const float b0 = B[(2 * h) * 32 + i], b1 = B[(2 * h + 1) * 32 + i];
    u32x2 ah, am, bq;
    ah[0] = pk(a0, a0); ah[1] = pk(a1, a1);
    am[0] = pk(a0 - __builtin_bit_cast(float, ah[0] & 0xffff0000u), a0 - __builtin_bit_cast(float, ah[0] & 0xffff0000u));
    am[1] = pk(a1 - __builtin_bit_cast(float, ah[1] & 0xffff0000u), a1 - __builtin_bit_cast(float, ah[1] & 0xffff0000u));
    const float b0h = __builtin_bit_cast(float, pk(b0, b0) & 0xffff0000u), b1h = __builtin_bit_cast(float, pk(b1, b1) & 0xffff0000u);
    bq[0] = pk(b0, b0 - b0h); bq[1] = pk(b1, b1 - b1h);
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, ah), __builtin_bit_cast(s16x4, bq), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, am), __builtin_bit_cast(s16x4, bq), acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) { const int row = (r & 3) + 8 * (r >> 2) + 4 * h; D[row * 32 + i] = acc[r]; }
}
int main() {
    float hA[128], hB[128], hD[1024], *dA, *dB, *dD;
    for (int t = 0; t < 128; t++) { hA[t] = sinf(t * 1.3f) * 2.f; hB[t] = cosf(t * 0.7f) * 0.3f; }
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    double me = 0, mr = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
        double ref = 0; for (int c = 0; c < 4; c++) ref += (double)hA[i * 4 + c] * hB[c * 32 + j];
        me = fmax(me, fabs(ref - hD[i * 32 + j])); mr = fmax(mr, fabs(ref));
    }
    printf("max err %.3g (max |ref| %.3g)  D[0][0..3] %g %g %g %g\n", me, mr, hD[0], hD[1], hD[2], hD[3]);
    return 0;
}
