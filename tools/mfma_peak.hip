// Sustained fp32 MFMA ceiling on this device: back-to-back v_mfma_f32_32x32x2_f32 on 4 accumulators per wave.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak ; run: /tmp/mfma_peak [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 1;
    int blocks = 256 * wps, iters = 20000;
    float *out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = 4.0 * iters * 4096.0 * blocks * 4;
        printf("waves/SIMD %d: %.3f ms  %.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
    }
    return 0;
}
