// GPU box: what a work queue costs on this chip, and what the dispatcher costs.  (1) an empty kernel of B blocks x 64 threads (the stage
// launch has 24 416): dispatch time; (2) G resident waves each taking K tickets with a returning device-scope atomicAdd from A counters
// (A = 1, 8, 32, 256; counters 256 bytes apart): time per ticket per wave = what a dynamic queue adds per item, and the aggregate rate.
// build: hipcc --offload-arch=gfx950 -O3 tools/dbg/queue_probe.hip -o tools/dbg/queue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 1234567) *p = 1; }
__global__ __launch_bounds__(64) void ticket_kernel(unsigned *ctr, int A, int K, unsigned *sink) {
    const int w = blockIdx.x;
    unsigned acc = 0;
    for (int k = 0; k < K; k++) {
        unsigned t = 0;
        if (threadIdx.x == 0) t = __hip_atomic_fetch_add(ctr + ((w + k) % A) * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __shfl(t, 0);
        acc += t;
    }
    if (acc == 0xffffffffu) *sink = acc;
}
static float run(void (*f)(void *), void *a) { return 0; }
int main() {
    int *d; hipMalloc(&d, 4);
    unsigned *ctr, *sink; hipMalloc(&ctr, 256 * 256); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 3072, 8192, 24416, 100000}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0); empty_kernel<<<blocks, 64>>>(nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("empty kernel, %6d blocks x 64 threads: %.1f us\n", blocks, best * 1e3);
    }
    for (int G : {256, 3072}) for (int A : {1, 8, 32, 256}) {
        const int K = 16;
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipMemset(ctr, 0, 256 * 256);
            hipEventRecord(e0); ticket_kernel<<<G, 64>>>(ctr, A, K, sink); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("tickets: %4d waves x %d tickets from %3d counters: %.1f us = %.2f us per ticket per wave, %.1f ns per ticket overall\n",
               G, K, A, best * 1e3, best * 1e3 / K, best * 1e6 / (G * K));
    }
    return 0;
}
