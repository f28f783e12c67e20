// How many single-wave workgroups with L bytes of LDS does the chip really run at once?  Every block stamps its start on the shared 100 MHz
// clock and then spins for ~20 us; blocks that start in the first 5 us were resident together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int LDS_WORDS, int NT>
__global__ __launch_bounds__(NT) void probe(long long *starts, int spin_ticks) {
    __shared__ unsigned arena[LDS_WORDS];
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) starts[blockIdx.x] = t0;
    arena[threadIdx.x] = threadIdx.x;
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (arena[(threadIdx.x + 1) % NT] == 12345) starts[0] = 0;
}
template <int LDS_WORDS, int NT>
void run(int blocks) {
    long long *d; (void)hipMalloc(&d, sizeof(long long) * blocks);
    std::vector<long long> h(blocks);
    int occ = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe<LDS_WORDS, NT>, NT, 0);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((probe<LDS_WORDS, NT>), dim3(blocks), dim3(NT), 0, 0, d, 2000);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), d, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    const long long t0 = *std::min_element(h.begin(), h.end());
    int early = 0; long long last = 0;
    for (long long v : h) { early += (v - t0) < 500; last = std::max(last, v - t0); }
    printf("LDS %6d B, %3d threads: API says %2d blocks/CU; %5d of %5d blocks started within 5 us (%.1f per CU); last start %.1f us\n",
           LDS_WORDS * 4, NT, occ, early, blocks, early / 256.0, last / 100.0);
    (void)hipFree(d);
}
int main() {
    run<64, 64>(16384); run<1024, 64>(16384); run<2100, 64>(16384); run<4096, 64>(16384); run<2100, 256>(4096); run<64, 256>(4096); run<8192, 256>(4096);
    return 0;
}
