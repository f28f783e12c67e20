// GPU box: is the device libm's sincos(x) bit-identical to its sin(x) and cos(x)?  (dbpost.hip's Clipper offset needs both of
// two_pi / steps; one call instead of two is only admissible if every bit agrees.)  build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// tools/dbg/sincos_probe.hip -o /tmp/sincos_probe;  prints the number of arguments (of 2^28 over (0, 7]) on which a bit differs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void probe(unsigned long long *bad, double lo, double hi, long n) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    // a low-discrepancy sweep plus a multiplicative scramble of the mantissa's low bits
    double x = lo + (hi - lo) * ((double)i + 0.5) / (double)n;
    x = x * (1.0 + 1e-9 * (double)((i * 2654435761u) & 1023));
    double s2, c2;
    sincos(x, &s2, &c2);
    const double s1 = sin(x), c1 = cos(x);
    if (__double_as_longlong(s1) != __double_as_longlong(s2) || __double_as_longlong(c1) != __double_as_longlong(c2)) atomicAdd(bad, 1ull);
}
int main() {
    unsigned long long *d, h = 0;
    hipMalloc(&d, 8); hipMemset(d, 0, 8);
    const long n = 1L << 28;
    probe<<<(unsigned)(n / 256), 256>>>(d, 1e-6, 7.0, n);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("sincos vs sin/cos over %ld arguments in (1e-6, 7]: %llu differ\n", n, h);
    return 0;
}
