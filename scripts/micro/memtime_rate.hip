// What does one s_memtime tick last on gfx950? (the LZ / block statistics of the diagnostic build count in these ticks)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0, n = 0;
    while (t - t0 < ticks) { t = __builtin_amdgcn_s_memtime(); n++; }
    out[0] = t - t0; out[1] = n;
    out[2] = wall_clock64();
}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 24);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (unsigned long long ticks : {10000000ull, 100000000ull}) {
        hipEventRecord(a); spin<<<1, 64>>>(ticks, d); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%llu ticks in %.3f ms -> %.1f MHz (%llu polls)\n", h[0], ms, h[0] / ms / 1e3, h[1]);
    }
    return 0;
}
