// Microbenchmark: cost of divergent gathers of 4 / 8 / 16 bytes per lane from an L2-resident buffer (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int BYTES, int MIS, int RUN, int MODE = 0>   // MODE (4-byte loads): 0 plain, 1 non-temporal, 2 agent-scope relaxed atomic (sc1), 3 system-scope (sc0 sc1)
__global__ __launch_bounds__(256) void gather(const uint8_t *__restrict__ buf, uint32_t mask, uint32_t *__restrict__ out, int iters) {
    // lane addresses: runs of RUN consecutive lanes read consecutive elements; the run's base is pseudo-random
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t h = (tid / RUN) * 0x9E3779B1u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        uint32_t off = ((h >> 4) & mask) * 16u + (tid % RUN) * BYTES + MIS;   // element offset inside the buffer
        const uint8_t *p = buf + off;
        if (BYTES == 4 && MODE == 1) acc += __builtin_nontemporal_load((const uint32_t *)p);
        else if (BYTES == 4 && MODE == 2) acc += __hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (BYTES == 4 && MODE == 3) acc += __hip_atomic_load((const uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (BYTES == 4) { uint32_t v; __builtin_memcpy(&v, p, 4); acc += v; }
        else if (BYTES == 8) { uint2 v; __builtin_memcpy(&v, p, 8); acc += v.x ^ v.y; }
        else if (BYTES == 12) { uint3 v; __builtin_memcpy(&v, __builtin_assume_aligned(p - MIS, 4), 12); acc += v.x ^ v.y ^ v.z; }
        else { uint4 v; if (MIS) __builtin_memcpy(&v, p, 16); else v = *(const uint4 *)p; acc += v.x ^ v.y ^ v.z ^ v.w; }
        h ^= acc & 1;   // dependent chain (like chain hops): next address needs this load
    }
    out[tid] = acc;
}

template <int BYTES, int MIS, int RUN, int MODE = 0>
void run(const char *name, const uint8_t *buf, uint32_t n_elems16, uint32_t *out) {
    const int blocks = 256 * 32, iters = 64;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gather<BYTES, MIS, RUN, MODE><<<blocks, 256>>>(buf, n_elems16 - 1, out, iters);
    CK(hipEventRecord(a));
    gather<BYTES, MIS, RUN, MODE><<<blocks, 256>>>(buf, n_elems16 - 1, out, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double lanes = (double)blocks * 256 * iters;
    printf("%-34s %8.3f ms  %7.2f G lane-gathers/s  %6.3f clk/lane/CU\n", name, ms, lanes / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / lanes);
}

int main(int argc, char **argv) {
    const size_t bytes = (argc > 1 ? atoi(argv[1]) : 2) << 20;   // buffer size in MiB (default 2: L2 resident)
    uint8_t *buf; uint32_t *out;
    CK(hipMalloc(&buf, bytes + 4096)); CK(hipMemset(buf, 1, bytes + 4096));
    CK(hipMalloc(&out, 256 * 32 * 256 * 4));
    const uint32_t n16 = (uint32_t)(bytes / 16);
    printf("buffer %zu MiB\n", bytes >> 20);
    run<4, 0, 1>("4 B  random", buf, n16, out);
    run<4, 0, 1, 1>("4 B  random, non-temporal", buf, n16, out);
    run<4, 0, 1, 2>("4 B  random, agent-scope load (sc1)", buf, n16, out);
    run<4, 0, 1, 3>("4 B  random, system-scope load (sc0 sc1)", buf, n16, out);
    run<8, 0, 1>("8 B  random aligned", buf, n16, out);
    run<8, 1, 1>("8 B  random misaligned", buf, n16, out);
    run<12, 0, 1>("12 B random dword-aligned", buf, n16, out);
    run<16, 0, 1>("16 B random aligned", buf, n16, out);
    run<16, 4, 1>("16 B random dword-aligned", buf, n16, out);
    run<16, 1, 1>("16 B random byte-misaligned", buf, n16, out);
    run<4, 0, 4>("4 B  runs of 4 lanes", buf, n16, out);
    run<4, 0, 16>("4 B  runs of 16 lanes", buf, n16, out);
    run<16, 0, 4>("16 B runs of 4 lanes", buf, n16, out);
    run<16, 0, 16>("16 B runs of 16 lanes", buf, n16, out);
    run<16, 0, 8>("16 B runs of 8 lanes, aligned", buf, n16, out);      // (round 4: what enc_cand's phase 3 issues -- 8-lane groups on 128 contiguous bytes)
    run<16, 4, 8>("16 B runs of 8 lanes, dword-aligned", buf, n16, out);
    run<16, 1, 8>("16 B runs of 8 lanes, byte-misaligned", buf, n16, out);
    run<8, 0, 8>("8 B  runs of 8 lanes, aligned", buf, n16, out);
    run<8, 1, 8>("8 B  runs of 8 lanes, byte-misaligned", buf, n16, out);
    run<4, 0, 8>("4 B  runs of 8 lanes", buf, n16, out);
    run<4, 0, 64>("4 B  coalesced wave", buf, n16, out);
    run<16, 0, 64>("16 B coalesced wave", buf, n16, out);
    return 0;
}
