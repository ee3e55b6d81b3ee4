// Two workgroups on different CUs hand a cache line back and forth (gfx950): which cache maintenance does a reader need
// to see what the other one wrote, when both sit on the same XCD (one L2) and when they do not?
//   A reads the line (it is now in A's L1), tells B; B rewrites the line, waits for the store's acknowledge, tells A;
//   A [mode: nothing | buffer_inv sc0 | buffer_inv sc1] reads the line again. Stale = A still sees the old value.
// Also reports the XCC id of every workgroup residue (s_getreg HW_REG_XCC_ID) to check the round-robin placement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15;
}

// a plain load (no scope bits: may be served by the CU's L1), which the compiler can neither cache nor widen
__device__ __forceinline__ uint32_t plain_load(const uint32_t *p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// the same load with scope bits: sc0 (workgroup), sc1 (agent), sc0 sc1 (system)
__device__ __forceinline__ uint32_t scoped_load(const uint32_t *p, int bits, uint64_t &cycles) {
    uint32_t v;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (bits == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (bits == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (bits == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    cycles += __builtin_amdgcn_s_memtime() - t0;
    return v;
}

__global__ void where(uint32_t *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// flag[0]: round A has read, flag[1]: round B has written. line: 32 dwords. partner = the blockIdx of B.
__global__ void pingpong(uint32_t *line, uint32_t *flag, uint32_t *stale, uint32_t *xccs, int rounds, int mode, uint32_t partner, int bits, unsigned long long *cyc) {
    const bool is_a = blockIdx.x == 0, is_b = blockIdx.x == partner;
    if (!is_a && !is_b) return;
    if (threadIdx.x == 0) xccs[is_a ? 0 : 1] = xcc_id();
    if (threadIdx.x >= 32) return;
    uint32_t bad = 0;
    uint64_t cycles = 0;
    for (int r = 1; r <= rounds; r++) {
        if (is_a) {
            const uint32_t keep = plain_load(&line[threadIdx.x]);  // in A's L1 now
            if (keep == 0xDEADBEEFu) bad += 1u << 20;
            __builtin_amdgcn_s_waitcnt(0);
            if (threadIdx.x == 0) __hip_atomic_store(&flag[0], (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 0) { uint32_t spins = 0; while (__hip_atomic_load(&flag[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)r && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(2); }
            __builtin_amdgcn_wave_barrier();
            if (mode == 1) asm volatile("buffer_inv sc0" ::: "memory");
            if (mode == 2) asm volatile("buffer_inv sc1" ::: "memory");
            const uint32_t v = scoped_load(&line[threadIdx.x], bits, cycles);
            if (v != (uint32_t)r) bad++;
        } else {
            if (threadIdx.x == 0) { uint32_t spins = 0; while (__hip_atomic_load(&flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)r && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(2); }
            __builtin_amdgcn_wave_barrier();
            line[threadIdx.x] = (uint32_t)r;
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            if (threadIdx.x == 0) __hip_atomic_store(&flag[1], (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (is_a) atomicAdd(stale, bad);
    if (is_a && threadIdx.x == 0) *cyc = cycles;
}

int main() {
    uint32_t *d;
    CK(hipMalloc(&d, 1 << 20));
    std::vector<uint32_t> h(4096);
    hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, 0, d);
    CK(hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost));
    int off = 0;
    for (int i = 0; i < 4096; i++) off += h[i] != h[i & 7];
    printf("XCC id of workgroups 0..7:");
    for (int i = 0; i < 8; i++) printf(" %u", h[i]);
    printf("   workgroups (of 4096) not on the XCC of their residue mod 8: %d\n", off);
    const int rounds = 20000;
    for (uint32_t partner : {8u, 1u}) {
        for (int mode = 0; mode < 3; mode++) {
            CK(hipMemset(d, 0, 4096));
            hipLaunchKernelGGL(pingpong, dim3(16), dim3(64), 0, 0, d + 256, d, d + 64, d + 128, rounds, mode, partner, 0, (unsigned long long *)(d + 192));
            CK(hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost));
            printf("partner workgroup %u (XCC %u vs %u), %-14s: %u stale lane-reads of %d\n", partner, h[128], h[129],
                   mode == 0 ? "no invalidate" : mode == 1 ? "buffer_inv sc0" : "buffer_inv sc1", h[64], rounds * 32);
        }
    }
    // scope bits on the load instead of an invalidate (same XCD): which ones get past the stale L1 line, and what they cost
    for (int bits = 0; bits < 4; bits++) {
        CK(hipMemset(d, 0, 4096));
        hipLaunchKernelGGL(pingpong, dim3(16), dim3(64), 0, 0, d + 256, d, d + 64, d + 128, rounds, 0, 8u, bits, (unsigned long long *)(d + 192));
        CK(hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost));
        const unsigned long long cy = *(unsigned long long *)&h[192];
        printf("same XCD, no invalidate, load %-8s: %u stale lane-reads of %d, %.0f cycles per load\n",
               bits == 0 ? "plain" : bits == 1 ? "sc0" : bits == 2 ? "sc1" : "sc0 sc1", h[64], rounds * 32, (double)cy / rounds);
    }
    return 0;
}
