// What does an LDS exchange cost on gfx950, and what would a SECOND predecessor per position cost enc_chain_kernel?
//   mode 0: one ds_wrxchg_rtn_b32 per position on a 16 Ki-entry table (today's chain step)
//   mode 1: two dependent exchanges per position, the second on a second table with the value the first returned
//   mode 2: one exchange + plain LDS traffic instead of the second exchange: a ds_write_b32 of the returned value into a batch
//           array, a ds_read_b32 gather from it, a ds_read_b32 of the table entry, a conditional ds_write_b32 into a second table
//   mode 3: plain ds_read_b32 gather + ds_write_b32 scatter per position (no atomic at all; the floor of the LDS for this pattern)
// Every wave runs batches of 32 steps issued back to back (as enc_chain_kernel does); W waves per workgroup all exchange at the
// same time (worst case for the unit), the dynamic LDS size sets how many workgroups a CU holds.
// Build: hipcc --offload-arch=gfx950 -O3 -o xchg_rate scripts/micro/xchg_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

extern __shared__ uint32_t lds[];

template <int MODE>
__global__ void rate(uint32_t *out, int n_batches, uint32_t spread) {
    uint32_t *tab = lds, *tab2 = lds + 16384, *barr = lds + 32768;   // (tab2 / barr only touched by the modes that have the room)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t k = threadIdx.x; k < 16384; k += blockDim.x) tab[k] = 0;
    if (MODE == 1 || MODE == 2) for (uint32_t k = threadIdx.x; k < 16384; k += blockDim.x) tab2[k] = 0;
    __syncthreads();
    uint32_t r = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    uint32_t *mb = barr + (wave & 1) * 2048;   // (waves beyond two share: timing only)
    for (int b = 0; b < n_batches; b++) {
        uint32_t key[32], old[32];
#pragma unroll
        for (int j = 0; j < 32; j++) {
            r = r * 1664525u + 1013904223u;
            key[j] = (r >> 10) & spread;
        }
        if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 32; j++) old[j] = tab[key[j]];
#pragma unroll
            for (int j = 0; j < 32; j++) tab[key[j]] = (uint32_t)(b * 2048 + j * 64 + lane + 1);
        } else {
#pragma unroll
            for (int j = 0; j < 32; j++)
                old[j] = __hip_atomic_exchange(&tab[key[j]], (uint32_t)(b * 2048 + j * 64 + lane + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 32; j++)
                old[j] ^= __hip_atomic_exchange(&tab2[key[j]], old[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) << 1;
        }
        if (MODE == 2) {
            uint32_t now[32];
#pragma unroll
            for (int j = 0; j < 32; j++) mb[j * 64 + lane] = old[j];
#pragma unroll
            for (int j = 0; j < 32; j++) now[j] = tab[key[j]];
            uint32_t o2[32];
#pragma unroll
            for (int j = 0; j < 32; j++) o2[j] = (old[j] & 1) ? mb[old[j] & 2047] : tab2[key[j]];
#pragma unroll
            for (int j = 0; j < 32; j++)
                if (now[j] == (uint32_t)(b * 2048 + j * 64 + lane + 1)) tab2[key[j]] = old[j];
#pragma unroll
            for (int j = 0; j < 32; j++) old[j] ^= o2[j] << 1;
        }
#pragma unroll
        for (int j = 0; j < 32; j++) acc += old[j];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static void run(const char *name, int waves, size_t lds_bytes, uint32_t spread, uint32_t *d_out) {
    const int n_batches = 256, blocks = 256 * 8;
    CK(hipFuncSetAttribute((const void *)rate<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    rate<MODE><<<blocks, waves * 64, lds_bytes>>>(d_out, 8, spread);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    rate<MODE><<<blocks, waves * 64, lds_bytes>>>(d_out, n_batches, spread);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double steps = (double)blocks * waves * n_batches * 32;   // wave-steps (64 positions each)
    const double per_cu_clk = ms * 1e-3 * 2.4e9 / (steps / 256.0);
    printf("%-44s waves/WG %d  LDS %3zu KB  spread %5u: %7.3f ms  %6.1f clk per 64 positions and CU  -> %5.2f ms for 753 M positions\n", name, waves,
           lds_bytes >> 10, spread + 1, ms, per_cu_clk, 753e6 / 64 / 256 * per_cu_clk / 2.4e9 * 1e3);
}

int main() {
    uint32_t *d_out;
    CK(hipMalloc(&d_out, 64));
    for (uint32_t spread : {16383u, 1023u}) {
        for (int w : {1, 2, 4}) run<0>("mode 0: one exchange", w, 64 << 10, spread, d_out);          // two workgroups per CU (today: 2 waves, one exchanging at a time)
        for (int w : {1, 2, 4, 8}) run<0>("mode 0: one exchange", w, 144 << 10, spread, d_out);      // one workgroup per CU
        for (int w : {1, 2, 4, 8}) run<1>("mode 1: two dependent exchanges", w, 144 << 10, spread, d_out);
        for (int w : {1, 2, 4, 8}) run<2>("mode 2: exchange + read/write for the second", w, 144 << 10, spread, d_out);
        for (int w : {1, 2, 4}) run<3>("mode 3: plain gather + scatter", w, 64 << 10, spread, d_out);
    }
    return 0;
}
