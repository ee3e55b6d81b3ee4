// In which order does one ds_wrxchg_rtn_b32 wave-instruction serialise lanes that hit the same LDS address (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void probe(const uint32_t *keys, uint32_t *olds, int n_steps, uint32_t mask_mode) {
    __shared__ uint32_t tab[16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < 16384; k += blockDim.x) tab[k] = 0;
    __syncthreads();
    if (wave != 0) return;
    for (int s = 0; s < n_steps; s++) {
        const uint32_t key = keys[(blockIdx.x * n_steps + s) * 64 + lane];
        const uint32_t mine = (uint32_t)(s * 64 + lane + 1);
        uint32_t old = 0xFFFFFFFFu;
        bool on = true;
        if (mask_mode == 1) on = (lane % 3) != 1;
        if (mask_mode == 2) on = (key & 1) == 0;
        if (on) old = __hip_atomic_exchange(&tab[key], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        olds[(blockIdx.x * n_steps + s) * 64 + lane] = old;
    }
}

int main() {
    const int blocks = 512, steps = 256;
    std::vector<uint32_t> keys((size_t)blocks * steps * 64), olds(keys.size());
    uint32_t *dk, *dol;
    CK(hipMalloc(&dk, keys.size() * 4)); CK(hipMalloc(&dol, keys.size() * 4));
    long total_bad = 0;
    for (int pattern = 0; pattern < 5; pattern++)
        for (uint32_t mm = 0; mm < 3; mm++) {
            uint32_t r = 12345 + pattern;
            for (size_t i = 0; i < keys.size(); i++) {
                r = r * 1664525u + 1013904223u;
                const int lane = i & 63;
                switch (pattern) {
                    case 0: keys[i] = 7; break;                                  // all lanes one address
                    case 1: keys[i] = (r >> 8) & 15; break;                      // heavy collisions
                    case 2: keys[i] = (r >> 8) & 255; break;                     // moderate
                    case 3: keys[i] = (r >> 8) & 16383; break;                   // rare
                    case 4: keys[i] = ((lane / 2) * 32 + ((r >> 20) & 1) * 8192) & 16383; break;   // pairs of lanes on one address, all pairs in one bank
                }
            }
            CK(hipMemcpy(dk, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
            probe<<<blocks, 256>>>(dk, dol, steps, mm);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(olds.data(), dol, olds.size() * 4, hipMemcpyDeviceToHost));
            long bad = 0, coll = 0;
            for (int b = 0; b < blocks; b++) {
                std::vector<uint32_t> tab(16384, 0);
                for (int s = 0; s < steps; s++)
                    for (int l = 0; l < 64; l++) {
                        const size_t i = ((size_t)b * steps + s) * 64 + l;
                        bool on = true;
                        if (mm == 1) on = (l % 3) != 1;
                        if (mm == 2) on = (keys[i] & 1) == 0;
                        if (!on) continue;
                        const uint32_t want = tab[keys[i]];
                        if (want > (uint32_t)(s * 64)) coll++;
                        if (olds[i] != want) bad++;
                        tab[keys[i]] = s * 64 + l + 1;
                    }
            }
            printf("pattern %d mask %u: %ld same-step predecessors, %ld lanes differ from ascending-lane order\n", pattern, mm, coll, bad);
            total_bad += bad;
        }
    printf(total_bad ? "ORDER NOT ASCENDING\n" : "ascending lane order everywhere\n");
    return 0;
}
