// How long does ONE wave wait for an LDS read it depends on? (the entropy decoder's step hangs on a table look-up: decode.hip)
// A chain of dependent reads p = lds[p] by one wave of a workgroup (the other lanes idle), with 0 / 4 / 8 dependent vector
// instructions between the reads; ds_read_b32 and ds_read_b64; also with a second, idle wave in the workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_latency scripts/micro/lds_latency.hip && /tmp/lds_latency
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WIDE, int PAD>
__global__ __launch_bounds__(128) void chase(uint32_t iters, unsigned long long *out, uint32_t *sink) {
    __shared__ uint2 tab[2048];
    for (uint32_t k = threadIdx.x; k < 2048; k += blockDim.x) tab[k] = make_uint2((k * 37 + 11) & 2047, k ^ 5);
    __syncthreads();
    if (threadIdx.x >= 64) return;   // (the second wave, if any, only keeps its slot)
    uint32_t p = threadIdx.x & 3, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < iters; i++) {
        if (WIDE) { const uint2 e = tab[p]; p = e.x; acc += e.y; }
        else p = ((const uint32_t *)tab)[2 * p];
#pragma unroll
        for (int j = 0; j < PAD; j++) asm volatile("v_add_u32 %0, %0, 0" : "+v"(p));   // dependent instructions between the reads
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (p + acc == 0xFFFFFFFFu) sink[0] = 1;
}

template <int WIDE, int PAD>
void run(const char *name, int threads, unsigned long long *d, uint32_t *s) {
    const uint32_t iters = 20000;
    unsigned long long h;
    chase<WIDE, PAD><<<1, threads>>>(iters, d, s);
    chase<WIDE, PAD><<<1, threads>>>(iters, d, s);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f cycles per link\n", name, (double)h / iters);
}

int main() {
    unsigned long long *d; uint32_t *s;
    hipMalloc(&d, 8); hipMalloc(&s, 4);
    run<0, 0>("ds_read_b32 -> ds_read_b32, one wave", 64, d, s);
    run<1, 0>("ds_read_b64 -> ds_read_b64, one wave", 64, d, s);
    run<0, 4>("ds_read_b32 + 4 dependent v_add", 64, d, s);
    run<0, 8>("ds_read_b32 + 8 dependent v_add", 64, d, s);
    run<1, 8>("ds_read_b64 + 8 dependent v_add", 64, d, s);
    run<1, 16>("ds_read_b64 + 16 dependent v_add", 64, d, s);
    run<1, 8>("ds_read_b64 + 8 dependent v_add, 2 waves in the WG", 128, d, s);
    return 0;
}
