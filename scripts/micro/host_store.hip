// Can a kernel's own stores carry the decoder's output to the host while it runs? 16-byte coalesced stores from every CU into
// hipHostMalloc'd (mapped, coherent) memory against the same bytes moved by a DMA after the kernel; also a kernel that writes every
// tile twice (device buffer + host image), which is what dec_lz would do.
// Build: hipcc --offload-arch=gfx950 -O3 -o host_store scripts/micro/host_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// every workgroup (256 threads) copies tiles of 4 KiB: reads 16 B per lane from src (device), stores to dst_a and (optionally) dst_b
__global__ void copy_tiles(const uint4 *__restrict__ src, uint4 *__restrict__ dst_a, uint4 *__restrict__ dst_b, size_t n16, int nt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = src[i];
        if (dst_a) {
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            if (nt) __builtin_nontemporal_store((v4u){v.x, v.y, v.z, v.w}, (v4u *)(dst_a + i));
            else dst_a[i] = v;
        }
        if (dst_b) dst_b[i] = v;
    }
}

int main() {
    const size_t bytes = 256u << 20, n16 = bytes / 16;
    uint4 *d_src, *d_dst, *h_map, *h_pin;
    CK(hipMalloc(&d_src, bytes)); CK(hipMalloc(&d_dst, bytes));
    CK(hipHostMalloc(&h_map, bytes, hipHostMallocMapped));
    CK(hipHostMalloc(&h_pin, bytes, hipHostMallocDefault));
    CK(hipMemset(d_src, 0x5A, bytes));
    memset(h_map, 0, bytes); memset(h_pin, 0, bytes);
    uint4 *d_map;
    CK(hipHostGetDevicePointer((void **)&d_map, h_map, 0));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto fn) {
        fn(); CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("%-78s %7.3f ms  %6.1f GB/s\n", name, best, bytes / best / 1e6);
    };
    timeit("DMA device -> pinned host (hipMemcpyAsync)", [&] { CK(hipMemcpyAsync(h_pin, d_src, bytes, hipMemcpyDeviceToHost, 0)); });
    timeit("kernel copy device -> device, 256 x 8 workgroups", [&] { copy_tiles<<<2048, 256>>>(d_src, d_dst, nullptr, n16, 0); });
    for (int wg : {64, 256, 1024, 4096})
        for (int nt : {0, 1}) {
            char nm[128];
            snprintf(nm, sizeof nm, "kernel stores -> mapped host, %4d workgroups of 256%s", wg, nt ? ", nontemporal" : "");
            timeit(nm, [&] { copy_tiles<<<wg, 256>>>(d_src, d_map, nullptr, n16, nt); });
        }
    timeit("kernel stores -> device AND mapped host, 2048 workgroups", [&] { copy_tiles<<<2048, 256>>>(d_src, d_map, d_dst, n16, 0); });
    timeit("kernel stores -> device AND mapped host, 256 workgroups", [&] { copy_tiles<<<256, 256>>>(d_src, d_map, d_dst, n16, 0); });
    // correctness of what arrived
    size_t bad = 0;
    for (size_t i = 0; i < n16; i += 4097) if (h_map[i].x != 0x5A5A5A5Au) bad++;
    printf("mapped image %s\n", bad ? "WRONG" : "ok");
    return 0;
}
