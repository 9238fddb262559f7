// Calibration of the FETCH_SIZE / WRITE_SIZE PMC counters on gfx950 for the access widths this library uses:
// every kernel streams the same 1 GiB buffer once (reads) or writes 1 GiB once, with 1 / 4 / 8 / 16 bytes per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o tools/bin/ubench_fetch
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/bin/ubench_fetch   (then WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T> __global__ void k_read(const T *p, size_t n, unsigned long long *sink)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n; i += stride) {
        T v = p[i];
        const unsigned char *b = reinterpret_cast<const unsigned char *>(&v);
        acc += b[0];
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}
template <typename T> __global__ void k_write(T *p, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    T v; __builtin_memset(&v, 1, sizeof(T));
    for (; i < n; i += stride) p[i] = v;
}
template <typename T> __global__ void k_read_rows(const T *p, size_t n, unsigned long long *sink)
{
    // misaligned by 3 elements of 4 bytes: the GFTT access pattern (wave row starts 12 bytes off a 64-byte line)
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x + 3;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n; i += stride) acc += (unsigned long long)p[i];
    if (acc == 0x123456789ull) sink[0] = acc;
}

int main()
{
    const size_t bytes = 1ull << 30;
    void *buf; unsigned long long *sink;
    hipMalloc(&buf, bytes + 4096); hipMalloc(&sink, 8);
    hipMemset(buf, 1, bytes + 4096);
    hipDeviceSynchronize();
    const dim3 g(256 * 32), b(256);
    hipLaunchKernelGGL(k_read<unsigned char>, g, b, 0, 0, (const unsigned char *)buf, bytes, sink);
    hipLaunchKernelGGL(k_read<uint32_t>, g, b, 0, 0, (const uint32_t *)buf, bytes / 4, sink);
    hipLaunchKernelGGL(k_read<uint2>, g, b, 0, 0, (const uint2 *)buf, bytes / 8, sink);
    hipLaunchKernelGGL(k_read<uint4>, g, b, 0, 0, (const uint4 *)buf, bytes / 16, sink);
    hipLaunchKernelGGL(k_read_rows<uint32_t>, g, b, 0, 0, (const uint32_t *)buf, bytes / 4, sink);
    hipLaunchKernelGGL(k_write<unsigned char>, g, b, 0, 0, (unsigned char *)buf, bytes);
    hipLaunchKernelGGL(k_write<uint32_t>, g, b, 0, 0, (uint32_t *)buf, bytes / 4);
    hipLaunchKernelGGL(k_write<uint2>, g, b, 0, 0, (uint2 *)buf, bytes / 8);
    hipLaunchKernelGGL(k_write<uint4>, g, b, 0, 0, (uint4 *)buf, bytes / 16);
    hipDeviceSynchronize();
    printf("done: every kernel moved %zu bytes\n", bytes);
    return 0;
}
