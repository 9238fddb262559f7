// micro-benchmarks for the integer / LDS instruction mix of k_lk (cycles per instruction per wave,
// 1 wave alone and 32 waves per CU) — run: hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o /tmp/ub2 && /tmp/ub2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint16_t u16_ua __attribute__((aligned(1)));
typedef uint32_t u32_ua __attribute__((aligned(1)));
typedef uint64_t u64_ua __attribute__((aligned(1)));
typedef short s2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void k_lds(double *out, int n, int misalign)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s[i] = (uint8_t)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane stride 4 bytes (conflict-free dword pattern) + misalign
    const volatile uint8_t *p = s + (wave & 1) * 2048 + lane * 4 + misalign;
    uint32_t acc = 0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) acc += *reinterpret_cast<const volatile uint8_t *>(p + k * 256);
            if (MODE == 1) acc += *reinterpret_cast<const volatile u16_ua *>(p + k * 256);
            if (MODE == 2) acc += *reinterpret_cast<const volatile u32_ua *>(p + k * 256);
            if (MODE == 3) { u64_ua v = *reinterpret_cast<const volatile u64_ua *>(p + lane * 4 + k * 512); acc += (uint32_t)v + (uint32_t)(v >> 32); }
            if (MODE == 4) { uint4 v; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(p + lane * 12 + k * 1024)) : "memory"); acc += v.x + v.y + v.z + v.w; }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x + 1] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (8.0 * n);
}

template <int MODE> __global__ void k_alu(double *out, int n)
{
    uint32_t a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 2654435761u + k;
    uint32_t b = threadIdx.x | 1, c = 0x00030005;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) a[k] = a[k] * b;                                               // v_mul_lo_u32
            if (MODE == 1) a[k] = __mul24((int)a[k], (int)b) + 1;                           // v_mad_i32_i24
            if (MODE == 2) a[k] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a[k]), __builtin_bit_cast(s2, c), (int)b, false);
            if (MODE == 3) a[k] = __builtin_amdgcn_perm(a[k], b, 0x05040100u);
            if (MODE == 4) a[k] += __builtin_amdgcn_update_dpp(0, (int)a[k], 0xB1, 0xf, 0xf, true);
            if (MODE == 5) a[k] = a[k] + b;                                               // v_add_u32
            if (MODE == 6) a[k] = __float_as_uint(__uint_as_float(a[k]) * 1.0001f);        // v_mul_f32
            if (MODE == 7) a[k] += __builtin_amdgcn_readlane((int)a[k], 16);
        }
    }
    long long t1 = clock64();
    uint32_t s = 0; for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x + 1] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (8.0 * n);
}

int main()
{
    double *d; hipMalloc(&d, sizeof(double) * (1 << 20));
    double h = 0;
    const int n = 2000;
#define RUN(name, kern, blocks, threads, ...) do { \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, n, ##__VA_ARGS__); hipDeviceSynchronize(); \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, n, ##__VA_ARGS__); hipDeviceSynchronize(); \
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); printf("%-44s %8.2f cyc/instr/wave\n", name, h); } while (0)
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int blocks = cfg ? 256 * 2 : 1, threads = cfg ? 1024 : 64;
        printf("---- %s\n", cfg ? "32 waves per CU (2 x 1024-thread blocks/CU)" : "one wave");
        RUN("ds_read_u8", k_lds<0>, blocks, threads, 0);
        RUN("ds_read_u16 aligned", k_lds<1>, blocks, threads, 0);
        RUN("ds_read_u16 odd (inside dword)", k_lds<1>, blocks, threads, 1);
        RUN("ds_read_u16 straddling dwords", k_lds<1>, blocks, threads, 3);
        RUN("ds_read_b32 aligned", k_lds<2>, blocks, threads, 0);
        RUN("ds_read_b32 misaligned +1", k_lds<2>, blocks, threads, 1);
        RUN("ds_read_b32 misaligned +3", k_lds<2>, blocks, threads, 3);
        RUN("ds_read_b64 aligned 8", k_lds<3>, blocks, threads, 0);
        RUN("ds_read_b64 aligned 4 only", k_lds<3>, blocks, threads, 4);
        RUN("ds_read_b128 aligned 16", k_lds<4>, blocks, threads, 0);
        RUN("ds_read_b128 aligned 8 only", k_lds<4>, blocks, threads, 8);
        RUN("ds_read_b128 aligned 4 only", k_lds<4>, blocks, threads, 4);
        RUN("v_mul_lo_u32", k_alu<0>, blocks, threads);
        RUN("v_mad_i32_i24", k_alu<1>, blocks, threads);
        RUN("v_dot2_i32_i16", k_alu<2>, blocks, threads);
        RUN("v_perm_b32", k_alu<3>, blocks, threads);
        RUN("v_add_u32_dpp", k_alu<4>, blocks, threads);
        RUN("v_add_u32", k_alu<5>, blocks, threads);
        RUN("v_mul_f32", k_alu<6>, blocks, threads);
        RUN("v_readlane+add", k_alu<7>, blocks, threads);
    }
    return 0;
}
