// Issue-slot sharing between VALU and the other instruction types of a wave (SALU, LDS, s_nop, branches)
// on gfx950: does a scalar instruction between two vector instructions cost VALU issue time?
// Same harness as ubench_valu.hip: W waves per SIMD on every CU, wall time by hipEvent, reported per
// GROUP (one VALU instruction + the extras named in the row) per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mix.hip -o tools/bin/ubench_mix && tools/bin/ubench_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CH 8
#define UNROLL 8

template <int KIND> __device__ __forceinline__ void step(uint32_t (&a)[CH], uint32_t (&s)[CH], uint32_t b, uint32_t c, const uint32_t *lds)
{
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        if (KIND == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        else if (KIND == 13) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 3" : "+v"(a[k]), "+s"(s[k]) : "v"(b) : "scc");
        else asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s[k]) : : "scc");
        if (KIND == 2) asm volatile("s_add_u32 %0, %0, 3\n\ts_xor_b32 %0, %0, 5" : "+s"(s[k]) : : "scc");
        if (KIND == 3) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(s[k]));
        if (KIND == 4) asm volatile("s_nop 0");
        if (KIND == 5) asm volatile("s_cmp_lg_u32 %0, 7\n\ts_cselect_b32 %0, %0, 9" : "+s"(s[k]) : : "scc");
        if (KIND == 6) { uint32_t t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((uint32_t)(uintptr_t)lds + ((threadIdx.x * 4 + k * 1024) & 8191))); asm volatile("s_waitcnt lgkmcnt(7)\n\tv_xor_b32 %0, %0, %1" : "+v"(a[(k + 1) & (CH - 1)]) : "v"(t)); }
        if (KIND == 7) asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 3" : "+s"(s[k]) : : "scc");
        if (KIND == 8) asm volatile("s_cmp_lg_u32 %0, 7\n\ts_cbranch_scc0 1f\n\ts_add_u32 %0, %0, 1\n1:" : "+s"(s[k]) : : "scc");   // never-taken branch
        if (KIND == 9) asm volatile("s_cmp_lg_u32 %0, 7\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 1\n1:" : "+s"(s[k]) : : "scc");   // always-taken branch
        if (KIND == 10) { uint32_t t; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(t) : "v"(a[k])); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[k]) : "s"(t) : "scc"); }
        if (KIND == 11) asm volatile("s_waitcnt lgkmcnt(0)");
    }
}

template <int KIND> __global__ void __launch_bounds__(256) k(uint32_t *out, int iters)
{
    __shared__ uint32_t lds[4096];
    uint32_t a[CH], s[CH];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = 1;
#pragma unroll
    for (int k2 = 0; k2 < CH; ++k2) { a[k2] = threadIdx.x * 2654435761u + k2; s[k2] = __builtin_amdgcn_readfirstlane(out[3] + k2 + 11); }
    const uint32_t b = out[0] | 3u, c = out[1] | 0x01000504u;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) step<KIND>(a, s, b, c, lds);
    }
    uint32_t x = 0;
#pragma unroll
    for (int k2 = 0; k2 < CH; ++k2) x ^= a[k2] ^ s[k2];
    if (x == 0x12345678u) out[2] = x;
}

template <int KIND> void run(const char *name, uint32_t *d_out, int ncu)
{
    const int iters = 10000;
    printf("%-44s", name);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int W : { 1, 2, 4, 8 }) {
        hipLaunchKernelGGL(k<KIND>, dim3(ncu * W), dim3(256), 0, 0, d_out, 10);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(ncu * W), dim3(256), 0, 0, d_out, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * UNROLL * CH * W;
        printf("  W=%d: %5.2f ns", W, 1e6 * ms / n);
    }
    printf("   per group per SIMD\n");
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    uint32_t *d_out; hipMalloc(&d_out, 64); hipMemset(d_out, 0, 64);
    printf("%s, %d CUs, W = waves per SIMD; group = one VALU instruction + the extras\n", p.name, ncu);
    run<0>("v_mad_i32_i24", d_out, ncu);
    run<1>("v_mad_i32_i24 + s_add_u32", d_out, ncu);
    run<2>("v_mad_i32_i24 + 2 dependent SALU", d_out, ncu);
    run<7>("v_mad_i32_i24 + 4 dependent SALU", d_out, ncu);
    run<3>("v_mad_i32_i24 + s_mul_i32", d_out, ncu);
    run<4>("v_mad_i32_i24 + s_nop 0", d_out, ncu);
    run<5>("v_mad_i32_i24 + s_cmp + s_cselect", d_out, ncu);
    run<8>("v_mad_i32_i24 + s_cmp + branch not taken", d_out, ncu);
    run<9>("v_mad_i32_i24 + s_cmp + branch taken", d_out, ncu);
    run<10>("v_mad_i32_i24 + v_readfirstlane + s_add", d_out, ncu);
    run<11>("v_mad_i32_i24 + s_waitcnt lgkmcnt(0)", d_out, ncu);
    run<6>("v_mad_i32_i24 + ds_read_b32 + v_xor", d_out, ncu);
    run<12>("v_add_u32", d_out, ncu);
    run<13>("v_add_u32 + s_add_u32", d_out, ncu);
    return 0;
}
