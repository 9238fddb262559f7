// f64 on the matrix cores against f64 on the vector ALU ("try v_mfma_f64_16x16x4_f64 for the blocked
// trailing update of the 60x60 factorisation"): issue rate of v_mfma_f64_16x16x4_f64 (512 fma per wave-instruction) against
// v_fma_f64 (64 fma per wave-instruction) with CH independent accumulator chains per wave, W waves per SIMD, and the latency of
// a dependent chain of each.  What the reduced camera system could gain from MFMA is bounded by the ratio of the two rates.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_f64.hip -o tools/bin/ubench_mfma_f64 && tools/bin/ubench_mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define ITERS 2048

template <int CH> __global__ void k_mfma(double *out, double a0, double b0)
{
    d4 acc[CH];
    for (int k = 0; k < CH; ++k) acc[k] = d4{ 0, 0, 0, 0 };
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < ITERS; ++i)
#pragma unroll
        for (int k = 0; k < CH; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    double s = 0;
    for (int k = 0; k < CH; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    if (s == 12345.678) out[0] = s;
}
template <int CH> __global__ void k_fma(double *out, double a0, double b0)
{
    double acc[CH];
    for (int k = 0; k < CH; ++k) acc[k] = k;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < ITERS; ++i)
#pragma unroll
        for (int k = 0; k < CH; ++k) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
    double s = 0;
    for (int k = 0; k < CH; ++k) s += acc[k];
    if (s == 12345.678) out[0] = s;
}

template <class F> static double run(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double *d; hipMalloc(&d, 8);
    printf("%s, %d CUs, clock %.0f MHz\n", p.name, cus, p.clockRate / 1e3);
    for (int w : { 1, 2, 4 }) {          // waves per SIMD: blocks of 256 threads (one wave per SIMD), w blocks per CU
        const int blocks = cus * w;
        const double waves = (double)blocks * 4;
        double t;
        t = run([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, d, 1.0, 1e-3); });
        printf("v_mfma_f64_16x16x4_f64, 4 chains, %d wave(s)/SIMD: %7.1f TFLOP/s  (%.2f ns per wave-instruction per SIMD)\n", w,
               2.0 * 16 * 16 * 4 * 4 * ITERS * waves / t / 1e12, t / (4.0 * ITERS * w) * 1e9);
        t = run([&] { hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(256), 0, 0, d, 1.0, 1e-3); });
        printf("v_fma_f64,              8 chains, %d wave(s)/SIMD: %7.1f TFLOP/s  (%.2f ns per wave-instruction per SIMD)\n", w,
               2.0 * 64 * 8 * ITERS * waves / t / 1e12, t / (8.0 * ITERS * w) * 1e9);
    }
    double t = run([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(cus), dim3(64), 0, 0, d, 1.0, 1e-3); });
    printf("dependent chain: v_mfma_f64_16x16x4_f64 %.1f ns per instruction", t / ITERS * 1e9);
    t = run([&] { hipLaunchKernelGGL(k_fma<1>, dim3(cus), dim3(64), 0, 0, d, 1.0, 1e-3); });
    printf(", v_fma_f64 %.1f ns per instruction\n", t / ITERS * 1e9);
    return 0;
}
