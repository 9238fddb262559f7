// VALU issue-rate calibration for the LK cost model ("2 or 4 cycles per wave64 VALU?").
// Each kernel runs N independent dependency chains of one instruction kind in a loop; the grid puts
// W waves on every SIMD (blocks of 256 threads = 1 wave per SIMD each, W blocks per CU).  Reported:
// shader cycles per wave-instruction per SIMD (= elapsed cycles * 1 / (instructions issued by ONE
// SIMD's waves)), so a saturated 2-cycle pipe reads 2.0 and a 4-cycle pipe 4.0 once W >= 2.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubv && /tmp/ubv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CH 8        // independent chains per lane
#define UNROLL 8

// inline asm: the compiler can neither fold the chains nor reorder across kinds
template <int KIND> __device__ __forceinline__ void step(uint32_t (&a)[CH], uint32_t b, uint32_t c)
{
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 2) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 4) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 5) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
        if (KIND == 6) asm volatile("v_rndne_f32 %0, %0\n\tv_cvt_i32_f32 %0, %0" : "+v"(a[k]));
        if (KIND == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 8) { uint32_t sreg; asm volatile("v_readlane_b32 %0, %1, 16" : "=s"(sreg) : "v"(a[k])); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "s"(sreg)); }
        if (KIND == 10) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 11) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 12) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 14) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[k]));
        if (KIND == 15) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 16) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 17) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 18) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(a[k]) : "v"(b));
        if (KIND == 19) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[k]));
        if (KIND == 20) asm volatile("v_ashrrev_i32 %0, 9, %0" : "+v"(a[k]));
        if (KIND == 21) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[k]) : "v"(b));
        if (KIND == 22) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 23) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 24) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
        if (KIND == 25) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
        if (KIND == 26) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*reinterpret_cast<double *>(&a[k & ~1])) : "v"(a[k]));
        if (KIND == 30) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 31) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 32) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 33) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 34) asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(a[k]));
        if (KIND == 35) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 36) asm volatile("v_mad_i16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 37) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (KIND == 38) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[k]) : "v"(b));
        if (KIND == 39) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[k]));
        if (KIND == 40) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b));
        if (KIND == 13) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&a[k & ~1])) : "v"(__hiloint2double((int)b, (int)c)));
    }
    if (KIND == 41 || KIND == 42 || KIND == 43) {
        double *d = reinterpret_cast<double *>(a);
        const double one = __hiloint2double((int)(c | 0x3ff00000u), (int)b);
#pragma unroll
        for (int k = 0; k < CH / 2; ++k) {
            if (KIND == 41) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[k]) : "v"(one)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[k]) : "v"(one)); }
            if (KIND == 42) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(one)); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(one)); }
            if (KIND == 43) for (int rep = 0; rep < 2; ++rep) {      // what dpp_f64 + add costs: no DPP form of v_add_f64 exists
                uint32_t lo, hi;
                asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                             : "=&v"(lo), "=&v"(hi) : "v"(a[2 * k]), "v"(a[2 * k + 1]));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(__hiloint2double((int)hi, (int)lo)));
            }
        }
    }
    if (KIND == 9) {
        double *d = reinterpret_cast<double *>(a);
        const double one = __hiloint2double((int)(c | 0x3ff00000u), (int)b);
#pragma unroll
        for (int k = 0; k < CH / 2; ++k) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(one)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(one)); }
    }
}

template <int KIND> __global__ void __launch_bounds__(256) k(uint32_t *out, int iters, long long *cyc)
{
    __attribute__((aligned(8))) uint32_t a[CH];
#pragma unroll
    for (int k2 = 0; k2 < CH; ++k2) a[k2] = threadIdx.x * 2654435761u + k2;
    const uint32_t b = out[0] | 3u, c = out[1] | 0x01000504u;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) step<KIND>(a, b, c);
    }
    const long long t1 = clock64();
    uint32_t s = 0;
#pragma unroll
    for (int k2 = 0; k2 < CH; ++k2) s ^= a[k2];
    if (s == 0x12345678u) out[2] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND> void run(const char *name, int instr_per_step, uint32_t *d_out, long long *d_cyc, int ncu)
{
    const int iters = 20000;
    printf("%-28s", name);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int W : { 1, 2, 4, 8 }) {
        hipLaunchKernelGGL(k<KIND>, dim3(ncu * W), dim3(256), 0, 0, d_out, 10, d_cyc);      // warm-up
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(ncu * W), dim3(256), 0, 0, d_out, iters, d_cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        long long cyc = 0;
        hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
        // one SIMD hosts W waves (one per resident block); each issues iters * UNROLL * CH * instr_per_step
        const double n_instr = (double)iters * UNROLL * CH * instr_per_step * W;
        // wall time of the whole grid -> ns per wave-instruction per SIMD; clock64 of block 0 next to it
        printf("  W=%d: %5.2f ns (%5.2f clk64)", W, 1e6 * ms / n_instr, (double)cyc / n_instr);
    }
    printf("   per wave-instruction per SIMD\n");
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    uint32_t *d_out; long long *d_cyc;
    hipMalloc(&d_out, 64); hipMemset(d_out, 0, 64); hipMalloc(&d_cyc, 8);
    printf("%s, %d CUs, W = waves per SIMD (blocks of 256 threads per CU)\n", p.name, ncu);
    run<0>("v_add_u32", 1, d_out, d_cyc, ncu);
    run<1>("v_fma_f32", 1, d_out, d_cyc, ncu);
    run<2>("v_mad_i32_i24", 1, d_out, d_cyc, ncu);
    run<3>("v_perm_b32", 1, d_out, d_cyc, ncu);
    run<4>("v_dot2_i32_i16", 1, d_out, d_cyc, ncu);
    run<10>("v_dot4_u32_u8", 1, d_out, d_cyc, ncu);
    run<5>("v_add_u32 dpp", 1, d_out, d_cyc, ncu);
    run<6>("v_rndne_f32 + v_cvt_i32_f32", 2, d_out, d_cyc, ncu);
    run<7>("v_mul_lo_u32", 1, d_out, d_cyc, ncu);
    run<8>("v_readlane_b32 + v_add_u32", 2, d_out, d_cyc, ncu);
    run<9>("v_add_f64", 1, d_out, d_cyc, ncu);
    run<11>("v_mul_i32_i24", 1, d_out, d_cyc, ncu);
    run<12>("v_and_b32", 1, d_out, d_cyc, ncu);
    run<14>("v_lshlrev_b32", 1, d_out, d_cyc, ncu);
    run<20>("v_ashrrev_i32", 1, d_out, d_cyc, ncu);
    run<17>("v_and_or_b32", 1, d_out, d_cyc, ncu);
    run<18>("v_add_lshl_u32", 1, d_out, d_cyc, ncu);
    run<21>("v_alignbyte_b32", 1, d_out, d_cyc, ncu);
    run<22>("v_mad_u32_u24", 1, d_out, d_cyc, ncu);
    run<15>("v_mul_f32", 1, d_out, d_cyc, ncu);
    run<16>("v_sub_f32", 1, d_out, d_cyc, ncu);
    run<23>("v_max_f32", 1, d_out, d_cyc, ncu);
    run<19>("v_cvt_f32_i32", 1, d_out, d_cyc, ncu);
    run<24>("v_mov_b32 dpp", 1, d_out, d_cyc, ncu);
    run<25>("v_sqrt_f32", 1, d_out, d_cyc, ncu);
    run<26>("v_cvt_f64_f32", 1, d_out, d_cyc, ncu);
    // round 3: packed 16-bit integer VALU, priced before k_lk's Scharr / A stage moves onto it
    run<30>("v_pk_add_i16", 1, d_out, d_cyc, ncu);
    run<31>("v_pk_sub_i16", 1, d_out, d_cyc, ncu);
    run<35>("v_pk_add_u16", 1, d_out, d_cyc, ncu);
    run<32>("v_pk_mad_i16", 1, d_out, d_cyc, ncu);
    run<33>("v_pk_mul_lo_u16", 1, d_out, d_cyc, ncu);
    run<34>("v_pk_lshlrev_b16", 1, d_out, d_cyc, ncu);
    run<36>("v_mad_i16", 1, d_out, d_cyc, ncu);
    run<37>("v_sad_u8", 1, d_out, d_cyc, ncu);
    run<38>("v_lshl_add_u32", 1, d_out, d_cyc, ncu);
    run<39>("v_bfe_u32", 1, d_out, d_cyc, ncu);
    run<40>("v_cndmask_b32", 1, d_out, d_cyc, ncu);
    run<41>("v_fma_f64", 1, d_out, d_cyc, ncu);
    run<42>("v_mul_f64", 1, d_out, d_cyc, ncu);
    run<43>("2 v_mov dpp + v_add_f64", 3, d_out, d_cyc, ncu);
    return 0;
}
