// micro-benchmarks to calibrate the cost model used for the f64 LM kernels (one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_fma_chain(double *out, int n)
{   // dependent f64 FMA chain
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { a = a * b + c; a = a * b + c; a = a * b + c; a = a * b + c; }
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (4.0 * n);
}
__global__ void k_fma_indep(double *out, int n)
{   // 8 independent chains
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x * 1e-3 + k;
    double b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = a[k] * b + c;
    long long t1 = clock64();
    double s = 0; for (int k = 0; k < 8; ++k) s += a[k];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (8.0 * n);
}
__global__ void k_div_chain(double *out, int n)
{
    double a = 1.0 + threadIdx.x * 1e-3;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { a = 1.0 / (a + 1.0); }
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / n;
}
__global__ void k_sqrt_chain(double *out, int n)
{
    double a = 1.0 + threadIdx.x * 1e-3;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { a = sqrt(a + 2.0); }
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / n;
}
__global__ void k_chase(const int *next, double *out, int n, int start)
{   // pointer chase in global memory (one lane active per wave is enough; all lanes same address)
    int p = start;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) p = next[p];
    long long t1 = clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / n;
}
__global__ void k_lds_chase(double *out, int n)
{
    __shared__ int nx[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) nx[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) p = nx[p];
    long long t1 = clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / n;
}
template <int CTRL> __device__ __forceinline__ double dppd(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k_rowsum(double *out, int n)
{
    double v = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        v += dppd<0xB1>(v); v += dppd<0x4E>(v); v += dppd<0x141>(v); v += dppd<0x140>(v);
        v = v * 0.0625;
    }
    long long t1 = clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / n;
}
__global__ void k_gather36(const double *A, const int *idx, double *out, int n, size_t stride)
{   // 36 plane loads per item (the Schur inner pattern), items consecutive per lane
    double acc = 0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        size_t u = (size_t)idx[i * 64 + threadIdx.x];
        double s = 0;
#pragma unroll
        for (int z = 0; z < 36; ++z) s += A[z * stride + u];
        acc += s;
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / n;
}

int main()
{
    double *d; hipMalloc(&d, 65 * 8);
    double h[65];
    auto rep = [&](const char *nm) { hipDeviceSynchronize(); hipMemcpy(h, d, 65 * 8, hipMemcpyDeviceToHost); printf("%-34s %8.1f cycles\n", nm, h[64]); };
    hipLaunchKernelGGL(k_fma_chain, 1, 64, 0, 0, d, 10000); rep("f64 FMA dependent (per op)");
    hipLaunchKernelGGL(k_fma_indep, 1, 64, 0, 0, d, 10000); rep("f64 FMA independent x8 (per op)");
    hipLaunchKernelGGL(k_div_chain, 1, 64, 0, 0, d, 2000); rep("f64 1/(a+1) dependent");
    hipLaunchKernelGGL(k_sqrt_chain, 1, 64, 0, 0, d, 2000); rep("f64 sqrt(a+2) dependent");
    hipLaunchKernelGGL(k_rowsum, 1, 64, 0, 0, d, 2000); rep("f64 16-lane DPP row sum");
    hipLaunchKernelGGL(k_lds_chase, 1, 64, 0, 0, d, 5000); rep("LDS dependent load");
    // global chase: small (L2-resident) and large (HBM)
    for (size_t n : { (size_t)1 << 12, (size_t)1 << 20, (size_t)1 << 26 }) {
        std::vector<int> nx(n);
        for (size_t i = 0; i < n; ++i) nx[i] = (int)((i * 1048583ull + 12345) % n);
        int *dn; hipMalloc(&dn, n * 4); hipMemcpy(dn, nx.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_chase, 1, 64, 0, 0, dn, d, 2000, 1);
        char nm[64]; snprintf(nm, sizeof nm, "global dependent load (%zu KB)", n * 4 / 1024); rep(nm);
        hipFree(dn);
    }
    {
        size_t stride = 2820, nitems = 64 * 50;
        double *A; hipMalloc(&A, 36 * stride * 8); hipMemset(A, 0, 36 * stride * 8);
        std::vector<int> idx(nitems);
        for (size_t i = 0; i < nitems; ++i) idx[i] = (int)((i * 3) % 2700);
        int *di; hipMalloc(&di, nitems * 4); hipMemcpy(di, idx.data(), nitems * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_gather36, 1, 64, 0, 0, A, di, d, 50, stride); rep("36 SoA plane loads + adds / item");
    }
    return 0;
}
