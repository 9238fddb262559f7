#include <hip/hip_runtime.h>
__global__ void k(int *o) {
    int v = threadIdx.x * 3 + 1;
    int l = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
    int r = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    o[threadIdx.x] = l; o[64 + threadIdx.x] = r;
}
int main() { int *d; hipMalloc(&d, 512); k<<<1,64>>>(d); int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("shr: %d %d %d ... %d %d | shl: %d %d ... %d %d\n", h[0], h[1], h[2], h[62], h[63], h[64], h[65], h[126], h[127]); return 0; }
