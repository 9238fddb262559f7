// synth.hip — HIP generator of the synthetic stereo stream (bench / test input,
// rendered straight into HBM so the timed region starts with resident frames).
// Input generation only; not part of the hot path.
#include <hip/hip_runtime.h>
#include "synth_scene.h"

__global__ void __launch_bounds__(256)
k_synth_render(const svs_synth_view *views, int w, int h, uint8_t *out)
{
    const svs_synth_view v = views[blockIdx.z];
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    out[(size_t)blockIdx.z * w * h + (size_t)y * w + x] = svs_synth_pixel(&v, x, y);
}

extern "C" int svslam_synth_render_batch(int device, int n, const svs_synth_view *views, int w, int h,
                                         void *d_out)
{
    if (n <= 0) return 0;
    if (hipSetDevice(device) != hipSuccess) return -1;
    svs_synth_view *dv = nullptr;
    if (hipMalloc(&dv, sizeof(svs_synth_view) * n) != hipSuccess) return -2;
    int rc = 0;
    if (hipMemcpy(dv, views, sizeof(svs_synth_view) * n, hipMemcpyHostToDevice) != hipSuccess) rc = -3;
    if (!rc) {
        dim3 blk(64, 4), grd((w + 63) / 64, (h + 3) / 4, n);
        hipLaunchKernelGGL(k_synth_render, grd, blk, 0, 0, dv, w, h, static_cast<uint8_t *>(d_out));
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = -4;
    }
    (void)hipFree(dv);
    return rc;
}
