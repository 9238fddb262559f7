// pipeline_capi.cpp — product build of the host pipeline: the reference-shaped
// Frontend/Backend/Map host logic bound to the HIP kernels through the C ABI.
#include "kernels_hip.h"
#define SVS_PIPE_KERNELS svs::HipKernels
#define SVS_PIPE_MAKE_KERNELS(lim) new svs::HipKernels(lim)
#define SVS_PIPE_IMAGES_ARE_DEVICE 1
#include "pipeline_capi_impl.h"

extern "C" void *svs_pipe_kernel_ctx(void *p) { return static_cast<PipeHandle *>(p)->kernels->ctx(); }
extern "C" void *svs_pipe_backend_ctx(void *p) { return static_cast<PipeHandle *>(p)->kernels->backend_ctx(); }
