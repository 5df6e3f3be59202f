// wavefront_kernels.hip -- wavefront (extend / shade / connect) pipeline.  Placeholder until the kernels land.
#include <hip/hip_runtime.h>

#include "cpugpupt_abi.h"
#include "device_scene.h"

namespace cgpt {
int CtxFail(cgpt_ctx* ctx, int code, const char* fmt, ...);

int LaunchWavefront(cgpt_ctx* ctx, const DevRenderArgs&, bool)
{
    CtxFail(ctx, CGPT_ERR_UNSUPPORTED, "CGPT_KERNEL_WAVEFRONT is not implemented yet");
    return -1;
}
}  // namespace cgpt
