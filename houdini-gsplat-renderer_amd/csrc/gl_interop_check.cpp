// gl_interop_check.cpp -- compile-only check of include/gsplat_gl_interop.h (it needs a GL host to run); not part of the library.
#define GSPLAT_WITH_GL_INTEROP
#include "gsplat_gl_interop.h"
#include "GSplatPrim.h"

int gsplat_gl_interop_check(unsigned int rgba_pbo, unsigned int depth_pbo, GSplatRenderContext* ctx)
{
    GSplatGLBuffer rgba, depth;
    if (!rgba.attach(rgba_pbo, true) || !depth.attach(depth_pbo, false)) return -1;
    ctx->target = static_cast<float*>(rgba.map(nullptr));
    ctx->target_is_device = 1;
    ctx->depth = static_cast<const float*>(depth.map(nullptr));
    ctx->depth_is_device = 1;
    rgba.unmap(nullptr);
    depth.unmap(nullptr);
    return 0;
}
