/*
 * gsplat_gl_interop.h -- device-resident hand-back to a GL host (header-only, C++).
 *
 * The reference blends straight into the viewport's bound framebuffer
 * (/root/reference/gsplat_plugin/src/GSplatRenderer.C:605-657).  The drop-in keeps the frame on the GPU the same way:
 * the RGBA target is a GL pixel-unpack buffer, and the opaque pass's depth is read by GL into a pixel-pack buffer; both
 * are registered with HIP once (hipGraphicsGLRegisterBuffer) and mapped around the frame, so GSplatRenderContext carries
 * DEVICE pointers (target_is_device = depth_is_device = 1) and nothing crosses PCIe.  INTEGRATION.md section 3 shows the
 * scene hook that uses it.
 *
 * Compile with -DGSPLAT_WITH_GL_INTEROP (needs a ROCm built with GL interop and a current GL context on the same GPU at
 * run time -- neither exists in this repo's CI, so the header is syntax-checked only: __graft_entry__.build()).
 */
#ifndef GSPLAT_GL_INTEROP_H
#define GSPLAT_GL_INTEROP_H
#ifdef GSPLAT_WITH_GL_INTEROP

#include <hip/hip_runtime_api.h>
#include <hip/hip_gl_interop.h>

#include <cstddef>

/* One GL buffer object (PBO) seen from HIP. */
class GSplatGLBuffer {
public:
    GSplatGLBuffer() = default;
    ~GSplatGLBuffer() { release(); }
    GSplatGLBuffer(const GSplatGLBuffer&) = delete;
    GSplatGLBuffer& operator=(const GSplatGLBuffer&) = delete;

    /* gl_buffer: a buffer object with storage already allocated (glBufferData); write_only: HIP overwrites it entirely
     * (the RGBA target) -> hipGraphicsRegisterFlagsWriteDiscard; otherwise read-only (the depth image). */
    bool attach(unsigned int gl_buffer, bool write_only)
    {
        release();
        const unsigned int flags = write_only ? hipGraphicsRegisterFlagsWriteDiscard : hipGraphicsRegisterFlagsReadOnly;
        return hipGraphicsGLRegisterBuffer(&res_, gl_buffer, flags) == hipSuccess;
    }
    /* device pointer valid until unmap(); GL must not touch the buffer in between */
    void* map(hipStream_t stream, size_t* bytes = nullptr)
    {
        if (!res_ || mapped_) return nullptr;
        if (hipGraphicsMapResources(1, &res_, stream) != hipSuccess) return nullptr;
        void* p = nullptr;
        size_t n = 0;
        if (hipGraphicsResourceGetMappedPointer(&p, &n, res_) != hipSuccess) { (void)hipGraphicsUnmapResources(1, &res_, stream); return nullptr; }
        mapped_ = true;
        if (bytes) *bytes = n;
        return p;
    }
    void unmap(hipStream_t stream)
    {
        if (res_ && mapped_) (void)hipGraphicsUnmapResources(1, &res_, stream);   /* orders the stream's work before GL's next use */
        mapped_ = false;
    }
    void release()
    {
        if (res_) { if (mapped_) (void)hipGraphicsUnmapResources(1, &res_, nullptr); (void)hipGraphicsUnregisterResource(res_); }
        res_ = nullptr;
        mapped_ = false;
    }

private:
    hipGraphicsResource* res_ = nullptr;
    bool mapped_ = false;
};

#endif /* GSPLAT_WITH_GL_INTEROP */
#endif
