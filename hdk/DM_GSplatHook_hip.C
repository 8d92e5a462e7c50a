/*
 * DM_GSplatHook_hip.C -- the scene render hook of the GSplat plugin over libgsplat_hip (HDK glue).
 *
 * Takes the place of the reference's /root/reference/gsplat_plugin/src/DM_GSplatHook.C: the same DM_SceneHook /
 * DM_SceneRenderHook pair registered for DM_HOOK_BEAUTY, DM_HOOK_AFTER_NATIVE at priority INT_MAX (:66-71), the same
 * three renderer phases per redraw (:30-39).  What differs is what `r` is asked for: the reference hands its renderer
 * the RE_RenderContext and draws into the bound framebuffer with GL; here the matrices are read out of `r` into a
 * GSplatRenderContext, the frame is composited by HIP kernels into a GL pixel-unpack buffer that HIP has mapped
 * (include/gsplat_gl_interop.h: nothing crosses PCIe), and the premultiplied result is drawn as ONE screen-filling
 * triangle with the reference's own blend function (src/GSplatRenderer.C:613-621), so compositing over the passes
 * before it is unchanged.  The opaque pass's depth is read by GL into a pixel-pack buffer and mapped the same way
 * (the reference tests against it with depth writes off, src/GSplatRenderer.C:595-610).
 *
 * EARLY-UNMAP HAZARD: with occlusion culling on (the default) gsr_render may write a frame that broke a depth horizon into
 * the target and overwrite it with the repaired frame before it returns; both writes are ordered on the hook's stream -- the one
 * handed to gsr_set_stream AND to map() / unmap() below.  A consumer that is ordered behind unmap() -- as here -- only ever sees the final frame; one that
 * unmapped (or read through another stream) before R.render() returned could see the first attempt.  Keep the unmap
 * after postRender().
 *
 * NOT BUILT IN THIS REPOSITORY (no HDK, no GL stack on the build or test machines: profiles/egl_probe_mi355x.txt): this file is
 * TYPE-CHECKED against stand-ins for the HDK classes it touches (tests/hdk_mock/, tests/test_hdk_glue.py), the interop header is
 * syntax-checked, and this path -- map, render to device pointers, unmap, textured draw -- has NEVER EXECUTED.
 * Everything between map() and unmap() is what the repo's -m gpu tests run through target_is_device = 1.  Build: hdk/build.sh.
 */
#include <DM/DM_RenderTable.h>
#include <DM/DM_SceneHook.h>
#include <DM/DM_VPortAgent.h>
#include <GUI/GUI_DisplayOption.h>
#include <RE/RE_Render.h>
#include <UT/UT_Matrix4.h>

#include <climits>
#include <cstring>

#define GSPLAT_WITH_GL_INTEROP
#include "gsplat_gl_interop.h"      /* this repo: include/ */
#include "GSplatRenderer.h"         /* this repo: include/ */

/* GR_PrimGsplat::render asks for the wireframe overlay of this redraw (display mode wireframe / wire-over) */
static bool theWireRequested = false;
void GSplatHipRequestWireOverlay(bool on) { theWireRequested = theWireRequested || on; }

/* ONE HIP stream for the whole plugin, owned next to the engine -- which is a process-wide singleton shared by every viewport
 * (GSplatRenderer::getInstance()).  A stream per viewport hook (round 5) was wrong twice over: in a quad view hook B re-bound the
 * engine to ITS stream, hook A -- whose cached "bound engine" still matched -- went on mapping and unmapping on a stream the kernels
 * no longer ran on (the depth map was not ordered before K1); and a closing viewport destroyed a stream the engine was still bound
 * to.  The stream lives as long as the process; the binding is redone whenever the engine object changes (single <-> multi GPU). */
static hipStream_t theHookStream = nullptr;
static const void* theBoundEngine = nullptr;     /* the engine whose public stream is theHookStream */
static hipStream_t hookStream(GSplatRenderer& R)
{
    if (!theHookStream && hipStreamCreateWithFlags(&theHookStream, hipStreamNonBlocking) != hipSuccess) theHookStream = nullptr;
    const void* engineNow = R.engine() ? static_cast<const void*>(R.engine()) : static_cast<const void*>(R.multi());
    if (engineNow != theBoundEngine) {           /* (once per engine: re-binding drains the context) */
        if (R.engine()) (void)gsr_set_stream(R.engine(), theHookStream);
        else if (R.multi()) (void)gsr_multi_set_stream(R.multi(), theHookStream);
        theBoundEngine = engineNow;
    }
    return theHookStream;
}

namespace {

/* the two pixel buffers of one viewport and the texture the result is drawn from */
struct ViewportBuffers {
    int width = 0, height = 0;
    unsigned rgbaPbo = 0, depthPbo = 0, tex = 0, vao = 0;
    GSplatGLBuffer rgba, depth;

    bool resize(int w, int h)
    {
        if (w == width && h == height && rgbaPbo) return true;
        release();
        glGenBuffers(1, &rgbaPbo);
        glBindBuffer(GL_PIXEL_UNPACK_BUFFER, rgbaPbo);
        glBufferData(GL_PIXEL_UNPACK_BUFFER, (GLsizeiptr)w * h * 16, nullptr, GL_STREAM_DRAW);
        glBindBuffer(GL_PIXEL_UNPACK_BUFFER, 0);
        glGenBuffers(1, &depthPbo);
        glBindBuffer(GL_PIXEL_PACK_BUFFER, depthPbo);
        glBufferData(GL_PIXEL_PACK_BUFFER, (GLsizeiptr)w * h * 4, nullptr, GL_STREAM_READ);
        glBindBuffer(GL_PIXEL_PACK_BUFFER, 0);
        glGenTextures(1, &tex);
        glBindTexture(GL_TEXTURE_2D, tex);
        glTexStorage2D(GL_TEXTURE_2D, 1, GL_RGBA32F, w, h);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
        glBindTexture(GL_TEXTURE_2D, 0);
        glGenVertexArrays(1, &vao);
        width = w; height = h;
        return rgba.attach(rgbaPbo, /*write_only=*/true) && depth.attach(depthPbo, /*write_only=*/false);
    }
    void release()
    {
        rgba.release(); depth.release();
        if (rgbaPbo) glDeleteBuffers(1, &rgbaPbo);
        if (depthPbo) glDeleteBuffers(1, &depthPbo);
        if (tex) glDeleteTextures(1, &tex);
        if (vao) glDeleteVertexArrays(1, &vao);
        rgbaPbo = depthPbo = tex = vao = 0; width = height = 0;
    }
};

/* a 3-vertex program that samples the frame 1:1 (gl_VertexID -> a triangle covering the viewport) */
unsigned blitProgram()
{
    static unsigned prog = 0;
    if (prog) return prog;
    const char* vs = "#version 330\nout vec2 uv;\nvoid main(){ vec2 p = vec2((gl_VertexID << 1) & 2, gl_VertexID & 2);"
                     " uv = p; gl_Position = vec4(p * 2.0 - 1.0, 0.0, 1.0); }";
    const char* fs = "#version 330\nuniform sampler2D frame;\nin vec2 uv;\nout vec4 o;\nvoid main(){ o = texture(frame, uv); }";
    auto compile = [](unsigned kind, const char* text) { unsigned s = glCreateShader(kind); glShaderSource(s, 1, &text, nullptr); glCompileShader(s); return s; };
    prog = glCreateProgram();
    glAttachShader(prog, compile(GL_VERTEX_SHADER, vs));
    glAttachShader(prog, compile(GL_FRAGMENT_SHADER, fs));
    glLinkProgram(prog);
    return prog;
}

}  // namespace

class GSplatHipSceneRenderHook : public DM_SceneRenderHook
{
public:
    GSplatHipSceneRenderHook(DM_VPortAgent& vport, DM_ViewportType view_mask) : DM_SceneRenderHook(vport, view_mask) {}
    ~GSplatHipSceneRenderHook() override
    {
        /* (the buffers are this viewport's; the stream is the plugin's: the engine stays bound to it) */
        if (theHookStream) (void)hipStreamSynchronize(theHookStream);
        myBuffers.release();
    }

    bool render(RE_RenderContext r, const DM_SceneHookData& hook_data) override
    {
        GSplatRenderer& R = GSplatRenderer::getInstance();
        GSplatRenderContext ctx{};
        /* the uniforms the reference's shader takes from Houdini (glH_ViewMatrix, glH_ProjectMatrix; object = identity
         * at SOP level).  UT_Matrix4F::data() -- row-major storage of a row-vector matrix -- is byte for byte the GL
         * column-major layout GSplatRenderContext documents */
        UT_Matrix4D view, proj;
        r->getMatrix(view);
        r->getProjectionMatrix(proj);
        const UT_Matrix4F v(view), pm(proj), ident(1.0f);
        std::memcpy(ctx.view, v.data(), 64);
        std::memcpy(ctx.obj_view, v.data(), 64);
        std::memcpy(ctx.proj, pm.data(), 64);
        std::memcpy(ctx.object, ident.data(), 64);
        std::memcpy(ctx.inv_object, ident.data(), 64);
        ctx.width = hook_data.view_width;
        ctx.height = hook_data.view_height;
        if (ctx.width <= 0 || ctx.height <= 0 || !myBuffers.resize(ctx.width, ctx.height)) {
            R.postRender();     /* the redraw's marks are consumed either way (src/GSplatRenderer.C:660-678) */
            theWireRequested = false;
            return true;
        }

        /* the opaque pass's depth attachment -> pack buffer (a copy inside the GPU) -> HIP */
        glBindBuffer(GL_PIXEL_PACK_BUFFER, myBuffers.depthPbo);
        glReadPixels(0, 0, ctx.width, ctx.height, GL_DEPTH_COMPONENT, GL_FLOAT, nullptr);
        glBindBuffer(GL_PIXEL_PACK_BUFFER, 0);
        /* ONE stream orders everything: the maps, the engine's kernels (gsr_set_stream: a context's frames are ordered on its
         * public stream -- its own stream is hipStreamNonBlocking and would NOT be ordered against stream 0) and the unmaps, which
         * is what hands the finished frame back to GL.  It is the plugin's, not this viewport's (hookStream above). */
        hipStream_t stream = hookStream(R);
        ctx.depth = static_cast<const float*>(myBuffers.depth.map(stream));
        ctx.depth_is_device = 1;
        ctx.target = static_cast<float*>(myBuffers.rgba.map(stream));
        ctx.target_is_device = 1;

        bool drew = false;
        if (ctx.target) {
            R.generateRenderGeometry(ctx);      /* re-stages only when the set of marked primitives changed */
            R.render(ctx, hook_data.disp_options->isObjectLevel());
            drew = R.query(GSplatRenderer::Q_CAN_RENDER) != 0;
            /* (several GPUs: the stitched frame and the whole cloud live on the root rank's context) */
            gsr_context* wireEngine = R.engine() ? R.engine() : (R.multi() ? gsr_multi_context(R.multi(), 0) : nullptr);
            if (theWireRequested && R.multi()) (void)gsr_multi_synchronize(R.multi());   /* the gather has landed before the outlines go on top */
            if (theWireRequested && wireEngine) {
                /* wire-over display: the reference draws the outlines and still includes the primitive in the splat pass
                 * (src/GR_GSplat.C:471-486) -- the outlines go ON TOP of the beauty frame just rendered, from the resident arrays */
                gsr_camera cam{};
                std::memcpy(cam.obj_view, ctx.obj_view, 64); std::memcpy(cam.object, ctx.object, 64);
                std::memcpy(cam.inv_object, ctx.inv_object, 64); std::memcpy(cam.view, ctx.view, 64); std::memcpy(cam.proj, ctx.proj, 64);
                cam.width = ctx.width; cam.height = ctx.height; cam.sh_order = 0;
                drew = (drew ? gsr_render_wire_over(wireEngine, &cam, ctx.target, 1) : gsr_render_wire(wireEngine, &cam, ctx.target, 1)) == GSR_OK || drew;
            }
        }
        R.postRender();
        theWireRequested = false;
        /* (after postRender: see the early-unmap hazard in the header comment).  Belt and braces: the frame's last kernel has
         * finished before GL may touch the buffers, whatever a driver makes of the stream argument of the unmap */
        if (R.engine()) (void)gsr_synchronize(R.engine());
        else if (R.multi()) (void)gsr_multi_synchronize(R.multi());
        myBuffers.rgba.unmap(stream);
        myBuffers.depth.unmap(stream);
        if (!drew) return true;

        /* hand-back: PBO -> texture inside the GPU, then one triangle with the reference's blend state */
        glBindBuffer(GL_PIXEL_UNPACK_BUFFER, myBuffers.rgbaPbo);
        glBindTexture(GL_TEXTURE_2D, myBuffers.tex);
        glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, ctx.width, ctx.height, GL_RGBA, GL_FLOAT, nullptr);
        glBindBuffer(GL_PIXEL_UNPACK_BUFFER, 0);
        r->pushBlendState();
        r->pushDepthState();
        r->blend(1);
        r->setBlendFunction(RE_SBLEND_ONE_MINUS_DST_ALPHA, RE_DBLEND_ONE);
        r->setAlphaBlendFunction(RE_SBLEND_ONE_MINUS_DST_ALPHA, RE_DBLEND_ONE);
        r->disableDepthTest();              /* the depth test has been applied per fragment by the kernels */
        r->disableDepthBufferWriting();
        glUseProgram(blitProgram());
        glActiveTexture(GL_TEXTURE0);
        glBindVertexArray(myBuffers.vao);
        glDrawArrays(GL_TRIANGLES, 0, 3);
        glBindVertexArray(0);
        glUseProgram(0);
        glBindTexture(GL_TEXTURE_2D, 0);
        r->popDepthState();
        r->popBlendState();
        return true;
    }

private:
    ViewportBuffers myBuffers;
};

class GSplatHipSceneHook : public DM_SceneHook
{
public:
    GSplatHipSceneHook(const char* name, int priority) : DM_SceneHook(name, priority, DM_HOOK_ALL_VIEWS) {}
    DM_SceneRenderHook* newSceneRender(DM_VPortAgent& vport, DM_SceneHookType, DM_SceneHookPolicy) override
    {
        return new GSplatHipSceneRenderHook(vport, DM_VIEWPORT_ALL);
    }
    void retireSceneRender(DM_VPortAgent&, DM_SceneRenderHook* hook) override { delete hook; }
};

/* the DSO entry point Houdini looks for (same name, same registration as the reference's) */
void newRenderHook(DM_RenderTable* table)
{
    table->registerSceneHook(new GSplatHipSceneHook("GSplat_RenderSceneHook", INT_MAX), DM_HOOK_BEAUTY, DM_HOOK_AFTER_NATIVE);
}
