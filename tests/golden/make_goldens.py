#!/usr/bin/env python3
"""Generate the golden fixtures that PIN the oracle: run the reference's OWN GLSL program
(read from /root/reference at run time, never copied into this repo) on a software GLES 3.0
rasteriser (SwiftShader, bundled in the `kaleido` wheel of the build container), with the
reference's texture layouts, blend state and instanced draw reproduced by hand, and store
inputs + the image it produced under tests/golden/*.npz.

Run in the BUILD CONTAINER only:   python tests/golden/make_goldens.py
(The GPU box has neither /root/reference nor needs it: tests only read the .npz files.)

What is reproduced from the reference's host side (paths relative to
/root/reference/gsplat_plugin):
  * texture layouts and sizes        src/GSplatRenderer.C:106-139 (closestSqrtPowerOf2 :155-163),
                                      pack loop :448-505 (4 RGBA32F texels / splat; 8 RGB16F texels / splat x2)
  * sorted-index texture (INT32)     src/GSplatRenderer.C:583-593
  * GL state                          :605-621  blend(ONE_MINUS_DST_ALPHA, ONE) colour+alpha, ADD; depth write off
  * uniforms                          :625-645
  * draw                              :647      6 vertices x N instances, triangles (0,1,2) (3,4,5)
The shader text is taken verbatim from shaders/GSplatShaderCoreLib.h and
shaders/GSplatShaderSource.h and made GLSL-ES-3.00 legal by the purely syntactic edits in
ES3_EDITS below (int literals in float context -> float literals; interface blocks -> plain
varyings; version/precision header).  No arithmetic is changed.  Every edit asserts that its
pattern is present, so a different reference revision fails loudly instead of silently.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

REF = "/root/reference/gsplat_plugin/shaders"
SWS = "/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/swiftshader"

# ----------------------------------------------------------------------------- shader text
def _raw_literal(text: str, name: str) -> str:
    m = re.search(r"const char\* const " + re.escape(name) + r'\s*=\s*R"glsl\((.*?)\)glsl";', text, re.S)
    assert m, f"raw string {name} not found in the reference"
    return m.group(1)


# (pattern, replacement, expected count) -- syntactic only
ES3_EDITS_CORE = [
    ("scale.x, 0, 0,", "scale.x, 0.0, 0.0,", 1),
    ("0, scale.y, 0,", "0.0, scale.y, 0.0,", 1),
    ("0, 0, scale.z", "0.0, 0.0, scale.z", 1),
    ("vec4(worldPos, 1)", "vec4(worldPos, 1.0)", 1),
    ("matrixP[0][0] / 2;", "matrixP[0][0] / 2.0;", 1),
    ("focal / viewPos.z, 0, -(focal * viewPos.x)", "focal / viewPos.z, 0.0, -(focal * viewPos.x)", 1),
    ("0, focal / viewPos.z, -(focal * viewPos.y)", "0.0, focal / viewPos.z, -(focal * viewPos.y)", 1),
    ("            0, 0, 0\n", "            0.0, 0.0, 0.0\n", 1),
]
ES3_EDITS_SH = [
    ("(2 * zz - xx - yy)", "(2.0 * zz - xx - yy)", 1),
    ("(3 * xx - yy)", "(3.0 * xx - yy)", 1),
    ("(4 * zz - xx - yy)", "(4.0 * zz - xx - yy)", 2),
    ("(2 * zz - 3 * xx - 3 * yy)", "(2.0 * zz - 3.0 * xx - 3.0 * yy)", 1),
    ("(xx - 3 * yy)", "(xx - 3.0 * yy)", 1),
    ("vec3(0,0,0)", "vec3(0.0,0.0,0.0)", 1),
]
ES3_EDITS_VS = [
    (re.compile(r"out parms\s*\{\s*vec4\s+pos;\s*vec3\s+color;\s*float\s+opacity;\s*\}\s*vsOut;"),
     "out vec4 v_pos; out vec3 v_color; out float v_opacity;", 1),
    ("vsOut.pos", "v_pos", None), ("vsOut.color", "v_color", None), ("vsOut.opacity", "v_opacity", None),
    ("vec2 quadPos = vec2(0,0);", "vec2 quadPos = vec2(0.0,0.0);", 1),
    ("quadPos = vec2(1,0);", "quadPos = vec2(1.0,0.0);", 1),
    ("quadPos = vec2(0,1);", "quadPos = vec2(0.0,1.0);", 1),
    ("quadPos = vec2(1,1);", "quadPos = vec2(1.0,1.0);", 1),
    ("quadPos = (quadPos * 2) - 1;", "quadPos = (quadPos * 2.0) - 1.0;", 1),
    ("quadPos *= 2;", "quadPos *= 2.0;", 1),
    ("mat4(1,0,0,0,0,-1,0,0,0,0,1,0,0,0,0,1)", "mat4(1.0,0.0,0.0,0.0,0.0,-1.0,0.0,0.0,0.0,0.0,1.0,0.0,0.0,0.0,0.0,1.0)", 1),
    ("vec4(centerViewPos, 1)", "vec4(centerViewPos, 1.0)", 1),
    ("centerClipPos.w <= 0)", "centerClipPos.w <= 0.0)", 1),
    ("gl_Position = vec4(0,0,0,0);", "gl_Position = vec4(0.0,0.0,0.0,0.0);", 1),
    ("vec4(quadPos, 0, 1)", "vec4(quadPos, 0.0, 1.0)", 1),
    ("* 2 / glH_ScreenSize", "* 2.0 / glH_ScreenSize", 1),
]
ES3_EDITS_FS = [
    (re.compile(r"in parms\s*\{\s*vec4\s+pos;\s*vec3\s+color;\s*float\s+opacity;\s*\}\s*fsIn;"),
     "in vec4 v_pos; in vec3 v_color; in float v_opacity;", 1),
    ("fsIn.pos", "v_pos", None), ("fsIn.color", "v_color", None), ("fsIn.opacity", "v_opacity", None),
]
ES3_EDITS_WIRE_VS = [
    (re.compile(r"out parms\s*\{\s*vec3\s+color;\s*\}\s*vsOut;"), "out vec3 v_color;", 1),
    ("vsOut.color", "v_color", None),
    ("vec2 quadPos = vec2(0,0);", "vec2 quadPos = vec2(0.0,0.0);", 1),
    ("quadPos = vec2(1,0);", "quadPos = vec2(1.0,0.0);", 1),
    ("quadPos = vec2(1,1);", "quadPos = vec2(1.0,1.0);", 1),
    ("quadPos = vec2(0,1);", "quadPos = vec2(0.0,1.0);", 1),
    ("quadPos = (quadPos * 2) - 1;", "quadPos = (quadPos * 2.0) - 1.0;", 1),
    ("quadPos *= 2;", "quadPos *= 2.0;", 1),
    ("mat4(1,0,0,0,0,-1,0,0,0,0,1,0,0,0,0,1)", "mat4(1.0,0.0,0.0,0.0,0.0,-1.0,0.0,0.0,0.0,0.0,1.0,0.0,0.0,0.0,0.0,1.0)", 1),
    ("vec4(centerWorldPos, 1)", "vec4(centerWorldPos, 1.0)", 2),   # one of the two sits in a comment
    ("* 2 / glH_ScreenSize", "* 2.0 / glH_ScreenSize", 1),
]
ES3_EDITS_WIRE_FS = [
    (re.compile(r"in parms\s*\{\s*vec3\s+color;\s*\}\s*fsIn;"), "in vec3 v_color;", 1),
    ("fsIn.color", "v_color", None),
]
ES3_HEADER = ("#version 300 es\nprecision highp float;\nprecision highp int;\n"
              "precision highp sampler2D;\nprecision highp isampler2D;\n")


def _apply(text: str, edits) -> str:
    for pat, rep, cnt in edits:
        if isinstance(pat, str):
            n = text.count(pat)
            assert n > 0 and (cnt is None or n == cnt), f"edit pattern {pat!r}: found {n}, expected {cnt}"
            text = text.replace(pat, rep)
        else:
            text, n = pat.subn(rep, text)
            assert n == cnt, f"edit pattern {pat.pattern!r}: found {n}, expected {cnt}"
    return text


def reference_shaders() -> tuple[str, str]:
    core_h = open(os.path.join(REF, "GSplatShaderCoreLib.h")).read()
    src_h = open(os.path.join(REF, "GSplatShaderSource.h")).read()
    core = _apply(_raw_literal(core_h, "GSplatCoreLib"), ES3_EDITS_CORE)
    sh = _apply(_raw_literal(core_h, "GSplatSphericalHarmonicsLib"), ES3_EDITS_SH)
    vs = _apply(_raw_literal(src_h, "_GSplatMainVertexShader"), ES3_EDITS_VS)
    fs = _apply(_raw_literal(src_h, "_GSplatMainFragmentShader"), ES3_EDITS_FS)
    # same assembly order as getFullShaderSrc("330", {core, sh, vs}) / {fs}  (GSplatShaderSource.h:9-15,291,315)
    return ES3_HEADER + core + sh + vs, ES3_HEADER + fs


def reference_wire_shaders() -> tuple[str, str]:
    core_h = open(os.path.join(REF, "GSplatShaderCoreLib.h")).read()
    src_h = open(os.path.join(REF, "GSplatShaderSource.h")).read()
    core = _apply(_raw_literal(core_h, "GSplatCoreLib"), ES3_EDITS_CORE)
    vs = _apply(_raw_literal(src_h, "_GSplatWireVertexShader"), ES3_EDITS_WIRE_VS)
    fs = _apply(_raw_literal(src_h, "_GSplatWireFragmentShader"), ES3_EDITS_WIRE_FS)
    return ES3_HEADER + core + vs, ES3_HEADER + fs     # getFullShaderSrc("330", {GSplatCoreLib, wire VS}) :89


# ----------------------------------------------------------------------------- EGL / GLES3
GL_VERTEX_SHADER, GL_FRAGMENT_SHADER, GL_COMPILE_STATUS, GL_LINK_STATUS = 0x8B31, 0x8B30, 0x8B81, 0x8B82
GL_TEXTURE_2D, GL_RGBA32F, GL_RGB16F, GL_R32I = 0x0DE1, 0x8814, 0x881B, 0x8235
GL_RGBA, GL_RGB, GL_RED_INTEGER, GL_FLOAT, GL_HALF_FLOAT, GL_INT = 0x1908, 0x1907, 0x8D94, 0x1406, 0x140B, 0x1404
GL_TEXTURE_MIN_FILTER, GL_TEXTURE_MAG_FILTER, GL_NEAREST = 0x2801, 0x2800, 0x2600
GL_TEXTURE_WRAP_S, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE = 0x2802, 0x2803, 0x812F
GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_FRAMEBUFFER_COMPLETE = 0x8D40, 0x8CE0, 0x8CD5
GL_BLEND, GL_ONE_MINUS_DST_ALPHA, GL_ONE, GL_FUNC_ADD = 0x0BE2, 0x0305, 1, 0x8006
GL_TRIANGLES, GL_COLOR_BUFFER_BIT, GL_TEXTURE0 = 4, 0x4000, 0x84C0
GL_UNPACK_ALIGNMENT, GL_PACK_ALIGNMENT, GL_DEPTH_TEST, GL_CULL_FACE = 0x0CF5, 0x0D05, 0x0B71, 0x0B44


class GLES:
    def __init__(self):
        self.gl = C.CDLL(os.path.join(SWS, "libGLESv2.so"), mode=C.RTLD_GLOBAL)
        self.egl = C.CDLL(os.path.join(SWS, "libEGL.so"), mode=C.RTLD_GLOBAL)
        egl = self.egl
        egl.eglGetDisplay.restype = C.c_void_p
        egl.eglGetDisplay.argtypes = [C.c_void_p]
        egl.eglCreatePbufferSurface.restype = C.c_void_p
        egl.eglCreateContext.restype = C.c_void_p
        dpy = egl.eglGetDisplay(None)
        assert dpy, "eglGetDisplay failed"
        major, minor = C.c_int(), C.c_int()
        assert egl.eglInitialize(C.c_void_p(dpy), C.byref(major), C.byref(minor))
        cfg_attr = (C.c_int * 13)(0x3033, 0x0001, 0x3040, 0x0040, 0x3024, 8, 0x3023, 8, 0x3022, 8, 0x3021, 8, 0x3038)
        cfg, ncfg = C.c_void_p(), C.c_int()
        assert egl.eglChooseConfig(C.c_void_p(dpy), cfg_attr, C.byref(cfg), 1, C.byref(ncfg)) and ncfg.value >= 1
        surf = egl.eglCreatePbufferSurface(C.c_void_p(dpy), cfg, (C.c_int * 5)(0x3057, 16, 0x3056, 16, 0x3038))
        assert surf
        assert egl.eglBindAPI(0x30A0)
        ctx = egl.eglCreateContext(C.c_void_p(dpy), cfg, None, (C.c_int * 3)(0x3098, 3, 0x3038))
        assert ctx, "no GLES3 context"
        assert egl.eglMakeCurrent(C.c_void_p(dpy), C.c_void_p(surf), C.c_void_p(surf), C.c_void_p(ctx))
        self.gl.glGetString.restype = C.c_char_p
        self.version = self.gl.glGetString(0x1F02).decode()
        self.renderer = self.gl.glGetString(0x1F01).decode()
        self.gl.glGetUniformLocation.argtypes = [C.c_uint, C.c_char_p]
        self.gl.glUniform2f.argtypes = [C.c_int, C.c_float, C.c_float]
        self.gl.glUniform3f.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
        self.gl.glClearColor.argtypes = [C.c_float] * 4

    def check(self, where=""):
        e = self.gl.glGetError()
        assert e == 0, f"GL error {hex(e)} {where}"

    def shader(self, kind, text):
        gl = self.gl
        s = gl.glCreateShader(kind)
        src = C.c_char_p(text.encode())
        gl.glShaderSource(s, 1, C.byref(src), None)
        gl.glCompileShader(s)
        ok = C.c_int()
        gl.glGetShaderiv(s, GL_COMPILE_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(8192)
            gl.glGetShaderInfoLog(s, 8192, None, log)
            raise RuntimeError("shader compile failed:\n" + log.value.decode())
        return s

    def program(self, vs_text, fs_text, tf_varyings=None):
        gl = self.gl
        p = gl.glCreateProgram()
        gl.glAttachShader(p, self.shader(GL_VERTEX_SHADER, vs_text))
        gl.glAttachShader(p, self.shader(GL_FRAGMENT_SHADER, fs_text))
        if tf_varyings:
            arr = (C.c_char_p * len(tf_varyings))(*[v.encode() for v in tf_varyings])
            gl.glTransformFeedbackVaryings(p, len(tf_varyings), arr, 0x8C8C)  # GL_INTERLEAVED_ATTRIBS
        gl.glLinkProgram(p)
        ok = C.c_int()
        gl.glGetProgramiv(p, GL_LINK_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(8192)
            gl.glGetProgramInfoLog(p, 8192, None, log)
            raise RuntimeError("program link failed:\n" + log.value.decode())
        return p

    def texture(self, unit, internal, w, h, fmt, typ, data: np.ndarray):
        gl = self.gl
        t = C.c_uint()
        gl.glGenTextures(1, C.byref(t))
        gl.glActiveTexture(GL_TEXTURE0 + unit)
        gl.glBindTexture(GL_TEXTURE_2D, t)
        gl.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)
        data = np.ascontiguousarray(data)
        gl.glTexImage2D(GL_TEXTURE_2D, 0, internal, w, h, 0, fmt, typ, C.c_void_p(data.ctypes.data))
        for pn, v in ((GL_TEXTURE_MIN_FILTER, GL_NEAREST), (GL_TEXTURE_MAG_FILTER, GL_NEAREST),
                      (GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE), (GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE)):
            gl.glTexParameteri(GL_TEXTURE_2D, pn, v)
        self.check("texture")
        return t


def closest_sqrt_pow2(n: int) -> int:  # src/GSplatRenderer.C:155-163
    if n <= 1:
        return 2
    return int(2 ** int(np.ceil(np.log2(np.float32(np.sqrt(np.float32(n)))))))


def h2f(bits: np.ndarray) -> np.ndarray:
    return bits.view(np.float16).astype(np.float32)


def render_reference_glsl(es: GLES, splats, cam, origin, perm, ss: int = 1, depth=None, window=None, capture=True):
    """One frame of the reference's main program on SwiftShader.

    ss (odd): the viewport/FBO is ss x larger than glH_ScreenSize while the uniform keeps the
    nominal size, and only the hi-res pixels whose centres coincide with nominal pixel centres
    (ss*i + ss//2) are kept.  Every shader input is unchanged, so each kept sample is exactly the
    fragment a nominal-resolution rasteriser would shade -- but SwiftShader's 4-bit sub-pixel
    vertex snapping (1/16 px) shrinks to 1/(16*ss) nominal px, i.e. ss=15 emulates the 8-bit
    sub-pixel precision of hardware rasterisers.
    depth: float32 [H,W] window depth (0..1) left by an opaque pass -> attached as the FBO's depth buffer, depth test
    ON, depth writes OFF (src/GSplatRenderer.C:595-610; the function is whatever the viewport had -- GL_LEQUAL).
    window = (x0, y0, w, h): only that window of the nominal frame is rasterised (the viewport is shifted so that the
    window lands in an FBO of its own size; every shader input is unchanged) -- full-size configs in affordable pieces.
    capture = False skips the transform-feedback capture (large scenes).
    Returns (image float32 [h,w,4] row 0 = bottom, vertex-stage capture float32 [n,6,12]:
    gl_Position xyzw, v_pos xyzw, v_color rgb, v_opacity -- in INSTANCE (sorted) order, or None)."""
    assert ss % 2 == 1
    gl = es.gl
    n = splats.n
    # (the reference skips the index upload when N is exactly 4^k -- SURVEY Q2; this harness always uploads)
    vs, fs = reference_shaders()
    if window is not None:
        # HARNESS edit (window rendering only): the viewport transform of a sub-window, applied to the finished clip
        # position -- x' = (x - cx*w) * W/ww is what a viewport of the full frame's size placed at -x0 would do, without
        # needing a viewport larger than the rasteriser's limit.  Nothing the reference computes is touched.
        marker = "gl_Position = out_vertex;"
        assert vs.count(marker) == 1
        vs = vs.replace(marker, marker + "\n gl_Position.xy = gl_Position.xy * harness_win.xy + harness_win.zw * gl_Position.w;")
        vs = vs.replace(ES3_HEADER, ES3_HEADER + "uniform vec4 harness_win;\n", 1)
    prog = es.program(vs, fs, tf_varyings=["gl_Position", "v_pos", "v_color", "v_opacity"])
    gl.glUseProgram(prog)
    origin = np.asarray(origin, np.float32)

    # ---- textures exactly as generateRenderGeometry packs them (:448-505)
    dim_a = closest_sqrt_pow2(n * 4)
    a = np.zeros((dim_a * dim_a, 4), np.float32)
    a[0:4 * n:4, :3] = splats.P - origin          # fl32(P - origin), pad 0
    a[1:4 * n:4, :3] = h2f(splats.Cd)
    a[1:4 * n:4, 3] = splats.alpha
    a[2:4 * n:4, :3] = h2f(splats.scale)
    a[3:4 * n:4, :] = h2f(splats.orient)
    es.texture(1, GL_RGBA32F, dim_a, dim_a, GL_RGBA, GL_FLOAT, a)
    order = cam.sh_order if splats.has_sh else 0
    dim_sh = 0
    if splats.has_sh:
        dim_sh = closest_sqrt_pow2(n * 8)
        t1 = np.zeros((dim_sh * dim_sh, 3), np.uint16)
        t2 = np.zeros((dim_sh * dim_sh, 3), np.uint16)
        sh = np.stack([splats.shx[:, :15], splats.shy[:, :15], splats.shz[:, :15]], axis=2)  # [n,15,3]
        for j in range(8):
            t1[j:8 * n:8] = sh[:, j]
        for j in range(8, 15):
            t2[j - 8:8 * n:8] = sh[:, j]
        es.texture(2, GL_RGB16F, dim_sh, dim_sh, GL_RGB, GL_HALF_FLOAT, t1)
        es.texture(3, GL_RGB16F, dim_sh, dim_sh, GL_RGB, GL_HALF_FLOAT, t2)
    dim_i = closest_sqrt_pow2(n)
    idx = np.zeros(dim_i * dim_i, np.int32)
    idx[:n] = perm
    es.texture(0, GL_R32I, dim_i, dim_i, GL_RED_INTEGER, GL_INT, idx)

    # ---- uniforms (:625-645 + Houdini built-ins consumed by the shader)
    def loc(name):
        return gl.glGetUniformLocation(prog, name.encode())

    def u1i(name, v):
        if loc(name) >= 0:
            gl.glUniform1i(loc(name), int(v))

    def umat(name, m):
        if loc(name) >= 0:
            arr = np.ascontiguousarray(m, np.float32)
            gl.glUniformMatrix4fv(loc(name), 1, 0, C.c_void_p(arr.ctypes.data))

    u1i("GSplatCount", n)
    u1i("GSplatVertexCount", 6)
    gl.glUniform3f(loc("GSplatOrigin"), float(origin[0]), float(origin[1]), float(origin[2]))
    u1i("GSplatShOrder", order)
    u1i("GSplatZOrderTexDim", dim_i)
    u1i("GSplatZOrderIntegerTexSampler", 0)
    u1i("GSplatPosColorAlphaScaleOrientTexDim", dim_a)
    u1i("GSplatPosColorAlphaScaleOrientTexSampler", 1)
    if order > 0:
        gl.glUniform3f(loc("WorldSpaceCameraPos"), *[float(x) for x in cam.cam_pos])
        u1i("GSplatShDeg1And2TexDim", dim_sh)
        u1i("GSplatShDeg1And2TexSampler", 2)
        if order > 2:
            u1i("GSplatShDeg3TexDim", dim_sh)
            u1i("GSplatShDeg3TexSampler", 3)
    umat("glH_ObjViewMatrix", cam.obj_view)
    umat("glH_ObjectMatrix", cam.object)
    umat("glH_InvObjectMatrix", cam.inv_object)
    umat("glH_ViewMatrix", cam.view)
    umat("glH_ProjectMatrix", cam.proj)
    gl.glUniform2f(loc("glH_ScreenSize"), float(cam.width), float(cam.height))
    es.check("uniforms")

    # ---- float render target cleared to transparent black
    wx0, wy0, ww, wh = window if window is not None else (0, 0, cam.width, cam.height)
    fb_tex = C.c_uint()
    gl.glGenTextures(1, C.byref(fb_tex))
    gl.glActiveTexture(GL_TEXTURE0 + 7)
    gl.glBindTexture(GL_TEXTURE_2D, fb_tex)
    gl.glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, ww * ss, wh * ss, 0, GL_RGBA, GL_FLOAT, None)
    fbo = C.c_uint()
    gl.glGenFramebuffers(1, C.byref(fbo))
    gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
    gl.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, fb_tex, 0)
    dtex = C.c_uint()
    if depth is not None:
        # the opaque pass's depth buffer: every nominal pixel replicated ss x ss
        d = np.ascontiguousarray(np.asarray(depth, np.float32)[wy0:wy0 + wh, wx0:wx0 + ww].repeat(ss, axis=0).repeat(ss, axis=1))
        gl.glGenTextures(1, C.byref(dtex))
        gl.glActiveTexture(GL_TEXTURE0 + 6)
        gl.glBindTexture(GL_TEXTURE_2D, dtex)
        gl.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)
        gl.glTexImage2D(GL_TEXTURE_2D, 0, 0x8CAC, ww * ss, wh * ss, 0, 0x1902, GL_FLOAT, C.c_void_p(d.ctypes.data))  # DEPTH_COMPONENT32F
        for pn, v in ((GL_TEXTURE_MIN_FILTER, GL_NEAREST), (GL_TEXTURE_MAG_FILTER, GL_NEAREST)):
            gl.glTexParameteri(GL_TEXTURE_2D, pn, v)
        gl.glFramebufferTexture2D(GL_FRAMEBUFFER, 0x8D00, GL_TEXTURE_2D, dtex, 0)                                     # DEPTH_ATTACHMENT
        es.check("depth attachment")
    assert gl.glCheckFramebufferStatus(GL_FRAMEBUFFER) == GL_FRAMEBUFFER_COMPLETE, "RGBA32F FBO incomplete"
    gl.glViewport(0, 0, ww * ss, wh * ss)
    if window is not None:
        sx, sy = cam.width / ww, cam.height / wh
        cx, cy = (2.0 * wx0 + ww) / cam.width - 1.0, (2.0 * wy0 + wh) / cam.height - 1.0
        gl.glUniform4f.argtypes = [C.c_int] + [C.c_float] * 4
        gl.glUniform4f(gl.glGetUniformLocation(prog, b"harness_win"), sx, sy, -cx * sx, -cy * sy)
    gl.glClearColor(0.0, 0.0, 0.0, 0.0)
    gl.glClear(GL_COLOR_BUFFER_BIT)

    # ---- GL state (:605-621): depth test against the opaque pass (if any), depth writes off, no culling
    if depth is not None:
        gl.glEnable(GL_DEPTH_TEST)
        gl.glDepthFunc(0x0203)        # GL_LEQUAL
        gl.glDepthMask(0)
    else:
        gl.glDisable(GL_DEPTH_TEST)
    gl.glDisable(GL_CULL_FACE)
    gl.glEnable(GL_BLEND)
    gl.glBlendFuncSeparate(GL_ONE_MINUS_DST_ALPHA, GL_ONE, GL_ONE_MINUS_DST_ALPHA, GL_ONE)
    gl.glBlendEquation(GL_FUNC_ADD)
    vao = C.c_uint()
    gl.glGenVertexArrays(1, C.byref(vao))
    gl.glBindVertexArray(vao)
    # vertex-stage capture (transform feedback), 12 floats per vertex
    vs_out = None
    if capture:
        tfb = C.c_uint()
        gl.glGenBuffers(1, C.byref(tfb))
        GL_TFB = 0x8C8E
        gl.glBindBuffer(GL_TFB, tfb)
        nbytes = n * 6 * 12 * 4
        gl.glBufferData(GL_TFB, C.c_ssize_t(nbytes), None, 0x88E9)  # GL_DYNAMIC_READ
        gl.glBindBufferBase(GL_TFB, 0, tfb)
        gl.glBeginTransformFeedback(GL_TRIANGLES)
    gl.glDrawArraysInstanced(GL_TRIANGLES, 0, 6, n)  # drawInstanced(..., splatCount) :647
    if capture:
        gl.glEndTransformFeedback()
    gl.glFinish()
    es.check("draw")
    if capture:
        gl.glMapBufferRange.restype = C.c_void_p
        gl.glMapBufferRange.argtypes = [C.c_uint, C.c_ssize_t, C.c_ssize_t, C.c_uint]
        ptr = gl.glMapBufferRange(GL_TFB, 0, nbytes, 0x0001)  # GL_MAP_READ_BIT
        assert ptr, "transform feedback buffer map failed"
        vs_out = np.frombuffer((C.c_float * (n * 72)).from_address(ptr), dtype=np.float32).reshape(n, 6, 12).copy()
        gl.glUnmapBuffer(GL_TFB)
        gl.glDeleteBuffers(1, C.byref(tfb))
    hi = np.zeros((wh * ss, ww * ss, 4), np.float32)
    gl.glPixelStorei(GL_PACK_ALIGNMENT, 1)
    gl.glReadPixels(0, 0, ww * ss, wh * ss, GL_RGBA, GL_FLOAT, C.c_void_p(hi.ctypes.data))
    es.check("readpixels")
    gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)
    if depth is not None:
        gl.glDepthMask(1)
        gl.glDisable(GL_DEPTH_TEST)
        gl.glDeleteTextures(1, C.byref(dtex))
    gl.glDeleteFramebuffers(1, C.byref(fbo))
    gl.glDeleteTextures(1, C.byref(fb_tex))
    out = np.ascontiguousarray(hi[ss // 2::ss, ss // 2::ss])
    assert out.shape == (wh, ww, 4)
    return out, vs_out


def render_reference_wire(es: GLES, splats, cam) -> np.ndarray:
    """the reference's WIRE program: 8-vertex line list per splat (src/GR_GSplat.C:374-421), depth test on"""
    gl = es.gl
    n = splats.n
    vs, fs = reference_wire_shaders()
    prog = es.program(vs, fs)
    gl.glUseProgram(prog)
    gl.glGetAttribLocation.argtypes = [C.c_uint, C.c_char_p]
    vao = C.c_uint()
    gl.glGenVertexArrays(1, C.byref(vao))
    gl.glBindVertexArray(vao)
    attrs = {"P": np.repeat(splats.P.astype(np.float32), 8, axis=0),
             "Cd": np.repeat(h2f(splats.Cd), 8, axis=0),
             "scale": np.repeat(h2f(splats.scale), 8, axis=0),
             "orient": np.repeat(h2f(splats.orient), 8, axis=0)}
    keep = []
    for name, arr in attrs.items():
        loc = gl.glGetAttribLocation(prog, name.encode())
        if loc < 0:
            continue
        arr = np.ascontiguousarray(arr, np.float32)
        keep.append(arr)
        buf = C.c_uint()
        gl.glGenBuffers(1, C.byref(buf))
        gl.glBindBuffer(0x8892, buf)                                     # GL_ARRAY_BUFFER
        gl.glBufferData(0x8892, C.c_ssize_t(arr.nbytes), C.c_void_p(arr.ctypes.data), 0x88E4)  # STATIC_DRAW
        gl.glEnableVertexAttribArray(loc)
        gl.glVertexAttribPointer(loc, arr.shape[1], GL_FLOAT, 0, 0, None)

    def loc(name):
        return gl.glGetUniformLocation(prog, name.encode())

    for name, m in (("glH_ObjViewMatrix", cam.obj_view), ("glH_ViewMatrix", cam.view), ("glH_ProjectMatrix", cam.proj)):
        if loc(name) >= 0:
            arr = np.ascontiguousarray(m, np.float32)
            gl.glUniformMatrix4fv(loc(name), 1, 0, C.c_void_p(arr.ctypes.data))
    gl.glUniform2f(loc("glH_ScreenSize"), float(cam.width), float(cam.height))
    fb_tex = C.c_uint()
    gl.glGenTextures(1, C.byref(fb_tex))
    gl.glActiveTexture(GL_TEXTURE0 + 7)
    gl.glBindTexture(GL_TEXTURE_2D, fb_tex)
    gl.glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, cam.width, cam.height, 0, GL_RGBA, GL_FLOAT, None)
    rb = C.c_uint()
    gl.glGenRenderbuffers(1, C.byref(rb))
    gl.glBindRenderbuffer(0x8D41, rb)
    gl.glRenderbufferStorage(0x8D41, 0x8CAC, cam.width, cam.height)     # DEPTH_COMPONENT32F
    fbo = C.c_uint()
    gl.glGenFramebuffers(1, C.byref(fbo))
    gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
    gl.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, fb_tex, 0)
    gl.glFramebufferRenderbuffer(GL_FRAMEBUFFER, 0x8D00, 0x8D41, rb)      # DEPTH_ATTACHMENT
    assert gl.glCheckFramebufferStatus(GL_FRAMEBUFFER) == GL_FRAMEBUFFER_COMPLETE
    gl.glViewport(0, 0, cam.width, cam.height)
    gl.glClearColor(0.0, 0.0, 0.0, 0.0)
    gl.glClearDepthf.argtypes = [C.c_float]
    gl.glClearDepthf(1.0)
    gl.glClear(GL_COLOR_BUFFER_BIT | 0x100)
    gl.glDisable(GL_BLEND)
    gl.glEnable(GL_DEPTH_TEST)
    gl.glDepthFunc(0x0201)                                               # GL_LESS
    gl.glDrawArrays(1, 0, 8 * n)                                         # GL_LINES
    gl.glFinish()
    es.check("wire draw")
    out = np.zeros((cam.height, cam.width, 4), np.float32)
    gl.glPixelStorei(GL_PACK_ALIGNMENT, 1)
    gl.glReadPixels(0, 0, cam.width, cam.height, GL_RGBA, GL_FLOAT, C.c_void_p(out.ctypes.data))
    es.check("wire readpixels")
    gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)
    gl.glDisable(GL_DEPTH_TEST)
    gl.glBindVertexArray(0)
    return out


# ----------------------------------------------------------------------------- cases
def cases(pkg):
    sc, cm = pkg.scenes, pkg.camera
    out = []
    # G1: one isolated splat near the view centre, order 0 (KAT: centre alpha = opacity * exp(-|q|^2))
    # (5 points, 4 of them far behind the camera: with a single point the reference's 2x2 attribute
    #  texture is too narrow for its own iuv+ivec2(2,0) fetches)
    s = sc.make_scene(5, seed=3, sh=False)
    cam1 = cm.make_camera(64, 64, sh_order=0, frame=0)
    s.P[:] = np.float32(cam1.cam_pos) * np.float32(3.0)
    s.P[2] = np.float32([0.03, -0.02, 0.1])
    s.scale[2] = sc.f16bits(np.float32([0.05, 0.02, 0.03]))
    s.alpha[:] = 0.8
    out.append(("g1_single", s, cam1, (0, 0, 0), 15))
    # G2: small SH3 cloud
    out.append(("g2_sh3_200", sc.make_scene(200, seed=12, sh=True, log_scale_range=(-4.0, -2.5)),
                cm.make_camera(128, 96, sh_order=3, frame=0), (0, 0, 0), 15))
    # G3: 2000 splats SH3, orbit frame 2
    out.append(("g3_sh3_2000", sc.make_scene(2000, seed=13, sh=True, log_scale_range=(-4.5, -3.0)),
                cm.make_camera(256, 192, sh_order=3, frame=2), (0, 0, 0), 9))
    # G4: camera INSIDE the cloud (w<=0 and near-plane culls), SH order 1, non-zero GSplatOrigin
    s = sc.make_scene(1500, seed=14, sh=True, log_scale_range=(-4.5, -3.0))
    out.append(("g4_inside_sh1", s, cm.make_camera(200, 150, sh_order=1, frame=5, distance=0.4, near=0.05),
                tuple(float(x) for x in s.barycenter()), 9))
    # G5: SH order 2, opaque-ish splats (occlusion / blend order)
    s = sc.make_scene(1200, seed=15, sh=True, log_scale_range=(-3.5, -2.5))
    s.alpha[:] = np.clip(s.alpha * 1.6, 0, 1)
    out.append(("g5_sh2_opaque", s, cm.make_camera(160, 160, sh_order=2, frame=9), (0, 0, 0), 15))
    # G6: object matrix != identity (rotation+non-uniform scale) -- exercises O and O^-1 paths
    ang = 0.6
    obj = np.array([[np.cos(ang) * 1.2, -np.sin(ang), 0, 0.1], [np.sin(ang) * 1.2, np.cos(ang), 0, -0.05],
                    [0, 0, 0.8, 0.02], [0, 0, 0, 1]], dtype=np.float64)
    out.append(("g6_object_xform", sc.make_scene(800, seed=16, sh=True, log_scale_range=(-4.0, -2.8)),
                cm.make_camera(160, 120, sh_order=3, frame=1, object_matrix=obj), (0, 0, 0), 15))
    # G7: a slice of BASELINE config 1 (anisotropic, SH degree 3, the C2 generator settings) at 640x360
    s, cfg = sc.make_config("C2", 20000)
    out.append(("g7_aniso_sh3_20k", s, cm.make_camera(640, 360, sh_order=3, frame=4), (0, 0, 0), 9))
    # C1: BASELINE config 0 -- 10k isotropic splats, SH degree 0, 512x512
    s, cfg = sc.make_config("C1")
    out.append(("c1_10k_iso", s, cm.make_camera(cfg["width"], cfg["height"], sh_order=0, frame=0), (0, 0, 0), 5))
    # ---- projections a Houdini viewport really produces (round 4).  The reference's covariance code assumes a
    # symmetric perspective frustum (shaders/GSplatShaderCoreLib.h:44-58: focal and both clamp limits from P00 alone, a division by
    # view z whatever the projection) and sandwiches the projection between two y flips (GSplatShaderSource.h:203-207, :281).  What
    # that does under other projections is what the contract has to do too:
    # G9: ORTHOGRAPHIC (Top / Front / Right views): clip w == 1, yet J still divides by view z
    s = sc.make_scene(1500, seed=19, sh=True, log_scale_range=(-2.6, -1.3))
    hw = 1.35
    out.append(("g9_ortho", s, cm.make_camera(192, 144, sh_order=3, frame=3,
                                             proj_matrix=cm.orthographic(-hw, hw, -hw * 144 / 192, hw * 144 / 192, 0.05, 60.0)), (0, 0, 0), 9))
    # G10: OFF-CENTRE frustum (a cropped / zoomed viewport: P02, P12 != 0).  The flip-Y sandwich is the identity only for P12 == 0:
    # here the vertical offset term comes out with its sign flipped (Q4), and J ignores both offsets
    near = 0.05
    a = near / 2.41421
    c = a * 150.0 / 200.0
    out.append(("g10_offcentre", sc.make_scene(1500, seed=20, sh=True, log_scale_range=(-4.2, -2.8)),
                cm.make_camera(200, 150, sh_order=3, frame=7, proj_matrix=cm.frustum(-0.6 * a, 1.4 * a, -1.3 * c, 0.7 * c, near, 1.0e4)), (0, 0, 0), 9))
    # G11: a wide lens in PORTRAIT format: limY = 1.3 / P00 is the HORIZONTAL limit (Q3), far inside the vertical field here, so
    # splats near the top and bottom edges have their view position clamped
    out.append(("g11_fov_aspect", sc.make_scene(1500, seed=21, sh=True, log_scale_range=(-4.0, -2.7), radius=1.6),
                cm.make_camera(120, 200, sh_order=2, frame=11, p00=1.1, distance=2.6), (0, 0, 0), 9))
    return out


def make_depth_golden(es, pkg, oracle, name="g8_depth_tested"):
    sc, cm = pkg.scenes, pkg.camera
    splats = sc.make_scene(3000, seed=18, sh=True, log_scale_range=(-4.2, -2.8))
    cam = cm.make_camera(256, 192, sh_order=2, frame=6)
    origin = (0.0, 0.0, 0.0)
    rec = oracle.preprocess(splats, cam, origin)
    perm = oracle.argsort(rec)
    z = np.sort(rec["zwin"][rec["visible"] == 1])
    def gap_mid(lo, hi):   # the middle of the widest gap between consecutive splat depths in z[lo:hi]: far from any tie
        k = lo + int(np.argmax(np.diff(z[lo:hi])))
        return float(0.5 * (z[k] + z[k + 1]))
    zmid = gap_mid(len(z) * 45 // 100, len(z) * 55 // 100)
    yy, xx = np.mgrid[0:cam.height, 0:cam.width]
    depth = np.full((cam.height, cam.width), 1.0, np.float32)
    depth[(xx // 32 + yy // 32) % 2 == 0] = zmid                       # a checkerboard "wall" through the middle of the cloud
    depth[:12] = 0.0                                                   # a strip where everything is hidden
    zq = gap_mid(len(z) * 20 // 100, len(z) * 30 // 100)
    depth[100:140, 60:200] = zq                                        # a nearer panel
    ss = 9
    img_gl, vs_sorted = render_reference_glsl(es, splats, cam, origin, perm, ss, depth=depth)
    vs_out = np.empty_like(vs_sorted)
    vs_out[perm] = vs_sorted
    img_or = oracle.render_depth(splats, cam, depth, origin)
    err = np.abs(img_gl - img_or)
    stats = dict(name=name, n=splats.n, size=(cam.width, cam.height), ss=ss, max_err=float(err.max()), mean_err=float(err.mean()),
                 frac_pixels_within_1e3=float((err.max(axis=2) <= 1e-3).mean()),
                 differs_from_untested=float(np.abs(img_gl - oracle.render(splats, cam, origin)).max()))
    print(stats)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), P=splats.P, Cd=splats.Cd, alpha=splats.alpha, scale=splats.scale,
                        orient=splats.orient, shx=splats.shx, shy=splats.shy, shz=splats.shz,
                        origin=np.asarray(origin, np.float32), image_reference_glsl=img_gl, supersample=np.int32(ss), vs_out=vs_out,
                        depth=depth, oracle_sha256=np.frombuffer(hashlib.sha256(img_or.tobytes()).digest(), dtype=np.uint8),
                        cam_obj_view=cam.obj_view, cam_object=cam.object, cam_inv_object=cam.inv_object, cam_view=cam.view,
                        cam_proj=cam.proj, cam_pos=cam.cam_pos, cam_whs=np.int32([cam.width, cam.height, cam.sh_order]))
    return stats


def make_c4_band_golden(es, pkg, oracle, name="c4_band_1080p", rows=(520, 584), ss=9, n_override=None):
    """BASELINE C4 (6M splats, SH 3, 1920x1080) through the reference GLSL: a full-width band of tile rows, rasterised
    in column blocks (SwiftShader's texture limit) with the viewport shifted, vertex snapping refined by ss."""
    import time
    sc, cm = pkg.scenes, pkg.camera
    splats, cfg = sc.make_config("C4", n_override)
    cam = cm.make_camera(cfg["width"], cfg["height"], sh_order=3, frame=0)
    origin = (0.0, 0.0, 0.0)
    rec = oracle.preprocess(splats, cam, origin)
    perm = oracle.argsort(rec)
    y0, y1 = rows
    band = np.zeros((y1 - y0, cam.width, 4), np.float32)
    block = 8192 // ss // 16 * 16
    for x0 in range(0, cam.width, block):
        w = min(block, cam.width - x0)
        t0 = time.time()
        img, _ = render_reference_glsl(es, splats, cam, origin, perm, ss, window=(x0, y0, w, y1 - y0), capture=False)
        band[:, x0:x0 + w] = img
        print("   columns", x0, x0 + w, "%.0f s" % (time.time() - t0), flush=True)
    img_or = oracle.render(splats, cam, origin, threads=oracle.max_threads())[y0:y1]
    err = np.abs(band - img_or)
    stats = dict(name=name, n=splats.n, size=(cam.width, cam.height), rows=rows, ss=ss, max_err=float(err.max()),
                 mean_err=float(err.mean()), frac_pixels_within_1e3=float((err.max(axis=2) <= 1e-3).mean()))
    print(stats)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), config="C4", n=np.int64(splats.n), rows=np.int32(rows), frame=np.int32(0),
                        band_reference_glsl=band, supersample=np.int32(ss),
                        oracle_band_sha256=np.frombuffer(hashlib.sha256(np.ascontiguousarray(img_or).tobytes()).digest(), dtype=np.uint8))
    return stats


def refresh_oracle_sha():
    """The arithmetic contract changed (oracle + kernels together): keep the reference-GLSL images and captures,
    re-check the new oracle against them with the acceptance rule of tests/helpers.py, re-pin its checksum."""
    sys.path.insert(0, os.path.dirname(HERE))
    import helpers
    oracle = ge.load_oracle()
    for name in helpers.golden_names():
        path = os.path.join(HERE, name + ".npz")
        d, s, c = helpers.load_golden(name)
        img = helpers.oracle_render_golden(oracle, d, s, c)
        print(name, helpers.check_against_golden(img, d["image_reference_glsl"]))
        arrays = {k: d[k] for k in d.files}
        arrays["oracle_sha256"] = np.frombuffer(hashlib.sha256(img.tobytes()).digest(), dtype=np.uint8)
        np.savez_compressed(path, **arrays)
    # the band of the full BASELINE C4 frame (the fixture holds only the reference image; inputs come from the seeded generator)
    path = os.path.join(HERE, "c4_band_1080p.npz")
    if os.path.exists(path):
        pkg = ge.load_package()
        d = np.load(path)
        splats, cfg = pkg.scenes.make_config("C4")
        cam = pkg.camera.make_camera(cfg["width"], cfg["height"], sh_order=3, frame=int(d["frame"]))
        y0, y1 = [int(v) for v in d["rows"]]
        band = oracle.render_rows(splats, cam, y0, y1)
        print("c4_band_1080p", helpers.check_against_golden(band, d["band_reference_glsl"]))
        arrays = {k: d[k] for k in d.files}
        arrays["oracle_band_sha256"] = np.frombuffer(hashlib.sha256(band.tobytes()).digest(), dtype=np.uint8)
        np.savez_compressed(path, **arrays)


def main():
    if "--refresh-oracle-sha" in sys.argv:
        return refresh_oracle_sha()
    pkg = ge.load_package()
    oracle = ge.load_oracle()
    es = GLES()
    print("GL:", es.version, "|", es.renderer)
    summary = []
    only = [a for a in sys.argv[1:] if not a.startswith("-")]   # optional: regenerate just these cases
    for name, splats, cam, origin, ss in cases(pkg):
        if only and name not in only:
            continue
        rec = oracle.preprocess(splats, cam, origin)
        perm = oracle.argsort(rec)  # (distance^2, index) ascending = the order the reference's argsort would upload
        img_gl, vs_sorted = render_reference_glsl(es, splats, cam, origin, perm, ss)
        img_nominal, _ = render_reference_glsl(es, splats, cam, origin, perm, 1)
        vs_out = np.empty_like(vs_sorted)
        vs_out[perm] = vs_sorted            # back to splat order
        img_or = oracle.render(splats, cam, origin)
        e1 = np.abs(img_nominal - img_or)
        print("   nominal-resolution raster (4-bit sub-pixel): max", float(e1.max()), "mean", float(e1.mean()),
              "frac<=1e-3", float((e1.max(axis=2) <= 1e-3).mean()))
        err = np.abs(img_gl - img_or)
        frac_ok = float((err.max(axis=2) <= 1e-3).mean())
        stats = dict(name=name, n=splats.n, size=(cam.width, cam.height), ss=ss, max_err=float(err.max()),
                     mean_err=float(err.mean()),
                     frac_pixels_within_1e3=frac_ok, p999=float(np.quantile(err, 0.999)),
                     covered=float((img_or[..., 3] > 0).mean()))
        print(stats)
        summary.append(stats)
        arrays = dict(P=splats.P, Cd=splats.Cd, alpha=splats.alpha, scale=splats.scale, orient=splats.orient,
                      origin=np.asarray(origin, np.float32), image_reference_glsl=img_gl,
                      image_reference_glsl_nominal=img_nominal.astype(np.float16), supersample=np.int32(ss),
                      vs_out=vs_out,
                      oracle_sha256=np.frombuffer(hashlib.sha256(img_or.tobytes()).digest(), dtype=np.uint8),
                      cam_obj_view=cam.obj_view, cam_object=cam.object, cam_inv_object=cam.inv_object,
                      cam_view=cam.view, cam_proj=cam.proj, cam_pos=cam.cam_pos,
                      cam_whs=np.int32([cam.width, cam.height, cam.sh_order]))
        if splats.has_sh:
            arrays.update(shx=splats.shx, shy=splats.shy, shz=splats.shz)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    # ---- G8: depth-tested against what an opaque pass left -- depth test on (LEQUAL), depth writes off,
    #      ONE depth per quad (all corners carry the centre's z and w, shaders/GSplatShaderSource.h:277-279)
    if not only or "g8_depth_tested" in only:
        summary.append(make_depth_golden(es, pkg, oracle))
    # ---- C4 at full resolution: a band of the 1920x1080 frame of the 6M-splat BASELINE scene (inputs are the seeded
    #      generator's, not stored)
    if "c4_band_1080p" in only:
        summary.append(make_c4_band_golden(es, pkg, oracle))
    if only and "w1_wire" not in only:
        return summary
    # ---- wireframe overlay: the reference's wire program as a LOOSE golden -- GL's diamond-exit
    #      line rule and the oracle's rule agree up to one pixel, not pixel for pixel
    sc, cm = pkg.scenes, pkg.camera
    ws = sc.make_scene(150, seed=17, sh=False, log_scale_range=(-3.6, -2.4))
    wcam = cm.make_camera(200, 150, sh_order=0, frame=3)
    wire_gl = render_reference_wire(es, ws, wcam)
    wire_or = oracle.render_wire(ws, wcam)
    g, o = wire_gl[..., 3] > 0, wire_or[..., 3] > 0

    def dil(m):
        out = m.copy()
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                out |= np.roll(np.roll(m, dy, 0), dx, 1)
        return out
    print("wire: GL px", int(g.sum()), "oracle px", int(o.sum()), "GL within 1px of oracle", float((g & dil(o)).sum() / g.sum()),
          "oracle within 1px of GL", float((o & dil(g)).sum() / o.sum()), "identical px", float((g & o).sum() / g.sum()))
    np.savez_compressed(os.path.join(HERE, "w1_wire.npz"), P=ws.P, Cd=ws.Cd, alpha=ws.alpha, scale=ws.scale,
                        orient=ws.orient, origin=np.zeros(3, np.float32), wire_reference_glsl=wire_gl.astype(np.float16),
                        cam_obj_view=wcam.obj_view, cam_object=wcam.object, cam_inv_object=wcam.inv_object,
                        cam_view=wcam.view, cam_proj=wcam.proj, cam_pos=wcam.cam_pos,
                        cam_whs=np.int32([wcam.width, wcam.height, wcam.sh_order]))
    return summary


if __name__ == "__main__":
    main()
