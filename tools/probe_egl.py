#!/usr/bin/env python3
"""Is there a way to get an OpenGL context on the AMD GPU of this machine, headless?   python tools/probe_egl.py

The hand-back of the drop-in (hdk/DM_GSplatHook_hip.C: GL pixel buffers mapped into HIP, include/gsplat_gl_interop.h) needs a
current GL >= 3.3 context on the SAME GPU the HIP context uses; the reference blends straight into Houdini's bound framebuffer
(/root/reference/gsplat_plugin/src/GSplatRenderer.C:605-657).  This probe walks every door a headless process has and says which
one closes where -- its output (profiles/egl_probe_*.txt) is the evidence behind INTEGRATION.md section 3, instead of an assertion:

  1. a system libEGL (libEGL.so.1): eglQueryDevicesEXT -> every EGLDevice with its extension string and DRM node ->
     eglGetPlatformDisplayEXT(EGL_PLATFORM_DEVICE_EXT) -> eglInitialize -> a desktop-GL >= 3.3 core context, made current
     surfaceless -> GL_VENDOR / GL_RENDERER;
  2. the render nodes (/dev/dri/renderD*) a GBM / surfaceless platform would need, and libgbm;
  3. GLX (libGL.so.1 + libGLX_mesa): needs an X server -- $DISPLAY;
  4. what HIP offers on its side: hipGraphicsGLRegisterBuffer / hipGLGetDevices in libamdhip64.
Exit code 0 = a GL context on an AMD device came up (the interop test can run), 1 = no door opens.  No dependency beyond ctypes."""
import ctypes as C
import ctypes.util
import glob
import os
import sys

EGL_EXTENSIONS, EGL_VENDOR, EGL_VERSION = 0x3055, 0x3053, 0x3054
EGL_PLATFORM_DEVICE_EXT = 0x313F
EGL_DRM_DEVICE_FILE_EXT, EGL_DRM_RENDER_NODE_FILE_EXT = 0x3233, 0x3377
EGL_OPENGL_API = 0x30A2
EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_BIT, EGL_NONE = 0x3033, 0x0001, 0x3040, 0x0008, 0x3038
EGL_CONTEXT_MAJOR_VERSION, EGL_CONTEXT_MINOR_VERSION = 0x3098, 0x30FB
EGL_CONTEXT_OPENGL_PROFILE_MASK, EGL_CONTEXT_OPENGL_CORE_PROFILE_BIT = 0x30FD, 0x00000001
GL_VENDOR, GL_RENDERER, GL_VERSION = 0x1F00, 0x1F01, 0x1F02


def say(*a):
    print(*a)
    sys.stdout.flush()


def try_egl() -> bool:
    name = ctypes.util.find_library("EGL")
    cands = [c for c in (name, "libEGL.so.1", "libEGL.so") if c]
    egl = None
    for c in cands:
        try:
            egl = C.CDLL(c)
            say(f"[egl] loaded {c}")
            break
        except OSError as e:
            say(f"[egl] {c}: {e}")
    if egl is None:
        say("[egl] no system libEGL: the EGL device platform is not available (the only libEGL in the image is SwiftShader's, a CPU "
            "rasteriser inside the kaleido wheel, which cannot share buffers with HIP)")
        return False
    egl.eglGetProcAddress.restype = C.c_void_p
    egl.eglGetProcAddress.argtypes = [C.c_char_p]
    egl.eglQueryString.restype = C.c_char_p
    egl.eglQueryString.argtypes = [C.c_void_p, C.c_int]
    client = egl.eglQueryString(None, EGL_EXTENSIONS)
    say("[egl] client extensions:", (client or b"(none)").decode())

    def proc(nm, restype, argtypes):
        p = egl.eglGetProcAddress(nm)
        return C.CFUNCTYPE(restype, *argtypes)(p) if p else None
    q_devices = proc(b"eglQueryDevicesEXT", C.c_uint, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)])
    q_dev_str = proc(b"eglQueryDeviceStringEXT", C.c_char_p, [C.c_void_p, C.c_int])
    get_disp = proc(b"eglGetPlatformDisplayEXT", C.c_void_p, [C.c_int, C.c_void_p, C.POINTER(C.c_int)])
    if not (q_devices and get_disp):
        say("[egl] EGL_EXT_device_enumeration / EGL_EXT_platform_device are missing")
        return False
    devs = (C.c_void_p * 32)()
    n = C.c_int(0)
    if not q_devices(32, devs, C.byref(n)):
        say("[egl] eglQueryDevicesEXT failed")
        return False
    say(f"[egl] {n.value} EGL device(s)")
    ok = False
    for i in range(n.value):
        ext = (q_dev_str(devs[i], EGL_EXTENSIONS) or b"").decode() if q_dev_str else ""
        node = (q_dev_str(devs[i], EGL_DRM_DEVICE_FILE_EXT) or b"").decode() if q_dev_str and "EGL_EXT_device_drm" in ext else ""
        say(f"[egl] device {i}: drm node '{node}', extensions: {ext}")
        dpy = get_disp(EGL_PLATFORM_DEVICE_EXT, devs[i], None)
        major, minor = C.c_int(0), C.c_int(0)
        egl.eglInitialize.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        if not dpy or not egl.eglInitialize(C.c_void_p(dpy), C.byref(major), C.byref(minor)):
            say(f"[egl] device {i}: eglInitialize failed (error 0x{egl.eglGetError():x})")
            continue
        say(f"[egl] device {i}: EGL {major.value}.{minor.value}, vendor {(egl.eglQueryString(C.c_void_p(dpy), EGL_VENDOR) or b'').decode()}")
        if not egl.eglBindAPI(EGL_OPENGL_API):
            say(f"[egl] device {i}: desktop OpenGL is not offered")
            continue
        cfg_attr = (C.c_int * 5)(EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_BIT, EGL_NONE)
        cfg, ncfg = C.c_void_p(), C.c_int(0)
        egl.eglChooseConfig.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        if not egl.eglChooseConfig(C.c_void_p(dpy), cfg_attr, C.byref(cfg), 1, C.byref(ncfg)) or ncfg.value < 1:
            say(f"[egl] device {i}: no OpenGL-capable config")
            continue
        ctx_attr = (C.c_int * 7)(EGL_CONTEXT_MAJOR_VERSION, 3, EGL_CONTEXT_MINOR_VERSION, 3, EGL_CONTEXT_OPENGL_PROFILE_MASK, EGL_CONTEXT_OPENGL_CORE_PROFILE_BIT, EGL_NONE)
        egl.eglCreateContext.restype = C.c_void_p
        egl.eglCreateContext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        ctx = egl.eglCreateContext(C.c_void_p(dpy), cfg, None, ctx_attr)
        if not ctx:
            say(f"[egl] device {i}: no OpenGL 3.3 core context (error 0x{egl.eglGetError():x})")
            continue
        egl.eglMakeCurrent.argtypes = [C.c_void_p] * 4
        if not egl.eglMakeCurrent(C.c_void_p(dpy), None, None, C.c_void_p(ctx)):
            say(f"[egl] device {i}: eglMakeCurrent (surfaceless) failed (error 0x{egl.eglGetError():x})")
            continue
        get_string = proc(b"glGetString", C.c_char_p, [C.c_uint])
        vendor = (get_string(GL_VENDOR) or b"").decode() if get_string else "?"
        renderer = (get_string(GL_RENDERER) or b"").decode() if get_string else "?"
        say(f"[egl] device {i}: CONTEXT CURRENT -- GL_VENDOR '{vendor}', GL_RENDERER '{renderer}', GL_VERSION '{(get_string(GL_VERSION) or b'').decode() if get_string else '?'}'")
        if any(k in (vendor + renderer).lower() for k in ("amd", "radeon", "ati ", "gfx9")):
            ok = True
    return ok


def main() -> int:
    say("== 1. EGL device platform")
    ok = False
    try:
        ok = try_egl()
    except Exception as e:  # noqa: BLE001
        say("[egl] probe raised:", repr(e))
    say("== 2. DRM render nodes / GBM")
    nodes = sorted(glob.glob("/dev/dri/*"))
    say("[drm] /dev/dri:", nodes if nodes else "absent (no render node is passed into this container)")
    say("[drm] /dev/kfd:", "present" if os.path.exists("/dev/kfd") else "absent")
    say("[drm] libgbm:", ctypes.util.find_library("gbm") or "not installed")
    dri = sorted(glob.glob("/usr/lib/x86_64-linux-gnu/dri/*radeonsi*"))
    say("[drm] Mesa radeonsi driver:", dri if dri else "not installed")
    say("== 3. GLX")
    say("[glx] libGL:", ctypes.util.find_library("GL") or "not installed", "| $DISPLAY:", os.environ.get("DISPLAY") or "unset (no X server: GLX cannot create a context)")
    say("== 4. HIP side of the interop")
    try:
        hip = C.CDLL("libamdhip64.so")
        for sym in ("hipGraphicsGLRegisterBuffer", "hipGraphicsMapResources", "hipGraphicsResourceGetMappedPointer", "hipGLGetDevices"):
            say(f"[hip] {sym}:", "exported" if hasattr(hip, sym) else "MISSING")
    except OSError as e:
        say("[hip] libamdhip64.so:", e)
    say("== verdict:", "a GL context on the AMD GPU came up: the interop test can run" if ok else
        "NO headless OpenGL context can be created on this machine (see above for the door that closed): include/gsplat_gl_interop.h and "
        "hdk/DM_GSplatHook_hip.C stay type-checked only")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
