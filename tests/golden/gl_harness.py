"""Run the REFERENCE's GLSL shaders off-screen in this container (never on the GPU box): a minimal EGL + OpenGL ES 3.0 binding
over the SwiftShader software renderer that ships inside the `kaleido` wheel of this image
(.../kaleido/executable/bin/swiftshader/libEGL.so + libGLESv2.so -- found by the round-5 probe: no moderngl / glfw / Xvfb /
OSMesa / Mesa-EGL here, but SwiftShader gives a surfaceless "OpenGL ES 3.0 SwiftShader 4.1.0.7" context with
OES_texture_float_linear and EXT_color_buffer_float).

The shader text is taken from /root/reference/viewer.py at generation time with `ast` (the VERTEX_SHADER / FRAGMENT_SHADER string
constants, viewer.py:376-631) and never stored.  The reference writes `#version 330` desktop GLSL; ES 3.00 is the same language
minus four liberties the text takes, patched mechanically by to_es300() and listed in every fixture's manifest (plus the
leading blank line of the Python string constants, which ES does not accept in front of #version):
  1. `#version 330`                         -> `#version 300 es` + highp default precisions
  2. `uniform float u_x = <literal>;`       -> `uniform float u_x;` and the literal is SET as the uniform's value by the harness
     (desktop GL initialises a uniform to its default; ES has no uniform initialisers)
  3. `vec2 pixel_size = 1.0 / u_resolution;` at global scope (viewer.py:413; not a constant expression)
                                            -> `vec2 pixel_size;` + the same assignment as the first statement of main()
  4. int * float products (`dy * pixel_size.y`, `x * pixel_size.x`) -> `float(dy) * ...` (ES has no implicit int -> float)
Nothing else is touched.  Only tests/golden/make_golden_dibr.py uses this file.
"""
from __future__ import annotations

import ast
import ctypes as C
import glob
import os
import re

import numpy as np

REF_VIEWER = "/root/reference/viewer.py"

GL = dict(VERTEX_SHADER=0x8B31, FRAGMENT_SHADER=0x8B30, COMPILE_STATUS=0x8B81, LINK_STATUS=0x8B82, TEXTURE_2D=0x0DE1,
          TEXTURE0=0x84C0, RGB=0x1907, RGBA=0x1908, RED=0x1903, RGB8=0x8051, R32F=0x822E, RGBA32F=0x8814, UNSIGNED_BYTE=0x1401,
          FLOAT=0x1406, LINEAR=0x2601, NEAREST=0x2600, REPEAT=0x2901, TEXTURE_MIN_FILTER=0x2801, TEXTURE_MAG_FILTER=0x2800,
          TEXTURE_WRAP_S=0x2802, TEXTURE_WRAP_T=0x2803, FRAMEBUFFER=0x8D40, COLOR_ATTACHMENT0=0x8CE0, FRAMEBUFFER_COMPLETE=0x8CD5,
          ARRAY_BUFFER=0x8892, STATIC_DRAW=0x88E4, TRIANGLE_STRIP=0x0005, COLOR_BUFFER_BIT=0x4000, UNPACK_ALIGNMENT=0x0CF5,
          PACK_ALIGNMENT=0x0D05, BLEND=0x0BE2)


def reference_shaders():
    """(VERTEX_SHADER, FRAGMENT_SHADER) string constants of the reference's viewer.py, read with ast (the module itself imports
    glfw / moderngl, absent here)."""
    with open(REF_VIEWER) as f:
        tree = ast.parse(f.read())
    out = {}
    for n in tree.body:
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) \
                and n.targets[0].id in ("VERTEX_SHADER", "FRAGMENT_SHADER") and isinstance(n.value, ast.Constant):
            out[n.targets[0].id] = (n.value.value, n.lineno, n.end_lineno)
    return out["VERTEX_SHADER"], out["FRAGMENT_SHADER"]


def to_es300(src: str):
    """-> (ES 3.00 source, {uniform: default literal}).  The four mechanical patches of the module docstring."""
    defaults = {}
    src = src.lstrip()                                     # (0) the constants start with a blank line; ES wants #version on line 1
    src = re.sub(r"#version\s+330", "#version 300 es\nprecision highp float;\nprecision highp int;\nprecision highp sampler2D;", src, count=1)

    def strip_default(m):
        defaults[m.group(2)] = float(m.group(3))
        return f"uniform {m.group(1)} {m.group(2)};"
    src = re.sub(r"uniform\s+(float|int)\s+(\w+)\s*=\s*([-0-9.eE]+)\s*;", strip_default, src)
    if re.search(r"vec2\s+pixel_size\s*=\s*1\.0\s*/\s*u_resolution\s*;", src):
        src = re.sub(r"vec2\s+pixel_size\s*=\s*1\.0\s*/\s*u_resolution\s*;", "vec2 pixel_size;", src, count=1)
        src = re.sub(r"void\s+main\s*\(\s*\)\s*\{", "void main() {\n        pixel_size = 1.0 / u_resolution;", src, count=1)
    src = src.replace("dy * pixel_size.y", "float(dy) * pixel_size.y").replace("x * pixel_size.x", "float(x) * pixel_size.x")
    return src, defaults


class Gles:
    """One surfaceless ES 3.0 context (pbuffer) for the life of the process."""

    def __init__(self):
        dirs = glob.glob("/usr/local/lib/python3*/dist-packages/kaleido/executable/bin/swiftshader") + \
            glob.glob("/opt/conda/lib/python3*/site-packages/kaleido/executable/bin/swiftshader")
        if not dirs:
            raise RuntimeError("SwiftShader (kaleido wheel) not found: no off-screen GL in this container")
        self.gl = C.CDLL(os.path.join(dirs[0], "libGLESv2.so"), mode=C.RTLD_GLOBAL)
        self.egl = C.CDLL(os.path.join(dirs[0], "libEGL.so"), mode=C.RTLD_GLOBAL)
        e = self.egl
        e.eglGetDisplay.restype = C.c_void_p
        e.eglGetDisplay.argtypes = [C.c_void_p]
        e.eglInitialize.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        e.eglChooseConfig.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        e.eglCreatePbufferSurface.restype = C.c_void_p
        e.eglCreatePbufferSurface.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        e.eglCreateContext.restype = C.c_void_p
        e.eglCreateContext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        e.eglMakeCurrent.argtypes = [C.c_void_p] * 4
        dpy = e.eglGetDisplay(None)
        maj, mnr = C.c_int(), C.c_int()
        if not e.eglInitialize(dpy, C.byref(maj), C.byref(mnr)):
            raise RuntimeError("eglInitialize failed")
        attrs = (C.c_int * 5)(0x3033, 1, 0x3040, 0x40, 0x3038)          # SURFACE_TYPE = PBUFFER, RENDERABLE_TYPE = ES3
        cfg, n = C.c_void_p(), C.c_int()
        if not e.eglChooseConfig(dpy, attrs, C.byref(cfg), 1, C.byref(n)) or n.value < 1:
            raise RuntimeError("no EGL config")
        surf = e.eglCreatePbufferSurface(dpy, cfg, (C.c_int * 5)(0x3057, 16, 0x3056, 16, 0x3038))
        e.eglBindAPI(0x30A0)
        ctx = e.eglCreateContext(dpy, cfg, None, (C.c_int * 3)(0x3098, 3, 0x3038))
        if not ctx or not e.eglMakeCurrent(dpy, surf, surf, ctx):
            raise RuntimeError("no ES 3 context")
        g = self.gl
        g.glGetString.restype = C.c_char_p
        g.glGetUniformLocation.argtypes = [C.c_uint, C.c_char_p]
        g.glUniform1f.argtypes = [C.c_int, C.c_float]
        g.glUniform2f.argtypes = [C.c_int, C.c_float, C.c_float]
        g.glUniform4f.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        g.glClearColor.argtypes = [C.c_float] * 4
        g.glTexImage2D.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
        g.glReadPixels.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
        g.glBufferData.argtypes = [C.c_uint, C.c_ssize_t, C.c_void_p, C.c_uint]
        g.glVertexAttribPointer.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_ubyte, C.c_int, C.c_void_p]
        g.glBindAttribLocation.argtypes = [C.c_uint, C.c_uint, C.c_char_p]
        self.version = g.glGetString(0x1F02).decode()
        self.renderer = g.glGetString(0x1F01).decode()

    def _check(self, what):
        err = self.gl.glGetError()
        if err:
            raise RuntimeError(f"GL error 0x{err:x} after {what}")

    def program(self, vs: str, fs: str) -> int:
        g = self.gl
        ids = []
        for kind, src in ((GL["VERTEX_SHADER"], vs), (GL["FRAGMENT_SHADER"], fs)):
            sh = g.glCreateShader(kind)
            b = src.encode()
            g.glShaderSource(sh, 1, C.byref(C.c_char_p(b)), None)
            g.glCompileShader(sh)
            ok = C.c_int()
            g.glGetShaderiv(sh, GL["COMPILE_STATUS"], C.byref(ok))
            if not ok.value:
                log = C.create_string_buffer(8192)
                g.glGetShaderInfoLog(sh, 8192, None, log)
                raise RuntimeError("shader compile failed:\n" + log.value.decode())
            ids.append(sh)
        prog = g.glCreateProgram()
        for sh in ids:
            g.glAttachShader(prog, sh)
        g.glBindAttribLocation(prog, 0, b"in_position")
        g.glBindAttribLocation(prog, 1, b"in_uv")
        g.glLinkProgram(prog)
        ok = C.c_int()
        g.glGetProgramiv(prog, GL["LINK_STATUS"], C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(8192)
            g.glGetProgramInfoLog(prog, 8192, None, log)
            raise RuntimeError("program link failed:\n" + log.value.decode())
        return prog

    def texture(self, arr: np.ndarray, unit: int) -> int:
        """uint8 [H,W,3] -> RGB8 (moderngl `texture((w,h), 3, dtype='f1')`, viewer.py:2385) or float32 [H,W] -> R32F
        (`texture((w,h), 1, dtype='f4')`, :2386); moderngl's defaults: LINEAR min / mag filter, REPEAT wrap; row 0 of the array is
        the texture's v = 0 row (write() of a top-down image)."""
        g = self.gl
        t = C.c_uint()
        g.glGenTextures(1, C.byref(t))
        g.glActiveTexture(GL["TEXTURE0"] + unit)
        g.glBindTexture(GL["TEXTURE_2D"], t)
        g.glPixelStorei(GL["UNPACK_ALIGNMENT"], 1)
        arr = np.ascontiguousarray(arr)
        h, w = arr.shape[:2]
        if arr.dtype == np.uint8:
            g.glTexImage2D(GL["TEXTURE_2D"], 0, GL["RGB8"], w, h, 0, GL["RGB"], GL["UNSIGNED_BYTE"], arr.ctypes.data)
        else:
            arr = arr.astype(np.float32)
            g.glTexImage2D(GL["TEXTURE_2D"], 0, GL["R32F"], w, h, 0, GL["RED"], GL["FLOAT"], arr.ctypes.data)
        for k, v in ((GL["TEXTURE_MIN_FILTER"], GL["LINEAR"]), (GL["TEXTURE_MAG_FILTER"], GL["LINEAR"]),
                     (GL["TEXTURE_WRAP_S"], GL["REPEAT"]), (GL["TEXTURE_WRAP_T"], GL["REPEAT"])):
            g.glTexParameteri(GL["TEXTURE_2D"], k, v)
        self._check("texture upload")
        return t.value

    def render(self, prog: int, uniforms: dict, out_w: int, out_h: int, viewport=None) -> np.ndarray:
        """The viewer's full-screen quad (viewer.py:1471-1476: TRIANGLE_STRIP, positions -1..1, uv 0..1) into an RGBA32F target of
        out_w x out_h, NO blending (the reference never enables BLEND for this quad: viewer.py:1304-1307 enable and disable it
        around the overlay only) -> float32 [out_h,out_w,4] with row 0 = TOP of the screen."""
        g = self.gl
        fbo, tex, vbo = C.c_uint(), C.c_uint(), C.c_uint()
        g.glGenTextures(1, C.byref(tex))
        g.glActiveTexture(GL["TEXTURE0"] + 7)
        g.glBindTexture(GL["TEXTURE_2D"], tex)
        g.glTexImage2D(GL["TEXTURE_2D"], 0, GL["RGBA32F"], out_w, out_h, 0, GL["RGBA"], GL["FLOAT"], None)
        g.glTexParameteri(GL["TEXTURE_2D"], GL["TEXTURE_MIN_FILTER"], GL["NEAREST"])
        g.glTexParameteri(GL["TEXTURE_2D"], GL["TEXTURE_MAG_FILTER"], GL["NEAREST"])
        g.glGenFramebuffers(1, C.byref(fbo))
        g.glBindFramebuffer(GL["FRAMEBUFFER"], fbo)
        g.glFramebufferTexture2D(GL["FRAMEBUFFER"], GL["COLOR_ATTACHMENT0"], GL["TEXTURE_2D"], tex, 0)
        if g.glCheckFramebufferStatus(GL["FRAMEBUFFER"]) != GL["FRAMEBUFFER_COMPLETE"]:
            raise RuntimeError("RGBA32F framebuffer incomplete")
        vp = viewport or (0, 0, out_w, out_h)
        g.glViewport(*[int(v) for v in vp])
        g.glDisable(GL["BLEND"])
        g.glClearColor(0.0, 0.0, 0.0, 0.0)
        g.glClear(GL["COLOR_BUFFER_BIT"])
        g.glUseProgram(prog)
        for name, val in uniforms.items():
            loc = g.glGetUniformLocation(prog, name.encode())
            if loc < 0:
                continue                                    # optimised out
            if isinstance(val, int):
                g.glUniform1i(loc, val)
            elif isinstance(val, float):
                g.glUniform1f(loc, val)
            elif len(val) == 2:
                g.glUniform2f(loc, *[float(v) for v in val])
            else:
                g.glUniform4f(loc, *[float(v) for v in val])
        quad = np.array([-1, -1, 0, 0, 1, -1, 1, 0, -1, 1, 0, 1, 1, 1, 1, 1], np.float32)          # viewer.py:1471-1476
        g.glGenBuffers(1, C.byref(vbo))
        g.glBindBuffer(GL["ARRAY_BUFFER"], vbo)
        g.glBufferData(GL["ARRAY_BUFFER"], quad.nbytes, quad.ctypes.data, GL["STATIC_DRAW"])
        g.glEnableVertexAttribArray(0)
        g.glEnableVertexAttribArray(1)
        g.glVertexAttribPointer(0, 2, GL["FLOAT"], 0, 16, C.c_void_p(0))
        g.glVertexAttribPointer(1, 2, GL["FLOAT"], 0, 16, C.c_void_p(8))
        g.glDrawArrays(GL["TRIANGLE_STRIP"], 0, 4)
        g.glFinish()
        out = np.empty((out_h, out_w, 4), np.float32)
        g.glPixelStorei(GL["PACK_ALIGNMENT"], 1)
        g.glReadPixels(0, 0, out_w, out_h, GL["RGBA"], GL["FLOAT"], out.ctypes.data)
        self._check("render")
        g.glBindFramebuffer(GL["FRAMEBUFFER"], 0)
        g.glDeleteFramebuffers(1, C.byref(fbo))
        g.glDeleteTextures(1, C.byref(tex))
        g.glDeleteBuffers(1, C.byref(vbo))
        return out[::-1].copy()                              # GL rows are bottom-up

    def delete_texture(self, t: int):
        self.gl.glDeleteTextures(1, C.byref(C.c_uint(t)))
