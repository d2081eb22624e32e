"""Headless OpenGL ES 3.0 through ctypes: what VTK asks of OpenGL for one view of the reference renderer.

BUILD-CONTAINER TOOL (like tools/make_golden.py): it needs an EGL + GLES 3.0 implementation; the image has one, SwiftShader
(software, conformant), shipped inside the `kaleido` wheel.  Nothing of the product, the tests or the bench imports this file;
tools/make_gl_golden.py uses it to write tests/golden/gl_raster.npz.

What is drawn follows the reference's calls, not this repository's contract (that is the point of the exercise):
  * camera: position (0, 0, 500), focal point origin, view-up +y, parallel projection, parallel scale 150, clipping range
    (0, 1500) (render3d.py:56-59, :136, :150-152) -> the MCDC matrix VTK composes in double and uploads as float:
    x_ndc = x / 150, y_ndc = y / 150, z_ndc = -2 (z - 500) / 1500 - 1;
  * geometry: vertices already rotated by M = Ry Rx Rz on the CPU (vtkTransformPolyDataFilter, render3d.py:140-145: double
    matrix x float point, stored as float);
  * state: white clear colour (:54), depth test GL_LEQUAL (VTK's default), no face culling, no blending, no multisampling;
  * material: colour (1,1,1), ambient 1, diffuse 0 (utils3d.py:61-64) -> unlit; texture GL_NEAREST (SetInterpolate(0), :32),
    GL_REPEAT (vtkTexture's default), no mipmaps; image row 0 at the BOTTOM (vtkJPEGReader hands VTK's lower-left origin), so
    the texture is uploaded bottom row first and t = v;
  * read-back: RGB as bytes (vtkWindowToImageFilter RGB), Z as float.  GLES 3.0 cannot read a depth attachment back, so every
    fragment also writes floatBitsToUint(gl_FragCoord.z) into an R32UI colour attachment (exact bits, core ES 3.0): the float
    window depth BEFORE the depth buffer's quantisation.  The 24-bit value a GL_DEPTH_COMPONENT read would return is derived
    from it (`quantise24`).  (This implementation advertises GL_NV_read_depth but refuses it on a renderbuffer attachment.)
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

SWIFTSHADER = Path("/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/swiftshader")

EGL_NONE = 0x3038
EGL_SURFACE_TYPE, EGL_PBUFFER_BIT = 0x3033, 0x0001
EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT = 0x3040, 0x0040
EGL_RED_SIZE, EGL_GREEN_SIZE, EGL_BLUE_SIZE, EGL_ALPHA_SIZE, EGL_DEPTH_SIZE = 0x3024, 0x3023, 0x3022, 0x3021, 0x3025
EGL_SAMPLE_BUFFERS, EGL_SAMPLES = 0x3032, 0x3031
EGL_WIDTH, EGL_HEIGHT = 0x3057, 0x3056
EGL_OPENGL_ES_API = 0x30A0
EGL_CONTEXT_CLIENT_VERSION = 0x3098

GL_TRIANGLES = 0x0004
GL_DEPTH_BUFFER_BIT, GL_COLOR_BUFFER_BIT = 0x0100, 0x4000
GL_LEQUAL = 0x0203
GL_CULL_FACE, GL_DEPTH_TEST, GL_BLEND, GL_DITHER, GL_SCISSOR_TEST = 0x0B44, 0x0B71, 0x0BE2, 0x0BD0, 0x0C11
GL_UNPACK_ALIGNMENT, GL_PACK_ALIGNMENT = 0x0CF5, 0x0D05
GL_TEXTURE_2D = 0x0DE1
GL_UNSIGNED_BYTE, GL_UNSIGNED_INT, GL_FLOAT = 0x1401, 0x1405, 0x1406
GL_DEPTH_COMPONENT, GL_RGB, GL_RGBA = 0x1902, 0x1907, 0x1908
GL_NEAREST, GL_REPEAT = 0x2600, 0x2901
GL_TEXTURE_MAG_FILTER, GL_TEXTURE_MIN_FILTER, GL_TEXTURE_WRAP_S, GL_TEXTURE_WRAP_T = 0x2800, 0x2801, 0x2802, 0x2803
GL_TEXTURE0 = 0x84C0
GL_ARRAY_BUFFER, GL_ELEMENT_ARRAY_BUFFER, GL_STATIC_DRAW = 0x8892, 0x8893, 0x88E4
GL_FRAGMENT_SHADER, GL_VERTEX_SHADER = 0x8B30, 0x8B31
GL_COMPILE_STATUS, GL_LINK_STATUS = 0x8B81, 0x8B82
GL_FRAMEBUFFER, GL_RENDERBUFFER = 0x8D40, 0x8D41
GL_COLOR_ATTACHMENT0, GL_COLOR_ATTACHMENT1, GL_DEPTH_ATTACHMENT = 0x8CE0, 0x8CE1, 0x8D00
GL_FRAMEBUFFER_COMPLETE = 0x8CD5
GL_RGBA8, GL_RGB8, GL_R32UI, GL_DEPTH_COMPONENT24 = 0x8058, 0x8051, 0x8236, 0x81A6
GL_RED_INTEGER = 0x8D94
GL_COLOR, GL_DEPTH = 0x1800, 0x1801
GL_SUBPIXEL_BITS, GL_SAMPLES, GL_DEPTH_BITS = 0x0D50, 0x80A9, 0x0D56
GL_VENDOR, GL_RENDERER, GL_VERSION, GL_EXTENSIONS = 0x1F00, 0x1F01, 0x1F02, 0x1F03
GL_READ_FRAMEBUFFER = 0x8CA8

VERTEX_SHADER = b"""#version 300 es
layout(location = 0) in vec3 vertexMC;
layout(location = 1) in vec2 tcoordMC;
uniform mat4 MCDCMatrix;
out vec2 tcoordVC;
void main() {
    tcoordVC = tcoordMC;
    gl_Position = MCDCMatrix * vec4(vertexMC, 1.0);
}
"""

FRAGMENT_SHADER = b"""#version 300 es
precision highp float;
precision highp int;
uniform sampler2D texture_0;
uniform int textured;
in vec2 tcoordVC;
layout(location = 0) out vec4 fragColour;
layout(location = 1) out uint fragDepthBits;
void main() {
    vec4 ambient = vec4(1.0, 1.0, 1.0, 1.0);                  /* colour (1,1,1) x ambient 1, diffuse 0 */
    if (textured != 0) ambient = ambient * texture(texture_0, tcoordVC);
    fragColour = ambient;
    fragDepthBits = floatBitsToUint(gl_FragCoord.z);
}
"""


def mcdc_matrix() -> np.ndarray:
    """VTK's model-to-device matrix for the reference camera, composed in double, handed to GL as float (column-major)."""
    near, far, scale = 0.0, 1500.0, 150.0
    view = np.eye(4)
    view[2, 3] = -500.0                                          # camera at z = +500 looking down -z, up +y
    proj = np.diag([1.0 / scale, 1.0 / scale, -2.0 / (far - near), 1.0])
    proj[2, 3] = -(far + near) / (far - near)
    return (proj @ view).astype(np.float32)


def quantise24(z: np.ndarray) -> np.ndarray:
    """The float a GL_DEPTH_COMPONENT / GL_FLOAT read returns from a 24-bit fixed-point depth buffer."""
    zi = np.floor(z.astype(np.float64) * 16777215.0 + 0.5)
    return (zi / 16777215.0).astype(np.float32)


def reference_postprocess(rgb_bottom_up: np.ndarray, z_bottom_up: np.ndarray) -> np.ndarray:
    """render3d.py:158-177 + :191 applied literally to one view's read-backs (both with GL's row 0 = bottom):
    RGB bytes into a float32 stack; Z x (-255) cast to unsigned char WITHOUT clamping (vtkImageShiftScale, :73-77: a C cast of a
    negative double, which wraps modulo 256 on every compiler VTK ships on); flip rows; / 255."""
    stack = np.empty((256, 256, 4), np.float32)
    stack[:, :, 0:3] = rgb_bottom_up
    scaled = (z_bottom_up.astype(np.float64) + 0.0) * -255.0
    stack[:, :, 3] = (np.trunc(scaled).astype(np.int64) & 255).astype(np.uint8)
    return np.flip(stack, axis=0) / np.float32(255)


class GLReference:
    def __init__(self, size: int = 256):
        self.size = size
        self.egl = C.CDLL(str(SWIFTSHADER / "libEGL.so"))
        self.gl = C.CDLL(str(SWIFTSHADER / "libGLESv2.so"))
        egl, gl = self.egl, self.gl
        egl.eglGetDisplay.restype = C.c_void_p
        egl.eglGetDisplay.argtypes = [C.c_void_p]
        egl.eglCreatePbufferSurface.restype = C.c_void_p
        egl.eglCreateContext.restype = C.c_void_p
        gl.glGetString.restype = C.c_char_p
        self.dpy = C.c_void_p(egl.eglGetDisplay(None))
        major, minor = C.c_int(), C.c_int()
        if not egl.eglInitialize(self.dpy, C.byref(major), C.byref(minor)):
            raise RuntimeError("eglInitialize failed")
        attrs = (C.c_int * 19)(EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT, EGL_RED_SIZE, 8,
                               EGL_GREEN_SIZE, 8, EGL_BLUE_SIZE, 8, EGL_ALPHA_SIZE, 8, EGL_DEPTH_SIZE, 24, EGL_SAMPLE_BUFFERS, 0,
                               EGL_SAMPLES, 0, EGL_NONE)
        cfg, ncfg = C.c_void_p(), C.c_int()
        if not egl.eglChooseConfig(self.dpy, attrs, C.byref(cfg), 1, C.byref(ncfg)) or ncfg.value < 1:
            raise RuntimeError("no EGL config")
        pb = (C.c_int * 5)(EGL_WIDTH, size, EGL_HEIGHT, size, EGL_NONE)
        self.surf = C.c_void_p(egl.eglCreatePbufferSurface(self.dpy, cfg, pb))
        egl.eglBindAPI(EGL_OPENGL_ES_API)
        ca = (C.c_int * 3)(EGL_CONTEXT_CLIENT_VERSION, 3, EGL_NONE)
        self.ctx = C.c_void_p(egl.eglCreateContext(self.dpy, cfg, None, ca))
        if not self.ctx.value or not egl.eglMakeCurrent(self.dpy, self.surf, self.surf, self.ctx):
            raise RuntimeError("EGL context failed")
        self.info = {k: (gl.glGetString(v) or b"").decode() for k, v in
                     (("vendor", GL_VENDOR), ("renderer", GL_RENDERER), ("version", GL_VERSION))}
        self.extensions = (gl.glGetString(GL_EXTENSIONS) or b"").decode().split()
        self._framebuffer()
        self.info["subpixel_bits"] = self._geti(GL_SUBPIXEL_BITS)
        self.info["samples"] = self._geti(GL_SAMPLES)
        self.info["depth_bits"] = self._geti(GL_DEPTH_BITS)
        self._program()
        self.check("init")

    def _geti(self, what: int) -> int:
        v = C.c_int(-1)
        self.gl.glGetIntegerv(what, C.byref(v))
        return v.value

    def check(self, where: str) -> None:
        e = self.gl.glGetError()
        if e:
            raise RuntimeError(f"GL error 0x{e:04x} at {where}")

    def _framebuffer(self) -> None:
        gl, s = self.gl, self.size
        rb = (C.c_uint * 3)()
        gl.glGenRenderbuffers(3, rb)
        for r, fmt in zip(rb, (GL_RGBA8, GL_R32UI, GL_DEPTH_COMPONENT24)):
            gl.glBindRenderbuffer(GL_RENDERBUFFER, r)
            gl.glRenderbufferStorage(GL_RENDERBUFFER, fmt, s, s)   # (no glRenderbufferStorageMultisample: samples = 0)
        fbo = C.c_uint()
        gl.glGenFramebuffers(1, C.byref(fbo))
        gl.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        gl.glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_RENDERBUFFER, rb[0])
        gl.glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT1, GL_RENDERBUFFER, rb[1])
        gl.glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, rb[2])
        if gl.glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError("framebuffer incomplete")
        bufs = (C.c_uint * 2)(GL_COLOR_ATTACHMENT0, GL_COLOR_ATTACHMENT1)
        gl.glDrawBuffers(2, bufs)
        gl.glViewport(0, 0, s, s)

    def _shader(self, kind: int, src: bytes) -> int:
        gl = self.gl
        sh = gl.glCreateShader(kind)
        p = C.c_char_p(src)
        gl.glShaderSource(sh, 1, C.byref(p), None)
        gl.glCompileShader(sh)
        ok = C.c_int()
        gl.glGetShaderiv(sh, GL_COMPILE_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(4096)
            gl.glGetShaderInfoLog(sh, 4096, None, log)
            raise RuntimeError("shader: " + log.value.decode())
        return sh

    def _program(self) -> None:
        gl = self.gl
        prog = gl.glCreateProgram()
        gl.glAttachShader(prog, self._shader(GL_VERTEX_SHADER, VERTEX_SHADER))
        gl.glAttachShader(prog, self._shader(GL_FRAGMENT_SHADER, FRAGMENT_SHADER))
        gl.glLinkProgram(prog)
        ok = C.c_int()
        gl.glGetProgramiv(prog, GL_LINK_STATUS, C.byref(ok))
        if not ok.value:
            raise RuntimeError("link failed")
        gl.glUseProgram(prog)
        m = mcdc_matrix()
        gl.glUniformMatrix4fv(gl.glGetUniformLocation(prog, b"MCDCMatrix"), 1, 0,
                              np.ascontiguousarray(m.T).ctypes.data_as(C.POINTER(C.c_float)))
        gl.glUniform1i(gl.glGetUniformLocation(prog, b"texture_0"), 0)
        self.u_textured = gl.glGetUniformLocation(prog, b"textured")
        self.prog = prog

    def set_mesh(self, uvs, tris, texture) -> None:
        """Index buffer, texture coordinates and the texture: per mesh, as in the reference (obj_to_actor)."""
        gl = self.gl
        self.n_idx = int(np.asarray(tris).size)
        idx = np.ascontiguousarray(tris, np.uint32)
        bufs = (C.c_uint * 3)()
        gl.glGenBuffers(3, bufs)
        self.vbo_pos, self.vbo_uv, ebo = bufs
        gl.glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ebo)
        gl.glBufferData(GL_ELEMENT_ARRAY_BUFFER, C.c_ssize_t(idx.nbytes), idx.ctypes.data_as(C.c_void_p), GL_STATIC_DRAW)
        textured = uvs is not None and texture is not None
        gl.glUniform1i(self.u_textured, 1 if textured else 0)
        if textured:
            uv = np.ascontiguousarray(uvs, np.float32)
            gl.glBindBuffer(GL_ARRAY_BUFFER, self.vbo_uv)
            gl.glBufferData(GL_ARRAY_BUFFER, C.c_ssize_t(uv.nbytes), uv.ctypes.data_as(C.c_void_p), GL_STATIC_DRAW)
            gl.glEnableVertexAttribArray(1)
            gl.glVertexAttribPointer(1, 2, GL_FLOAT, 0, 0, None)
            tex = np.ascontiguousarray(np.asarray(texture, np.uint8)[::-1])      # VTK image origin: lower left
            t = C.c_uint()
            gl.glGenTextures(1, C.byref(t))
            gl.glActiveTexture(GL_TEXTURE0)
            gl.glBindTexture(GL_TEXTURE_2D, t)
            gl.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)
            gl.glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB8, tex.shape[1], tex.shape[0], 0, GL_RGB, GL_UNSIGNED_BYTE,
                            tex.ctypes.data_as(C.c_void_p))
            for k, v in ((GL_TEXTURE_MIN_FILTER, GL_NEAREST), (GL_TEXTURE_MAG_FILTER, GL_NEAREST),
                         (GL_TEXTURE_WRAP_S, GL_REPEAT), (GL_TEXTURE_WRAP_T, GL_REPEAT)):
                gl.glTexParameteri(GL_TEXTURE_2D, k, v)
        else:
            gl.glDisableVertexAttribArray(1)
        self.check("set_mesh")

    def draw(self, verts_view: np.ndarray):
        """One view: vertices already in view orientation (float32) -> (rgb u8 [256,256,3], z f32 [256,256]), GL rows
        (row 0 = bottom)."""
        gl, s = self.gl, self.size
        v = np.ascontiguousarray(verts_view, np.float32)
        gl.glBindBuffer(GL_ARRAY_BUFFER, self.vbo_pos)
        gl.glBufferData(GL_ARRAY_BUFFER, C.c_ssize_t(v.nbytes), v.ctypes.data_as(C.c_void_p), GL_STATIC_DRAW)
        gl.glEnableVertexAttribArray(0)
        gl.glVertexAttribPointer(0, 3, GL_FLOAT, 0, 0, None)
        for cap in (GL_CULL_FACE, GL_BLEND, GL_DITHER, GL_SCISSOR_TEST):
            gl.glDisable(cap)
        gl.glEnable(GL_DEPTH_TEST)
        gl.glDepthFunc(GL_LEQUAL)
        gl.glDepthMask(1)
        white = (C.c_float * 4)(1.0, 1.0, 1.0, 1.0)
        one_bits = (C.c_uint * 4)(0x3F800000, 0, 0, 0)
        one = C.c_float(1.0)
        gl.glClearBufferfv(GL_COLOR, 0, white)
        gl.glClearBufferuiv(GL_COLOR, 1, one_bits)
        gl.glClearBufferfv(GL_DEPTH, 0, C.byref(one))
        gl.glDrawElements(GL_TRIANGLES, self.n_idx, GL_UNSIGNED_INT, None)
        gl.glFinish()
        gl.glPixelStorei(GL_PACK_ALIGNMENT, 1)
        rgba = np.empty((s, s, 4), np.uint8)
        gl.glReadBuffer(GL_COLOR_ATTACHMENT0)
        gl.glReadPixels(0, 0, s, s, GL_RGBA, GL_UNSIGNED_BYTE, rgba.ctypes.data_as(C.c_void_p))
        zbits = np.empty((s, s), np.uint32)
        gl.glReadBuffer(GL_COLOR_ATTACHMENT1)
        gl.glReadPixels(0, 0, s, s, GL_RED_INTEGER, GL_UNSIGNED_INT, zbits.ctypes.data_as(C.c_void_p))
        self.check("draw")
        return rgba[:, :, :3].copy(), zbits.view(np.float32)
