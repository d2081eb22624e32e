#!/usr/bin/env python3
"""What multisampling would do to the reference's views.  The reference never sets vtkRenderWindow.MultiSamples
(render3d.py:60-65), so whether its views are multisampled is VTK's default for the installed version / platform
(vtkOpenGLRenderWindow's global maximum of multisamples: 0 or 8) - not knowable here, and the renderer contract models NO
multisampling ("assumed, not checkable").  This probe draws two of the golden scenes with the software OpenGL
at 0 and at 4 samples per pixel (its maximum; resolve by glBlitFramebuffer, as VTK does before reading back) and reports how many
pixels change and by how much: the size of that assumption.  BUILD-CONTAINER TOOL (SwiftShader).
usage: python tools/experiments/gl_msaa_probe.py  -> profiles/r06_gl_msaa_probe.txt"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tools"))
import gl_reference as G  # noqa: E402
from make_gl_golden import rotation, scenes  # noqa: E402

GL_DRAW_FRAMEBUFFER, GL_READ_FRAMEBUFFER, GL_NEAREST_FILTER = 0x8CA9, 0x8CA8, 0x2600


def multisample_target(gl, size, samples):
    """(multisampled FBO with RGBA8 + DEPTH24, single-sample FBO with RGBA8 to resolve into)"""
    rb = (C.c_uint * 3)()
    gl.glGenRenderbuffers(3, rb)
    gl.glBindRenderbuffer(G.GL_RENDERBUFFER, rb[0])
    gl.glRenderbufferStorageMultisample(G.GL_RENDERBUFFER, samples, G.GL_RGBA8, size, size)
    gl.glBindRenderbuffer(G.GL_RENDERBUFFER, rb[1])
    gl.glRenderbufferStorageMultisample(G.GL_RENDERBUFFER, samples, G.GL_DEPTH_COMPONENT24, size, size)
    gl.glBindRenderbuffer(G.GL_RENDERBUFFER, rb[2])
    gl.glRenderbufferStorage(G.GL_RENDERBUFFER, G.GL_RGBA8, size, size)
    fbo = (C.c_uint * 2)()
    gl.glGenFramebuffers(2, fbo)
    gl.glBindFramebuffer(G.GL_FRAMEBUFFER, fbo[0])
    gl.glFramebufferRenderbuffer(G.GL_FRAMEBUFFER, G.GL_COLOR_ATTACHMENT0, G.GL_RENDERBUFFER, rb[0])
    gl.glFramebufferRenderbuffer(G.GL_FRAMEBUFFER, G.GL_DEPTH_ATTACHMENT, G.GL_RENDERBUFFER, rb[1])
    assert gl.glCheckFramebufferStatus(G.GL_FRAMEBUFFER) == G.GL_FRAMEBUFFER_COMPLETE
    gl.glBindFramebuffer(G.GL_FRAMEBUFFER, fbo[1])
    gl.glFramebufferRenderbuffer(G.GL_FRAMEBUFFER, G.GL_COLOR_ATTACHMENT0, G.GL_RENDERBUFFER, rb[2])
    assert gl.glCheckFramebufferStatus(G.GL_FRAMEBUFFER) == G.GL_FRAMEBUFFER_COMPLETE
    return fbo[0], fbo[1]


def draw_msaa(g, ms_fbo, resolve_fbo, verts_view):
    gl, s = g.gl, g.size
    v = np.ascontiguousarray(verts_view, np.float32)
    gl.glBindFramebuffer(G.GL_FRAMEBUFFER, ms_fbo)
    one_buf = (C.c_uint * 1)(G.GL_COLOR_ATTACHMENT0)
    gl.glDrawBuffers(1, one_buf)
    gl.glViewport(0, 0, s, s)
    gl.glBindBuffer(G.GL_ARRAY_BUFFER, g.vbo_pos)
    gl.glBufferData(G.GL_ARRAY_BUFFER, C.c_ssize_t(v.nbytes), v.ctypes.data_as(C.c_void_p), G.GL_STATIC_DRAW)
    gl.glEnableVertexAttribArray(0)
    gl.glVertexAttribPointer(0, 3, G.GL_FLOAT, 0, 0, None)
    for cap in (G.GL_CULL_FACE, G.GL_BLEND, G.GL_DITHER, G.GL_SCISSOR_TEST):
        gl.glDisable(cap)
    gl.glEnable(G.GL_DEPTH_TEST)
    gl.glDepthFunc(G.GL_LEQUAL)
    white = (C.c_float * 4)(1.0, 1.0, 1.0, 1.0)
    one = C.c_float(1.0)
    gl.glClearBufferfv(G.GL_COLOR, 0, white)
    gl.glClearBufferfv(G.GL_DEPTH, 0, C.byref(one))
    gl.glDrawElements(G.GL_TRIANGLES, g.n_idx, G.GL_UNSIGNED_INT, None)
    gl.glBindFramebuffer(GL_READ_FRAMEBUFFER, ms_fbo)
    gl.glBindFramebuffer(GL_DRAW_FRAMEBUFFER, resolve_fbo)
    gl.glBlitFramebuffer(0, 0, s, s, 0, 0, s, s, G.GL_COLOR_BUFFER_BIT, GL_NEAREST_FILTER)
    gl.glBindFramebuffer(G.GL_FRAMEBUFFER, resolve_fbo)
    gl.glReadBuffer(G.GL_COLOR_ATTACHMENT0)
    rgba = np.empty((s, s, 4), np.uint8)
    gl.glPixelStorei(G.GL_PACK_ALIGNMENT, 1)
    gl.glReadPixels(0, 0, s, s, G.GL_RGBA, G.GL_UNSIGNED_BYTE, rgba.ctypes.data_as(C.c_void_p))
    g.check("msaa draw")
    return rgba[:, :, :3].copy()


def main():
    g = G.GLReference(256)
    # the fragment shader writes two outputs; with one draw buffer the second is discarded - same shader, same sampling
    ms_fbo, resolve_fbo = multisample_target(g.gl, 256, 4)
    all_sc = scenes()
    print(f"OpenGL: {g.info}; multisampled target: 4 samples per pixel (GL_MAX_SAMPLES of this implementation), resolved by glBlitFramebuffer")
    for name in ("face40", "face224", "coarse"):
        sc = all_sc[name]
        g.set_mesh(sc["uvs"], sc["tris"], sc["tex"])
        changed = covered = 0
        absdiff, edge = [], 0
        for rx, ry, rz in sc["poses"]:
            m = rotation(rx, ry, rz)
            v = sc["verts"].astype(np.float64)
            vv = np.stack([(m[k, 0] * v[:, 0] + m[k, 1] * v[:, 1]) + m[k, 2] * v[:, 2] for k in range(3)], 1).astype(np.float32)
            rgb4 = draw_msaa(g, ms_fbo, resolve_fbo, vv)
            g._framebuffer()                                   # back to the single-sample target of the golden file
            rgb0, z0 = g.draw(vv)
            cov = z0 < 1
            d = np.abs(rgb4.astype(np.int32) - rgb0.astype(np.int32)).max(-1)
            # silhouette = covered pixels with an uncovered 4-neighbour, or the reverse
            nb = np.zeros_like(cov)
            nb[1:] |= cov[:-1] != cov[1:]; nb[:-1] |= cov[:-1] != cov[1:]; nb[:, 1:] |= cov[:, :-1] != cov[:, 1:]; nb[:, :-1] |= cov[:, :-1] != cov[:, 1:]
            changed += int((d > 0).sum())
            edge += int(((d > 0) & nb).sum())
            covered += int(cov.sum())
            absdiff.append(d[d > 0])
        ad = np.concatenate(absdiff) if absdiff else np.zeros(0)
        n_px = len(sc["poses"]) * 256 * 256
        print(f"{name:8s} {len(sc['poses'])} views: pixels whose RGB changes with 4 samples {changed} = {100 * changed / n_px:.2f} % of all, "
              f"{100 * changed / max(covered, 1):.1f} % relative to the covered ones; on the silhouette {edge}, inside {changed - edge}; "
              f"change in byte units: median {np.median(ad) if len(ad) else 0:.0f}, 90th percentile {np.percentile(ad, 90) if len(ad) else 0:.0f}, max {ad.max() if len(ad) else 0}")


if __name__ == "__main__":
    main()
