// Scalar arithmetic of the multi-view rasteriser, written once for all kernels of raster.hip
// (transform, classify, tile) so they agree bit for bit on every pixel.  Plain C99 subset.
// The tests do NOT share this file: their checker, oracle/raster.c, is an independent restatement.
//
// What it restates (reference: src/mvlm/utils/render3d.py, all of it executed inside
// the third-party VTK/OpenGL stack there):
//   * per pose M = Ry*Rx*Rz applied to every vertex (render3d.py:140-144, vtkTransform
//     keeps a double 4x4 and vtkTransformPolyDataFilter writes float points),
//   * orthographic camera at (0,0,500) looking down -z, ParallelScale 150 => the 256x256
//     window shows x,y in [-150,150] (render3d.py:50, :150-152), clip range (0,1500)
//     => z-buffer value (500 - z_view)/1500 (render3d.py:136),
//   * unlit material, nearest-texel texture, white background (utils3d.py:32, :61-64;
//     render3d.py:54),
//   * z-buffer -> vtkImageShiftScale(scale=-255) -> unsigned char (render3d.py:73-77,
//     :166-170): (256 - trunc(255*z)) mod 256.
// OpenGL leaves sub-pixel snapping and the fill rule to the implementation; this build
// fixes them the way GPUs do: vertices snapped to 1/256 pixel, integer edge functions,
// a tie rule that gives every shared edge to exactly one triangle, depth test LEQUAL
// (VTK's default) => among equal depths the later-drawn triangle wins.
#ifndef MVLM_RASTER_MATH_H
#define MVLM_RASTER_MATH_H

#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define RM_FN __host__ __device__ static inline
#else
#define RM_FN static inline
#endif

#define RM_SIZE 256          /* image side (general_pipeline.py:57) */
#define RM_SUB 256           /* sub-pixel steps per pixel */
#define RM_HALF 128
#define RM_TILE 16           /* binning tile side in pixels */
#define RM_TILES (RM_SIZE / RM_TILE)
#define RM_COORD_LIM (1 << 22)

typedef struct {
    int32_t X, Y;  /* window coords, 1/256 pixel, y up (GL convention) */
    float z;       /* z-buffer value in [0,1] when inside the clip range */
    float pad;
} rm_vert;

/* `sub_bits`: vertices snap to 2^-sub_bits pixel (GL_SUBPIXEL_BITS of the OpenGL to be matched: 8 on GPUs, the default; >= 4
 * by the standard).  The lattice stays 1/256 pixel: a coarser snap is a multiple of it, every integer below scales by a
 * power of two and every float made from a ratio of them is unchanged. */
RM_FN rm_vert rm_transform(const double* m /*[9] row-major*/, float vx, float vy, float vz, int sub_bits) {
    const double x = vx, y = vy, z = vz;
    const float xv = (float)((m[0] * x + m[1] * y) + m[2] * z);
    const float yv = (float)((m[3] * x + m[4] * y) + m[5] * z);
    const float zv = (float)((m[6] * x + m[7] * y) + m[8] * z);
    const float k = 256.0f / 300.0f;
    const float sub = (float)(1 << sub_bits);
    float fx = floorf(((xv + 150.0f) * k) * sub + 0.5f);
    float fy = floorf(((yv + 150.0f) * k) * sub + 0.5f);
    const float lim = (float)(RM_COORD_LIM >> (8 - sub_bits));
    fx = fx < -lim ? -lim : (fx > lim ? lim : fx);
    fy = fy < -lim ? -lim : (fy > lim ? lim : fy);
    rm_vert o;
    o.X = (int32_t)fx * (RM_SUB >> sub_bits);
    o.Y = (int32_t)fy * (RM_SUB >> sub_bits);
    o.z = (500.0f - zv) / 1500.0f;
    o.pad = 0.0f;
    return o;
}

typedef struct {
    int32_t X0, Y0, X1, Y1, X2, Y2;
    float z0, z1, z2;
    float farea;
    int32_t ix0, ix1, iy0, iy1; /* covered pixel-centre range, clipped to the window; empty if ix0 > ix1 */
    int32_t swapped;            /* 1 if vertices 1 and 2 were exchanged to make the winding positive */
    int32_t valid;
} rm_tri;

RM_FN int32_t rm_floor_div(int32_t a, int32_t b) { /* b > 0 */
    int32_t q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

RM_FN rm_tri rm_setup(rm_vert a, rm_vert b, rm_vert c) {
    rm_tri t;
    int64_t area = (int64_t)(b.X - a.X) * (c.Y - a.Y) - (int64_t)(b.Y - a.Y) * (c.X - a.X);
    t.swapped = 0;
    if (area < 0) {
        rm_vert s = b;
        b = c;
        c = s;
        area = -area;
        t.swapped = 1;
    }
    t.X0 = a.X; t.Y0 = a.Y; t.X1 = b.X; t.Y1 = b.Y; t.X2 = c.X; t.Y2 = c.Y;
    t.z0 = a.z; t.z1 = b.z; t.z2 = c.z;
    t.farea = (float)area;
    int32_t minx = a.X < b.X ? a.X : b.X; minx = minx < c.X ? minx : c.X;
    int32_t maxx = a.X > b.X ? a.X : b.X; maxx = maxx > c.X ? maxx : c.X;
    int32_t miny = a.Y < b.Y ? a.Y : b.Y; miny = miny < c.Y ? miny : c.Y;
    int32_t maxy = a.Y > b.Y ? a.Y : b.Y; maxy = maxy > c.Y ? maxy : c.Y;
    /* pixel i has its centre at i*256+128: centres inside [min,max] */
    t.ix0 = -rm_floor_div(-(minx - RM_HALF), RM_SUB);
    t.ix1 = rm_floor_div(maxx - RM_HALF, RM_SUB);
    t.iy0 = -rm_floor_div(-(miny - RM_HALF), RM_SUB);
    t.iy1 = rm_floor_div(maxy - RM_HALF, RM_SUB);
    if (t.ix0 < 0) t.ix0 = 0;
    if (t.iy0 < 0) t.iy0 = 0;
    if (t.ix1 > RM_SIZE - 1) t.ix1 = RM_SIZE - 1;
    if (t.iy1 > RM_SIZE - 1) t.iy1 = RM_SIZE - 1;
    t.valid = area != 0 && t.ix0 <= t.ix1 && t.iy0 <= t.iy1;
    return t;
}

/* A pixel centre exactly on the directed edge (dx,dy) of a counter-clockwise triangle (y up) belongs to the triangle whose
 * LEFT or BOTTOM edge that is: the top-left rule in the flipped image the reference returns.  OpenGL leaves the choice open;
 * this is the rule of the OpenGL implementation the contract was checked against (tests/golden/gl_raster.npz). */
#define RM_OWNS(dx, dy) ((dy) < 0 || ((dy) == 0 && (dx) > 0))

/* edge function for the directed edge (ax,ay)->(bx,by) at point (px,py) plus the
 * ownership rule for points exactly on the edge */
RM_FN int rm_edge(int32_t ax, int32_t ay, int32_t bx, int32_t by, int32_t px, int32_t py, int64_t* w) {
    const int32_t dx = bx - ax, dy = by - ay;
    const int64_t e = (int64_t)dx * (py - ay) - (int64_t)dy * (px - ax);
    *w = e;
    if (e > 0) return 1;
    if (e < 0) return 0;
    return RM_OWNS(dx, dy);
}

/* coverage + barycentric weights of pixel (i,j) (j counts up from the bottom row) */
RM_FN int rm_cover(const rm_tri* t, int i, int j, float* b0, float* b1, float* b2) {
    const int32_t px = i * RM_SUB + RM_HALF, py = j * RM_SUB + RM_HALF;
    int64_t w0, w1, w2;
    const int in0 = rm_edge(t->X1, t->Y1, t->X2, t->Y2, px, py, &w0);
    const int in1 = rm_edge(t->X2, t->Y2, t->X0, t->Y0, px, py, &w1);
    const int in2 = rm_edge(t->X0, t->Y0, t->X1, t->Y1, px, py, &w2);
    if (!(in0 && in1 && in2)) return 0;
    *b0 = (float)w0 / t->farea;
    *b1 = (float)w1 / t->farea;
    *b2 = (float)w2 / t->farea;
    return 1;
}

/* a vertex attribute at a covered pixel: the plane through vertex 0, as a GL rasteriser evaluates it - a constant
 * attribute stays constant bit for bit (coplanar screen-parallel triangles tie exactly and LEQUAL decides; a triangle with
 * one texture coordinate has one texel) */
RM_FN float rm_interp(float b0, float b1, float b2, float a0, float a1, float a2) {
    (void)b0;
    return a0 + (b1 * (a1 - a0) + b2 * (a2 - a0));
}

/* depth-test key: smaller wins; equal depth -> later triangle wins (GL_LEQUAL) */
RM_FN uint64_t rm_key(float z, uint32_t tri) {
    union { float f; uint32_t u; } c;
    c.f = z;
    return ((uint64_t)c.u << 32) | (uint64_t)(0xFFFFFFFFu - tri);
}
#define RM_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull
RM_FN uint32_t rm_key_tri(uint64_t k) { return 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFu); }
RM_FN float rm_key_z(uint64_t k) {
    union { float f; uint32_t u; } c;
    c.u = (uint32_t)(k >> 32);
    return c.f;
}

/* z-buffer value -> the depth plane the network sees (before /255) */
RM_FN int rm_depth_u8(float z) {
    const int t = (int)(255.0 * (double)z);
    return (256 - t) & 255;
}

/* Build-defined "geometry" plane (absent from the reference renderer): flat
 * two-sided head-light shading |n.z| / |n| of the winning triangle in view space, as 0..255.
 * The normal comes from the snapped window coordinates (1/256 px) and the z-buffer values
 * converted to the same unit: 1 z-buffer unit = 1500 model units = 1500 * 256/300 * 256 steps. */
RM_FN int rm_geometry_u8(const rm_tri* t) {
    const float kz = -327680.0f;
    const float ax = (float)(t->X1 - t->X0), ay = (float)(t->Y1 - t->Y0), az = (t->z1 - t->z0) * kz;
    const float bx = (float)(t->X2 - t->X0), by = (float)(t->Y2 - t->Y0), bz = (t->z2 - t->z0) * kz;
    const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
    if (!(len > 0.0f)) return 0;
    const float s = fabsf(nz) / len;
    return (int)(s * 255.0f + 0.5f);
}

/* nearest texel with GL_REPEAT wrapping; image rows are stored top row first while
 * v = 0 is the bottom of the image (vtkJPEGReader / OBJ convention) */
RM_FN int rm_texel(float u, float v, int tw, int th) {
    float uu = u - floorf(u), vv = v - floorf(v);
    int tx = (int)(uu * (float)tw), ty = (int)(vv * (float)th);
    if (tx > tw - 1) tx = tw - 1;
    if (ty > th - 1) ty = th - 1;
    if (tx < 0) tx = 0;
    if (ty < 0) ty = 0;
    return (th - 1 - ty) * tw + tx;
}

#endif
