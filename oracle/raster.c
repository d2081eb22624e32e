/*
 * Oracle: CPU multi-view renderer (TEST INFRASTRUCTURE - see oracle/__init__.py).
 *
 * A plain z-buffer restatement of what the reference obtains from VTK/OpenGL in
 * src/mvlm/utils/render3d.py:114-177 + :191 and src/mvlm/utils/utils3d.py:26-64.
 * The arithmetic itself lives in the third-party `vtk` package (absent here), so
 * this file follows the *specification* written down in DESIGN.md ("Renderer
 * contract") rather than executable reference code - PARITY UNPINNED against VTK at pixel
 * level.  What VTK asks of OpenGL, however, was drawn by a real OpenGL (SwiftShader GLES 3.0,
 * tools/make_gl_golden.py -> tests/golden/gl_raster.npz) and this file is held against that
 * (tests/test_gl_contract.py): fill rule, pixel centres, depth mapping and byte conversion,
 * LEQUAL, clip planes, texel addressing and flip below are the ones that GL shows; what GL
 * leaves to the implementation (sub-pixel bits, guard band) is listed in DESIGN.md section 5.1:
 *   - per pose M = Ry*Rx*Rz (render3d.py:140-144) applied in double, points kept
 *     as float (vtkTransformPolyDataFilter on float points),
 *   - orthographic camera at z=+500 looking down -z, parallel scale 150, 256x256
 *     window (render3d.py:50, :150-152): window x = (x+150)*256/300, same for y,
 *   - depth = (500 - z)/1500 for the (0,1500) clip range (render3d.py:136),
 *   - vertices snapped to 1/256 pixel (2^-bits: oracle_render_bits), pixel centres at
 *     (i+0.5, j+0.5), a centre on a shared edge owned by the triangle whose left / bottom
 *     edge it is, attributes interpolated as the plane through the first vertex, depth
 *     test LEQUAL in draw order (VTK default) i.e. a later triangle replaces an equal depth,
 *   - unlit white material modulated by the nearest texel (utils3d.py:32, :61-64),
 *     white background (render3d.py:54),
 *   - depth plane = (256 - trunc(255*z)) mod 256 (vtkImageShiftScale scale -255 to
 *     unsigned char, render3d.py:73-77), image flipped vertically (:177), /255 (:191).
 * Triangles are drawn one after another into a full-frame z-buffer - deliberately a
 * different organisation from the GPU's tile-binned kernel it checks.
 *
 * build: gcc -O2 -ffp-contract=off -fPIC -shared oracle/raster.c -o oracle/_build/liboracle_raster.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define N 256

typedef struct { int32_t x, y; float z; } SV;

static SV project(const double* m, const float* v, int bits) {
    double x = v[0], y = v[1], z = v[2];
    float xv = (float)((m[0] * x + m[1] * y) + m[2] * z);
    float yv = (float)((m[3] * x + m[4] * y) + m[5] * z);
    float zv = (float)((m[6] * x + m[7] * y) + m[8] * z);
    const float k = 256.0f / 300.0f;
    const float sub = (float)(1 << bits); /* 256: the contract's 1/256 pixel */
    float fx = floorf(((xv + 150.0f) * k) * sub + 0.5f);
    float fy = floorf(((yv + 150.0f) * k) * sub + 0.5f);
    const float lim = (float)(1 << (14 + bits)); /* 16384 pixels */
    if (fx < -lim) fx = -lim;
    if (fx > lim) fx = lim;
    if (fy < -lim) fy = -lim;
    if (fy > lim) fy = lim;
    SV s;
    s.x = (int32_t)fx;
    s.y = (int32_t)fy;
    s.z = (500.0f - zv) / 1500.0f;
    return s;
}

/* signed doubled area of (a, b, p); > 0 when p is left of a->b */
static int64_t orient(SV a, SV b, int32_t px, int32_t py) {
    return (int64_t)(b.x - a.x) * (py - a.y) - (int64_t)(b.y - a.y) * (px - a.x);
}

/* who gets a pixel centre lying exactly on edge a->b of a counter-clockwise triangle (y up): the triangle whose LEFT or
 * BOTTOM edge it is - the top-left rule in the flipped image the reference returns.  OpenGL leaves the choice to the
 * implementation; this is what the GL this contract was checked against does (tests/golden/gl_raster.npz). */
static int owns(SV a, SV b) {
    int32_t dx = b.x - a.x, dy = b.y - a.y;
    return dy < 0 || (dy == 0 && dx > 0);
}

static int32_t fdiv(int32_t a, int32_t b) {
    int32_t q = a / b;
    if ((a % b != 0) && (a < 0)) q -= 1;
    return q;
}

/*
 * verts f32[V,3], uvs f32[V,2] or NULL, tris i32[T,3], tex u8[th,tw,3] or NULL,
 * rot f64[n_views,9] -> out f32[n_views,256,256,4]
 */
/* build-defined geometry plane: |n.z|/|n| of the triangle in view space */
static int geometry_u8(SV a, SV b, SV c) {
    const float kz = -327680.0f; /* z-buffer unit -> 1/256 pixel: -1500 * 256/300 * 256 */
    float ax = (float)(b.x - a.x), ay = (float)(b.y - a.y), az = (b.z - a.z) * kz;
    float bx = (float)(c.x - a.x), by = (float)(c.y - a.y), bz = (c.z - a.z) * kz;
    float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    float len = sqrtf((nx * nx + ny * ny) + nz * nz);
    if (!(len > 0.0f)) return 0;
    return (int)((fabsf(nz) / len) * 255.0f + 0.5f);
}

/* `subpixel_bits`: the vertex lattice is 2^-bits pixel.  The contract (and the HIP rasteriser) is 8 - what GPUs report as
 * GL_SUBPIXEL_BITS; OpenGL only demands >= 4, and the software GL that tests/golden/gl_raster.npz was drawn with
 * (tools/make_gl_golden.py) uses 4, so the comparison with it runs this same code at 4. */
int oracle_render_bits(const float* verts, const float* uvs, int n_verts, const int32_t* tris, int n_tris,
                       const uint8_t* tex, int th, int tw, const double* rot, int n_views, int shading, int subpixel_bits,
                       float* out) {
    const int32_t S = 1 << subpixel_bits, H = S / 2;
    if (subpixel_bits < 1 || subpixel_bits > 8 || (shading == 1 && subpixel_bits != 8)) return 2; /* geometry_u8 is in 1/256 px */
    SV* sv = (SV*)malloc(sizeof(SV) * (size_t)n_verts);
    float* zbuf = (float*)malloc(sizeof(float) * N * N);
    int32_t* owner = (int32_t*)malloc(sizeof(int32_t) * N * N);
    float* bary = (float*)malloc(sizeof(float) * N * N * 3);
    if (!sv || !zbuf || !owner || !bary) return 1;
    for (int view = 0; view < n_views; ++view) {
        const double* m = rot + 9 * view;
        for (int i = 0; i < n_verts; ++i) sv[i] = project(m, verts + 3 * i, subpixel_bits);
        for (int p = 0; p < N * N; ++p) {
            zbuf[p] = 2.0f;
            owner[p] = -1;
        }
        for (int t = 0; t < n_tris; ++t) {
            int ia = tris[3 * t], ib = tris[3 * t + 1], ic = tris[3 * t + 2];
            SV a = sv[ia], b = sv[ib], c = sv[ic];
            int64_t area = orient(a, b, c.x, c.y);
            int flipped = 0;
            if (area == 0) continue;
            if (area < 0) {
                SV s = b; b = c; c = s;
                area = -area;
                flipped = 1;
            }
            int32_t minx = a.x, maxx = a.x, miny = a.y, maxy = a.y;
            if (b.x < minx) minx = b.x;
            if (c.x < minx) minx = c.x;
            if (b.x > maxx) maxx = b.x;
            if (c.x > maxx) maxx = c.x;
            if (b.y < miny) miny = b.y;
            if (c.y < miny) miny = c.y;
            if (b.y > maxy) maxy = b.y;
            if (c.y > maxy) maxy = c.y;
            int i0 = -fdiv(-(minx - H), S), i1 = fdiv(maxx - H, S);
            int j0 = -fdiv(-(miny - H), S), j1 = fdiv(maxy - H, S);
            if (i0 < 0) i0 = 0;
            if (j0 < 0) j0 = 0;
            if (i1 > N - 1) i1 = N - 1;
            if (j1 > N - 1) j1 = N - 1;
            const float fa = (float)area;
            for (int j = j0; j <= j1; ++j)
                for (int i = i0; i <= i1; ++i) {
                    int32_t px = i * S + H, py = j * S + H;
                    int64_t w0 = orient(b, c, px, py), w1 = orient(c, a, px, py), w2 = orient(a, b, px, py);
                    if (w0 < 0 || w1 < 0 || w2 < 0) continue;
                    if (w0 == 0 && !owns(b, c)) continue;
                    if (w1 == 0 && !owns(c, a)) continue;
                    if (w2 == 0 && !owns(a, b)) continue;
                    float b0 = (float)w0 / fa, b1 = (float)w1 / fa, b2 = (float)w2 / fa;
                    /* attributes as a plane through vertex a (what a GL rasteriser evaluates): a constant attribute stays
                     * constant bit for bit, so coplanar screen-parallel triangles tie exactly and LEQUAL decides */
                    float z = a.z + (b1 * (b.z - a.z) + b2 * (c.z - a.z));
                    if (!(z >= 0.0f && z <= 1.0f)) continue;
                    int p = j * N + i;
                    if (z <= zbuf[p]) {
                        zbuf[p] = z;
                        owner[p] = flipped ? -2 - t : t; /* remember the winding fix */
                        bary[3 * p] = b0;
                        bary[3 * p + 1] = b1;
                        bary[3 * p + 2] = b2;
                    }
                }
        }
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i) {
                int p = j * N + i;
                float r = 255.f, g = 255.f, bl = 255.f, z = 1.0f;
                if (owner[p] != -1) {
                    int flipped = owner[p] < -1;
                    int t = flipped ? -2 - owner[p] : owner[p];
                    int ia = tris[3 * t], ib = tris[3 * t + 1], ic = tris[3 * t + 2];
                    if (flipped) { int s = ib; ib = ic; ic = s; }
                    z = zbuf[p];
                    if (shading == 1) {
                        r = g = bl = (float)geometry_u8(sv[ia], sv[ib], sv[ic]);
                    } else if (tex && uvs) {
                        float b0 = bary[3 * p], b1 = bary[3 * p + 1], b2 = bary[3 * p + 2];
                        float u0 = uvs[2 * ia], v0 = uvs[2 * ia + 1];
                        float u = u0 + (b1 * (uvs[2 * ib] - u0) + b2 * (uvs[2 * ic] - u0));
                        float v = v0 + (b1 * (uvs[2 * ib + 1] - v0) + b2 * (uvs[2 * ic + 1] - v0));
                        (void)b0;
                        float uu = u - floorf(u), vv = v - floorf(v);
                        int tx = (int)(uu * (float)tw), ty = (int)(vv * (float)th);
                        if (tx > tw - 1) tx = tw - 1;
                        if (ty > th - 1) ty = th - 1;
                        if (tx < 0) tx = 0;
                        if (ty < 0) ty = 0;
                        const uint8_t* q = tex + ((size_t)(th - 1 - ty) * tw + tx) * 3;
                        r = q[0]; g = q[1]; bl = q[2];
                    }
                }
                int d8 = (256 - (int)(255.0 * (double)z)) & 255;
                float* o = out + (((size_t)view * N + (N - 1 - j)) * N + i) * 4;
                o[0] = r / 255.0f;
                o[1] = g / 255.0f;
                o[2] = bl / 255.0f;
                o[3] = (float)d8 / 255.0f;
            }
    }
    free(sv);
    free(zbuf);
    free(owner);
    free(bary);
    return 0;
}

int oracle_render(const float* verts, const float* uvs, int n_verts, const int32_t* tris, int n_tris,
                  const uint8_t* tex, int th, int tw, const double* rot, int n_views, int shading, float* out) {
    return oracle_render_bits(verts, uvs, n_verts, tris, n_tris, tex, th, tw, rot, n_views, shading, 8, out);
}
