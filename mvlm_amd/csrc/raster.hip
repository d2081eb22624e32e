// Batched multi-view triangle rasteriser: every camera pose of a mesh in one launch set.
// Replaces the reference's per-pose VTK offscreen loop (src/mvlm/utils/render3d.py:139-170:
// CPU re-transform of all vertices + GL draw + two read-backs per pose).
//
// Pipeline (all views at once, nothing returns to the host):
//   1. transform   one thread per (view, vertex): M*v in double -> snapped window coords + z
//   2. classify    one thread per (view, triangle): covered pixel-centre box.  Most triangles of
//                  a 100k-face head are sub-pixel: empty boxes are culled, boxes of <= 16 pixel
//                  centres are resolved on the spot with a 64-bit atomicMin of the
//                  (depth, id) key per covered pixel, larger triangles are counted into
//                  16x16-pixel tiles and appended to the view's "big" list
//   3. scan        one workgroup per view: exclusive prefix sum over its 256 tile counters
//   4. bin fill    big triangles only: ids scattered into the per-tile lists
//   5. tile raster one workgroup per (view, tile), one thread per pixel: starts from the key the
//                  small triangles left, stages the tile's big-triangle list through LDS in
//                  chunks of 256 set-up triangles (each thread sets one up), every pixel walks
//                  the chunk (LDS broadcast reads) and keeps the winning key in a register;
//                  then shades (nearest texel, unlit) and writes its RGBD texel; a tile row is
//                  256 contiguous bytes of the [N,256,256,4] f32 stack.
// The key (depth bits << 32 | ~triangle id) makes the result independent of the order in which
// atomics and tiles run: least depth wins, the later-drawn triangle wins ties (GL_LEQUAL).
// Bandwidth-bound on its REAL traffic, measured (round 4): 0.74 GB through the L2s per 96 views at
// 4.5 TB/s = the 165 us it took - 100 MB of pixels, 160 MB of texels (a 3-byte nearest texel costs a 64-byte line; every
// sample lies in another one), 175 MB for the key plane (clear, atomics, read back), 216 MB of transformed vertices.
// Neither fewer instructions (the 24-bit path below: 3 %) nor more loads in flight per thread (TPT / PPT: nothing) nor
// keeping the keys in LDS with every survivor binned (built: 210 us - a cull pass, a survivor list, a fill pass and a
// per-tile gather cost more than the plane they save) beat this structure; fewer bytes per vertex do.
#include "common.h"
#include "raster_math.h"

namespace {

constexpr int TILES = RM_TILES * RM_TILES;  // 256 per view

// Workgroup -> (view, chunk of that view's work) so that every view is worked on by ONE XCD: consecutive block ids
// go round-robin over the 8 XCDs, each with its own L2; a view's transformed vertices (0.8 MB), key plane (0.5 MB) and
// bins then live in one L2 instead of being fetched by all eight (speed heuristic only: any placement is correct).
// The grid holds ceil(n_views / 8) * 8 * chunks_per_view workgroups; surplus ones return at once.
__device__ inline bool view_chunk(int chunks_per_view, int n_views, int* view, int* chunk) {
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    *view = (j / chunks_per_view) * 8 + xcd;
    *chunk = j % chunks_per_view;
    return *view < n_views;
}
inline unsigned view_chunk_grid(int chunks_per_view, int n_views) { return unsigned((n_views + 7) / 8 * 8) * unsigned(chunks_per_view); }

// a transformed vertex as stored (rm_vert without its padding word: a quarter less traffic on the largest scratch array)
struct vert12 {
    int32_t X, Y;
    float z;
};
__device__ inline rm_vert load_vert(const vert12* __restrict__ p, int i) {
    const vert12 v = p[i];
    rm_vert o;
    o.X = v.X;
    o.Y = v.Y;
    o.z = v.z;
    o.pad = 0.f;
    return o;
}

__global__ void transform_kernel(const float* __restrict__ verts, int n_verts, const double* __restrict__ rot,
                                 int n_views, int sub_bits, vert12* __restrict__ tv) {
    int view, chunk;
    if (!view_chunk((n_verts + 255) / 256, n_views, &view, &chunk)) return;
    const int v = chunk * 256 + int(threadIdx.x);
    if (v >= n_verts) return;
    double m[9];
    for (int k = 0; k < 9; ++k) m[k] = rot[view * 9 + k];
    const rm_vert o = rm_transform(m, verts[3 * v], verts[3 * v + 1], verts[3 * v + 2], sub_bits);
    vert12 w;
    w.X = o.X;
    w.Y = o.Y;
    w.z = o.z;
    tv[size_t(view) * n_verts + v] = w;
}

__device__ inline rm_tri load_tri(const vert12* tvv, const int32_t* tris, int t) {
    const int a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    return rm_setup(load_vert(tvv, a), load_vert(tvv, b), load_vert(tvv, c));
}

// ---- 24-bit path ---------------------------------------------------------------------------------------------------------
// rm_setup / rm_cover evaluate the area and the three edge functions in 64-bit integers (window coordinates reach
// +-2^22 sub-pixel steps), and a 64-bit or a full 32-bit integer multiply runs at a quarter of the vector rate.  A
// triangle whose vertices lie within 2^14 steps (64 pixels) of each other
// in x and in y - every triangle of a dense scan - has |dx|, |dy| < 2^14 and, for a pixel centre inside its bounding box,
// |px - ax|, |py - ay| < 2^14: every product is below 2^28 and fits the full-rate 24-bit multiply (v_mul_i32_i24), every
// edge value and the area are below 2^29.  The integers are THE SAME as rm_setup's / rm_cover's, so are the floats made
// from them: a kernel may take either path per triangle and the image does not change by a bit (oracle/raster.c, the
// tests' checker, knows only the 64-bit form).  Worth 3-5 % of a render.
constexpr int RM_SMALL_EXTENT = 1 << 14;

struct tri24 {
    int32_t X0, Y0, X1, Y1, X2, Y2;  // after the winding swap (vertices 1 and 2)
    float z0, z1, z2;
    float farea;
    int32_t swapped;
};

__device__ inline bool small_extent(const rm_vert& a, const rm_vert& b, const rm_vert& c) {
    const int32_t minx = min(a.X, min(b.X, c.X)), maxx = max(a.X, max(b.X, c.X));
    const int32_t miny = min(a.Y, min(b.Y, c.Y)), maxy = max(a.Y, max(b.Y, c.Y));
    return maxx - minx < RM_SMALL_EXTENT && maxy - miny < RM_SMALL_EXTENT;
}

// rm_setup without the bounding box; false for a zero-area triangle
__device__ inline bool setup24(rm_vert a, rm_vert b, rm_vert c, tri24* t) {
    int32_t area = __mul24(b.X - a.X, c.Y - a.Y) - __mul24(b.Y - a.Y, c.X - a.X);
    t->swapped = 0;
    if (area < 0) {
        const rm_vert s = b;
        b = c;
        c = s;
        area = -area;
        t->swapped = 1;
    }
    t->X0 = a.X; t->Y0 = a.Y; t->X1 = b.X; t->Y1 = b.Y; t->X2 = c.X; t->Y2 = c.Y;
    t->z0 = a.z; t->z1 = b.z; t->z2 = c.z;
    t->farea = float(area);
    return area != 0;
}

// the geometry plane's shade (rm_geometry_u8 reads the swapped vertices and depths only)
__device__ inline int geometry24(const tri24* t) {
    rm_tri r = {};
    r.X0 = t->X0; r.Y0 = t->Y0; r.X1 = t->X1; r.Y1 = t->Y1; r.X2 = t->X2; r.Y2 = t->Y2;
    r.z0 = t->z0; r.z1 = t->z1; r.z2 = t->z2;
    return rm_geometry_u8(&r);
}

__device__ inline int edge24(int32_t ax, int32_t ay, int32_t bx, int32_t by, int32_t px, int32_t py, int32_t* w) {
    const int32_t dx = bx - ax, dy = by - ay;
    const int32_t e = __mul24(dx, py - ay) - __mul24(dy, px - ax);
    *w = e;
    if (e > 0) return 1;
    if (e < 0) return 0;
    return RM_OWNS(dx, dy);
}

// rm_cover for a pixel centre inside the triangle's bounding box (T: tri24, or an rm_tri of small extent)
template <class T>
__device__ inline int cover24(const T* t, int i, int j, float* b0, float* b1, float* b2) {
    const int32_t px = i * RM_SUB + RM_HALF, py = j * RM_SUB + RM_HALF;
    int32_t w0, w1, w2;
    const int in0 = edge24(t->X1, t->Y1, t->X2, t->Y2, px, py, &w0);
    const int in1 = edge24(t->X2, t->Y2, t->X0, t->Y0, px, py, &w1);
    const int in2 = edge24(t->X0, t->Y0, t->X1, t->Y1, px, py, &w2);
    if (!(in0 && in1 && in2)) return 0;
    *b0 = float(w0) / t->farea;
    *b1 = float(w1) / t->farea;
    *b2 = float(w2) / t->farea;
    return 1;
}

constexpr int SMALL_PIXELS = 16;  // triangles covering at most this many pixel centres skip the bins

// Classify every (view, triangle): cull / resolve small ones with atomics / count big ones.
// Order of the tests: the pixel-centre box first (integer shifts: most triangles of a dense scan lie between pixel
// centres and leave here), then the area; triangles of small extent take the 24-bit path (above).
// Returns true for a big triangle (the caller appends it to the view's list).
__device__ inline bool classify_one(int view, int t, const rm_vert& va, const rm_vert& vb, const rm_vert& vc,
                                    unsigned long long* __restrict__ keys, int* __restrict__ counts) {
    // rm_setup's box: pixel i has its centre at 256 i + 128; ceil / floor of a division by 256 as arithmetic shifts
    const int32_t minx = min(va.X, min(vb.X, vc.X)), maxx = max(va.X, max(vb.X, vc.X));
    const int32_t miny = min(va.Y, min(vb.Y, vc.Y)), maxy = max(va.Y, max(vb.Y, vc.Y));
    const int ix0 = max((minx + (RM_SUB - 1 - RM_HALF)) >> 8, 0), ix1 = min((maxx - RM_HALF) >> 8, RM_SIZE - 1);
    const int iy0 = max((miny + (RM_SUB - 1 - RM_HALF)) >> 8, 0), iy1 = min((maxy - RM_HALF) >> 8, RM_SIZE - 1);
    static_assert(RM_SUB == 256 && RM_HALF == 128, "the shifts above divide by RM_SUB");
    if (ix0 > ix1 || iy0 > iy1) return false;
    const int w = ix1 - ix0 + 1, h = iy1 - iy0 + 1;
    const bool small = maxx - minx < RM_SMALL_EXTENT && maxy - miny < RM_SMALL_EXTENT;
    if (w * h <= SMALL_PIXELS) {
        unsigned long long* kv = keys + size_t(view) * RM_SIZE * RM_SIZE;
        if (small) {
            tri24 tr;
            if (!setup24(va, vb, vc, &tr)) return false;
            for (int j = iy0; j <= iy1; ++j)
                for (int ii = ix0; ii <= ix1; ++ii) {
                    float b0, b1, b2;
                    if (!cover24(&tr, ii, j, &b0, &b1, &b2)) continue;
                    const float z = rm_interp(b0, b1, b2, tr.z0, tr.z1, tr.z2);
                    if (!(z >= 0.0f && z <= 1.0f)) continue;  // near / far clip (render3d.py:136)
                    atomicMin(&kv[j * RM_SIZE + ii], (unsigned long long)rm_key(z, uint32_t(t)));
                }
        } else {  // a long sliver that crosses the window's edge: few pixel centres inside, vertices far apart
            const rm_tri tr = rm_setup(va, vb, vc);
            if (!tr.valid) return false;
            for (int j = tr.iy0; j <= tr.iy1; ++j)
                for (int ii = tr.ix0; ii <= tr.ix1; ++ii) {
                    float b0, b1, b2;
                    if (!rm_cover(&tr, ii, j, &b0, &b1, &b2)) continue;
                    const float z = rm_interp(b0, b1, b2, tr.z0, tr.z1, tr.z2);
                    if (!(z >= 0.0f && z <= 1.0f)) continue;
                    atomicMin(&kv[j * RM_SIZE + ii], (unsigned long long)rm_key(z, uint32_t(t)));
                }
        }
        return false;
    }
    // big: valid means a non-zero area (rm_setup)
    if (small) {
        if (__mul24(vb.X - va.X, vc.Y - va.Y) == __mul24(vb.Y - va.Y, vc.X - va.X)) return false;
    } else if (int64_t(vb.X - va.X) * (vc.Y - va.Y) == int64_t(vb.Y - va.Y) * (vc.X - va.X)) {
        return false;
    }
    const int tx0 = ix0 / RM_TILE, tx1 = ix1 / RM_TILE, ty0 = iy0 / RM_TILE, ty1 = iy1 / RM_TILE;
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&counts[view * TILES + ty * RM_TILES + tx], 1);
    return true;
}

// A thread takes TPT triangles and has all their loads (indices, then three vertices each) in flight together.  Measured
// (tools/raster_bench.py, 96 views of the bench mesh): TPT 1 / 2 / 4 = 164 / 161 / 163 us per render - the pass does not
// wait for latency (without its key atomics it still takes 43 of its 59 us: 9.5 M threads x four 12-byte gathers through
// the L1s), so TPT stays 1.  The ids of a wave's big triangles are appended to the view's list with ONE atomicAdd per
// wave: a coarse mesh - every triangle big - made 3 000 returning atomics on one counter per view, 54 us for 3 042
// triangles (now 6).
constexpr int CLASSIFY_TPT = 1;
template <int TPT>
__global__ __launch_bounds__(256) void classify_kernel(const vert12* __restrict__ tv, const int32_t* __restrict__ tris, int n_verts,
                                                       int n_tris, int n_views, unsigned long long* __restrict__ keys,
                                                       int* __restrict__ counts, int* __restrict__ n_big, int* __restrict__ big_list) {
    int view, chunk;
    if (!view_chunk((n_tris + 256 * TPT - 1) / (256 * TPT), n_views, &view, &chunk)) return;
    const vert12* const tvv = tv + size_t(view) * n_verts;
    int t[TPT];
    int ia[TPT], ib[TPT], ic[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        t[k] = (chunk * TPT + k) * 256 + int(threadIdx.x);
        const int tt = t[k] < n_tris ? t[k] : 0;  // (unconditional loads: all of them leave together)
        ia[k] = tris[3 * tt];
        ib[k] = tris[3 * tt + 1];
        ic[k] = tris[3 * tt + 2];
    }
    rm_vert va[TPT], vb[TPT], vc[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        va[k] = load_vert(tvv, ia[k]);
        vb[k] = load_vert(tvv, ib[k]);
        vc[k] = load_vert(tvv, ic[k]);
    }
    const int lane = int(threadIdx.x) & 63;
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const bool big = t[k] < n_tris && classify_one(view, t[k], va[k], vb[k], vc[k], keys, counts);
        const unsigned long long m = __ballot(big);
        if (m) {  // (wave-uniform)
            const int leader = __ffsll(static_cast<long long>(m)) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&n_big[view], __popcll(m));
            base = __shfl(base, leader);
            if (big) big_list[size_t(view) * n_tris + base + __popcll(m & ((1ull << lane) - 1ull))] = t[k];
        }
    }
}

// Scatter the big triangles' ids into the per-tile lists (offsets from the scan).  FILL_WGS workgroups per view stride
// over the view's big list: a grid sized for "every triangle is big" spent 11 us on dispatching 37 000 workgroups that
// found nothing to do (a dense scan has a handful of big triangles per view).
constexpr int FILL_WGS = 32;
__global__ void bin_fill_kernel(const vert12* __restrict__ tv, const int32_t* __restrict__ tris, int n_verts,
                                int n_tris, int n_views, const int* __restrict__ n_big,
                                const int* __restrict__ big_list, const int* __restrict__ offsets,
                                int* __restrict__ cursors, int* __restrict__ bins, int cap,
                                int* __restrict__ overflow) {
    int view, wg;
    if (!view_chunk(FILL_WGS, n_views, &view, &wg)) return;
    const int n = n_big[view];
    for (int k = wg * 256 + int(threadIdx.x); k < n; k += FILL_WGS * 256) {
        const int t = big_list[size_t(view) * n_tris + k];
        const rm_tri tr = load_tri(tv + size_t(view) * n_verts, tris, t);
        const int tx0 = tr.ix0 / RM_TILE, tx1 = tr.ix1 / RM_TILE, ty0 = tr.iy0 / RM_TILE, ty1 = tr.iy1 / RM_TILE;
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) {
                const int tile = view * TILES + ty * RM_TILES + tx;
                const int pos = offsets[tile] + atomicAdd(&cursors[tile], 1);
                if (pos < cap)
                    bins[size_t(view) * cap + pos] = t;
                else
                    *overflow = 1;
            }
    }
}

__global__ void scan_kernel(const int* __restrict__ counts, int* __restrict__ offsets, int cap,
                            int* __restrict__ overflow) {
    __shared__ int s[TILES];
    const int view = blockIdx.x, t = threadIdx.x;
    s[t] = counts[view * TILES + t];
    __syncthreads();
    for (int d = 1; d < TILES; d <<= 1) {
        const int v = t >= d ? s[t - d] : 0;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    offsets[view * TILES + t] = s[t] - counts[view * TILES + t];
    if (t == TILES - 1 && s[t] > cap) *overflow = 1;
}

// One workgroup per (view, 16x16-pixel tile), one thread per pixel.  More pixels per thread with the loads of a stage
// (key, triangle, vertices and texture coordinates, texel) in flight together were built and measured (2 and 4 pixels:
// 161 / 163 us per render against 164): the kernel does not wait for latency, it moves its bytes at the rate the memory
// system gives - without the texel fetch 43 of its 76 us, without any shading 37 = key plane in, pixels out at 4 TB/s.
__global__ __launch_bounds__(256) void tile_kernel(const vert12* __restrict__ tv, const int32_t* __restrict__ tris,
                                                   const float* __restrict__ uvs, const uint8_t* __restrict__ tex,
                                                   int tex_w, int tex_h, int n_verts, const int* __restrict__ counts,
                                                   const int* __restrict__ offsets, const int* __restrict__ bins,
                                                   int cap, unsigned long long* __restrict__ keys,
                                                   int shading, int n_views, const int* __restrict__ overflow,
                                                   int* __restrict__ overflow_host, float* __restrict__ out) {
    __shared__ rm_tri s_tri[256];  // (valid == 2: small extent, the 24-bit edge functions apply)
    __shared__ int s_id[256];
    int view, tile;
    if (!view_chunk(TILES, n_views, &view, &tile)) return;
    const int vt = view * TILES + tile;
    const int tid = threadIdx.x;
    const int i = (tile % RM_TILES) * RM_TILE + (tid & (RM_TILE - 1));
    const int j = (tile / RM_TILES) * RM_TILE + (tid >> 4);
    const vert12* const tvv = tv + size_t(view) * n_verts;
    const int n = min(counts[vt], cap - offsets[vt]);
    const int* const list = bins + size_t(view) * cap + offsets[vt];

    unsigned long long* const key_slot = keys + (size_t(view) * RM_SIZE + j) * RM_SIZE + i;
    uint64_t best = *key_slot;  // what the small triangles left
    // The key plane is handed back EMPTY: this kernel reads every slot of the rendered views exactly once, so it also clears
    // what classify dirtied - instead of a 67 MB fill in front of every render (14 us + a launch gap at 128 views)
    if (best != RM_KEY_EMPTY) *key_slot = RM_KEY_EMPTY;
    // and the first workgroup carries the overflow flag of the kernels before it to the host's pinned word (no copy node)
    if (blockIdx.x == 0 && tid == 0) {
        __atomic_store_n(overflow_host, *overflow, __ATOMIC_RELAXED);
        __threadfence_system();
    }

    // ---- phase B: every pixel walks the tile's big triangles, set up into LDS a chunk at a time ----
    for (int base = 0; base < n; base += 256) {
        const int m = min(256, n - base);
        __syncthreads();
        if (tid < m) {
            const int t = list[base + tid];
            const rm_vert a = load_vert(tvv, tris[3 * t]), b = load_vert(tvv, tris[3 * t + 1]), c = load_vert(tvv, tris[3 * t + 2]);
            rm_tri tr = rm_setup(a, b, c);
            if (tr.valid && small_extent(a, b, c)) tr.valid = 2;
            s_id[tid] = t;
            s_tri[tid] = tr;
        }
        __syncthreads();
        for (int k = 0; k < m; ++k) {
            const rm_tri* t = &s_tri[k];
            if (i < t->ix0 || i > t->ix1 || j < t->iy0 || j > t->iy1) continue;
            float b0, b1, b2;
            if (!(t->valid == 2 ? cover24(t, i, j, &b0, &b1, &b2) : rm_cover(t, i, j, &b0, &b1, &b2))) continue;
            const float z = rm_interp(b0, b1, b2, t->z0, t->z1, t->z2);
            if (!(z >= 0.0f && z <= 1.0f)) continue;  // near / far clip (render3d.py:136)
            const uint64_t key = rm_key(z, uint32_t(s_id[k]));
            best = key < best ? key : best;
        }
    }

    // ---- phase C: shade the winner ----
    float4 px = make_float4(1.f, 1.f, 1.f, float(rm_depth_u8(1.0f)) / 255.0f);  // white background, far plane
    if (best != RM_KEY_EMPTY) {
        const int t = int(rm_key_tri(best));
        int a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
        const rm_vert va = load_vert(tvv, a), vb = load_vert(tvv, b), vc = load_vert(tvv, c);
        // the winner's barycentric weights at this pixel (it is covered: inside the triangle's box)
        float b0 = 0.f, b1 = 0.f, b2 = 0.f;
        bool swapped;
        float r = 255.f, g = 255.f, bl = 255.f;
        if (small_extent(va, vb, vc)) {
            tri24 tr;
            setup24(va, vb, vc, &tr);
            swapped = tr.swapped != 0;
            cover24(&tr, i, j, &b0, &b1, &b2);
            if (shading == 1) r = g = bl = float(geometry24(&tr));
        } else {
            const rm_tri tr = rm_setup(va, vb, vc);
            swapped = tr.swapped != 0;
            rm_cover(&tr, i, j, &b0, &b1, &b2);
            if (shading == 1) r = g = bl = float(rm_geometry_u8(&tr));
        }
        if (swapped) {
            const int s = b;
            b = c;
            c = s;
        }
        if (shading != 1 && tex && uvs) {
            const float u = rm_interp(b0, b1, b2, uvs[2 * a], uvs[2 * b], uvs[2 * c]);
            const float v = rm_interp(b0, b1, b2, uvs[2 * a + 1], uvs[2 * b + 1], uvs[2 * c + 1]);
            // one (unaligned) 4-byte load per texel: the buffer carries 4 spare bytes behind the last one (api.hip)
            uint32_t rgb;
            __builtin_memcpy(&rgb, tex + size_t(rm_texel(u, v, tex_w, tex_h)) * 3, 4);
            r = float(rgb & 255u);
            g = float((rgb >> 8) & 255u);
            bl = float((rgb >> 16) & 255u);
        }
        px = make_float4(r / 255.0f, g / 255.0f, bl / 255.0f, float(rm_depth_u8(rm_key_z(best))) / 255.0f);
    }
    // np.flip(axis=1): GL row j (bottom-up) is image row 255 - j (render3d.py:177)
    reinterpret_cast<float4*>(out)[(size_t(view) * RM_SIZE + (RM_SIZE - 1 - j)) * RM_SIZE + i] = px;
}

}  // namespace

extern "C" int mvlm_render(mvlm_ctx* ctx, const mvlm_mesh* mesh, const double* rot_host, int n_views, float* out_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, mesh && rot_host && out_dev && n_views > 0, "render: bad arguments");
    MVLM_REQUIRE(ctx, mesh->n_verts > 0 && mesh->n_tris > 0, "render: empty mesh");
    if (mvlm_mesh_wait_ready(ctx, mesh, ctx->stream)) return 1;  // the upload runs on a stream of its own
    const int V = mesh->n_verts, T = mesh->n_tris;
    const int cap = 4 * T + 16384;  // tile-list entries per view; larger lists raise an error
    auto* tv = static_cast<vert12*>(ctx->get_scratch("raster.tv", size_t(n_views) * V * sizeof(vert12)));
    auto* rot = static_cast<double*>(ctx->get_scratch("raster.rot", size_t(n_views) * 9 * sizeof(double)));
    // counts | cursors | n_big | overflow in one block so a single memset clears them
    const size_t n_ctr = size_t(n_views) * TILES;
    const size_t ctr_ints = 2 * n_ctr + n_views + 4;
    auto* ctr = static_cast<int*>(ctx->get_scratch("raster.ctr", ctr_ints * sizeof(int)));
    auto* offsets = static_cast<int*>(ctx->get_scratch("raster.off", n_ctr * sizeof(int)));
    auto* bins = static_cast<int*>(ctx->get_scratch("raster.bins", size_t(n_views) * cap * sizeof(int)));
    auto* big_list = static_cast<int*>(ctx->get_scratch("raster.big", size_t(n_views) * T * sizeof(int)));
    const size_t key_bytes = size_t(n_views) * RM_SIZE * RM_SIZE * sizeof(unsigned long long);
    auto* keys = static_cast<unsigned long long*>(ctx->get_scratch("raster.keys", key_bytes));
    MVLM_REQUIRE(ctx, tv && rot && ctr && offsets && bins && big_list && keys, "render: scratch allocation failed");
    int* counts = ctr;
    int* cursors = ctr + n_ctr;
    int* n_big = ctr + 2 * n_ctr;
    int* overflow = n_big + n_views;
    MVLM_CHECK_HIP(ctx, hipMemcpyAsync(rot, rot_host, size_t(n_views) * 9 * sizeof(double), hipMemcpyHostToDevice,
                                       ctx->stream));
    MVLM_CHECK_HIP(ctx, hipMemsetAsync(ctr, 0, ctr_ints * sizeof(int), ctx->stream));
    // The key plane is EMPTY between renders (tile_kernel clears what it reads); it is filled here only when the buffer is new
    // or a render did not run to its end.  `clean_cap` = the capacity it was last known clean at (a reallocation only grows it).
    const size_t keys_cap = ctx->scratch["raster.keys"].second;
    if (ctx->raster_keys_clean_cap != keys_cap)
        MVLM_CHECK_HIP(ctx, hipMemsetAsync(keys, 0xFF, keys_cap, ctx->stream));  // RM_KEY_EMPTY everywhere
    ctx->raster_keys_clean_cap = 0;  // (until this call has enqueued its tile kernel)
    if (!ctx->render_overflow_host)
        MVLM_CHECK_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->render_overflow_host), sizeof(int)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->render_profiling) {  // HIP events on the launch stream around the five kernels of this call
        if (ctx->render_event_cursor + 2 > ctx->render_events.size()) ctx->render_events.resize(ctx->render_event_cursor + 2, nullptr);
        for (int k = 0; k < 2; ++k) {
            hipEvent_t& e = ctx->render_events[ctx->render_event_cursor + k];
            if (!e) MVLM_CHECK_HIP(ctx, hipEventCreate(&e));
        }
        e0 = ctx->render_events[ctx->render_event_cursor];
        e1 = ctx->render_events[ctx->render_event_cursor + 1];
        ctx->render_event_cursor += 2;
        MVLM_CHECK_HIP(ctx, hipEventRecord(e0, ctx->stream));
    }
    hipLaunchKernelGGL(transform_kernel, dim3(view_chunk_grid((V + 255) / 256, n_views)), dim3(256), 0, ctx->stream,
                       mesh->verts, V, rot, n_views, ctx->render_subpixel_bits, tv);
    hipLaunchKernelGGL(classify_kernel<CLASSIFY_TPT>, dim3(view_chunk_grid((T + 256 * CLASSIFY_TPT - 1) / (256 * CLASSIFY_TPT), n_views)), dim3(256), 0,
                       ctx->stream, tv, mesh->tris, V, T, n_views, keys, counts, n_big, big_list);
    hipLaunchKernelGGL(scan_kernel, dim3(n_views), dim3(TILES), 0, ctx->stream, counts, offsets, cap, overflow);
    hipLaunchKernelGGL(bin_fill_kernel, dim3(view_chunk_grid(FILL_WGS, n_views)), dim3(256), 0, ctx->stream, tv,
                       mesh->tris, V, T, n_views, n_big, big_list, offsets, cursors, bins, cap, overflow);
    hipLaunchKernelGGL(tile_kernel, dim3(view_chunk_grid(TILES, n_views)), dim3(256), 0, ctx->stream, tv, mesh->tris,
                       mesh->uvs, mesh->tex, mesh->tex_w, mesh->tex_h, V, counts, offsets, bins, cap, keys,
                       ctx->render_shading, n_views, overflow, ctx->render_overflow_host, out_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    ctx->raster_keys_clean_cap = keys_cap;
    if (e1) {
        MVLM_CHECK_HIP(ctx, hipEventRecord(e1, ctx->stream));
        ctx->render_prof.push_back({n_views, V, T, e0, e1});
    }
    // (the overflow flag has travelled to pinned host memory inside the tile kernel; mvlm_render_check examines it after the
    // caller's own synchronisation point)
    return 0;
}

// The rotations of the last mvlm_render on this context, f64[n_views, 9] on the device (its scratch: valid until the next
// render): the estimator's rays of the same views need the same table (estimator3d.py:57) and read it from here.
extern "C" int mvlm_render_rotations_dev(mvlm_ctx* ctx, const double** rot_dev) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    MVLM_REQUIRE(ctx, rot_dev, "render_rotations_dev: null pointer");
    const auto it = ctx->scratch.find("raster.rot");
    *rot_dev = it == ctx->scratch.end() ? nullptr : static_cast<const double*>(it->second.first);
    MVLM_REQUIRE(ctx, *rot_dev, "render_rotations_dev: nothing has been rendered on this context");
    return 0;
}

extern "C" int mvlm_render_check(mvlm_ctx* ctx) {
    MVLM_ENTER(ctx);
    MVLM_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MVLM_REQUIRE(ctx, !ctx->render_overflow_host || *ctx->render_overflow_host == 0,
                 "render: per-view tile lists overflowed (mesh has too many screen-filling triangles)");
    return 0;
}

extern "C" int mvlm_render_set_profiling(mvlm_ctx* ctx, int enabled) {
    MVLM_ENTER(ctx);
    ctx->render_profiling = enabled != 0;
    ctx->render_prof.clear();
    ctx->render_event_cursor = 0;
    return 0;
}

extern "C" int mvlm_render_get_profile(mvlm_ctx* ctx, int32_t* n_views, int32_t* n_verts, int32_t* n_tris, float* ms,
                                       int cap) {
    std::lock_guard<std::mutex> lk(ctx->mu);  // returns a record count, -1 on failure; clears the records
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
    int n = 0;
    for (const auto& r : ctx->render_prof) {
        if (n >= cap) break;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -1;
        n_views[n] = r.n_views;
        n_verts[n] = r.n_verts;
        n_tris[n] = r.n_tris;
        ms[n] = t;
        ++n;
    }
    ctx->render_prof.clear();
    ctx->render_event_cursor = 0;
    return n;
}

extern "C" int mvlm_set_render_subpixel_bits(mvlm_ctx* ctx, int bits) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    MVLM_REQUIRE(ctx, bits >= 4 && bits <= 8, "render: subpixel bits must be 4..8 (GL_SUBPIXEL_BITS of the OpenGL to match; 8 = GPUs)");
    ctx->render_subpixel_bits = bits;
    return 0;
}

extern "C" int mvlm_set_render_shading(mvlm_ctx* ctx, int shading) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    MVLM_REQUIRE(ctx, shading == 0 || shading == 1, "render: shading must be 0 (unlit texture) or 1 (geometry)");
    ctx->render_shading = shading;
    return 0;
}
