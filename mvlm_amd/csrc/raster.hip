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
// HBM-bound by design: per view it reads the mesh once per pass and writes 1 MiB.
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

__global__ void transform_kernel(const float* __restrict__ verts, int n_verts, const double* __restrict__ rot,
                                 int n_views, rm_vert* __restrict__ tv) {
    int view, chunk;
    if (!view_chunk((n_verts + 255) / 256, n_views, &view, &chunk)) return;
    const int v = chunk * 256 + int(threadIdx.x);
    if (v >= n_verts) return;
    double m[9];
    for (int k = 0; k < 9; ++k) m[k] = rot[view * 9 + k];
    tv[size_t(view) * n_verts + v] = rm_transform(m, verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]);
}

__device__ inline rm_tri load_tri(const rm_vert* tvv, const int32_t* tris, int t) {
    const int a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    return rm_setup(tvv[a], tvv[b], tvv[c]);
}

constexpr int SMALL_PIXELS = 16;  // triangles covering at most this many pixel centres skip the bins

// Classify every (view, triangle): cull / resolve small ones with atomics / count big ones.
__global__ void classify_kernel(const rm_vert* __restrict__ tv, const int32_t* __restrict__ tris, int n_verts,
                                int n_tris, int n_views, unsigned long long* __restrict__ keys,
                                int* __restrict__ counts, int* __restrict__ n_big, int* __restrict__ big_list) {
    int view, chunk;
    if (!view_chunk((n_tris + 255) / 256, n_views, &view, &chunk)) return;
    const int t = chunk * 256 + int(threadIdx.x);
    if (t >= n_tris) return;
    const rm_tri tr = load_tri(tv + size_t(view) * n_verts, tris, t);
    if (!tr.valid) return;
    const int w = tr.ix1 - tr.ix0 + 1, h = tr.iy1 - tr.iy0 + 1;
    if (w * h <= SMALL_PIXELS) {
        unsigned long long* kv = keys + size_t(view) * RM_SIZE * RM_SIZE;
        for (int j = tr.iy0; j <= tr.iy1; ++j)
            for (int ii = tr.ix0; ii <= tr.ix1; ++ii) {
                float b0, b1, b2;
                if (!rm_cover(&tr, ii, j, &b0, &b1, &b2)) continue;
                const float z = rm_interp(b0, b1, b2, tr.z0, tr.z1, tr.z2);
                if (!(z >= 0.0f && z <= 1.0f)) continue;  // near / far clip (render3d.py:136)
                atomicMin(&kv[j * RM_SIZE + ii], (unsigned long long)rm_key(z, uint32_t(t)));
            }
        return;
    }
    big_list[size_t(view) * n_tris + atomicAdd(&n_big[view], 1)] = t;
    const int tx0 = tr.ix0 / RM_TILE, tx1 = tr.ix1 / RM_TILE, ty0 = tr.iy0 / RM_TILE, ty1 = tr.iy1 / RM_TILE;
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&counts[view * TILES + ty * RM_TILES + tx], 1);
}

// Scatter the big triangles' ids into the per-tile lists (offsets from the scan).
__global__ void bin_fill_kernel(const rm_vert* __restrict__ tv, const int32_t* __restrict__ tris, int n_verts,
                                int n_tris, int n_views, const int* __restrict__ n_big,
                                const int* __restrict__ big_list, const int* __restrict__ offsets,
                                int* __restrict__ cursors, int* __restrict__ bins, int cap,
                                int* __restrict__ overflow) {
    int view, chunk;
    if (!view_chunk((n_tris + 255) / 256, n_views, &view, &chunk)) return;
    const int k = chunk * 256 + int(threadIdx.x);
    if (k >= n_big[view]) return;
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

__global__ __launch_bounds__(256) void tile_kernel(const rm_vert* __restrict__ tv, const int32_t* __restrict__ tris,
                                                   const float* __restrict__ uvs, const uint8_t* __restrict__ tex,
                                                   int tex_w, int tex_h, int n_verts, const int* __restrict__ counts,
                                                   const int* __restrict__ offsets, const int* __restrict__ bins,
                                                   int cap, const unsigned long long* __restrict__ keys,
                                                   int shading, int n_views, float* __restrict__ out) {
    __shared__ rm_tri s_tri[256];
    __shared__ int s_id[256];
    int view, tile;
    if (!view_chunk(TILES, n_views, &view, &tile)) return;
    const int vt = view * TILES + tile;
    const int tid = threadIdx.x;
    const int i = (tile % RM_TILES) * RM_TILE + (tid & (RM_TILE - 1));
    const int j = (tile / RM_TILES) * RM_TILE + (tid >> 4);
    const rm_vert* tvv = tv + size_t(view) * n_verts;
    const int n = min(counts[vt], cap - offsets[vt]);
    const int* list = bins + size_t(view) * cap + offsets[vt];

    uint64_t best = keys[(size_t(view) * RM_SIZE + j) * RM_SIZE + i];  // what the small triangles left
    for (int base = 0; base < n; base += 256) {
        const int m = min(256, n - base);
        __syncthreads();
        if (tid < m) {
            const int t = list[base + tid];
            s_id[tid] = t;
            s_tri[tid] = load_tri(tvv, tris, t);
        }
        __syncthreads();
        for (int k = 0; k < m; ++k) {
            const rm_tri* t = &s_tri[k];
            if (i < t->ix0 || i > t->ix1 || j < t->iy0 || j > t->iy1) continue;
            float b0, b1, b2;
            if (!rm_cover(t, i, j, &b0, &b1, &b2)) continue;
            const float z = rm_interp(b0, b1, b2, t->z0, t->z1, t->z2);
            if (!(z >= 0.0f && z <= 1.0f)) continue;  // near / far clip (render3d.py:136)
            const uint64_t key = rm_key(z, uint32_t(s_id[k]));
            best = key < best ? key : best;
        }
    }

    float4 px = make_float4(1.f, 1.f, 1.f, float(rm_depth_u8(1.0f)) / 255.0f);  // white background, far plane
    if (best != RM_KEY_EMPTY) {
        const int t = int(rm_key_tri(best));
        int a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
        const rm_tri tr = rm_setup(tvv[a], tvv[b], tvv[c]);
        if (tr.swapped) {
            const int s = b;
            b = c;
            c = s;
        }
        float b0 = 0.f, b1 = 0.f, b2 = 0.f;
        rm_cover(&tr, i, j, &b0, &b1, &b2);
        float r = 255.f, g = 255.f, bl = 255.f;
        if (shading == 1) {
            r = g = bl = float(rm_geometry_u8(&tr));
        } else if (tex && uvs) {
            const float u = rm_interp(b0, b1, b2, uvs[2 * a], uvs[2 * b], uvs[2 * c]);
            const float v = rm_interp(b0, b1, b2, uvs[2 * a + 1], uvs[2 * b + 1], uvs[2 * c + 1]);
            const uint8_t* tp = tex + size_t(rm_texel(u, v, tex_w, tex_h)) * 3;
            r = float(tp[0]);
            g = float(tp[1]);
            bl = float(tp[2]);
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
    auto* tv = static_cast<rm_vert*>(ctx->get_scratch("raster.tv", size_t(n_views) * V * sizeof(rm_vert)));
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
    MVLM_CHECK_HIP(ctx, hipMemsetAsync(keys, 0xFF, key_bytes, ctx->stream));  // RM_KEY_EMPTY everywhere
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
                       mesh->verts, V, rot, n_views, tv);
    hipLaunchKernelGGL(classify_kernel, dim3(view_chunk_grid((T + 255) / 256, n_views)), dim3(256), 0, ctx->stream, tv,
                       mesh->tris, V, T, n_views, keys, counts, n_big, big_list);
    hipLaunchKernelGGL(scan_kernel, dim3(n_views), dim3(TILES), 0, ctx->stream, counts, offsets, cap, overflow);
    hipLaunchKernelGGL(bin_fill_kernel, dim3(view_chunk_grid((T + 255) / 256, n_views)), dim3(256), 0, ctx->stream, tv,
                       mesh->tris, V, T, n_views, n_big, big_list, offsets, cursors, bins, cap, overflow);
    hipLaunchKernelGGL(tile_kernel, dim3(view_chunk_grid(TILES, n_views)), dim3(256), 0, ctx->stream, tv, mesh->tris,
                       mesh->uvs, mesh->tex, mesh->tex_w, mesh->tex_h, V, counts, offsets, bins, cap, keys,
                       ctx->render_shading, n_views, out_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    if (e1) {
        MVLM_CHECK_HIP(ctx, hipEventRecord(e1, ctx->stream));
        ctx->render_prof.push_back({n_views, V, T, e0, e1});
    }
    // the overflow flag travels to pinned host memory without stalling the stream; it is
    // examined by mvlm_render_check (after the caller's own synchronisation point)
    if (!ctx->render_overflow_host)
        MVLM_CHECK_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->render_overflow_host), sizeof(int)));
    MVLM_CHECK_HIP(ctx, hipMemcpyAsync(ctx->render_overflow_host, overflow, sizeof(int), hipMemcpyDeviceToHost,
                                       ctx->stream));
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

extern "C" int mvlm_set_render_shading(mvlm_ctx* ctx, int shading) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    MVLM_REQUIRE(ctx, shading == 0 || shading == 1, "render: shading must be 0 (unlit texture) or 1 (geometry)");
    ctx->render_shading = shading;
    return 0;
}
