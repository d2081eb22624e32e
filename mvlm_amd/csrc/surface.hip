// Snap landmarks to the closest point of the triangle surface: every (landmark, triangle) pair is
// tested exactly in float64 (two passes: per-chunk winners, then the winner), replacing the reference's
// vtkCleanPolyData + vtkCellLocator.FindClosestPoint loop
// (src/mvlm/utils/estimator3d.py:252-285).  100k triangles x 84 landmarks is 8.4 M
// point-triangle tests - far cheaper on the GPU than building a locator.
#include "common.h"

namespace {

struct V3 {
    double x, y, z;
};
__device__ inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3 madd(V3 a, V3 d, double t) { return {a.x + t * d.x, a.y + t * d.y, a.z + t * d.z}; }

// closest point on triangle (a,b,c) to p: walk the Voronoi regions (vertex, edge, face)
__device__ V3 closest_on_triangle(V3 p, V3 a, V3 b, V3 c) {
    const V3 ab = sub(b, a), ac = sub(c, a), ap = sub(p, a);
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) return a;
    const V3 bp = sub(p, b);
    const double d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) return b;
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) return madd(a, ab, d1 / (d1 - d3));
    const V3 cp = sub(p, c);
    const double d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) return c;
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) return madd(a, ac, d2 / (d2 - d6));
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) return madd(b, sub(c, b), (d4 - d3) / ((d4 - d3) + (d5 - d6)));
    const double denom = 1.0 / (va + vb + vc);
    const double v = vb * denom, w = vc * denom;
    return {a.x + ab.x * v + ac.x * w, a.y + ab.y * v + ac.y * w, a.z + ab.z * v + ac.z * w};
}

// Pass 1: one workgroup per (landmark, chunk of PROJECT_CHUNK triangles) -> the chunk's nearest
// triangle (squared distance, id).  73-84 landmarks alone would leave 2/3 of the chip idle
// (one workgroup per landmark: 310 us on 73 workgroups); chunking gives ~50x more workgroups.
constexpr int PROJECT_CHUNK = 2048;

__global__ __launch_bounds__(256) void project_partial_kernel(const float* __restrict__ verts,
                                                              const int32_t* __restrict__ tris, int n_tris,
                                                              const double* __restrict__ pts, int n_chunks,
                                                              double* __restrict__ part_d, int* __restrict__ part_t) {
    const int lm = blockIdx.y, chunk = blockIdx.x;
    const V3 p = {pts[lm * 3], pts[lm * 3 + 1], pts[lm * 3 + 2]};
    double best = INFINITY;
    int best_t = 0x7fffffff;
    const int t_end = min(n_tris, (chunk + 1) * PROJECT_CHUNK);
    for (int t = chunk * PROJECT_CHUNK + threadIdx.x; t < t_end; t += blockDim.x) {
        const int ia = tris[3 * t], ib = tris[3 * t + 1], ic = tris[3 * t + 2];
        const V3 a = {verts[3 * ia], verts[3 * ia + 1], verts[3 * ia + 2]};
        const V3 b = {verts[3 * ib], verts[3 * ib + 1], verts[3 * ib + 2]};
        const V3 c = {verts[3 * ic], verts[3 * ic + 1], verts[3 * ic + 2]};
        const V3 q = closest_on_triangle(p, a, b, c);
        const V3 d = sub(q, p);
        const double d2 = dot(d, d);
        if (d2 < best) {  // strided ascending t: first minimum per thread
            best = d2;
            best_t = t;
        }
    }
    // (distance, id) minimum over the workgroup: lowest triangle id on ties
    for (int s = 32; s >= 1; s >>= 1) {
        const double od = __shfl_down(best, s);
        const int ot = __shfl_down(best_t, s);
        if (od < best || (od == best && ot < best_t)) {
            best = od;
            best_t = ot;
        }
    }
    __shared__ double s_d[4];
    __shared__ int s_t[4];
    if ((threadIdx.x & 63) == 0) {
        s_d[threadIdx.x >> 6] = best;
        s_t[threadIdx.x >> 6] = best_t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_d[w] < best || (s_d[w] == best && s_t[w] < best_t)) {
                best = s_d[w];
                best_t = s_t[w];
            }
        part_d[size_t(lm) * n_chunks + chunk] = best;
        part_t[size_t(lm) * n_chunks + chunk] = best_t;
    }
}

// Pass 2: one wavefront per landmark picks the nearest chunk winner and re-evaluates the closest
// point on that triangle.  A landmark with no finite distance to any triangle (NaN / Inf input)
// is passed through unchanged, so a non-finite value stays visible instead of leaving the output
// uninitialised.
__global__ __launch_bounds__(64) void project_final_kernel(const float* __restrict__ verts,
                                                           const int32_t* __restrict__ tris,
                                                           const double* __restrict__ pts, int n_chunks,
                                                           const double* __restrict__ part_d,
                                                           const int* __restrict__ part_t, double* __restrict__ out) {
    const int lm = blockIdx.x;
    double best = INFINITY;
    int best_t = 0x7fffffff;
    for (int c = threadIdx.x; c < n_chunks; c += 64) {
        const double od = part_d[size_t(lm) * n_chunks + c];
        const int ot = part_t[size_t(lm) * n_chunks + c];
        if (od < best || (od == best && ot < best_t)) {
            best = od;
            best_t = ot;
        }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const double od = __shfl_down(best, s);
        const int ot = __shfl_down(best_t, s);
        if (od < best || (od == best && ot < best_t)) {
            best = od;
            best_t = ot;
        }
    }
    if (threadIdx.x != 0) return;
    const V3 p = {pts[lm * 3], pts[lm * 3 + 1], pts[lm * 3 + 2]};
    V3 q = p;
    if (best_t != 0x7fffffff) {
        const int ia = tris[3 * best_t], ib = tris[3 * best_t + 1], ic = tris[3 * best_t + 2];
        const V3 a = {verts[3 * ia], verts[3 * ia + 1], verts[3 * ia + 2]};
        const V3 b = {verts[3 * ib], verts[3 * ib + 1], verts[3 * ib + 2]};
        const V3 c = {verts[3 * ic], verts[3 * ic + 1], verts[3 * ic + 2]};
        q = closest_on_triangle(p, a, b, c);
    }
    out[lm * 3] = q.x;
    out[lm * 3 + 1] = q.y;
    out[lm * 3 + 2] = q.z;
}

// ---- first intersection of ray segments with the surface -------------------------------------
// The reference clips every view ray to its first hit with the mesh through a vtkOBBTree
// (src/mvlm/visualization/ray_visualizer.py:172-192).  Along a view ray that hit is the point the
// depth buffer of that view stores, so this is also the exact (sub-pixel) form of unprojecting a
// heatmap maximum through the rendered depth: one 3-D surface point per (landmark, view).
// RAYS_PER_GROUP rays share every triangle fetch; Moeller-Trumbore in float64, two-sided, segment
// parameter t in [0,1], smallest t wins, lowest triangle id on ties.
constexpr int RAYS_PER_GROUP = 8;

__device__ inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__global__ __launch_bounds__(256) void clip_rays_kernel(const float* __restrict__ verts, const int32_t* __restrict__ tris,
                                                        int n_tris, const double* __restrict__ starts,
                                                        const double* __restrict__ ends, int n_rays,
                                                        double* __restrict__ new_ends, uint8_t* __restrict__ hit) {
    const int r0 = blockIdx.x * RAYS_PER_GROUP;
    V3 o[RAYS_PER_GROUP], d[RAYS_PER_GROUP];
    double best[RAYS_PER_GROUP];
    int best_t[RAYS_PER_GROUP];
#pragma unroll
    for (int r = 0; r < RAYS_PER_GROUP; ++r) {
        const int ray = min(r0 + r, n_rays - 1);  // the tail group repeats its last ray
        o[r] = {starts[ray * 3], starts[ray * 3 + 1], starts[ray * 3 + 2]};
        const V3 e = {ends[ray * 3], ends[ray * 3 + 1], ends[ray * 3 + 2]};
        d[r] = sub(e, o[r]);
        best[r] = INFINITY;
        best_t[r] = 0x7fffffff;
    }
    for (int t = threadIdx.x; t < n_tris; t += blockDim.x) {
        const int ia = tris[3 * t], ib = tris[3 * t + 1], ic = tris[3 * t + 2];
        const V3 a = {verts[3 * ia], verts[3 * ia + 1], verts[3 * ia + 2]};
        const V3 b = {verts[3 * ib], verts[3 * ib + 1], verts[3 * ib + 2]};
        const V3 c = {verts[3 * ic], verts[3 * ic + 1], verts[3 * ic + 2]};
        const V3 e1 = sub(b, a), e2 = sub(c, a);
#pragma unroll
        for (int r = 0; r < RAYS_PER_GROUP; ++r) {
            const V3 pvec = cross(d[r], e2);
            const double det = dot(e1, pvec);
            if (det == 0.0) continue;  // parallel to the plane or a degenerate triangle
            const double inv = 1.0 / det;
            const V3 tvec = sub(o[r], a);
            const double u = dot(tvec, pvec) * inv;
            if (!(u >= 0.0 && u <= 1.0)) continue;
            const V3 qvec = cross(tvec, e1);
            const double v = dot(d[r], qvec) * inv;
            if (!(v >= 0.0 && u + v <= 1.0)) continue;
            const double tt = dot(e2, qvec) * inv;
            if (!(tt >= 0.0 && tt <= 1.0)) continue;
            if (tt < best[r]) {  // ascending t per thread: first minimum keeps the lowest id
                best[r] = tt;
                best_t[r] = t;
            }
        }
    }
    __shared__ double s_d[RAYS_PER_GROUP][4];
    __shared__ int s_t[RAYS_PER_GROUP][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < RAYS_PER_GROUP; ++r) {
        double bd = best[r];
        int bt = best_t[r];
        for (int s = 32; s >= 1; s >>= 1) {
            const double od = __shfl_down(bd, s);
            const int ot = __shfl_down(bt, s);
            if (od < bd || (od == bd && ot < bt)) {
                bd = od;
                bt = ot;
            }
        }
        if (lane == 0) {
            s_d[r][wave] = bd;
            s_t[r][wave] = bt;
        }
    }
    __syncthreads();
    if (threadIdx.x < RAYS_PER_GROUP && r0 + int(threadIdx.x) < n_rays) {
        const int r = threadIdx.x, ray = r0 + r;
        double bd = s_d[r][0];
        int bt = s_t[r][0];
        for (int w = 1; w < 4; ++w)
            if (s_d[r][w] < bd || (s_d[r][w] == bd && s_t[r][w] < bt)) {
                bd = s_d[r][w];
                bt = s_t[r][w];
            }
        const bool found = bt != 0x7fffffff;
        const V3 p0 = {starts[ray * 3], starts[ray * 3 + 1], starts[ray * 3 + 2]};
        const V3 p1 = {ends[ray * 3], ends[ray * 3 + 1], ends[ray * 3 + 2]};
        const V3 q = found ? madd(p0, sub(p1, p0), bd) : p1;
        new_ends[ray * 3] = q.x;
        new_ends[ray * 3 + 1] = q.y;
        new_ends[ray * 3 + 2] = q.z;
        if (hit) hit[ray] = found ? 1 : 0;
    }
}

}  // namespace

extern "C" int mvlm_clip_rays_to_mesh(mvlm_ctx* ctx, const mvlm_mesh* mesh, const double* starts_dev,
                                      const double* ends_dev, int n_rays, double* new_ends_dev, uint8_t* hit_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, mesh && starts_dev && ends_dev && new_ends_dev && n_rays > 0, "clip_rays_to_mesh: bad arguments");
    MVLM_REQUIRE(ctx, mesh->n_tris > 0, "clip_rays_to_mesh: empty mesh");
    if (mvlm_mesh_wait_ready(ctx, mesh, ctx->stream)) return 1;
    const int groups = (n_rays + RAYS_PER_GROUP - 1) / RAYS_PER_GROUP;
    hipLaunchKernelGGL(clip_rays_kernel, dim3(groups), dim3(256), 0, ctx->stream, mesh->verts, mesh->tris, mesh->n_tris,
                       starts_dev, ends_dev, n_rays, new_ends_dev, hit_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int mvlm_project_to_surface(mvlm_ctx* ctx, const mvlm_mesh* mesh, const double* pts_dev, int n_points,
                                       double* out_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, mesh && pts_dev && out_dev && n_points > 0, "project_to_surface: bad arguments");
    MVLM_REQUIRE(ctx, mesh->n_tris > 0, "project_to_surface: empty mesh");
    if (mvlm_mesh_wait_ready(ctx, mesh, ctx->stream)) return 1;
    const int n_chunks = (mesh->n_tris + PROJECT_CHUNK - 1) / PROJECT_CHUNK;
    auto* part_d = static_cast<double*>(ctx->get_scratch("project.part_d", size_t(n_points) * n_chunks * sizeof(double)));
    auto* part_t = static_cast<int*>(ctx->get_scratch("project.part_t", size_t(n_points) * n_chunks * sizeof(int)));
    MVLM_REQUIRE(ctx, part_d && part_t, "project_to_surface: scratch allocation failed");
    MVLM_REQUIRE(ctx, n_points <= 65535, "project_to_surface: at most 65535 points per call");
    hipLaunchKernelGGL(project_partial_kernel, dim3(n_chunks, n_points), dim3(256), 0, ctx->stream, mesh->verts,
                       mesh->tris, mesh->n_tris, pts_dev, n_chunks, part_d, part_t);
    hipLaunchKernelGGL(project_final_kernel, dim3(n_points), dim3(64), 0, ctx->stream, mesh->verts, mesh->tris, pts_dev,
                       n_chunks, part_d, part_t, out_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}
