// Snap landmarks to the closest point of the triangle surface: the result of testing every (landmark, triangle) pair
// exactly in float64 (an upper bound per landmark, per-chunk winners among the triangles the bound cannot exclude, then
// the winner), replacing the reference's
// vtkCleanPolyData + vtkCellLocator.FindClosestPoint loop
// (src/mvlm/utils/estimator3d.py:252-285).  100k triangles x 84 landmarks is 8.4 M
// point-triangle tests - far cheaper on the GPU than building a locator.
#include <algorithm>

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
    // (d1 - d3 = |ab|^2: a triangle with a == b is the segment ac - vtkCleanPolyData turns it into a line cell - and belongs
    //  to the edge-ac branch below, not to a 0 / 0 here)
    if (vc <= 0 && d1 >= 0 && d3 <= 0 && d1 > d3) return madd(a, ab, d1 / (d1 - d3));
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

// Pass 0: an upper bound on every landmark's squared distance to the surface.  The nearest vertex THAT A TRIANGLE USES
// (vert_tri[v] = lowest id of a triangle with corner v, filled by project_soup_kernel; VERT_UNUSED = a stray point of the
// file, which the .ply / .vtk / .stl / .wrl readers and mvlm_mesh_upload keep and which is no point of the surface - the
// reference drops such points in vtkCleanPolyData and searches cells only, estimator3d.py:258-270) names a candidate
// triangle; the bound is the exact walk's OWN squared distance to that triangle (project_bound_pick_kernel), i.e. a value
// the search below will meet again - so whatever the walk makes of a degenerate triangle, the bound never lies below the
// winner.  A candidate whose walk gives no finite distance leaves "no bound" (0x7f7f... = 1.4e306: nothing is skipped), and
// so does a non-finite landmark, whose comparisons are all false.
// BOUND_SPLIT workgroups per landmark stride over the vertices (0.6 MB, L2-resident), four loads in flight per thread;
// every workgroup leaves its (distance, vertex) minimum, lowest vertex on ties.
constexpr int BOUND_SPLIT = 8;
constexpr int VERT_UNUSED = 0x7f7f7f7f;  // what hipMemsetAsync(.., 0x7f, ..) leaves: above every triangle id
__global__ __launch_bounds__(256) void project_bound_kernel(const float* __restrict__ verts, int n_verts,
                                                            const int* __restrict__ vert_tri, const double* __restrict__ pts,
                                                            double* __restrict__ cand_d, int* __restrict__ cand_v) {
    __shared__ double s_d[4];
    __shared__ int s_v[4];
    const int lm = blockIdx.x;
    const V3 p = {pts[lm * 3], pts[lm * 3 + 1], pts[lm * 3 + 2]};
    double best = INFINITY;
    int best_v = 0x7fffffff;
    const int stride = BOUND_SPLIT * 256;
    for (int v0 = blockIdx.y * 256 + threadIdx.x; v0 < n_verts; v0 += 4 * stride) {
        V3 q[4];
        int used[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = min(v0 + k * stride, n_verts - 1);  // (a repeated vertex changes no minimum)
            q[k] = V3{verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]};
            used[k] = vert_tri[v];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const V3 d = sub(q[k], p);
            const double d2 = dot(d, d);
            if (used[k] != VERT_UNUSED && d2 < best) {  // ascending v per thread: first minimum (NaN never wins)
                best = d2;
                best_v = min(v0 + k * stride, n_verts - 1);
            }
        }
    }
    for (int s = 32; s >= 1; s >>= 1) {
        const double od = __shfl_down(best, s);
        const int ov = __shfl_down(best_v, s);
        if (od < best || (od == best && ov < best_v)) {
            best = od;
            best_v = ov;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_d[threadIdx.x >> 6] = best;
        s_v[threadIdx.x >> 6] = best_v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_d[w] < best || (s_d[w] == best && s_v[w] < best_v)) {
                best = s_d[w];
                best_v = s_v[w];
            }
        cand_d[lm * BOUND_SPLIT + blockIdx.y] = best;
        cand_v[lm * BOUND_SPLIT + blockIdx.y] = best_v;
    }
}

// ... and the bound itself: one thread per landmark takes the nearest used vertex of the BOUND_SPLIT candidates and walks
// the triangle that vertex stands for.
__global__ __launch_bounds__(64) void project_bound_pick_kernel(const float* __restrict__ verts, const int32_t* __restrict__ tris,
                                                                const int* __restrict__ vert_tri, const double* __restrict__ pts,
                                                                int n_points, const double* __restrict__ cand_d,
                                                                const int* __restrict__ cand_v, unsigned long long* __restrict__ ub) {
    const int lm = blockIdx.x * 64 + threadIdx.x;
    if (lm >= n_points) return;
    double best = INFINITY;
    int best_v = 0x7fffffff;
    for (int k = 0; k < BOUND_SPLIT; ++k) {
        const double od = cand_d[lm * BOUND_SPLIT + k];
        const int ov = cand_v[lm * BOUND_SPLIT + k];
        if (od < best || (od == best && ov < best_v)) {
            best = od;
            best_v = ov;
        }
    }
    unsigned long long bound = 0x7f7f7f7f7f7f7f7full;  // "no bound"
    if (best_v != 0x7fffffff) {
        const int t = vert_tri[best_v];
        const int ia = tris[3 * t], ib = tris[3 * t + 1], ic = tris[3 * t + 2];
        const V3 p = {pts[lm * 3], pts[lm * 3 + 1], pts[lm * 3 + 2]};
        const V3 a = {verts[3 * ia], verts[3 * ia + 1], verts[3 * ia + 2]};
        const V3 b = {verts[3 * ib], verts[3 * ib + 1], verts[3 * ib + 2]};
        const V3 c = {verts[3 * ic], verts[3 * ic + 1], verts[3 * ic + 2]};
        const V3 d = sub(closest_on_triangle(p, a, b, c), p);
        const double d2 = dot(d, d);
        if (d2 < 1e300) bound = (unsigned long long)__double_as_longlong(d2);  // (false for NaN / inf)
    }
    ub[lm] = bound;
}

// The triangles once more, de-indexed and component-major: soup[e * n_tris + t], e = 0..8 = a.xyz, b.xyz, c.xyz.
// Pass 1 reads every triangle once per GROUP of landmarks (60 times for 478 landmarks); through the index list that is
// four scattered loads per triangle and group, and the L1s' rate for scattered lines - not the arithmetic - set its
// time (7 us per 1 024 triangles and CU).  One gather pass per call (3.6 MB for 99 458 triangles) makes all of them
// coalesced streams.
// The same pass notes for every vertex the lowest triangle that uses it (vert_tri, cleared to VERT_UNUSED before).
__global__ __launch_bounds__(256) void project_soup_kernel(const float* __restrict__ verts, const int32_t* __restrict__ tris,
                                                           int n_tris, float* __restrict__ soup, int* __restrict__ vert_tri) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tris) return;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int i = tris[3 * t + e];
        atomicMin(&vert_tri[i], t);
#pragma unroll
        for (int d = 0; d < 3; ++d) soup[size_t(3 * e + d) * n_tris + t] = verts[3 * i + d];
    }
}

// Pass 1: one workgroup per (chunk of PROJECT_CHUNK triangles, group of PROJECT_GROUP landmarks) -> per landmark the
// chunk's nearest triangle (squared distance, id).  A thread reads a triangle ONCE and tests it against the group's
// landmarks (rounds 1-3: one workgroup per landmark and chunk, every pair paid its own four gathers and the full
// Voronoi walk: 135 us for 478 landmarks x 99 458 triangles, at the rate of the L1s and of the f64 pipe at once).
// Most pairs end at a sphere test: every point of triangle (a, b, c) lies within R = max(|ab|, |ac|) of a, so its
// distance to p is at least |pa| - R; with the landmark's upper bound u2 (pass 0) the triangle can be neither the
// winner nor tied with it once |pa| > u + R, which |pa|^2 > 2 (u2 + R^2) implies (2 u R <= u2 + R^2): no root.  The
// comparison carries a relative margin of 1e-12 (the exact walk's own rounding is of the order 1e-16), so the result
// is what the walk over ALL triangles gives, ties (lowest id) included.  The landmarks' coordinates and bounds are
// wave-uniform: they are read through scalar loads and stay in scalar registers (an LDS copy cost a read latency per
// pair: 100 us for the pass).
constexpr int PROJECT_CHUNK = 1024;
constexpr int PROJECT_GROUP = 8;

__global__ __launch_bounds__(256) void project_partial_kernel(const float* __restrict__ soup, int n_tris,
                                                              const double* __restrict__ pts, int n_points,
                                                              const unsigned long long* __restrict__ ub, int n_chunks,
                                                              int n_sub, double* __restrict__ part_d, int* __restrict__ part_t) {
    __shared__ double s_d[PROJECT_GROUP][4];
    __shared__ int s_t[PROJECT_GROUP][4];
    const int chunk = blockIdx.x, lm0 = blockIdx.y * PROJECT_GROUP;
    const int n_lm = min(PROJECT_GROUP, n_points - lm0);
    double px[PROJECT_GROUP], py[PROJECT_GROUP], pz[PROJECT_GROUP], u2[PROJECT_GROUP];  // (uniform: scalar registers)
#pragma unroll
    for (int g = 0; g < PROJECT_GROUP; ++g) {
        const int lm = lm0 + min(g, n_lm - 1);
        px[g] = pts[lm * 3];
        py[g] = pts[lm * 3 + 1];
        pz[g] = pts[lm * 3 + 2];
        u2[g] = __longlong_as_double((long long)ub[lm]);  // ("no bound" = 1.4e306 never skips; nor do NaN / inf landmarks)
    }
    double best[PROJECT_GROUP];
    int best_t[PROJECT_GROUP];
#pragma unroll
    for (int g = 0; g < PROJECT_GROUP; ++g) {
        best[g] = INFINITY;
        best_t[g] = 0x7fffffff;
    }
    // A workgroup walks n_sub pieces of PROJECT_CHUNK triangles (the host picks n_sub so that the grid still fills the chip:
    // the eight reductions at the end are then paid once per several thousand triangles); a thread's TPT triangles of a
    // piece are TPT x 9 coalesced loads in flight together.
    constexpr int TPT = PROJECT_CHUNK / 256;
    for (int sc = 0; sc < n_sub; ++sc) {
        const int t0 = (chunk * n_sub + sc) * PROJECT_CHUNK;
        if (t0 >= n_tris) break;  // (uniform)
        float xyz[TPT][9];
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const int t = min(t0 + k * 256 + int(threadIdx.x), n_tris - 1);
#pragma unroll
            for (int e = 0; e < 9; ++e) xyz[k][e] = soup[size_t(e) * n_tris + t];
        }
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const int t = t0 + k * 256 + int(threadIdx.x);
            if (t >= n_tris) break;
            const V3 a = {xyz[k][0], xyz[k][1], xyz[k][2]};
            const V3 b = {xyz[k][3], xyz[k][4], xyz[k][5]};
            const V3 c = {xyz[k][6], xyz[k][7], xyz[k][8]};
            const V3 ab = sub(b, a), ac = sub(c, a);
            const double r2ab = dot(ab, ab), r2ac = dot(ac, ac);
            const double R2 = r2ab > r2ac ? r2ab : r2ac;
#pragma unroll
            for (int g = 0; g < PROJECT_GROUP; ++g) {
                const V3 p = {px[g], py[g], pz[g]};
                const V3 pa = sub(a, p);
                if (dot(pa, pa) > (u2[g] + R2) * (2.0 * (1.0 + 1e-12))) continue;  // farther than the bound allows (false for NaN / inf)
                const V3 q = closest_on_triangle(p, a, b, c);
                const V3 d = sub(q, p);
                const double d2 = dot(d, d);
                if (d2 < best[g]) {  // ascending t per thread: first minimum
                    best[g] = d2;
                    best_t[g] = t;
                }
            }
        }
    }
    // (distance, id) minimum over the workgroup, per landmark: lowest triangle id on ties
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < PROJECT_GROUP; ++g) {
        double bd = best[g];
        int bt = best_t[g];
        for (int s = 32; s >= 1; s >>= 1) {
            const double od = __shfl_down(bd, s);
            const int ot = __shfl_down(bt, s);
            if (od < bd || (od == bd && ot < bt)) {
                bd = od;
                bt = ot;
            }
        }
        if (lane == 0) {
            s_d[g][wave] = bd;
            s_t[g][wave] = bt;
        }
    }
    __syncthreads();
    if (int(threadIdx.x) < n_lm) {
        const int g = threadIdx.x;
        double bd = s_d[g][0];
        int bt = s_t[g][0];
        for (int w = 1; w < 4; ++w)
            if (s_d[g][w] < bd || (s_d[g][w] == bd && s_t[g][w] < bt)) {
                bd = s_d[g][w];
                bt = s_t[g][w];
            }
        part_d[size_t(lm0 + g) * n_chunks + chunk] = bd;
        part_t[size_t(lm0 + g) * n_chunks + chunk] = bt;
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
    // pieces of PROJECT_CHUNK triangles, n_sub of them per workgroup: as many as leave >= ~768 workgroups (3 per CU)
    const int n_pieces = (mesh->n_tris + PROJECT_CHUNK - 1) / PROJECT_CHUNK, n_groups = (n_points + PROJECT_GROUP - 1) / PROJECT_GROUP;
    // (measured, 478 landmarks x 98 pieces: 1 / 2 / 3 / 4 / 7 pieces per workgroup = 70 / 56 / 51 / 51 / 51 us)
    const int n_sub = std::max(1, std::min(8, int(long(n_pieces) * n_groups / 768)));
    const int n_chunks = (n_pieces + n_sub - 1) / n_sub;
    auto* part_d = static_cast<double*>(ctx->get_scratch("project.part_d", size_t(n_points) * n_chunks * sizeof(double)));
    auto* part_t = static_cast<int*>(ctx->get_scratch("project.part_t", size_t(n_points) * n_chunks * sizeof(int)));
    auto* ub = static_cast<unsigned long long*>(ctx->get_scratch("project.ub", size_t(n_points) * sizeof(unsigned long long)));
    auto* soup = static_cast<float*>(ctx->get_scratch("project.soup", size_t(9) * mesh->n_tris * sizeof(float)));
    auto* vert_tri = static_cast<int*>(ctx->get_scratch("project.vert_tri", size_t(mesh->n_verts) * sizeof(int)));
    auto* cand_d = static_cast<double*>(ctx->get_scratch("project.cand_d", size_t(n_points) * BOUND_SPLIT * sizeof(double)));
    auto* cand_v = static_cast<int*>(ctx->get_scratch("project.cand_v", size_t(n_points) * BOUND_SPLIT * sizeof(int)));
    MVLM_REQUIRE(ctx, part_d && part_t && ub && soup && vert_tri && cand_d && cand_v, "project_to_surface: scratch allocation failed");
    MVLM_REQUIRE(ctx, n_points <= 65535, "project_to_surface: at most 65535 points per call");
    // the de-indexed triangles and the vertex -> triangle table depend on the mesh only: kept in the context's scratch from call
    // to call of the same upload (a pipeline snaps every scan once, bench and folders of one mesh many times)
    if (ctx->snap_mesh_uid != mesh->uid || ctx->snap_soup != soup || ctx->snap_vert_tri != vert_tri) {
        MVLM_CHECK_HIP(ctx, hipMemsetAsync(vert_tri, 0x7f, size_t(mesh->n_verts) * sizeof(int), ctx->stream));
        hipLaunchKernelGGL(project_soup_kernel, dim3((mesh->n_tris + 255) / 256), dim3(256), 0, ctx->stream, mesh->verts, mesh->tris,
                           mesh->n_tris, soup, vert_tri);
        ctx->snap_mesh_uid = mesh->uid;
        ctx->snap_soup = soup;
        ctx->snap_vert_tri = vert_tri;
    }
    hipLaunchKernelGGL(project_bound_kernel, dim3(n_points, BOUND_SPLIT), dim3(256), 0, ctx->stream, mesh->verts, mesh->n_verts,
                       vert_tri, pts_dev, cand_d, cand_v);
    hipLaunchKernelGGL(project_bound_pick_kernel, dim3((n_points + 63) / 64), dim3(64), 0, ctx->stream, mesh->verts, mesh->tris,
                       vert_tri, pts_dev, n_points, cand_d, cand_v, ub);
    hipLaunchKernelGGL(project_partial_kernel, dim3(n_chunks, n_groups), dim3(256), 0, ctx->stream, soup, mesh->n_tris, pts_dev,
                       n_points, ub, n_chunks, n_sub, part_d, part_t);
    hipLaunchKernelGGL(project_final_kernel, dim3(n_points), dim3(64), 0, ctx->stream, mesh->verts, mesh->tris, pts_dev,
                       n_chunks, part_d, part_t, out_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}
