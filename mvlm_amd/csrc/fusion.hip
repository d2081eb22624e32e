// Heatmap maxima -> 3-D view rays -> per-landmark consensus, one wavefront per landmark.
// Replaces three pure-Python double loops of the reference:
//   Estimator3D.estimate_landmark_lines            src/mvlm/utils/estimator3d.py:31-90
//   filter_lines_based_on_heatmap_value_*          estimator3d.py:140-155
//   compute_intersection_between_lines_ransac      estimator3d.py:92-137
//   compute_intersection_between_lines (pinv 3x3)  src/mvlm/utils/utils3d.py:99-124
// The arithmetic keeps the reference's dtypes: pixel -> model units in float32 (numpy
// float32 scalars with weak python ints), everything after that in float64.  The build
// uses -ffp-contract=off so no fused multiply-add changes a rounding.
#include "common.h"

namespace {

constexpr int MAX_VIEWS = 1024;

__global__ void lines_kernel(const float* __restrict__ maxima, const double* __restrict__ rot, int n_views, int nl,
                             float img_size, double* __restrict__ starts, double* __restrict__ ends) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_views * nl) return;
    const int v = i % n_views;
    float y = maxima[size_t(i) * 3 + 0];
    float x = maxima[size_t(i) * 3 + 1];
    // estimator3d.py:66-67 (heatmap size == image size) and :73-78, all float32
    y = y / img_size * img_size;
    x = x / img_size * img_size;
    const float px = (x / img_size) * 300.0f + (-150.0f);
    const float py = ((img_size - 1.0f - y) / img_size) * 300.0f + (-150.0f);
    // points = t.T @ [px, py, +-500, 1]  with t[0:3,0:3] = M (estimator3d.py:56-57, :83)
    const double* m = rot + size_t(v) * 9;
    const double dx = px, dy = py;
    for (int k = 0; k < 3; ++k) {
        const double base = m[0 * 3 + k] * dx + m[1 * 3 + k] * dy;
        starts[size_t(i) * 3 + k] = base + m[2 * 3 + k] * 500.0;
        ends[size_t(i) * 3 + k] = base + m[2 * 3 + k] * -500.0;
    }
}

// One wave per landmark.  np.quantile(values, q) with method "linear" on a float32 vector
// (numpy >= 2: q is cast to the array dtype, the interpolation runs in float32), then
// mask = values > threshold (estimator3d.py:141-143); or the absolute test (:151-152).
__global__ __launch_bounds__(64) void mask_kernel(const float* __restrict__ maxima, int n_views, int mode, float q,
                                                  float thr_abs, uint8_t* __restrict__ mask,
                                                  int* __restrict__ count) {
    __shared__ float vals[MAX_VIEWS];
    const int lm = blockIdx.x, lane = threadIdx.x;
    for (int v = lane; v < n_views; v += 64) vals[v] = maxima[(size_t(lm) * n_views + v) * 3 + 2];
    __syncthreads();
    float thr = thr_abs;
    if (mode == MVLM_MODE_QUANTILE) {
        const float vi = float(n_views - 1) * q;
        int lo = int(floorf(vi)), hi = lo + 1;
        if (vi >= float(n_views - 1)) lo = hi = n_views - 1;
        if (vi < 0.f) lo = hi = 0;
        if (hi > n_views - 1) hi = n_views - 1;
        const float gamma = vi - floorf(vi);
        // order statistics by rank counting (stable for ties); NaN anywhere -> NaN result
        float a = 0.f, b = 0.f;
        bool have_a = false, have_b = false, any_nan = false;
        for (int i = lane; i < n_views; i += 64) {
            const float x = vals[i];
            if (x != x) any_nan = true;
            int rank = 0;
            for (int j = 0; j < n_views; ++j) {
                const float yv = vals[j];
                rank += (yv < x) || (yv == x && j < i);
            }
            if (rank == lo) {
                a = x;
                have_a = true;
            }
            if (rank == hi) {
                b = x;
                have_b = true;
            }
        }
        // broadcast the two order statistics
        const unsigned long long ma = __ballot(have_a), mb = __ballot(have_b);
        any_nan = __any(any_nan);
        a = __shfl(a, ma ? __ffsll((long long)ma) - 1 : 0);
        b = __shfl(b, mb ? __ffsll((long long)mb) - 1 : 0);
        // numpy _lerp: a + (b-a)*t, replaced by b - (b-a)*(1-t) where t >= 0.5
        const float diff = b - a;
        float r = a + diff * gamma;
        if (gamma >= 0.5f) r = b - diff * (1.0f - gamma);
        thr = any_nan ? NAN : r;
    }
    int local = 0;
    for (int v = lane; v < n_views; v += 64) {
        const bool keep = vals[v] > thr;
        mask[size_t(lm) * n_views + v] = keep;
        local += keep;
    }
    for (int s = 32; s >= 1; s >>= 1) local += __shfl_xor(local, s);
    if (lane == 0) count[lm] = local;
}

__device__ inline double wave_sum(double v) {
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

// Moore-Penrose inverse of a symmetric 3x3 applied to c, through a cyclic Jacobi
// eigen-decomposition; eigenvalues with |l| <= 1e-15 * max|l| are dropped, which is
// np.linalg.pinv's default cutoff on the singular values (utils3d.py:123).
__device__ void pinv3_apply(const double s_in[6], const double c[3], double p[3]) {
    double a[3][3] = {{s_in[0], s_in[3], s_in[4]}, {s_in[3], s_in[1], s_in[5]}, {s_in[4], s_in[5], s_in[2]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off == 0.0) break;
        for (int pi = 0; pi < 2; ++pi)
            for (int qi = pi + 1; qi < 3; ++qi) {
                const double apq = a[pi][qi];
                if (apq == 0.0) continue;
                // After a few sweeps an off-diagonal element that no longer registers beside BOTH of its diagonal elements
                // (|a_pp| + 100 |a_pq| == |a_pp| in double arithmetic) is set to zero instead of rotated (the classical cyclic
                // Jacobi termination): its rotation would move the eigenvalues by less than an ulp, and without this the
                // two-sided update never leaves an exact 0.0 behind, so every solve ran all 30 sweeps - 45 of the kernel's
                // 68 us, on one wavefront per landmark (round 5).
                if (sweep > 2) {
                    const double g = 100.0 * fabs(apq);
                    if (fabs(a[pi][pi]) + g == fabs(a[pi][pi]) && fabs(a[qi][qi]) + g == fabs(a[qi][qi])) {
                        a[pi][qi] = 0.0;
                        a[qi][pi] = 0.0;
                        continue;
                    }
                }
                const double theta = (a[qi][qi] - a[pi][pi]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) {  // columns p,q of A
                    const double akp = a[k][pi], akq = a[k][qi];
                    a[k][pi] = cs * akp - sn * akq;
                    a[k][qi] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; ++k) {  // rows p,q of A
                    const double apk = a[pi][k], aqk = a[qi][k];
                    a[pi][k] = cs * apk - sn * aqk;
                    a[qi][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][pi], vkq = v[k][qi];
                    v[k][pi] = cs * vkp - sn * vkq;
                    v[k][qi] = sn * vkp + cs * vkq;
                }
            }
    }
    const double l0 = a[0][0], l1 = a[1][1], l2 = a[2][2];
    const double lmax = fmax(fabs(l0), fmax(fabs(l1), fabs(l2)));
    const double cutoff = 1e-15 * lmax;
    const double lam[3] = {l0, l1, l2};
    p[0] = p[1] = p[2] = 0.0;
    for (int e = 0; e < 3; ++e) {
        if (!(fabs(lam[e]) > cutoff)) continue;
        const double proj = (v[0][e] * c[0] + v[1][e] * c[1] + v[2][e] * c[2]) / lam[e];
        p[0] += v[0][e] * proj;
        p[1] += v[1][e] * proj;
        p[2] += v[2][e] * proj;
    }
}

struct Line {
    double ax, ay, az, bx, by, bz;
};

__device__ inline Line load_line(const double* s, const double* e, size_t i) {
    return {s[i * 3], s[i * 3 + 1], s[i * 3 + 2], e[i * 3], e[i * 3 + 1], e[i * 3 + 2]};
}

// accumulate one line's contribution to S (6 unique entries) and c (utils3d.py:101-120)
__device__ inline void lsq_accum(const Line& l, double acc[9]) {
    const double sx = l.bx - l.ax, sy = l.by - l.ay, sz = l.bz - l.az;
    const double len = sqrt(sx * sx + sy * sy + sz * sz);
    const double nx = sx / len, ny = sy / len, nz = sz / len;
    acc[0] += nx * nx - 1;
    acc[1] += ny * ny - 1;
    acc[2] += nz * nz - 1;
    acc[3] += nx * ny;
    acc[4] += nx * nz;
    acc[5] += ny * nz;
    acc[6] += l.ax * (nx * nx - 1) + l.ay * (nx * ny) + l.az * (nx * nz);
    acc[7] += l.ax * (nx * ny) + l.ay * (ny * ny - 1) + l.az * (ny * nz);
    acc[8] += l.ax * (nx * nz) + l.ay * (ny * nz) + l.az * (nz * nz - 1);
}

__device__ inline void lsq_solve(double acc[9], double p[3]) {
    for (int k = 0; k < 9; ++k) acc[k] = wave_sum(acc[k]);
    pinv3_apply(acc, acc + 6, p);
}

// squared point-line distance (estimator3d.py:109-111)
__device__ inline double sq_dist(const Line& l, const double p[3]) {
    const double ux = p[0] - l.ax, uy = p[1] - l.ay, uz = p[2] - l.az;
    const double wx = p[0] - l.bx, wy = p[1] - l.by, wz = p[2] - l.bz;
    const double cx = uy * wz - uz * wy, cy = uz * wx - ux * wz, cz = ux * wy - uy * wx;
    const double bx = l.bx - l.ax, by = l.by - l.ay, bz = l.bz - l.az;
    const double r = sqrt(cx * cx + cy * cy + cz * cz) / sqrt(bx * bx + by * by + bz * bz);
    return r * r;
}

__global__ __launch_bounds__(64) void solve_kernel(const double* __restrict__ starts, const double* __restrict__ ends,
                                                   const uint8_t* __restrict__ mask, const int* __restrict__ count,
                                                   const int* __restrict__ draws, int n_views,
                                                   double* __restrict__ out, double* __restrict__ err) {
    __shared__ int sel[MAX_VIEWS];
    __shared__ uint8_t inl[MAX_VIEWS];
    const int lm = blockIdx.x, lane = threadIdx.x;
    const double* S = starts + size_t(lm) * n_views * 3;
    const double* E = ends + size_t(lm) * n_views * 3;
    // compact the surviving views in view order (pa[idx], estimator3d.py:145-146)
    int k = 0;
    for (int base = 0; base < n_views; base += 64) {
        const int v = base + lane;
        const bool keep = v < n_views && mask[size_t(lm) * n_views + v];
        const unsigned long long bal = __ballot(keep);
        if (keep) sel[k + __popcll(bal & ((1ull << lane) - 1))] = v;
        k += __popcll(bal);
    }
    __syncthreads();
    (void)count;
    double p[3];
    double e_out = 0.0;
    if (k < 3) {
        // "Not enough points": plain least squares over what is left (estimator3d.py:174-176)
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = lane; i < k; i += 64) lsq_accum(load_line(S, E, sel[i]), acc);
        lsq_solve(acc, p);
    } else {
        // one draw of 8 lines with replacement (estimator3d.py:105-107)
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (lane < 8) {
            int d = draws[lm * 8 + lane];
            d = d < 0 ? 0 : (d > k - 1 ? k - 1 : d);
            lsq_accum(load_line(S, E, sel[d]), acc);
        }
        lsq_solve(acc, p);
        // inliers among all surviving lines (:109-113)
        int n_in = 0;
        for (int i = lane; i < k; i += 64) {
            const bool in = sq_dist(load_line(S, E, sel[i]), p) < 100.0;
            inl[i] = in;
            n_in += in;
        }
        for (int s = 32; s >= 1; s >>= 1) n_in += __shfl_xor(n_in, s);
        __syncthreads();
        double best_error = 100000000.0;
        bool used = false;
        if (double(n_in) > double(k) / 3.0) {
            for (int z = 0; z < 9; ++z) acc[z] = 0.0;
            for (int i = lane; i < k; i += 64)
                if (inl[i]) lsq_accum(load_line(S, E, sel[i]), acc);
            lsq_solve(acc, p);
            double dsum = 0.0;
            for (int i = lane; i < k; i += 64)
                if (inl[i]) dsum += sq_dist(load_line(S, E, sel[i]), p);
            dsum = wave_sum(dsum);
            const double sum_squared = dsum / double(n_in);
            if (sum_squared < best_error) {
                best_error = sum_squared;
                used = true;
            }
        }
        if (!used) {
            // "Ransac failed - estimating from all lines"; the error stays 1e8 (:131-133)
            for (int z = 0; z < 9; ++z) acc[z] = 0.0;
            for (int i = lane; i < k; i += 64) lsq_accum(load_line(S, E, sel[i]), acc);
            lsq_solve(acc, p);
        }
        e_out = best_error;
    }
    if (lane == 0) {
        out[lm * 3 + 0] = p[0];
        out[lm * 3 + 1] = p[1];
        out[lm * 3 + 2] = p[2];
        err[lm] = e_out;
    }
}

}  // namespace

extern "C" int mvlm_estimate_lines(mvlm_ctx* ctx, const float* maxima_dev, const double* rot_dev, int n_views,
                                   int n_landmarks, int image_size, double* starts_dev, double* ends_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, maxima_dev && rot_dev && starts_dev && ends_dev, "estimate_lines: null pointer");
    MVLM_REQUIRE(ctx, n_views > 0 && n_landmarks > 0 && image_size > 0, "estimate_lines: empty problem");
    const int total = n_views * n_landmarks;
    hipLaunchKernelGGL(lines_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, maxima_dev, rot_dev, n_views,
                       n_landmarks, float(image_size), starts_dev, ends_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int mvlm_consensus_mask(mvlm_ctx* ctx, const float* maxima_dev, int n_views, int n_landmarks, int mode,
                                   double q, double thr, uint8_t* mask_dev, int32_t* count_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, maxima_dev && mask_dev && count_dev, "consensus_mask: null pointer");
    MVLM_REQUIRE(ctx, n_views > 0 && n_views <= MAX_VIEWS && n_landmarks > 0, "consensus_mask: 1..1024 views supported");
    MVLM_REQUIRE(ctx, mode == MVLM_MODE_QUANTILE || mode == MVLM_MODE_ABSOLUTE,
                 "Unknown mode for line matching in Estimator");
    MVLM_REQUIRE(ctx, mode != MVLM_MODE_QUANTILE || (q >= 0.0 && q <= 1.0), "Quantiles must be in the range [0, 1]");
    hipLaunchKernelGGL(mask_kernel, dim3(n_landmarks), dim3(64), 0, ctx->stream, maxima_dev, n_views, mode, float(q),
                       float(thr), mask_dev, count_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int mvlm_consensus_solve(mvlm_ctx* ctx, const double* starts_dev, const double* ends_dev,
                                    const uint8_t* mask_dev, const int32_t* count_dev, const int32_t* draws_dev,
                                    int n_views, int n_landmarks, double* out_dev, double* err_dev) {
    MVLM_ENTER(ctx);
    MVLM_REQUIRE(ctx, starts_dev && ends_dev && mask_dev && draws_dev && out_dev && err_dev,
                 "consensus_solve: null pointer");
    MVLM_REQUIRE(ctx, n_views > 0 && n_views <= MAX_VIEWS && n_landmarks > 0, "consensus_solve: 1..1024 views supported");
    hipLaunchKernelGGL(solve_kernel, dim3(n_landmarks), dim3(64), 0, ctx->stream, starts_dev, ends_dev, mask_dev,
                       count_dev, draws_dev, n_views, out_dev, err_dev);
    MVLM_CHECK_HIP(ctx, hipGetLastError());
    return 0;
}
