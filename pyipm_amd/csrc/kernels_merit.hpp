// kernels_merit.hpp — the merit-function pieces of the line search on the device.
// Reference: merit function phi and its directional derivative pyipm.py:670-721, the backtracking search :1438-1565, the
// merit-parameter update :1727-1735, the KKT report :958-991, the barrier update :1804-1814.  All reductions are
// HBM-bound passes over O(n + me + mi) numbers, deterministic (fixed partition, fixed order of the partial sums: the
// Armijo decisions of a run do not depend on the launch), one D2H of a handful of doubles per call.
#pragma once
#include "ctx.hpp"

namespace pyipm {

constexpr int MERIT_NQ = 16;      // quantities of k_merit_info
constexpr int MERIT_NB = 64;      // partial blocks

// block-wide sum of one value per thread (256 threads), result valid in thread 0; fixed order
__device__ __forceinline__ double merit_block_sum(double v, double* red) {
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double merit_block_min(double v, double* red) {
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
}

// Partial sums of block b over its fixed share of every vector:  part[b * MERIT_NQ + q]
//   q0  sum |ce|                 q1  sum |ci - s|            (||c||_1 of phi, dphi and the nu update)
//   q2  df . dx                  q3  sum ds / (s + eps)      (dphi = q2 - nu (q0 + q1) - mu q3;  nu threshold: (q2 - mu q3) / ((1 - rho)(q0 + q1)))
//   q4  sum log s                                            (phi = f + nu (q0 + q1) - mu q4)
//   q5  sum g_x^2   q6  sum (g_s s)^2   q7  sum ce^2   q8  sum (ci - s)^2     (KKT report, from g = -grad of the staged point)
//   q9  sum s lda_i              q10 min s lda_i             (barrier update)
//   q11 sum dx^2                 q12 sum ds^2                ("search direction unreliable" test)
// dz == NULL leaves q2, q3, q11, q12 at zero; g == NULL leaves q5, q6 at zero.
__global__ __launch_bounds__(256) void k_merit_info(
    double* __restrict__ part, Geo geo, const double* __restrict__ df, const double* __restrict__ ce,
    const double* __restrict__ ci, const double* __restrict__ s, const double* __restrict__ lda,
    const double* __restrict__ dz, const double* __restrict__ g, double eps)
{
    __shared__ double red[4];
    const int64_t n = geo.n, me = geo.me, mi = geo.mi;
    const int b = blockIdx.x, nb = gridDim.x;
    double q[MERIT_NQ];
    #pragma unroll
    for (int k = 0; k < MERIT_NQ; ++k) q[k] = 0.0;
    q[10] = 1.0e308;
    auto range = [&](int64_t len, int64_t* a, int64_t* e) {
        const int64_t per = (len + nb - 1) / nb;
        *a = (int64_t)b * per; *e = *a + per; if (*e > len) *e = len; if (*a > len) *a = len;
    };
    int64_t a, e;
    range(n, &a, &e);
    for (int64_t i = a + threadIdx.x; i < e; i += 256) {
        if (dz) { const double d = dz[i]; q[2] = fma(df[i], d, q[2]); q[11] = fma(d, d, q[11]); }
        if (g) { const double v = g[i]; q[5] = fma(v, v, q[5]); }
    }
    range(me, &a, &e);
    for (int64_t i = a + threadIdx.x; i < e; i += 256) {
        const double c = ce[i];
        q[0] += fabs(c); q[7] = fma(c, c, q[7]);
    }
    range(mi, &a, &e);
    for (int64_t i = a + threadIdx.x; i < e; i += 256) {
        const double si = s[i], r = ci[i] - si, li = lda[me + i];
        q[1] += fabs(r); q[8] = fma(r, r, q[8]);
        q[4] += log(si);
        const double sl = si * li;
        q[9] += sl; q[10] = fmin(q[10], sl);
        if (dz) { const double d = dz[n + i]; q[3] += d / (si + eps); q[12] = fma(d, d, q[12]); }
        if (g) { const double v = g[n + i] * si; q[6] = fma(v, v, q[6]); }
    }
    #pragma unroll
    for (int k = 0; k < MERIT_NQ; ++k) {
        const double t = (k == 10) ? merit_block_min(q[k], red) : merit_block_sum(q[k], red);
        if (threadIdx.x == 0) part[b * MERIT_NQ + k] = t;
    }
}

// out[q] = the MERIT_NB partials of quantity q added in block order (min for q10); one block of MERIT_NQ threads... 64
__global__ __launch_bounds__(64) void k_merit_info_final(double* __restrict__ out, const double* __restrict__ part, int nblocks)
{
    const int k = threadIdx.x;
    if (k >= MERIT_NQ) return;
    double t = (k == 10) ? 1.0e308 : 0.0;
    for (int b = 0; b < nblocks; ++b) { const double v = part[b * MERIT_NQ + k]; t = (k == 10) ? fmin(t, v) : t + v; }
    out[k] = t;
}

// out[p] = a_p . b_p for up to 8 pairs of device vectors (f(x) = 1/2 x'Qx + c'x and the like): one block per pair,
// fixed order.
struct DotPairs { const double* a[8]; const double* b[8]; int64_t len[8]; };
__global__ __launch_bounds__(1024) void k_dots(double* __restrict__ out, DotPairs dp)
{
    __shared__ double red[16];
    const int p = blockIdx.x;
    const double* __restrict__ a = dp.a[p];
    const double* __restrict__ b = dp.b[p];
    double t = 0.0;
    for (int64_t i = threadIdx.x; i < dp.len[p]; i += 1024) t = fma(a[i], b[i], t);
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int w = 0; w < 16; ++w) r += red[w];
        out[p] = r;
    }
}

// |c + a d| - |c| without the cancellation: while c + a d keeps the sign of c the change IS sign(c) a d (one rounding,
// relative to the change -- for the a ~ 1e-12 of the last backtracking steps the subtraction of two numbers of size |c| would
// leave eps |c|, many times the change itself); across a sign change the two magnitudes are comparable and subtract safely.
__device__ __forceinline__ double abs_change(double c, double d, double a) {
    const double t = a * d, cn = fma(a, d, c);
    if (c > 0.0 && cn >= 0.0) return t;
    if (c < 0.0 && cn <= 0.0) return -t;
    return fabs(cn) - fabs(c);
}

// The merit function of a QP along the ray, as a DIFFERENCE (pyipm.py:670-721 evaluated at x + a dx, s + a ds minus at x, s;
// f quadratic, constraints affine, so everything is a closed form in a):
//   out[k] = a g1 + a^2/2 g2 + nu sum(|ce + a dce| - |ce|) + nu sum(|r + a dr| - |r|) - mu sum log1p(a ds / s)
// with r = ci - s, dr = dci - ds, g1 = df . dx, g2 = dx' Q dx (gq[0], gq[1] on the device), dce = Je' dx, dci = Ji' dx.  The
// difference is formed element by element, so its error is relative to the CHANGE of phi, not to phi -- what keeps the
// Armijo test meaningful for the tiny steps of the last iterations.  One block per candidate a = alphas[k]: every
// backtracking candidate a tau^k of a search is evaluated by one launch.
__global__ __launch_bounds__(256) void k_merit_ray(
    double* __restrict__ out, const double* __restrict__ alphas, Geo geo, const double* __restrict__ gq,
    const double* __restrict__ ce, const double* __restrict__ dce, const double* __restrict__ ci,
    const double* __restrict__ dci, const double* __restrict__ s, const double* __restrict__ ds, double nu, double mu)
{
    __shared__ double red[4];
    const double a = alphas[blockIdx.x];
    const int64_t me = geo.me, mi = geo.mi;
    double se = 0.0, si = 0.0, sl = 0.0;
    for (int64_t i = threadIdx.x; i < me; i += 256) se += abs_change(ce[i], dce[i], a);
    for (int64_t i = threadIdx.x; i < mi; i += 256) {
        const double s0 = s[i], d = ds[i], r = ci[i] - s0, dr = dci[i] - d;
        si += abs_change(r, dr, a);
        sl += log1p(a * (d / s0));
    }
    se = merit_block_sum(se, red);
    si = merit_block_sum(si, red);
    sl = merit_block_sum(sl, red);
    if (threadIdx.x == 0) {
        double v = a * gq[0] + (0.5 * a * a) * gq[1];
        if (me) v = v + nu * se;
        if (mi) v = v + nu * si - mu * sl;
        out[blockIdx.x] = v;
    }
}

}  // namespace pyipm
