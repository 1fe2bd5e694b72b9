// kernels_lbfgs.hpp — L-BFGS search direction (include/pyipm_lbfgs.h; /root/reference/pyipm.py:1007-1182).
//
// The Gram matrix J'J and the factorisation of G reuse k_update and the block LDL' unchanged; what is here
// is the HBM-bound remainder: two passes over J (J'V with skinny V of 2m+1 columns, then J u with ONE vector:
// everything the 2m x 2m system needs from J X00 is P'X00 with P = J'W already at hand, see lbfgs_impl.hpp), the
// skinny-skinny reductions, the 2m x 2m dense solve and the element-wise glue.
//
// Layouts.  JT: p_pad x n_pad column-major, JT[j + k*ldj] = J[k][j]  (= the caller's row-major J with padded
// rows: no transposition on staging, and exactly the operand layout k_update wants for C += JT * JT').
// Skinny matrices over n (V, T/E): row-major n x rr (a row is one short contiguous read, uniform per wave in
// the J'V pass).  Skinny matrices over p (P/R): column-major ld = p_pad (each column is a right-hand side of
// the block solves).
#pragma once
#include "ctx.hpp"

namespace pyipm {

constexpr int LB_CC = 17;         // skinny columns per pass over J (2m+1 = 17 for m = 8: one pass)
constexpr int LB_GCH = 32;        // rows per LDS chunk in the skinny-skinny reduction
constexpr int LB_GBLK = 256;      // blocks (= partial sums) of that reduction

// V[k][0] = g[k] ; V[k][1+c] = cS * S[k][c] ; V[k][1+m+c] = cY * Y[k][c]
__global__ __launch_bounds__(256) void k_lb_pack(double* __restrict__ V, int rr, const double* __restrict__ g,
                                                 const double* __restrict__ S, int64_t ldS,
                                                 const double* __restrict__ Y, int64_t ldY, int64_t n, int m,
                                                 double cS, double cY)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * rr) return;
    const int64_t k = idx / rr;
    const int c = (int)(idx - k * rr);
    double v;
    if (c == 0) v = g[k];
    else if (c <= m) v = cS * S[k * ldS + (c - 1)];
    else v = cY * Y[k * ldY + (c - 1 - m)];
    V[idx] = v;
}

__global__ __launch_bounds__(256) void k_lb_sigma(double* __restrict__ sig, const double* __restrict__ s,
                                                  const double* __restrict__ lam_i, double eps, int64_t mi)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < mi) sig[i] = lam_i[i] / (s[i] + eps);
}

// part[(split*rr + c)*ldp + j] = sum_{k in split} JT[j + k*ldj] * V[k*rr + c]      (J'V, one thread per j)
// grid (ldp/256, nsplit, ceil(rr/LB_CC)).  The V rows are wave-uniform (scalar loads, one wide load per row when the
// chunk is full); the next four J values are already in flight while the current four are consumed.
template <bool FULL>
__device__ __forceinline__ void tall_tn_body(double* __restrict__ part, int64_t ldp, const double* __restrict__ JT,
                                             int64_t ldj, const double* __restrict__ V, int rr, int64_t k0, int64_t k1,
                                             int64_t j, int c0, int nc)
{
    double acc[LB_CC];
    #pragma unroll
    for (int u = 0; u < LB_CC; ++u) acc[u] = 0.0;
    int64_t k = k0;
    if (k + 4 <= k1) {
        const double* col = JT + j + k * ldj;
        double a0 = col[0], a1 = col[ldj], a2 = col[2 * ldj], a3 = col[3 * ldj];
        for (; k + 4 <= k1; k += 4) {
            const int64_t kn = (k + 8 <= k1) ? k + 4 : k;                // past the end: re-read the current four (harmless)
            const double* cn = JT + j + kn * ldj;
            const double b0 = cn[0], b1 = cn[ldj], b2 = cn[2 * ldj], b3 = cn[3 * ldj];
            const double* vr = V + k * rr + c0;
            #pragma unroll
            for (int u = 0; u < LB_CC; ++u)
                if (FULL || u < nc)
                    acc[u] = fma(a3, vr[3 * rr + u], fma(a2, vr[2 * rr + u], fma(a1, vr[rr + u], fma(a0, vr[u], acc[u]))));
            a0 = b0; a1 = b1; a2 = b2; a3 = b3;
        }
    }
    for (; k < k1; ++k) {
        const double a = JT[j + k * ldj];
        const double* vr = V + k * rr + c0;
        #pragma unroll
        for (int u = 0; u < LB_CC; ++u) if (FULL || u < nc) acc[u] = fma(a, vr[u], acc[u]);
    }
    #pragma unroll
    for (int u = 0; u < LB_CC; ++u)
        if (FULL || u < nc) part[((int64_t)blockIdx.y * rr + c0 + u) * ldp + j] = acc[u];
}

__global__ __launch_bounds__(256) void k_tall_tn(double* __restrict__ part, int64_t ldp, const double* __restrict__ JT,
                                                 int64_t ldj, const double* __restrict__ V, int rr, int64_t n,
                                                 int64_t kper)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ldp) return;                     // ldp is a multiple of 128, not of the block size
    const int c0 = blockIdx.z * LB_CC;
    const int64_t k0 = (int64_t)blockIdx.y * kper;
    int64_t k1 = k0 + kper; if (k1 > n) k1 = n;
    const int nc = rr - c0 < LB_CC ? rr - c0 : LB_CC;
    if (nc == LB_CC) tall_tn_body<true>(part, ldp, JT, ldj, V, rr, k0, k1, j, c0, nc);
    else tall_tn_body<false>(part, ldp, JT, ldj, V, rr, k0, k1, j, c0, nc);
}

// P[c*ldp + j] = sum_split part[(split*rr + c)*ldp + j]        grid (ldp/256, rr)
__global__ __launch_bounds__(256) void k_tall_tn_reduce(double* __restrict__ P, const double* __restrict__ part,
                                                        int64_t ldp, int rr, int nsplit)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ldp) return;
    const int c = blockIdx.y;
    double t = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) t += part[((int64_t)sp * rr + c) * ldp + j];
    P[(int64_t)c * ldp + j] = t;
}

// Right-hand sides of the block solves, R from P (both column-major, ld = ldp; P = J'V is kept):
//   column 0 :  J'g_x + [0 ; -zeta g_s / Sigma] - zeta g2     (= zeta (B' inv(A) g1 - g2); the factor is of zeta G)
//   column c :  -J'W_c                                          (= -zeta (J'W_c / zeta))
__global__ __launch_bounds__(256) void k_lb_rhs(double* __restrict__ R, const double* __restrict__ P, int64_t ldp, int rr, int64_t p, int64_t me,
                                                const double* __restrict__ g, int64_t n, int64_t mi,
                                                const double* __restrict__ sig, double zeta)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (j >= ldp) return;
    double v = 0.0;
    if (j < p) {
        v = P[(int64_t)c * ldp + j];
        if (c == 0) {
            v -= zeta * g[n + mi + j];
            if (j >= me) v -= zeta * g[n + (j - me)] / sig[j - me];
        } else {
            v = -v;
        }
    }
    R[(int64_t)c * ldp + j] = v;
}

// out[k] = sum_j JT[j + k*ldj] * u[j]        (J u: one streaming pass, a wave per row, four rows in flight)
__global__ __launch_bounds__(256) void k_jvec(double* __restrict__ out, const double* __restrict__ JT, int64_t ldj,
                                              const double* __restrict__ u, int64_t p, int64_t n)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t k0 = ((int64_t)blockIdx.x * 4 + wave) * 4; k0 < n; k0 += (int64_t)gridDim.x * 16) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const double* c0 = JT + k0 * ldj;
        const bool h1 = k0 + 1 < n, h2 = k0 + 2 < n, h3 = k0 + 3 < n;
        for (int64_t j = lane; j < p; j += 64) {
            const double uj = u[j];
            a0 = fma(c0[j], uj, a0);
            if (h1) a1 = fma(c0[j + ldj], uj, a1);
            if (h2) a2 = fma(c0[j + 2 * ldj], uj, a2);
            if (h3) a3 = fma(c0[j + 3 * ldj], uj, a3);
        }
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64);
            a2 += __shfl_xor(a2, off, 64); a3 += __shfl_xor(a3, off, 64);
        }
        if (lane == 0) {
            out[k0] = a0;
            if (h1) out[k0 + 1] = a1;
            if (h2) out[k0 + 2] = a2;
            if (h3) out[k0 + 3] = a3;
        }
    }
}

// part[b][a*ne + e] = sum_{k in block b's chunks} V(k, a) * E(k, e)     a < r, e < ne, k < n
// with V(k, a) = V[k*vrs + (voff + a)*vcs], E(k, e) = E[k*ers + e*ecs]: skinny-skinny products over a long index for
// row-major operands over n (W'V) and column-major ones over p (P'R).  Deterministic: fixed block count, a second
// pass sums the partials.
__global__ __launch_bounds__(256) void k_small_gram(double* __restrict__ part, const double* __restrict__ V, int64_t vrs,
                                                    int64_t vcs, int voff, const double* __restrict__ E, int64_t ers,
                                                    int64_t ecs, int ne, int r, int64_t n)
{
    extern __shared__ double sm[];
    double* sV = sm;                       // LB_GCH x r
    double* sE = sm + LB_GCH * r;          // LB_GCH x ne
    const int nout = r * ne;
    constexpr int MAXO = 16;               // r*ne <= 64*65 = 4160 <= 256*17
    double acc[MAXO + 1];
    #pragma unroll
    for (int u = 0; u <= MAXO; ++u) acc[u] = 0.0;
    const int64_t nchunks = (n + LB_GCH - 1) / LB_GCH;
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int64_t k0 = ch * LB_GCH;
        const int rows = (int)(n - k0 < LB_GCH ? n - k0 : LB_GCH);
        __syncthreads();
        for (int t = threadIdx.x; t < LB_GCH * r; t += 256) {
            const int kk = t / r, a = t - kk * r;
            sV[t] = kk < rows ? V[(k0 + kk) * vrs + (int64_t)(voff + a) * vcs] : 0.0;
        }
        for (int t = threadIdx.x; t < LB_GCH * ne; t += 256) {
            const int kk = t / ne, e = t - kk * ne;
            sE[t] = kk < rows ? E[(k0 + kk) * ers + (int64_t)e * ecs] : 0.0;
        }
        __syncthreads();
        #pragma unroll
        for (int u = 0; u <= MAXO; ++u) {
            const int o = threadIdx.x + u * 256;
            if (o < nout) {
                const int a = o / ne, e = o - a * ne;
                double t = acc[u];
                #pragma unroll 8
                for (int kk = 0; kk < LB_GCH; ++kk) t = fma(sV[kk * r + a], sE[kk * ne + e], t);
                acc[u] = t;
            }
        }
    }
    #pragma unroll
    for (int u = 0; u <= MAXO; ++u) {
        const int o = threadIdx.x + u * 256;
        if (o < nout) part[(int64_t)blockIdx.x * nout + o] = acc[u];
    }
}

__global__ __launch_bounds__(256) void k_small_gram_reduce(double* __restrict__ out, const double* __restrict__ part,
                                                           int nout, int nblk)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= nout) return;
    double t = 0.0;
    for (int b = 0; b < nblk; ++b) t += part[(int64_t)b * nout + o];
    out[o] = t;
}

// x = inv(M) b for one small dense system (r <= 64), LU with partial pivoting in LDS — the algorithm LAPACK's
// gesv applies behind the reference's sym_solve (pyipm.py:18-20).  One wave; lane i owns row i.
//   M[i][j] = (M1 ? M1[i*ld1 + off1 + j] : 0) + s2 * M2[i*r + j] ;   b[i] = B[i*ldb]
// info[0] = smallest |pivot| (0: singular -> x = NaN).
__global__ __launch_bounds__(64) void k_small_solve(double* __restrict__ x, double* __restrict__ info,
                                                    const double* __restrict__ M1, int ld1, int off1,
                                                    const double* __restrict__ M2, double s2,
                                                    const double* __restrict__ B, int ldb, int r)
{
    __shared__ double a[64][65];
    __shared__ double b[64];
    const int i = threadIdx.x;
    for (int j = 0; j < r; ++j) {
        double v = 0.0;
        if (i < r) {
            if (M1) v = M1[i * ld1 + off1 + j];
            if (M2) v += s2 * M2[i * r + j];
        }
        a[i][j] = v;
    }
    b[i] = i < r ? B[i * ldb] : 0.0;
    __syncthreads();
    double pmin = 1.0e308;
    for (int k = 0; k < r; ++k) {
        double best = (i >= k && i < r) ? fabs(a[i][k]) : -1.0;
        int who = i;
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int ow = __shfl_xor(who, off, 64);
            if (ob > best || (ob == best && ow < who)) { best = ob; who = ow; }
        }
        { const double bm = best > 0.0 ? best : 0.0; if (bm < pmin) pmin = bm; }      // NaN after a zero pivot counts as 0
        if (who != k) {                         // swap rows k and who: lane j moves column j
            if (i < r) { const double t = a[k][i]; a[k][i] = a[who][i]; a[who][i] = t; }
            if (i == 0) { const double t = b[k]; b[k] = b[who]; b[who] = t; }
        }
        __syncthreads();
        if (i > k && i < r) {
            const double f = a[i][k] / a[k][k];
            for (int j = k + 1; j < r; ++j) a[i][j] = fma(-f, a[k][j], a[i][j]);
            b[i] = fma(-f, b[k], b[i]);
        }
        __syncthreads();
    }
    for (int k = r - 1; k >= 0; --k) {
        const double xk = b[k] / a[k][k];
        __syncthreads();
        if (i == k) b[k] = xk;
        if (i < k) b[i] = fma(-a[i][k], xk, b[i]);
        __syncthreads();
    }
    if (i < r) x[i] = b[i];
    if (i == 0) info[0] = pmin;
}

// Hs = (W'V + sgn * P_w'R) / zeta, sgn = -1 on column 0, +1 elsewhere:
//   column 0     = W'Zg_x  = (W'g_x - (J'W)'y) / zeta
//   columns 1..  = W'X01_x = (W'W + (J'W)'X00) / zeta            (X00 = -inv(zeta G) J'W carries the minus)
__global__ __launch_bounds__(256) void k_lb_hs(double* __restrict__ Hs, const double* __restrict__ Ha,
                                               const double* __restrict__ Hb, int r, int rr, double zeta)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= r * rr) return;
    const int e = o % rr;
    Hs[o] = (e == 0 ? Ha[o] - Hb[o] : Ha[o] + Hb[o]) / zeta;
}

// out[k] = a0 * E[k*rr] - sum_{c<r} E[k*rr + 1 + c] * v[c]                    (unconstrained: zeta g - W c)
// with Ju: out[k] = (E[k*rr] - sum_c E[k*rr + 1 + c] v[c] - Ju[k]) * a0      (x rows of Zg - X10 v11, a0 = 1/zeta:
//                                                                              (g_x - W v11 - J u) / zeta)
__global__ __launch_bounds__(256) void k_lb_comb_x(double* __restrict__ out, const double* __restrict__ E, int rr,
                                                   int64_t n, const double* __restrict__ v, int r, double a0,
                                                   const double* __restrict__ Ju)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double* row = E + k * rr;
    if (Ju) {
        double t = row[0] - Ju[k];
        for (int c = 0; c < r; ++c) t = fma(-row[1 + c], v[c], t);
        out[k] = t * a0;
    } else {
        double t = a0 * row[0];
        for (int c = 0; c < r; ++c) t = fma(-row[1 + c], v[c], t);
        out[k] = t;
    }
}

// u[j] = R[j] + sum_c R[j + (1+c)*ldr] v[c]  (= y - (-X00) v11), kept for the J u pass;  dz_lambda[j] = sgn*u[j];
// dz_s[i] = (g_s[i] + u[me+i]) / Sigma_i      (rows of Zg - X10 v11 below x; uses X01_s = -X00_i / Sigma)
__global__ __launch_bounds__(256) void k_lb_comb_ls(double* __restrict__ dz, double* __restrict__ uout,
                                                    const double* __restrict__ R, int64_t ldr,
                                                    int64_t p, int64_t me, int64_t n, int64_t mi,
                                                    const double* __restrict__ g, const double* __restrict__ sig,
                                                    const double* __restrict__ v, int r, double sgn)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    double u = R[j];
    for (int c = 0; c < r; ++c) u = fma(R[j + (int64_t)(1 + c) * ldr], v[c], u);
    uout[j] = u;
    dz[n + mi + j] = sgn * u;
    if (j >= me) dz[n + (j - me)] = (g[n + (j - me)] + u) / sig[j - me];
}

// A = sum of the split-K partial Gram matrices (ns buffers of `count` doubles, stride `count`; fixed order)
__global__ __launch_bounds__(256) void k_lb_ksum(double* __restrict__ A, const double* __restrict__ Cs, int64_t count, int ns)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double t = 0.0;
    for (int sp = 0; sp < ns; ++sp) t += Cs[(int64_t)sp * count + i];
    A[i] = t;
}

// Diagonal of zeta*G on top of the Gram launch; identity on the padding.
__global__ __launch_bounds__(256) void k_lb_gram_diag(double* __restrict__ A, int64_t ld, int64_t p, int64_t me,
                                                      const double* __restrict__ sig, double zeta, double reg_e)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ld) return;
    double* d = A + j + j * ld;
    if (j < me) *d += zeta * reg_e;
    else if (j < p) *d += zeta / sig[j - me];
    else *d = 1.0;
}

}  // namespace pyipm
