// kernels_lbfgs.hpp — L-BFGS search direction (include/pyipm_lbfgs.h; /root/reference/pyipm.py:1007-1182).
//
// The Gram matrix J'J and the factorisation of G reuse k_update and the block LDL' unchanged; what is here
// is the HBM-bound remainder: two passes over J (J'V and J R with skinny V, R of <= 2m+1 columns), the
// skinny-skinny reductions over n, the 2m x 2m dense solve and the element-wise glue.
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
// grid (ldp/256, nsplit, ceil(rr/LB_CC))
__global__ __launch_bounds__(256) void k_tall_tn(double* __restrict__ part, int64_t ldp, const double* __restrict__ JT,
                                                 int64_t ldj, const double* __restrict__ V, int rr, int64_t n,
                                                 int64_t kper)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= ldp) return;                     // ldp is a multiple of 128, not of the block size
    const int c0 = blockIdx.z * LB_CC;
    const int64_t k0 = (int64_t)blockIdx.y * kper;
    int64_t k1 = k0 + kper; if (k1 > n) k1 = n;
    double acc[LB_CC];
    #pragma unroll
    for (int u = 0; u < LB_CC; ++u) acc[u] = 0.0;
    const int nc = rr - c0 < LB_CC ? rr - c0 : LB_CC;
    int64_t k = k0;
    for (; k + 4 <= k1; k += 4) {             // four independent loads in flight per thread
        const double a0 = JT[j + k * ldj], a1 = JT[j + (k + 1) * ldj], a2 = JT[j + (k + 2) * ldj], a3 = JT[j + (k + 3) * ldj];
        const double* vr = V + k * rr + c0;
        #pragma unroll
        for (int u = 0; u < LB_CC; ++u)
            if (u < nc) acc[u] = fma(a3, vr[3 * rr + u], fma(a2, vr[2 * rr + u], fma(a1, vr[rr + u], fma(a0, vr[u], acc[u]))));
    }
    for (; k < k1; ++k) {
        const double a = JT[j + k * ldj];
        const double* vr = V + k * rr + c0;
        #pragma unroll
        for (int u = 0; u < LB_CC; ++u) if (u < nc) acc[u] = fma(a, vr[u], acc[u]);
    }
    #pragma unroll
    for (int u = 0; u < LB_CC; ++u)
        if (u < nc) part[((int64_t)blockIdx.y * rr + c0 + u) * ldp + j] = acc[u];
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

// Right-hand sides of the block solves, in place on P (column-major, ld = ldp):
//   column 0 :  J'g_x + [0 ; -zeta g_s / Sigma] - zeta g2     (= zeta (B' inv(A) g1 - g2); the factor is of zeta G)
//   column c :  -J'W_c                                          (= -zeta (J'W_c / zeta))
__global__ __launch_bounds__(256) void k_lb_rhs(double* __restrict__ P, int64_t ldp, int rr, int64_t p, int64_t me,
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
    P[(int64_t)c * ldp + j] = v;
}

// T[k*rr + c] = sum_j JT[j + k*ldj] * R[j + c*ldr]       (J R)
// One block = 64 rows k; J is walked in 64 x 64 tiles staged through LDS (the global read is coalesced along j, the
// compute wants one row per thread), the matching 64 x nc slab of R beside it; the next tile is already in flight
// (registers) while the current one is consumed.  Thread (kr, cg): row kr, columns cg, cg+4, ...
// grid (ceil(n/64), ceil(rr/LB_CC)), block 256.
constexpr int LB_NNU = (LB_CC + 3) / 4;
constexpr int LB_NNR = 64;
__global__ __launch_bounds__(256) void k_tall_nn(double* __restrict__ T, int rr, const double* __restrict__ JT,
                                                 int64_t ldj, const double* __restrict__ R, int64_t ldr, int64_t p,
                                                 int64_t n)
{
    __shared__ double tile[64][65];
    __shared__ double Rs[64][LB_CC + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kr = lane, cg = wave;
    const int c0 = blockIdx.y * LB_CC;
    const int nc = rr - c0 < LB_CC ? rr - c0 : LB_CC;
    const int64_t k0 = (int64_t)blockIdx.x * 64;
    double acc[LB_NNU];
    #pragma unroll
    for (int u = 0; u < LB_NNU; ++u) acc[u] = 0.0;
    double pre[16];
    #pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t k = k0 + wave * 16 + i;
        pre[i] = (k < n && lane < p) ? JT[lane + k * ldj] : 0.0;
    }
    for (int64_t j0 = 0; j0 < p; j0 += 64) {
        #pragma unroll
        for (int i = 0; i < 16; ++i) tile[wave * 16 + i][lane] = pre[i];
        for (int t = tid; t < 64 * nc; t += 256) {
            const int jj = t & 63, c = t >> 6;
            Rs[jj][c] = (j0 + jj < p) ? R[(j0 + jj) + (int64_t)(c0 + c) * ldr] : 0.0;
        }
        __syncthreads();
        if (j0 + 64 < p) {
            #pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t k = k0 + wave * 16 + i, j = j0 + 64 + lane;
                pre[i] = (k < n && j < p) ? JT[j + k * ldj] : 0.0;
            }
        }
        #pragma unroll 8
        for (int jj = 0; jj < 64; ++jj) {
            const double a = tile[kr][jj];
            #pragma unroll
            for (int u = 0; u < LB_NNU; ++u) if (cg + 4 * u < nc) acc[u] = fma(a, Rs[jj][cg + 4 * u], acc[u]);
        }
        __syncthreads();
    }
    if (k0 + kr < n) {
        #pragma unroll
        for (int u = 0; u < LB_NNU; ++u) if (cg + 4 * u < nc) T[(k0 + kr) * rr + c0 + cg + 4 * u] = acc[u];
    }
}

// E = inv(A_x) applied:  E[k][0] = (V[k][0] - T[k][0]) / zeta   (Zg_x)
//                        E[k][c] = (V[k][c] + T[k][c]) / zeta   (X01_x), in place on T
__global__ __launch_bounds__(256) void k_lb_E(double* __restrict__ T, const double* __restrict__ V, int rr, int64_t n,
                                              double zeta, int have_T)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * rr) return;
    const int c = (int)(idx % rr);
    const double t = have_T ? T[idx] : 0.0;
    T[idx] = (c == 0 ? V[idx] - t : V[idx] + t) / zeta;
}

// part[b][a*ne + e] = sum_{k in block b's chunks} V[k*rr + 1 + a] * E[k*ldE + e]     a < r, e < ne
// (W'E over the n variables; deterministic: fixed block count, second pass sums the partials)
__global__ __launch_bounds__(256) void k_small_gram(double* __restrict__ part, const double* __restrict__ V, int rr,
                                                    const double* __restrict__ E, int ldE, int ne, int r, int64_t n)
{
    extern __shared__ double sm[];
    double* sV = sm;                       // LB_GCH x r
    double* sE = sm + LB_GCH * r;          // LB_GCH x ne
    const int nout = r * ne;
    constexpr int MAXO = 16;               // r*ne <= 64*65 = 4160 <= 256*17: see the host check
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
            sV[t] = kk < rows ? V[(k0 + kk) * rr + 1 + a] : 0.0;
        }
        for (int t = threadIdx.x; t < LB_GCH * ne; t += 256) {
            const int kk = t / ne, e = t - kk * ne;
            sE[t] = kk < rows ? E[(k0 + kk) * ldE + e] : 0.0;
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
        if (best < pmin) pmin = best;
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

// out[k] = a0 * E[k*rr] - sum_{c<r} E[k*rr + 1 + c] * v[c]          (x rows of Zg - X10 v11, or zeta g - W c)
__global__ __launch_bounds__(256) void k_lb_comb_x(double* __restrict__ out, const double* __restrict__ E, int rr,
                                                   int64_t n, const double* __restrict__ v, int r, double a0)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double* row = E + k * rr;
    double t = a0 * row[0];
    for (int c = 0; c < r; ++c) t = fma(-row[1 + c], v[c], t);
    out[k] = t;
}

// u[j] = R[j] + sum_c R[j + (1+c)*ldr] v[c]  (= y - (-X00) v11);  dz_lambda[j] = sgn*u[j];
// dz_s[i] = (g_s[i] + u[me+i]) / Sigma_i      (rows of Zg - X10 v11 below x; uses X01_s = -X00_i / Sigma)
__global__ __launch_bounds__(256) void k_lb_comb_ls(double* __restrict__ dz, const double* __restrict__ R, int64_t ldr,
                                                    int64_t p, int64_t me, int64_t n, int64_t mi,
                                                    const double* __restrict__ g, const double* __restrict__ sig,
                                                    const double* __restrict__ v, int r, double sgn)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    double u = R[j];
    for (int c = 0; c < r; ++c) u = fma(R[j + (int64_t)(1 + c) * ldr], v[c], u);
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
