// kernels_solve.hpp — K5 block substitutions with the block-LDL' factor (HBM-bound).
// Replaces the triangular solves inside scipy.linalg.solve reached from
// /root/reference/pyipm.py:911-914,1720-1721.
//
//   forward : y_k = b_k - sum_{j<k} Lb[k,j] y_j          (unit BLOCK lower triangular: no
//   diagonal: z_k = inv(T_k) y_k                            in-tile triangular solve at all)
//   backward: x_k = z_k - sum_{i>k} Lb[i,k]' x_i
// organised per panel (nb columns): a one-workgroup kernel resolves the nb x nb diagonal
// block, streaming kernels handle everything below it.
//
// Several right-hand sides: every kernel takes a vector stride and reads its right-hand side index from the
// last grid dimension (v + index*vstride); single-vector callers launch that dimension as 1 with stride 0.
#pragma once
#include "ctx.hpp"

namespace pyipm {

// In-panel forward substitution on the nbw x nbw diagonal block.  blockDim = nbw.
__global__ void k_fwd_diag(const double* __restrict__ A, int64_t ld, int64_t lc0, int64_t c0,
                           int nbw, double* __restrict__ v, int64_t vstride)
{
    extern __shared__ double y[];
    const int tid = threadIdx.x;
    v += (int64_t)blockIdx.y * vstride;
    y[tid] = v[c0 + tid];
    const int nt = nbw / TB;
    for (int u = 0; u + 1 < nt; ++u) {
        __syncthreads();
        if (tid >= (u + 1) * TB) {
            const double* col = A + (c0 + tid) + (lc0 + (int64_t)u * TB) * ld;
            double acc = 0.0;
            #pragma unroll 8
            for (int k = 0; k < TB; ++k) acc = fma(col[(int64_t)k * ld], y[u * TB + k], acc);
            y[tid] -= acc;
        }
    }
    __syncthreads();
    v[c0 + tid] = y[tid];
}

// Rows below the panel:  v[i] -= sum_{k<nbw} A[i, lc0+k] * v[c0+k].   One thread per row.
__global__ __launch_bounds__(256) void k_fwd_gemv(const double* __restrict__ A, int64_t ld, int64_t lc0,
                                                  int64_t c0, int nbw, int64_t row_begin, int64_t Npad,
                                                  double* __restrict__ v, int64_t vstride)
{
    extern __shared__ double y[];
    v += (int64_t)blockIdx.y * vstride;
    for (int k = threadIdx.x; k < nbw; k += 256) y[k] = v[c0 + k];
    __syncthreads();
    const int64_t i = row_begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad) return;
    const double* row = A + i + lc0 * ld;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    #pragma unroll 4                                   // 16 loads in flight per thread instead of 4 (rolled: one latency per trip)
    for (int k = 0; k < nbw; k += 4) {
        acc0 = fma(row[(int64_t)(k + 0) * ld], y[k + 0], acc0);
        acc1 = fma(row[(int64_t)(k + 1) * ld], y[k + 1], acc1);
        acc2 = fma(row[(int64_t)(k + 2) * ld], y[k + 2], acc2);
        acc3 = fma(row[(int64_t)(k + 3) * ld], y[k + 3], acc3);
    }
    v[i] -= (acc0 + acc1) + (acc2 + acc3);
}

// z_k = inv(T_k) y_k for the tiles of one panel, refined nref times against T_k itself
// (r = y - T z ; z += inv(T) r): the explicit inverse alone leaves a residual of cond(T_k)*eps.
// grid = tiles, block = 64.
__global__ __launch_bounds__(64) void k_diag_apply(const double* __restrict__ Dinv, const double* __restrict__ Tsave,
                                                   const double* __restrict__ Tflag, int nref, int64_t tile0,
                                                   int64_t c0, double* __restrict__ v, int64_t vstride)
{
    __shared__ double y[TB];
    __shared__ double w[TB];
    const int lane = threadIdx.x;
    v += (int64_t)blockIdx.y * vstride;
    const int64_t base = c0 + (int64_t)blockIdx.x * TB;
    const double y0 = v[base + lane];
    y[lane] = y0;
    __syncthreads();
    const double* X = Dinv + (tile0 + blockIdx.x) * (int64_t)(TB * TB);
    const double* T = Tsave + (tile0 + blockIdx.x) * (int64_t)(TB * TB);
    double z = 0.0;
    #pragma unroll 8
    for (int j = 0; j < TB; ++j) z = fma(X[j * TB + lane], y[j], z);
    if (Tflag[tile0 + blockIdx.x] == 0.0) nref = 0;      // well-conditioned tile
    for (int it = 0; it < nref; ++it) {
        w[lane] = z;
        __syncthreads();
        double r = y0;
        #pragma unroll 8
        for (int j = 0; j < TB; ++j) r = fma(-T[j * TB + lane], w[j], r);
        __syncthreads();
        w[lane] = r;
        __syncthreads();
        #pragma unroll 8
        for (int j = 0; j < TB; ++j) z = fma(X[j * TB + lane], w[j], z);
        __syncthreads();
    }
    v[base + lane] = z;
}

// Backward, rows below the panel: part[chunk][k] = sum_{i in chunk} A[i, lc0+k] * v[i].
// grid = (nbw, nchunk), block 256; deterministic (no atomics).
__global__ __launch_bounds__(256) void k_bwd_dot(const double* __restrict__ A, int64_t ld, int64_t lc0,
                                                 int nb, int64_t row_begin, int64_t Npad,
                                                 const double* __restrict__ v, double* __restrict__ part,
                                                 int64_t vstride, int64_t pstride)
{
    __shared__ double red[4];
    v += (int64_t)blockIdx.z * vstride;
    part += (int64_t)blockIdx.z * pstride;
    const int k = blockIdx.x;
    const int64_t r0 = row_begin + (int64_t)blockIdx.y * ROWCHUNK;
    int64_t r1 = r0 + ROWCHUNK; if (r1 > Npad) r1 = Npad;
    const double* col = A + (lc0 + k) * ld;
    // all ROWCHUNK/256 loads of a thread in flight at once (a rolled loop waits out the memory latency every trip:
    // 8 x ~1.2 us was the whole kernel); same products in the same order, so the same bits
    constexpr int PER = ROWCHUNK / 256;
    double ca[PER], va[PER];
    #pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = r0 + threadIdx.x + 256 * u;
        const bool ok = i < r1;
        ca[u] = ok ? col[i] : 0.0;
        va[u] = ok ? v[i] : 0.0;
    }
    double acc = 0.0;
    #pragma unroll
    for (int u = 0; u < PER; ++u) acc = fma(ca[u], va[u], acc);
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(int64_t)blockIdx.y * nb + k] = (red[0] + red[1]) + (red[2] + red[3]);
}

// In-panel backward substitution.  blockDim = nbw.
__global__ void k_bwd_diag(const double* __restrict__ A, int64_t ld, int64_t lc0, int64_t c0,
                           int nbw, int nb, const double* __restrict__ part, int nchunk,
                           double* __restrict__ v, int64_t vstride, int64_t pstride)
{
    extern __shared__ double x[];
    const int tid = threadIdx.x;
    v += (int64_t)blockIdx.y * vstride;
    part += (int64_t)blockIdx.y * pstride;
    double t = 0.0;
    {   // partial sums of the rows below: independent loads four at a time, added in the original order
        int c = 0;
        for (; c + 4 <= nchunk; c += 4) {
            const double p0 = part[(int64_t)c * nb + tid], p1 = part[(int64_t)(c + 1) * nb + tid];
            const double p2 = part[(int64_t)(c + 2) * nb + tid], p3 = part[(int64_t)(c + 3) * nb + tid];
            t += p0; t += p1; t += p2; t += p3;
        }
        for (; c < nchunk; ++c) t += part[(int64_t)c * nb + tid];
    }
    x[tid] = v[c0 + tid] - t;
    const int nt = nbw / TB;
    for (int u = nt - 1; u >= 1; --u) {
        __syncthreads();
        if (tid < u * TB) {
            const double* col = A + (c0 + (int64_t)u * TB) + (lc0 + tid) * ld;
            double acc = 0.0;
            #pragma unroll                                  // all 64 loads in flight: the recursion has nt - 1 latency-bound steps
            for (int i = 0; i < TB; ++i) acc = fma(col[i], x[u * TB + i], acc);
            x[tid] -= acc;
        }
    }
    __syncthreads();
    v[c0 + tid] = x[tid];
}

// ---------------------------------------------------------------------------------------------
// Backward sweep without the in-panel recursion (single right-hand side).
//
// k_bwd_diag resolves the panel's own nb x nb unit block-triangular block by nb/64 - 1 dependent steps in ONE
// workgroup: 11 us x 128 panels was more than the streaming part of the whole sweep at N = 32768.  Instead
// MT_p = inv(Lb_pp)' is formed ONCE per factorisation for every panel in parallel (k_panel_inv; Lb_pp is unit BLOCK
// lower triangular with entries bounded by the pivoting, nb <= 1024), and k_bwd_apply computes
// x_p = MT_p (z_p - t) as one dense product (t = the partial column sums of k_bwd_dot, added in chunk order).
// (Tried and dropped, r02: dots and product in ONE launch with an arrival counter -- the few fat blocks that keep the
// atomics cheap stream at a fraction of k_bwd_dot's rate: 8.1 ms per sweep against 2.6.)
// ---------------------------------------------------------------------------------------------
// v[i] = b[i] on the rows of the panels this rank owns, 0 elsewhere: the ranks' vectors sum to b (distributed sweeps)
__global__ __launch_bounds__(256) void k_mask_owned(double* __restrict__ v, const double* __restrict__ b, Geo g)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.Npad) return;
    v[i] = (g.owner(i / g.nb) == g.rank) ? b[i] : 0.0;
}

// MT (nb x nb per local panel, column-major, ld = nb):  MT[j + i*nb] = inv(Lb_pp)[i][j]  (upper triangular, unit
// block diagonal).  Tiles: M[t][t] = I,  M[t][u] = -( L[t][u] + sum_{u<k<t} L[t][k] M[k][u] ),  t > u.
// grid = local panels lp0, lp0 + 1, ...; block 256.
__global__ __launch_bounds__(256) void k_panel_inv(const double* __restrict__ A, int64_t ld, Geo g, double* __restrict__ MT, int64_t lp0)
{
    __shared__ double La[TB][TB + 1];
    __shared__ double Mb[TB][TB + 1];
    const int64_t lp = lp0 + blockIdx.x, p = lp * g.world + g.rank;
    if (p >= g.npanels) return;
    const int64_t c0 = g.panel_c0(p), lc0 = lp * (int64_t)g.nb;
    const int nbw = (int)g.panel_w(p), nt = nbw / TB, nb = g.nb;
    double* M = MT + lp * (int64_t)nb * nb;
    const int tid = threadIdx.x, r = tid & 63, cq = tid >> 6;
    // identity diagonal tiles, zero below the block diagonal (the consumer reads whole columns of MT)
    for (int e = tid; e < nbw * nbw; e += 256) {
        const int j = e % nbw, i = e / nbw;
        if ((j >> 6) >= (i >> 6)) M[j + (int64_t)i * nb] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int d = 1; d < nt; ++d)                        // by block sub-diagonal: M[t][u] needs M[k][u], u < k < t (closer ones)
        for (int u = 0; u + d < nt; ++u) {
            const int t = u + d;
            double acc[16];
            #pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = A[(c0 + t * TB + r) + (lc0 + u * TB + cq * 16 + c) * ld];    // L[t][u]
            for (int k = u + 1; k < t; ++k) {
                __syncthreads();
                for (int e = tid; e < TB * TB; e += 256) {
                    const int i = e & 63, j = e >> 6;
                    La[i][j] = A[(c0 + t * TB + i) + (lc0 + k * TB + j) * ld];                               // L[t][k][i][j]
                    Mb[i][j] = M[(u * TB + j) + (int64_t)(k * TB + i) * nb];                               // M[k][u][i][j] = MT[(u,j),(k,i)]
                }
                __syncthreads();
                #pragma unroll 4
                for (int kk = 0; kk < TB; ++kk) {
                    const double a = La[r][kk];
                    #pragma unroll
                    for (int c = 0; c < 16; ++c) acc[c] = fma(a, Mb[kk][cq * 16 + c], acc[c]);
                }
            }
            #pragma unroll
            for (int c = 0; c < 16; ++c) M[(u * TB + cq * 16 + c) + (int64_t)(t * TB + r) * nb] = -acc[c];   // MT[(u,c),(t,r)] = M[t][u][r][c]
            __syncthreads();                             // (writes of this tile visible to the block before a later tile reads them)
        }
}

// x_p = MT (z_p - sum of the partial column sums).  One block of 1024 threads: thread t takes column j = t % nbw and
// the i-range part t / nbw of 1024 / nbw parts, its loads of MT 32 at a time in flight (two memory latencies for the
// whole product at nb = 256: MT_p is cold, a rolled loop paid ~1 us per 16 rows), partial results joined through LDS.
__global__ __launch_bounds__(1024) void k_bwd_apply(const double* __restrict__ MT, int64_t c0, int nbw, int nb,
                                                    const double* __restrict__ part, int nchunk, double* __restrict__ v)
{
    __shared__ double ys[1024];                              // nbw <= 1024
    __shared__ double acc[1024];
    const int tid = threadIdx.x;
    for (int j = tid; j < nbw; j += 1024) {
        double t = 0.0;
        int c = 0;
        for (; c + 4 <= nchunk; c += 4) {                    // independent loads four at a time, added in chunk order
            const double p0 = part[(int64_t)c * nb + j], p1 = part[(int64_t)(c + 1) * nb + j];
            const double p2 = part[(int64_t)(c + 2) * nb + j], p3 = part[(int64_t)(c + 3) * nb + j];
            t += p0; t += p1; t += p2; t += p3;
        }
        for (; c < nchunk; ++c) t += part[(int64_t)c * nb + j];
        ys[j] = v[c0 + j] - t;
    }
    __syncthreads();
    int parts = 1024 / nbw;                                  // nbw in {128, 256, ..., 1024}
    if (parts > nbw / 32) parts = nbw / 32;                  // a part is a whole number of 32-row load groups
    const int j = tid % nbw, pt = tid / nbw;
    double a = 0.0;
    if (pt < parts) {
        const int len = nbw / parts;                         // rows per part (a multiple of 64 unless nbw < 1024 / 16 ...)
        const int i0 = pt * len, i1 = i0 + len;
        const int lo = (j >> 6) << 6;                        // MT[j][i] = 0 for i < tile(j): skip those rows
        const double* m = MT + j;
        for (int i = i0; i < i1; i += 32) {                  // 32 loads in flight (1024 threads leave 128 VGPRs each)
            if (i + 32 <= lo) continue;
            double mv[32];
            #pragma unroll
            for (int q = 0; q < 32; ++q) mv[q] = m[(int64_t)(i + q) * nb];
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            #pragma unroll
            for (int q = 0; q < 32; q += 4) {
                a0 = fma(mv[q], ys[i + q], a0); a1 = fma(mv[q + 1], ys[i + q + 1], a1);
                a2 = fma(mv[q + 2], ys[i + q + 2], a2); a3 = fma(mv[q + 3], ys[i + q + 3], a3);
            }
            a += (a0 + a1) + (a2 + a3);
        }
    }
    acc[tid] = a;
    __syncthreads();
    if (tid < nbw) {
        double x = 0.0;
        for (int q = 0; q < parts; ++q) x += acc[q * nbw + tid];
        v[c0 + tid] = x;
    }
}

}  // namespace pyipm
