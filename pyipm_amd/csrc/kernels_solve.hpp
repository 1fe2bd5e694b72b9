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
                                                  double* __restrict__ v, int64_t vstride,
                                                  int64_t a0, int64_t a1, int64_t b0, int64_t b1)   // rows where L can be non-zero
{
    extern __shared__ double y[];
    {   // structural zeros of the KKT factor (active_ranges): these rows of L are exact zeros, nothing to subtract
        const int64_t i0 = row_begin + (int64_t)blockIdx.x * 256, i1 = i0 + 256;
        if (!((i1 > a0 && i0 < a1) || (i1 > b0 && i0 < b1))) return;
    }
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
                                                 int64_t vstride, int64_t pstride,
                                                 int64_t a0, int64_t a1, int64_t b0, int64_t b1)     // rows where L can be non-zero
{
    __shared__ double red[4];
    v += (int64_t)blockIdx.z * vstride;
    part += (int64_t)blockIdx.z * pstride;
    const int k = blockIdx.x;
    const int64_t r0 = row_begin + (int64_t)blockIdx.y * ROWCHUNK;
    int64_t r1 = r0 + ROWCHUNK; if (r1 > Npad) r1 = Npad;
    if (!((r1 > a0 && r0 < a1) || (r1 > b0 && r0 < b1))) {          // a chunk of structural zeros (active_ranges)
        if (threadIdx.x == 0) part[(int64_t)blockIdx.y * nb + k] = 0.0;
        return;
    }
    const double* col = A + (lc0 + k) * ld;
    // all loads of a thread in flight at once (a rolled loop waits out the memory latency every trip), 16 bytes each:
    // a thread takes rows (2 t, 2 t + 1) + 512 u -- row_begin, ROWCHUNK and the leading dimension are even, so the pairs
    // are aligned; a wave reads 1 KB per instruction
    typedef double d2_t __attribute__((ext_vector_type(2)));
    constexpr int PER = ROWCHUNK / 512;
    d2_t ca[PER], va[PER];
    #pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = r0 + 2 * threadIdx.x + 512 * u;
        if (i + 1 < r1) {
            ca[u] = *reinterpret_cast<const d2_t*>(col + i);
            va[u] = *reinterpret_cast<const d2_t*>(v + i);
        } else {
            ca[u].x = (i < r1) ? col[i] : 0.0; ca[u].y = 0.0;
            va[u].x = (i < r1) ? v[i] : 0.0;   va[u].y = 0.0;
        }
    }
    double acc = 0.0;
    #pragma unroll
    for (int u = 0; u < PER; ++u) acc = fma(ca[u].y, va[u].y, fma(ca[u].x, va[u].x, acc));
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

// In-panel backward substitution on 1024 threads (panels of at most 4 tiles).  k_bwd_diag lets thread k walk down column k
// of every tile -- 64 loads per step that no other thread shares a cache line with, three dependent steps: ~9 us, half of
// the backward sweep.  Here the six in-panel tiles are requested up front, coalesced (lane = row), and go through shared
// memory once per step, where thread (column j, row quarter q) reads them transposed: a step is 16 multiply-adds and two
// barriers.  (A first attempt -- one launch per panel with the far-row dots of the NEXT panel riding along, wave w reducing
// 64 rows of column w, w + 16, ... -- took 20-50 us per launch: 52 wave reductions per wave are ~9 us of DPP traffic
// alone, and 1024-thread dot blocks stream at half the rate of k_bwd_dot's.)
__global__ __launch_bounds__(1024) void k_bwd_diag4(const double* __restrict__ A, int64_t ld, int64_t lc0, int64_t c0,
                                                    int nbw, int nb, const double* __restrict__ part, int nchunk,
                                                    double* __restrict__ v, int64_t vstride, int64_t pstride)
{
    __shared__ double Ls[3][TB][TB + 1];             // the tiles of the current step, [t][i][j]
    __shared__ double x[4 * TB], ps[4][4 * TB];
    const int tid = threadIdx.x, k = tid & 255, q = tid >> 8;
    v += (int64_t)blockIdx.y * vstride;
    part += (int64_t)blockIdx.y * pstride;
    const int nt = nbw / TB;
    // tiles (u, t), t < u <= 3, in the order (1,0) (2,0) (2,1) (3,0) (3,1) (3,2); element e = tid + 1024 r: row e & 63, column e >> 6
    double lt[6][4];
    {
        int idx = 0;
        #pragma unroll
        for (int u = 1; u <= 3; ++u)
            #pragma unroll
            for (int t = 0; t < u; ++t, ++idx)
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = tid + 1024 * r;
                    lt[idx][r] = (u < nt) ? A[(c0 + (int64_t)u * TB + (e & 63)) + (lc0 + t * TB + (e >> 6)) * ld] : 0.0;
                }
    }
    {   // partial sums of the rows below: four interleaved chains per column, combined in a fixed order
        double t = 0.0;
        if (k < nbw) for (int c = q; c < nchunk; c += 4) t += part[(int64_t)c * nb + k];
        ps[q][k] = t;
    }
    __syncthreads();
    if (tid < nbw) x[tid] = v[c0 + tid] - ((ps[0][tid] + ps[1][tid]) + (ps[2][tid] + ps[3][tid]));
    #pragma unroll
    for (int u = 3; u >= 1; --u) {
        if (u < nt) {
            const int base = u * (u - 1) / 2;              // first tile of step u in lt[]
            #pragma unroll
            for (int t = 0; t < 3; ++t)
                if (t < u) {
                    #pragma unroll
                    for (int r = 0; r < 4; ++r) { const int e = tid + 1024 * r; Ls[t][e & 63][e >> 6] = lt[base + t][r]; }
                }
            __syncthreads();                                // tiles staged, x of tile u final
            double acc = 0.0;
            if (k < u * TB) {
                const int t = k >> 6, j = k & 63;
                #pragma unroll
                for (int i = 0; i < 16; ++i) acc = fma(Ls[t][16 * q + i][j], x[u * TB + 16 * q + i], acc);
            }
            ps[q][k] = acc;
            __syncthreads();
            if (tid < u * TB) x[tid] -= (ps[0][tid] + ps[1][tid]) + (ps[2][tid] + ps[3][tid]);
        }
    }
    __syncthreads();
    if (tid < nbw) v[c0 + tid] = x[tid];
}

// (Tried and dropped, r02, to shorten the backward sweep -- 2.6 ms exposed at N = 32768, 128 x (k_bwd_dot 9 us + k_bwd_diag 11 us):
//  (a) dots and in-panel block in ONE launch with an arrival counter: the few fat blocks that keep the atomics cheap stream
//      at a fraction of k_bwd_dot's rate, 8.1 ms;  (b) the in-panel recursion replaced by a dense product with the inverse
//      of the panel's unit block-triangular block, formed once per factorisation on fp64 MFMA beside it: the one-block
//      kernel takes the same ~10 us -- it is the dependent launch, not the three recursion steps, that costs -- and the
//      128 extra inversions per step load the GPU during the factorisation.  Kept: 16-byte loads in k_bwd_dot, 9.6 -> 8.9 us.)

// v[i] = b[i] on the rows of the panels this rank owns, 0 elsewhere: the ranks' vectors sum to b (distributed sweeps)
__global__ __launch_bounds__(256) void k_mask_owned(double* __restrict__ v, const double* __restrict__ b, Geo g)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.Npad) return;
    v[i] = (g.owner(i / g.nb) == g.rank) ? b[i] : 0.0;
}

}  // namespace pyipm
