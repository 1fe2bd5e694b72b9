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

// ---------------------------------------------------------------------------------------------
// The whole backward sweep in ONE launch (round 3).  Per panel the launches above are k_bwd_dot (HBM-bound, 7.6 us on
// average at N = 32768) and k_bwd_diag4 (one workgroup, 8.6 us of dependent latencies): 127 x 16 us = 2.06 ms for 4.3 GB,
// 26 % of what HBM delivers, and 24 x 10 us of config 2's 2.9 ms.  Here the sweep is RIGHT-looking and device-driven:
//   * workgroup 0 walks the panels from the last to the first; for panel s it waits until the contributions of all later
//     panels have reached y_s, resolves the panel's own diagonal block (the three steps of k_bwd_diag4), stores x_s and
//     raises flag[s];
//   * every other wave owns fixed groups of 8 columns (group g -> wave g mod #waves) and, as soon as flag[t] is up, subtracts
//     L[rows of panel t, its columns]' x_t from its entries of y -- 2 KB per column, read coalesced (lane = 4 rows), reduced
//     across the wave by a butterfly that folds 8 columns in 10 shuffles.  The columns of panel t - 1 (the only ones
//     workgroup 0 waits for immediately) come first and are counted apart (near[t]); everything else of step t is needed
//     two panels later (a progress word per workgroup).  L is static: the loads of a step are issued BEFORE its flag is polled.
// A flag hand-off between workgroups costs 0.6 us when nothing is fenced (tools/ubench/pingpong.hip; 2.7 us with
// __threadfence on both sides: buffer_wbl2 / buffer_inv), so every value that crosses workgroups inside this kernel (x, y,
// flags, counters) is read and written with relaxed agent-scope atomics -- they bypass the non-coherent cache levels -- and
// ordered by s_waitcnt alone.  Per panel workgroup 0 spends (tools/sweep_clock.py, N = 6144): 2.8 us in the recursion of the
// diagonal block (LDS-bound: 32 ds_read per thread and step; from the registers with __shfl_xor folds it took 4.4 us -- 84
// ds_bpermute), 0.5 us publishing x, and 3.7 us until the near sums are back (flag seen, x loaded, 8 columns folded, sum
// stored, seen): 6.9 us a panel where two dependent launches took 10-16.  (Also tried: the recursion on the matrix pipe, the
// tiles loaded as v_mfma_f64_16x16x4 A operands -- 16 columns x 32 bytes per instruction: the uncoalesced 192 KB took one
// CU ~5 us to fetch and every poll queued behind them, 11.4 us a panel.)  Polling the VALUES of x instead of the flag saved
// nothing on 32 waves and cost bandwidth on 4080.  Ownership is static and every sum has a fixed order: deterministic.  Every poll carries a
// timeout (a workgroup that is not resident would otherwise hang the GPU): on expiry *err is set, the sweep ends and
// the vector is poisoned with NaN.
struct SweepGeo {
    int64_t Npad, ld, n, mi, me;
    int nb, npanels, skip;              // skip: structural zeros of the KKT factor (active_ranges on the host)
};
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// can L[rows [r0, r1), columns of the panel [cA, cB)] be non-zero?  (active_ranges, pyipm_newton.hip)
__device__ __forceinline__ bool sweep_active(const SweepGeo& sg, int64_t cA, int64_t cB, int64_t r0, int64_t r1) {
    if (!sg.skip || sg.mi == 0) return true;
    const int64_t s0 = sg.n, s1 = sg.n + sg.mi, i0 = sg.n + sg.mi + sg.me;
    if (cB <= s0) return (r0 < s0) || (r1 > s1);
    if (cA >= s0 && cB <= s1) return r1 > i0 + (cA - s0) && r0 < i0 + (cB - s0);
    return true;
}
// poll *p until it reaches `want`; false on timeout or when another workgroup gave up
__device__ __forceinline__ bool sweep_wait(const unsigned* p, unsigned want, unsigned* err, unsigned long long timeout) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if ((unsigned long long)wall_clock64() - t0 > timeout) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    }
}

__global__ __launch_bounds__(1024) void k_bwd_sweep(const double* __restrict__ A, SweepGeo sg, double* v, unsigned* sync,
                                                    unsigned* err, unsigned long long timeout,
                                                    double* nearbuf,                    // Npad doubles, filled with NaN before the launch
                                                    unsigned long long* dbg)            // diagnostics (NULL normally): workgroup 0's phases, tools/sweep_clock.py
{
    __shared__ double Ls[3][TB][TB + 1];
    __shared__ double x[4 * TB], ps[4][4 * TB];
    __shared__ int ok_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = sg.npanels, nb = sg.nb;
    unsigned* flag = sync; unsigned* nearc = sync + P; unsigned* prog = sync + 2 * P;     // prog[w]: steps column-owner wave w has completed
    const int gpp = nb / 8;                                    // groups of 8 columns per panel
    if (blockIdx.x == 0) {
        const int k = tid & 255, q = tid >> 8;
        if (tid == 0) ok_s = 1;
        __syncthreads();
        for (int s = P - 1; s >= 0; --s) {
            const int64_t c0 = (int64_t)s * nb;
            int64_t w64 = sg.Npad - c0; if (w64 > nb) w64 = nb;
            const int nbw = (int)w64, nt = nbw / TB;
            // tiles (u, t), t < u <= 3, of the panel's diagonal block: requested up front, coalesced (see k_bwd_diag4)
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
                            lt[idx][r] = (u < nt) ? A[(c0 + (int64_t)u * TB + (e & 63)) + (c0 + t * TB + (e >> 6)) * sg.ld] : 0.0;
                        }
            }
            if (dbg && tid == 0) dbg[4 * s + 0] = wall_clock64();
            if (tid < nbw) {
                // y_s with every contribution but the one of panel s + 1, which arrives as values, not as a count: a column's
                // entry of nearbuf turns from NaN into its sum (one round trip instead of counter + load; should the sum BE
                // NaN -- a broken factor -- the count near[s + 1] says so)
                const double yv = ld_agent(v + c0 + tid);
                double nv = 0.0;
                if (s + 1 < P) {
                    const unsigned long long t0 = wall_clock64();
                    for (;;) {
                        nv = ld_agent(nearbuf + c0 + tid);
                        if (nv == nv) break;
                        if (__hip_atomic_load(nearc + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)gpp) { nv = ld_agent(nearbuf + c0 + tid); break; }
                        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok_s = 0; break; }
                        if ((unsigned long long)wall_clock64() - t0 > timeout) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok_s = 0; break; }
                    }
                }
                x[tid] = yv - nv;
            }
            __syncthreads();
            if (!ok_s) break;
            if (dbg && tid == 0) dbg[4 * s + 1] = wall_clock64();
            #pragma unroll
            for (int u = 3; u >= 1; --u) {
                if (u < nt) {
                    const int base = u * (u - 1) / 2;
                    #pragma unroll
                    for (int t = 0; t < 3; ++t)
                        if (t < u) {
                            #pragma unroll
                            for (int r = 0; r < 4; ++r) { const int e = tid + 1024 * r; Ls[t][e & 63][e >> 6] = lt[base + t][r]; }
                        }
                    __syncthreads();
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
            if (dbg && tid == 0) dbg[4 * s + 2] = wall_clock64();
            if (tid < nbw) st_agent(v + c0 + tid, x[tid]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (dbg && tid == 0) dbg[4 * s + 3] = wall_clock64();
            if (tid == 0) __hip_atomic_store(flag + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // for the NEXT panel (s - 1): the owners of ITS columns have finished step s + 1 -- thread j polls the progress word of
            // the wave that owns group j of that panel (ownership is static: group g -> wave g mod #waves).  Until round 5 this
            // waited for EVERY owner to finish the step (a progress word per workgroup): where the sweep is bandwidth-bound --
            // the first half at N = 32768: 64 ... 32 MB of far columns per step against 6.5 us of chain -- the chain then ran at
            // the pace of the slowest owner of columns it would not need for a hundred panels (3.1 us of this wait per panel
            // against 1.5 at N = 6144); now those owners fall behind and catch up while the chain is the bound.
            // (One counter per step took 4080 atomics on one address: 27 us a step.)
            if (s + 1 < P && s >= 1 && tid < gpp) {
                const int nwv_o = ((int)gridDim.x - 1 - (gpp + 15) / 16) * 16;
                if (!sweep_wait(prog + ((s - 1) * gpp + tid) % nwv_o, (unsigned)(P - (s + 1)), err, timeout)) ok_s = 0;
            }
            __syncthreads();
            if (!ok_s) break;
        }
        __syncthreads();
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
            for (int64_t i = tid; i < sg.Npad; i += 1024) st_agent(v + i, __builtin_nan(""));
        return;
    }
    // ---- column owners ----
    const int nearb_ = (gpp + 15) / 16;
    const bool specialist_ = (int)blockIdx.x <= nearb_;
    const int gw_ = ((int)blockIdx.x - 1 - nearb_) * 16 + wave;
    auto publish = [&](unsigned steps_done) {                   // (wave-uniform) this wave has completed that many steps
        if (lane == 0 && !specialist_) __hip_atomic_store(prog + gw_, steps_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // The first `nearb` workgroups are NEAR specialists: wave j of theirs takes, at every step t, group j of the columns of
    // panel t - 1 -- the sums workgroup 0 is waiting for -- and nothing else, so it is already polling flag[t] when it goes
    // up (as an ordinary owner it might still be on its far group of step t + 1: workgroup 0 waited 4.5 us per panel at
    // N = 32768 against 2.1 at N = 6144).  Their sums go to nearbuf, never into y: no conflict with the columns' owners.
    const int nearb = (gpp + 15) / 16;
    const bool specialist = (int)blockIdx.x <= nearb;
    const int nwv = specialist ? 1 : ((int)gridDim.x - 1 - nearb) * 16;
    const int gw = specialist ? ((int)blockIdx.x - 1) * 16 + wave : ((int)blockIdx.x - 1 - nearb) * 16 + wave;
    if (specialist) {
        publish((unsigned)P);                                  // (nothing of theirs is ever waited for through the progress words)
        if (gw >= gpp) return;
    }
    typedef double d2_t __attribute__((ext_vector_type(2)));
    for (int t = P - 1; t >= 1; --t) {
        const int64_t r0 = (int64_t)t * nb;
        int64_t w64 = sg.Npad - r0; if (w64 > nb) w64 = nb;
        const int nbw = (int)w64;
        const int glim = t * gpp;                              // groups [0, glim) lie left of panel t
        if (!specialist && gw >= glim - gpp) break;            // nothing left for this wave, now or later
        // my groups left of panel t - 1, the highest first: g = gw + k nwv  (a specialist: the one group glim - gpp + gw)
        int kmax = specialist ? 0 : (glim - gpp - 1 - gw) / nwv;
        const bool in_rows = 4 * lane < nbw;
        bool have_x = false;
        double xr[4] = {0.0, 0.0, 0.0, 0.0};
        for (int kk = kmax; kk >= 0; --kk) {
            const int g = specialist ? glim - gpp + gw : gw + kk * nwv;
            const int64_t j0 = (int64_t)g * 8;
            const int64_t cp = (j0 / nb) * nb;                 // panel of these columns
            const bool nearg = specialist;
            const bool act = sweep_active(sg, cp, cp + nb, r0, r0 + nbw);
            d2_t la[8][2];
            if (act) {
                #pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (in_rows) {
                        const double* src = A + (r0 + 4 * lane) + (j0 + c) * sg.ld;
                        la[c][0] = *reinterpret_cast<const d2_t*>(src);
                        la[c][1] = *reinterpret_cast<const d2_t*>(src + 2);
                    } else { la[c][0] = d2_t{0.0, 0.0}; la[c][1] = d2_t{0.0, 0.0}; }
                }
            }
            if (!have_x) {
                // x_t: one word polled per wave, then PLAIN loads -- the first wave of an XCD brings the 2 KB into its L2 for
                // the other 500 (no copy of these lines can be stale: nobody reads them before the flag, and polling the
                // values themselves from 4080 waves costs more memory traffic than it saves in round trips)
                if (!sweep_wait(flag + t, 1u, err, timeout)) { publish((unsigned)P); return; }
                asm volatile("" ::: "memory");
                if (in_rows) {
                    const d2_t x01 = *reinterpret_cast<const d2_t*>(v + r0 + 4 * lane), x23 = *reinterpret_cast<const d2_t*>(v + r0 + 4 * lane + 2);
                    xr[0] = x01.x; xr[1] = x01.y; xr[2] = x23.x; xr[3] = x23.y;
                }
                have_x = true;
            }
            if (act) {
                double a[8];
                #pragma unroll
                for (int c = 0; c < 8; ++c)
                    a[c] = fma(la[c][1].y, xr[3], fma(la[c][1].x, xr[2], fma(la[c][0].y, xr[1], la[c][0].x * xr[0])));
                // fold 8 columns x 64 lanes: after the xor-32 / 16 / 8 steps a lane holds ONE column's partial sum
                double b4[4], b2[2];
                const bool h5 = (lane & 32) != 0, h4 = (lane & 16) != 0, h3 = (lane & 8) != 0;
                #pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double send = h5 ? a[i] : a[i + 4], keep = h5 ? a[i + 4] : a[i];
                    b4[i] = keep + __shfl_xor(send, 32, 64);
                }
                #pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const double send = h4 ? b4[i] : b4[i + 2], keep = h4 ? b4[i + 2] : b4[i];
                    b2[i] = keep + __shfl_xor(send, 16, 64);
                }
                double r;
                { const double send = h3 ? b2[0] : b2[1], keep = h3 ? b2[1] : b2[0]; r = keep + __shfl_xor(send, 8, 64); }
                r += __shfl_xor(r, 4, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 1, 64);
                if ((lane & 7) == 0) {
                    const int c = (h5 ? 4 : 0) + (h4 ? 2 : 0) + (h3 ? 1 : 0);
                    if (nearg) st_agent(nearbuf + j0 + c, r);          // workgroup 0 subtracts it (and is polling for it)
                    else { double* yp = v + j0 + c; st_agent(yp, ld_agent(yp) - r); }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (nearg && lane < 8) {
                st_agent(nearbuf + j0 + lane, 0.0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (nearg && lane == 0) __hip_atomic_fetch_add(nearc + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!specialist) publish((unsigned)(P - t));
    }
    if (!specialist) publish((unsigned)P);
}

// ---------------------------------------------------------------------------------------------
// The forward sweep in one launch (the mirror of k_bwd_sweep; refined solves and right-hand sides of their own run it --
// the forward pass of a plain step trails the factorisation panel by panel instead).  Workgroup 0 walks the panels first to
// last: for panel s it waits until the four row chunks of the panel have every earlier panel's contribution (a progress word
// per chunk), resolves the panel's diagonal block (y_u -= sum_{t<u} L[u,t] y_t, a lane per row: no transposition, the 16
// waves' partial sums meet in shared memory) and raises flag[s].  Every other workgroup owns 64-row chunks of the vector for
// the whole launch: wave q of its eight takes an eighth of the panel's columns (32 loads of 512 contiguous bytes per lane,
// issued before the flag is polled), the eight partial sums are added in a fixed order and subtracted from the chunk.
// 512 threads per workgroup, two per CU.  Same hand-over rules as k_bwd_sweep: relaxed agent-scope atomics for everything
// that crosses workgroups inside the launch, s_waitcnt for order, a timeout on every poll.
__global__ __launch_bounds__(512, 4) void k_fwd_sweep(const double* __restrict__ A, SweepGeo sg, double* v, unsigned* sync,
                                                      unsigned* err, unsigned long long timeout)
{
    __shared__ double xs[4 * TB];
    __shared__ double ps[8][TB];
    __shared__ int ok_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = sg.npanels, nb = sg.nb;
    unsigned* flag = sync; unsigned* prog = sync + P;           // prog[c]: steps applied to the 64-row chunk c
    const int cpp = nb / TB;                                    // chunks per panel
    const int nchunks = (int)(sg.Npad / TB);
    if (tid == 0) ok_s = 1;
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int s = 0; s < P; ++s) {
            const int64_t c0 = (int64_t)s * nb;
            int64_t w64 = sg.Npad - c0; if (w64 > nb) w64 = nb;
            const int nbw = (int)w64, nt = nbw / TB;
            // tiles (u, t), t < u <= 3: element e = tid + 512 r is row e & 63 = lane, column e >> 6 = wave + 8 r
            double lt[6][8];
            {
                int idx = 0;
                #pragma unroll
                for (int u = 1; u <= 3; ++u)
                    #pragma unroll
                    for (int t = 0; t < u; ++t, ++idx)
                        #pragma unroll
                        for (int r = 0; r < 8; ++r)
                            lt[idx][r] = (u < nt) ? A[(c0 + (int64_t)u * TB + lane) + (c0 + t * TB + wave + 8 * r) * sg.ld] : 0.0;
            }
            if (s > 0 && tid < nt && !sweep_wait(prog + (int)(c0 / TB) + tid, (unsigned)s, err, timeout)) ok_s = 0;
            __syncthreads();
            if (!ok_s) break;
            if (tid < nbw) xs[tid] = ld_agent(v + c0 + tid);
            __syncthreads();
            #pragma unroll
            for (int u = 1; u <= 3; ++u) {
                if (u < nt) {
                    const int base = u * (u - 1) / 2;
                    double acc = 0.0;
                    #pragma unroll
                    for (int t = 0; t < 3; ++t)
                        if (t < u) {
                            #pragma unroll
                            for (int r = 0; r < 8; ++r) acc = fma(lt[base + t][r], xs[t * TB + wave + 8 * r], acc);
                        }
                    ps[wave][lane] = acc;
                    __syncthreads();
                    if (tid < TB) {
                        double t8 = ((ps[0][tid] + ps[1][tid]) + (ps[2][tid] + ps[3][tid])) + ((ps[4][tid] + ps[5][tid]) + (ps[6][tid] + ps[7][tid]));
                        xs[u * TB + tid] -= t8;
                    }
                    __syncthreads();
                }
            }
            if (tid < nbw) st_agent(v + c0 + tid, xs[tid]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
            for (int64_t i = tid; i < sg.Npad; i += 512) st_agent(v + i, __builtin_nan(""));
        return;
    }
    // ---- row owners: chunk c = cpp + (blockIdx - 1) + k (gridDim - 1) ----
    const int nown = (int)gridDim.x - 1;
    for (int s = 0; s + 1 < P; ++s) {
        const int64_t c0 = (int64_t)s * nb;                     // (every panel but the last is nb wide)
        const int cfirst = (s + 1) * cpp;                       // chunks below panel s
        const int cbeg = cpp + ((int)blockIdx.x - 1);
        // my first chunk below panel s
        int c = cbeg;
        if (c < cfirst) c += ((cfirst - c + nown - 1) / nown) * nown;
        if (c >= nchunks) return;                               // nothing below this panel is mine, now or later
        const int kc = nb / 8;                                  // columns per wave: 8, 16, 24 or 32
        const int64_t jc0 = c0 + (int64_t)wave * kc;
        bool have_y = false;
        for (; c < nchunks; c += nown) {
            const int64_t r0 = (int64_t)c * TB;
            const bool act = sweep_active(sg, c0, c0 + nb, r0, r0 + TB);
            double la[32];
            if (act) {
                #pragma unroll
                for (int j = 0; j < 32; ++j) la[j] = (j < kc) ? A[(r0 + lane) + (jc0 + j) * sg.ld] : 0.0;
            }
            if (!have_y) {
                // (all eight waves agree before anyone leaves: a wave returning alone would leave the others at the barrier)
                if (!sweep_wait(flag + s, 1u, err, timeout)) ok_s = 0;
                __syncthreads();
                if (!ok_s) return;
                asm volatile("" ::: "memory");
                have_y = true;
            }
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            if (act) {
                // y_s: the wave's 32 entries, the same for every lane (scalar loads; first touch of these lines in the launch)
                #pragma unroll
                for (int j = 0; j < 32; j += 4) {
                    const double y0 = (j < kc) ? v[jc0 + j] : 0.0, y1 = (j + 1 < kc) ? v[jc0 + j + 1] : 0.0;
                    const double y2 = (j + 2 < kc) ? v[jc0 + j + 2] : 0.0, y3 = (j + 3 < kc) ? v[jc0 + j + 3] : 0.0;
                    a0 = fma(la[j], y0, a0); a1 = fma(la[j + 1], y1, a1);
                    a2 = fma(la[j + 2], y2, a2); a3 = fma(la[j + 3], y3, a3);
                }
            }
            ps[wave][lane] = (a0 + a1) + (a2 + a3);
            __syncthreads();
            if (wave == 0) {
                if (act) {
                    const double t8 = ((ps[0][lane] + ps[1][lane]) + (ps[2][lane] + ps[3][lane])) + ((ps[4][lane] + ps[5][lane]) + (ps[6][lane] + ps[7][lane]));
                    double* vp = v + r0 + lane;
                    st_agent(vp, ld_agent(vp) - t8);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (lane == 0) __hip_atomic_store(prog + c, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
    }
}

// v[i] = b[i] on the rows of the panels this rank owns, 0 elsewhere: the ranks' vectors sum to b (distributed sweeps)
__global__ __launch_bounds__(256) void k_mask_owned(double* __restrict__ v, const double* __restrict__ b, Geo g)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.Npad) return;
    v[i] = (g.owner(i / g.nb) == g.rank) ? b[i] : 0.0;
}

}  // namespace pyipm
