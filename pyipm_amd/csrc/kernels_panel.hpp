// Factorisation of a diagonal block, "stepped" schedule (round 2): one launch per diagonal tile, one launch per panel for
// the rows below.
//
// What sits on the critical path is the sequence of tile inversions of a diagonal block -- a panel's (256 x 256, the
// per-panel / multi-GPU driver) or a whole group's (1024 x 1024, factor_group: the single-rank schedule) -- each tile
// needing the tile before it eliminated from its own 64 rows.  The first schedule ran, per tile, an inversion launch and
// a scaling (+ in-panel update) launch over ALL rows below -- two dependent launches per tile, the second one as long as
// its K loop over the earlier tiles of the panel -- and a pending update between two panels of a group.  Here the rows are
// split by what waits for them:
//   * k_tile_step, launch t: block 0 eliminates tile t-1 from the 64 rows of tile t (one scaling product, one K = 64
//     update of the diagonal tile, operands from registers) and inverts that tile straight out of shared memory; blocks
//     1.. do the same right-looking step for the other row tiles INSIDE the diagonal block (a far row tile's column
//     tiles split over grid.y).  One dependent launch per tile.
//   * k_panel_rest, once per panel: every 64-row strip below the diagonal block runs all of the panel's stages in one
//     launch (the first schedule went over the slab once per tile, in separate launches that each waited for a tile).
// Every entry sees the same operations in the same order as before (scaling = the k_panel_scale product and refinement;
// updates = MFMA groups of 4 columns, ascending): the factor is bit for bit the one of the first schedule.
#pragma once
#include "kernels_factor.hpp"

namespace pyipm {

// L = S inv(T) for one 64-row strip (wave: 16 rows x 64 columns), nref refinement steps against T; the arithmetic of
// k_panel_scale with sign = +1.  X holds inv(T) on entry (staged, synchronised) and on exit; sb = S in the B-operand map.
// one refinement step of strip_scale (four barriers)
template <bool COH = false>
__device__ __forceinline__ void strip_refine_step(double (&X)[TB][TB + 2], const double* __restrict__ Tinv,
                                                  const double* __restrict__ Tsave, const double (&sb)[16],
                                                  int tid, int l15, int l4, double4_t (&acc)[4]);

// (COH: inv(T) / T come from another workgroup of the same launch, k_tile_chain -- kernels_factor.hpp:ldg_c)
template <bool REFINE = true, bool COH = false>
__device__ __forceinline__ void strip_scale(double (&X)[TB][TB + 2], const double* __restrict__ Tinv,
                                            const double* __restrict__ Tsave, int nref, const double (&sb)[16],
                                            int tid, int l15, int l4, double4_t (&acc)[4])
{
    #pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    #pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        #pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double a = X[t * 16 + l15][ks * 4 + l4];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sb[ks], acc[t], 0, 0, 0);
        }
    }
    if (REFINE)
        for (int it = 0; it < nref; ++it) strip_refine_step<COH>(X, Tinv, Tsave, sb, tid, l15, l4, acc);
}

template <bool COH>
__device__ __forceinline__ void strip_refine_step(double (&X)[TB][TB + 2], const double* __restrict__ Tinv,
                                                  const double* __restrict__ Tsave, const double (&sb)[16],
                                                  int tid, int l15, int l4, double4_t (&acc)[4])
{
    {
        __syncthreads();
        PYIPM_STAGE_TILE_C(X, -1.0, Tsave, COH)
        __syncthreads();
        double4_t res[4];                                            // R = S - L T
        #pragma unroll
        for (int t = 0; t < 4; ++t)
            #pragma unroll
            for (int r = 0; r < 4; ++r) res[t][r] = sb[4 * t + r];
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double lop = acc[ks >> 2][ks & 3];
            #pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double a = X[t * 16 + l15][ks * 4 + l4];
                res[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, lop, res[t], 0, 0, 0);
            }
        }
        __syncthreads();
        PYIPM_STAGE_TILE_C(X, 1.0, Tinv, COH)
        __syncthreads();
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) {                            // L += R inv(T)
            const double rop = res[ks >> 2][ks & 3];
            #pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double a = X[t * 16 + l15][ks * 4 + l4];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rop, acc[t], 0, 0, 0);
            }
        }
    }
}

// C[i][c] += sum_{k < 64} L[i][k] Wn[c][k] for the strip: L from the scaling accumulators (their C/D map is the B-operand
// map), Wn[c][k] = wn[c + k * ldw] (the -S rows of the target column tile, 64 x 64).
template <int KS = 8, bool COH = false>   // k-steps (of 4 columns) whose Wn operands are in flight together
__device__ __forceinline__ void strip_update(double4_t (&c2)[4], const double4_t (&acc)[4], const double* __restrict__ wn,
                                             int64_t ldw, int l15, int l4)
{
    #pragma unroll
    for (int h = 0; h < 16 / KS; ++h) {
        double wa[4][KS];
        #pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            #pragma unroll
            for (int t = 0; t < 4; ++t) wa[t][ks] = ldg_c<COH>(wn + (16 * t + l15) + (int64_t)(4 * (KS * h + ks) + l4) * ldw);
        #pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const double lop = acc[(KS * h + ks) >> 2][(KS * h + ks) & 3];
            #pragma unroll
            for (int t = 0; t < 4; ++t)
                c2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[t][ks], lop, c2[t], 0, 0, 0);
        }
    }
}

// Launch t of a diagonal block of nt tiles (t = 0 .. nt-1), grid = (nt - t, ny).  Block (b, y) works on row tile t + b.
//   t = 0: block 0 inverts tile 0; block b > 0 saves -S of its rows in column tile 0 (W, read by the stages that follow).
//   t > 0: stage t-1 for the rows of tile t + b: L = S inv(T[t-1]) -> A; column tiles t .. t+b of those rows += L Wn'
//          (the ones with (v - t) % ny == y: a row tile far below the diagonal has many); -S of the now final column tile t
//          -> W; block 0 (whose only column tile is the diagonal tile t) then inverts it.
// The diagonal block is a panel's (nt = nb / 64, the per-panel schedule) or a whole group's (factor_group: the panels of a
// group are then chained tile to tile, and what used to be the pending update between them happens here, stage by stage).
__global__ __launch_bounds__(256, 2) void k_tile_step(
    double* __restrict__ A, int64_t ld, int64_t c0, int64_t lc0, int t,      // block: first global / local column; the step
    double* __restrict__ W, int64_t ldw,                                     // its -S buffer: W[row + k * ldw], k < 64 nt
    double* __restrict__ Dinv, double* __restrict__ Tsv, double* __restrict__ Tflag,    // of the block's first tile
    double refine_cond, int nref, DevStats* __restrict__ st, int64_t Nreal, double pivtol_rel,
    const unsigned long long* __restrict__ anorm_bits, int64_t neg_from, unsigned long long* __restrict__ dbg, int blocked)
{
    __shared__ TileScratch sm;
    // inv(T[t-1]) for the scaling lives where the tile inversion will put its stage (and the first bytes of colbuf): it is
    // dead before the diagonal tile is written there
    static_assert(sizeof(TileScratch) >= sizeof(double) * TB * (TB + 2), "X must fit into the tile scratch");
    double (&X)[TB][TB + 2] = *reinterpret_cast<double (*)[TB][TB + 2]>(&sm);
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x, y = blockIdx.y, ny = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t TT = (int64_t)TB * TB;
    if (b == 0 && y != 0) return;
    if (t == 0) {
        if (b == 0) {
            tile_invert_dev(sm, A, ld, c0, lc0, Dinv, Tsv, Tflag, refine_cond, st, Nreal, pivtol_rel, anorm_bits, neg_from, dbg,
                            false, blocked != 0);
        } else if (y == 0) {
            double tmp[TB * TB / 256];
            const int64_t r0 = c0 + (int64_t)b * TB;
            #pragma unroll
            for (int q = 0; q < TB * TB / 256; ++q) {
                const int e = tid + 256 * q;
                tmp[q] = A[(r0 + (e & 63)) + (lc0 + (e >> 6)) * ld];
            }
            #pragma unroll
            for (int q = 0; q < TB * TB / 256; ++q) {
                const int e = tid + 256 * q;
                W[(r0 + (e & 63)) + (int64_t)(e >> 6) * ldw] = -tmp[q];
            }
        }
        return;
    }
    const int tp = t - 1, it = t + b;                                // the stage applied; this block's row tile
#define PYIPM_TS_STAMP(k_) if (dbg && b == 0 && tid == 0) dbg[200 + (k_)] = clock64();     /* diagnostics: tools/tile_clock.py */
    PYIPM_TS_STAMP(0)
    if (y > it - t) return;                                          // fewer column tiles than y-blocks
    const int64_t i = c0 + (int64_t)it * TB + wave * 16 + l15;        // this lane's (global) row
    if (nref > 0 && Tflag[tp] == 0.0) nref = 0;
    PYIPM_STAGE_TILE(X, 1.0, Dinv + tp * TT)
    double sb[16];
    #pragma unroll
    for (int ks = 0; ks < 16; ++ks) sb[ks] = -W[i + (int64_t)(tp * TB + ks * 4 + l4) * ldw];    // S from its saved negative: the
    __syncthreads();                                     // y-blocks of a row tile all need it, and block y = 0 overwrites A with L
    PYIPM_TS_STAMP(1)
    double4_t acc[4];
    strip_scale(X, Dinv + tp * TT, Tsv + tp * TT, nref, sb, tid, l15, l4, acc);
    if (dbg && b == 0) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[3][3])); PYIPM_TS_STAMP(2) }
    if (y == 0) {
        double gmax = 0.0;
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                A[i + (lc0 + tp * TB + tt * 16 + l4 + 4 * r) * ld] = acc[tt][r];
                gmax = fmax(gmax, fabs(acc[tt][r]));
            }
        gmax = wave_max(gmax);
        if (lane == 0) atomicMax(&st->growth_bits, (unsigned long long)__double_as_longlong(gmax));
    }
    PYIPM_TS_STAMP(3)
    if (b == 0) {
        // The critical block: its one column tile is the diagonal tile t, and the Wn operand of that update is -S of ITS OWN
        // 64 rows -- already in the registers of its four waves (sb, B-operand map).  It goes through shared memory into the
        // A-operand map instead of being read back from the W buffer in two dependent batches of global loads (two memory
        // round trips on the chain: 9400 of the prologue's 18500 cycles -> 7000 of 16100, tools/tile_clock.py).  Same values
        // (W holds -S exactly), same products in the same order: the same bits.
        double4_t c2[4];
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            #pragma unroll
            for (int r = 0; r < 4; ++r) c2[tt][r] = A[i + (lc0 + t * TB + tt * 16 + l4 + 4 * r) * ld];
        __syncthreads();                                             // every wave is done with X (inv(T))
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) X[ks * 4 + l4][wave * 16 + l15] = -sb[ks];      // Wn[c][k] = -S[c][k], stored [k][c]
        __syncthreads();
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double lop = acc[ks >> 2][ks & 3];
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                c2[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[ks * 4 + l4][16 * tt + l15], lop, c2[tt], 0, 0, 0);
        }
        __syncthreads();                                             // Wn read: the tile goes where it was
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = tt * 16 + l4 + 4 * r;
                A[i + (lc0 + t * TB + c) * ld] = c2[tt][r];
                sm.stage[wave * 16 + l15][c] = c2[tt][r];
            }
    }
    for (int v = t + y; v <= it && b != 0; v += ny) {
        double4_t c2[4];
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            #pragma unroll
            for (int r = 0; r < 4; ++r) c2[tt][r] = A[i + (lc0 + v * TB + tt * 16 + l4 + 4 * r) * ld];
        strip_update(c2, acc, W + (c0 + (int64_t)v * TB) + (int64_t)(tp * TB) * ldw, ldw, l15, l4);
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = tt * 16 + l4 + 4 * r;
                A[i + (lc0 + v * TB + c) * ld] = c2[tt][r];
                if (b == 0) sm.stage[wave * 16 + l15][c] = c2[tt][r];             // (v == t == it: the diagonal tile)
                else if (v == t) W[i + (int64_t)(t * TB + c) * ldw] = -c2[tt][r];
            }
    }
    PYIPM_TS_STAMP(4)
#undef PYIPM_TS_STAMP
    if (b == 0)
        tile_invert_dev(sm, A, ld, c0 + (int64_t)t * TB, lc0 + (int64_t)t * TB, Dinv + t * TT, Tsv + t * TT, Tflag + t, refine_cond,
                        st, Nreal, pivtol_rel, anorm_bits, neg_from, dbg, /*from_stage=*/true, blocked != 0);
}

// k_tile_step on 512 threads (round 5).  Same arithmetic, same order per entry -- the same bits -- with the work of the
// critical block split over eight waves:
//   * block 0: waves 0..3 are the chain (scaling product, update of the diagonal tile, the eliminations of the blocked
//     inversion); waves 4..7 fetch the diagonal tile while the scaling runs (the chain used to wait out a memory round trip
//     for it after the scaling: requesting it earlier from the same waves spilled registers), save the tile for the
//     refinement, form the column maxima and run the inversion's updates and commits beside the next elimination
//     (tile_blocked8.hpp).  The -S' operand of the diagonal update sits in a buffer of its own, so no barrier separates
//     the two products.
//   * blocks 1..: upb = 2: two (row tile, y) units of k_tile_step per block, one per half (the footprint per CU of two
//     256-thread blocks: what fits beside a bulk update); upb = 1: one unit per block, the other half leaves at once (an
//     idle GPU: two units on one CU share its matrix pipes and the launch would end after its critical block).
// grid = (1 + ceil((nt - t - 1) * ny / upb)); ny as for k_tile_step.
struct TileScratch8 {
    TileScratch ts;
    double X[TB][TB + 2];        // inv(T[t-1]) (block 0; unit of half 0)
    double Wn[TB][TB + 2];       // -S' of the critical rows, [k][c] (block 0); inv(T[t-1]) of the unit of half 1
    Blocked8Scratch ex;
};

__global__ __launch_bounds__(512) void k_tile_step8(
    double* __restrict__ A, int64_t ld, int64_t c0, int64_t lc0, int t, int nt, int ny, int upb,
    double* __restrict__ W, int64_t ldw,
    double* __restrict__ Dinv, double* __restrict__ Tsv, double* __restrict__ Tflag,
    double refine_cond, int nref, DevStats* __restrict__ st, int64_t Nreal, double pivtol_rel,
    const unsigned long long* __restrict__ anorm_bits, int64_t neg_from, unsigned long long* __restrict__ dbg, int blocked)
{
    __shared__ TileScratch8 sm;
    __builtin_amdgcn_s_setprio(3);
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t TT = (int64_t)TB * TB;
    if (blockIdx.x == 0) {
        // ------------------------------------------------ the critical block ------------------------------------------------
        if (half) __builtin_amdgcn_s_setprio(1);                     // the chain's waves win the issue arbitration on their SIMDs
        if (t == 0) {
            tile_invert_dev<true>(sm.ts, A, ld, c0, lc0, Dinv, Tsv, Tflag, refine_cond, st, Nreal, pivtol_rel, anorm_bits, neg_from, dbg,
                                  false, blocked != 0, &sm.ex);
            return;
        }
        const int tp = t - 1;
#define PYIPM_TS_STAMP(k_) if (dbg && threadIdx.x == 0) dbg[200 + (k_)] = clock64();
        PYIPM_TS_STAMP(0)
        const int64_t i = c0 + (int64_t)t * TB + wave * 16 + l15;    // this lane's (global) row: the same map in both halves
        if (nref > 0 && Tflag[tp] == 0.0) nref = 0;                  // (uniform over the block)
        double4_t acc[4];
        double sb[16], c2h[16];
        if (half == 0) {
            // S first, inv(T) behind it: waiting for the tile that goes to shared memory then covers both round trips
            #pragma unroll
            for (int ks = 0; ks < 16; ++ks) sb[ks] = W[i + (int64_t)(tp * TB + ks * 4 + l4) * ldw];
            asm volatile("" ::: "memory");
            PYIPM_STAGE_TILE(sm.X, 1.0, Dinv + tp * TT)
            #pragma unroll
            for (int ks = 0; ks < 16; ++ks) sb[ks] = -sb[ks];
        } else {
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)                           // the diagonal tile as it stands, in the C/D map of the update
                #pragma unroll
                for (int r = 0; r < 4; ++r) c2h[4 * tt + r] = A[i + (lc0 + t * TB + tt * 16 + l4 + 4 * r) * ld];
        }
        __syncthreads();                                             // inv(T) is staged (nobody waits for the other loads here)
        PYIPM_TS_STAMP(1)
        if (half == 0) {
            // the scaling product; the -S' operand of the diagonal update goes to its buffer in the shadow of the MFMAs
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[tt] = (double4_t){0.0, 0.0, 0.0, 0.0};
            #pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                #pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const double a = sm.X[tt * 16 + l15][ks * 4 + l4];
                    acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sb[ks], acc[tt], 0, 0, 0);
                }
            }
            #pragma unroll
            for (int ks = 0; ks < 16; ++ks) sm.Wn[ks * 4 + l4][wave * 16 + l15] = -sb[ks];       // Wn[c][k] = -S[c][k], stored [k][c]
        } else {
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                #pragma unroll
                for (int r = 0; r < 4; ++r) sm.ts.stage[wave * 16 + l15][tt * 16 + l4 + 4 * r] = c2h[4 * tt + r];
        }
        for (int it = 0; it < nref; ++it) {                          // refinement of the block solve: four barriers a step,
            if (half == 0) strip_refine_step(sm.X, Dinv + tp * TT, Tsv + tp * TT, sb, tid, l15, l4, acc);   // the helpers keep count
            else { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
        }
        __syncthreads();                                             // the diagonal tile and -S' are in shared memory
        if (half == 0) {
            if (dbg) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[3][3])); PYIPM_TS_STAMP(2) }
            double4_t c2[4];
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                #pragma unroll
                for (int r = 0; r < 4; ++r) c2[tt][r] = sm.ts.stage[wave * 16 + l15][tt * 16 + l4 + 4 * r];
            #pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const double lop = acc[ks >> 2][ks & 3];
                #pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    c2[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(sm.Wn[ks * 4 + l4][16 * tt + l15], lop, c2[tt], 0, 0, 0);
            }
            double gmax = 0.0;                                       // L goes out while the matrix pipe works on the update
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    A[i + (lc0 + tp * TB + tt * 16 + l4 + 4 * r) * ld] = acc[tt][r];
                    gmax = fmax(gmax, fabs(acc[tt][r]));
                }
            gmax = wave_max(gmax);
            if (lane == 0) atomicMax(&st->growth_bits, (unsigned long long)__double_as_longlong(gmax));
            PYIPM_TS_STAMP(3)
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = tt * 16 + l4 + 4 * r;
                    A[i + (lc0 + t * TB + c) * ld] = c2[tt][r];
                    sm.ts.stage[wave * 16 + l15][c] = c2[tt][r];
                }
        }
        PYIPM_TS_STAMP(4)
#undef PYIPM_TS_STAMP
        tile_invert_dev<true>(sm.ts, A, ld, c0 + (int64_t)t * TB, lc0 + (int64_t)t * TB, Dinv + t * TT, Tsv + t * TT, Tflag + t, refine_cond,
                              st, Nreal, pivtol_rel, anorm_bits, neg_from, dbg, /*from_stage=*/true, blocked != 0, &sm.ex);
        return;
    }
    // ------------------------------------------------ the other row tiles -----------------------------------------------
    const int unit = upb * ((int)blockIdx.x - 1) + half;
    const int b = 1 + unit / ny, y = unit % ny;
    const bool live = b < nt - t && half < upb;                      // (uniform per half)
    if (t == 0) {
        if (live && y == 0) {
            double tmp[TB * TB / 256];
            const int64_t r0 = c0 + (int64_t)b * TB;
            #pragma unroll
            for (int q = 0; q < TB * TB / 256; ++q) {
                const int e = tid + 256 * q;
                tmp[q] = A[(r0 + (e & 63)) + (lc0 + (e >> 6)) * ld];
            }
            #pragma unroll
            for (int q = 0; q < TB * TB / 256; ++q) {
                const int e = tid + 256 * q;
                W[(r0 + (e & 63)) + (int64_t)(e >> 6) * ldw] = -tmp[q];
            }
        }
        return;
    }
    const int tp = t - 1, it = t + b;
    if (!live || y > it - t) return;                                 // (a half that leaves is not waited for)
    double (&X)[TB][TB + 2] = half ? sm.Wn : sm.X;
    const int64_t i = c0 + (int64_t)it * TB + wave * 16 + l15;
    if (nref > 0 && Tflag[tp] == 0.0) nref = 0;
    PYIPM_STAGE_TILE(X, 1.0, Dinv + tp * TT)
    double sb[16];
    #pragma unroll
    for (int ks = 0; ks < 16; ++ks) sb[ks] = -W[i + (int64_t)(tp * TB + ks * 4 + l4) * ldw];
    __syncthreads();
    double4_t acc[4];
    strip_scale(X, Dinv + tp * TT, Tsv + tp * TT, nref, sb, tid, l15, l4, acc);
    if (y == 0) {
        double gmax = 0.0;
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                A[i + (lc0 + tp * TB + tt * 16 + l4 + 4 * r) * ld] = acc[tt][r];
                gmax = fmax(gmax, fabs(acc[tt][r]));
            }
        gmax = wave_max(gmax);
        if (lane == 0) atomicMax(&st->growth_bits, (unsigned long long)__double_as_longlong(gmax));
    }
    for (int v = t + y; v <= it; v += ny) {
        double4_t c2[4];
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            #pragma unroll
            for (int r = 0; r < 4; ++r) c2[tt][r] = A[i + (lc0 + v * TB + tt * 16 + l4 + 4 * r) * ld];
        strip_update(c2, acc, W + (c0 + (int64_t)v * TB) + (int64_t)(tp * TB) * ldw, ldw, l15, l4);
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = tt * 16 + l4 + 4 * r;
                A[i + (lc0 + v * TB + c) * ld] = c2[tt][r];
                if (v == t) W[i + (int64_t)(t * TB + c) * ldw] = -c2[tt][r];
            }
    }
}

// (Round 3 built the tile steps of a sub-panel as ONE launch -- k_tile_chain, in the history at commit 435e38f: a
// critical workgroup inverting tile after tile, one static owner workgroup per row tile applying the stages, release /
// acquire fences and flags in between, the same device functions in the same order: the SAME BITS as this schedule on
// every shape tried.  It was slower, 27 us a tile against 22.5 and 3.6 against 2.8 ms at N = 6144, for three measured
// reasons (tools/chain_clock.py at the time): every acquire is a buffer_inv that empties the XCD's L2 for all sixteen
// workgroups on it (the stage in front of an inversion took 8.7 us instead of 6.7); inlined into a loop, the inversion's
// ~100 lane-constant registers were hoisted and 154 VGPRs spilled, behind a real call it ran 16.5 instead of 12.5 us; and
// a far row tile has up to fifteen column tiles to update per stage, alone in its workgroup, so every launch ended
// 35-130 us after its last inversion.  What it would take: operands read with per-access coherent loads instead of
// fences, the far rows split over several workgroups as k_tile_step's grid.y does, the inversion's constants recomputed
// per call.  Removed)

// The rows below the diagonal block, 64 per block: all nt stages of the strip in one launch, right-looking.  The column
// tile a stage scales is the last one the stage before it updated and stays in registers in between; the other column
// tiles of the strip's 64 x 256 slab go through L2 (a whole slab in registers -- 128 of them -- plus the operands of the
// refinement needs more than the 256 registers a strip can have if it is to fit beside a resident block of the bulk
// update; the launch would then wait for whole CUs to drain).
__global__ __launch_bounds__(256, 2) void k_panel_rest(
    double* __restrict__ A, int64_t ld, int64_t c0, int64_t lc0, int nt, int64_t row_begin,
    double* __restrict__ W, int64_t ldw,
    const double* __restrict__ Dinv, const double* __restrict__ Tsv, const double* __restrict__ Tflag, int nref,
    int64_t hole0, int64_t hole1, unsigned long long* __restrict__ growth_bits)
{
    __shared__ double X[TB][TB + 2];
    const int64_t r0 = row_begin + (int64_t)blockIdx.x * TB;
    if (r0 >= hole0 && r0 + TB <= hole1) return;          // rows identically zero (KKT structure): L = 0 already, W never read
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t TT = (int64_t)TB * TB;
    const int64_t i = r0 + wave * 16 + l15;
    double sb[16];                                         // S of the stage's column tile, B-operand map
    #pragma unroll
    for (int ks = 0; ks < 16; ++ks) sb[ks] = A[i + (lc0 + ks * 4 + l4) * ld];
    double gmax = 0.0;
    for (int t = 0; t < nt; ++t) {
        if (t > 0) __syncthreads();                        // everyone is done with the previous inv(T)
        PYIPM_STAGE_TILE(X, 1.0, Dinv + t * TT)
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) W[i + (int64_t)(t * TB + ks * 4 + l4) * ldw] = -sb[ks];
        __syncthreads();
        const int nr = (nref > 0 && Tflag[t] != 0.0) ? nref : 0;
        double4_t acc[4];
        strip_scale(X, Dinv + t * TT, Tsv + t * TT, nr, sb, tid, l15, l4, acc);
        #pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            #pragma unroll
            for (int r = 0; r < 4; ++r) {
                A[i + (lc0 + t * TB + tt * 16 + l4 + 4 * r) * ld] = acc[tt][r];
                gmax = fmax(gmax, fabs(acc[tt][r]));
            }
        for (int v = nt - 1; v > t; --v) {                 // (independent entries: any order of the column tiles gives the same bits)
            double4_t c2[4];
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                #pragma unroll
                for (int r = 0; r < 4; ++r) c2[tt][r] = A[i + (lc0 + v * TB + tt * 16 + l4 + 4 * r) * ld];
            strip_update<4>(c2, acc, W + (c0 + (int64_t)v * TB) + (int64_t)(t * TB) * ldw, ldw, l15, l4);
            if (v == t + 1) {                              // the next stage's tile: C/D map == B-operand map, stays in registers
                #pragma unroll
                for (int ks = 0; ks < 16; ++ks) sb[ks] = c2[ks >> 2][ks & 3];
            } else {
                #pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    #pragma unroll
                    for (int r = 0; r < 4; ++r) A[i + (lc0 + v * TB + tt * 16 + l4 + 4 * r) * ld] = c2[tt][r];
            }
        }
    }
    gmax = wave_max(gmax);
    if (lane == 0 && growth_bits) atomicMax(growth_bits, (unsigned long long)__double_as_longlong(gmax));
}

}  // namespace pyipm
