// tile_blocked.hpp -- the 64x64 tile inversion in blocks of 16 pivots (round 3).
//
// tile_invert_dev (kernels_factor.hpp) inverts a diagonal tile by 64 symmetric sweeps, each one a rank-1 update of the
// whole tile behind a dependent chain of ~1000 cycles (relative-addressed register read, two v_readlane, an LDS write ->
// read round trip of the pivot row, the reciprocal, an exec-mask switch for the pivot lane: DESIGN.md section 4, lessons
// 2 and 6 of round 2).  N sequential pivots of that length are the floor under everything that is not bulk work.
//
// Here the same sweep operator is applied 16 pivots at a time -- a BLOCK sweep with the 16x16 diagonal micro-block P:
//     B[R,R] -= X B[K,R],   B[R,K] <- X = B[R,K] inv(P),   B[K,K] <- -inv(P)          (R = the other 48 rows)
// after which the working matrix is what 16 single sweeps leave (the unswept part is the Schur complement, the swept
// part -inv of what was swept), so the four block sweeps end at -inv(T) like the 64 single ones.
//   * P is factored P = L D L' entirely in registers, one lane per row, the pivot row broadcast by DP-ALU DPP
//     (v_fmac_f64_dpp row_newbcast: no LDS round trip, no v_readlane, no polling) and M = inv(L) formed by the same
//     instructions on the augmented columns (microblock_asm.inc, tools/gen/gen_microblock.py).  Every wave does this
//     redundantly, so no barrier separates it from what follows.
//   * everything else is fp64 MFMA (v_mfma_f64_16x16x4_f64), one 16-row tile per wave:
//         U = W M'   (the columns as plain elimination would have updated them),   Lr = U inv(D),   X = Lr M,
//         C(t,t') -= X_t W_t''   for the lower block triangle,    -inv(P) = -M' inv(D) M.
// Pivoting.  The fast path takes the diagonal pivots in their natural order and checks AFTERWARDS what Bunch-Kaufman
// would have checked before each pivot: |l_ij| <= 1/alpha for every multiplier of a not yet eliminated row of the TILE
// (in the micro-block: read off the elimination; in the rows below: Lr above), and |d_j| above the pivot's own rejection
// threshold.  If any test fails -- or anything is not finite -- the block is NOT committed (the working matrix in shared
// memory is only written after all four waves agree) and tile_invert_dev's general loop takes over from exactly that
// state: 2x2 pivots, off-diagonal choices, static pivots all live there, unchanged.  Definite and quasi-definite tiles
// (every tile of the benchmark family) never leave the fast path.
#pragma once
#include "ctx.hpp"
#include "microblock_asm.inc"

namespace pyipm {

typedef double double4_tb __attribute__((ext_vector_type(4)));

struct BlockedScratch {
    double Ms[16][17];          // a[] of the 16 lanes after the elimination (row i: M_i[c] for c < i; rest unused)
    double Ds[16];              // the 16 pivots
    double ptol[TB];            // rejection threshold of each pivot (pivtol_rel x the original maximum of its column)
    int fail;                   // a wave found a multiplier beyond the Bunch-Kaufman bound in its rows
};

#define PYIPM_BK_INV_ALPHA 1.5615528128088303   /* 8 / (1 + sqrt(17)) */

// The thread index through an opaque move.  Inside the loop of a persistent kernel (k_tile_chain, kernels_chain.hpp) the
// ~100 lane constants of a tile inversion would otherwise be hoisted out of the loop and spill (round 3's k_tile_chain: 154
// spilled VGPRs); behind this nothing derived from the index is loop-invariant, so each tile recomputes them as each launch
// of k_tile_step does.
template <bool LOOPED>
__device__ __forceinline__ int tile_tid() {
    int t = (int)threadIdx.x;
    if constexpr (LOOPED) asm volatile("" : "+v"(t));
    return t;
}

// v_rcp_f64 and two Newton steps, exactly as microblock_asm.inc computes 1/d (and as pivot_recip in kernels_factor.hpp)
__device__ __forceinline__ double blocked_recip(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}

// Per-lane constants of a tile's blocked sweep, computed once (every shared-memory address of the four blocks is one of
// these bases plus an immediate; a single wave issues one instruction per ~5 cycles whatever it is, so the ~200
// instructions of address arithmetic and operand masks per block were a quarter of its time).
struct BlockedLane {
    int aoff[16];               // bytes: entry (i15, c) of a diagonal micro-block relative to its corner (lower-valid storage)
    int oRow, oCol;             // bytes: &stage[i15][q], &stage[q][i15]
    int oRowT, oColT;           // ... of the wave's own row tile: &stage[16 t + i15][q], &stage[q][16 t + i15]
    double mlt[4], mgt[4], meq[4];   // 1.0 / 0.0: (4 s + q < i15), (4 s + q > i15), (4 s + q == i15)
};

// One block sweep, micro-block KB (compile time).  Returns false (uniform) when the block was NOT committed.
template <int KB, int STRIDE, bool LOOPED = false>
__device__ __forceinline__ bool tile_blocked_block(double (&stage)[TB][STRIDE], BlockedScratch& bs, double* __restrict__ dsave,
                                                   const BlockedLane& L, unsigned long long* __restrict__ dbg)
{
#define PYIPM_TB_STAMP(ph_) if (dbg && lane == 0) dbg[8 + 32 * KB + 8 * wave + (ph_)] = clock64();
#define PYIPM_TB_LD(off_) (*reinterpret_cast<const double*>(sb + (off_)))
#define PYIPM_TB_ST(off_) (*reinterpret_cast<double*>(sb + (off_)))
    constexpr int k0 = 16 * KB;
    constexpr int D8 = (int)sizeof(double), S8 = STRIDE * D8;
    const int tid = tile_tid<LOOPED>(), lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i15 = lane & 15, q = lane >> 4;
    const int t = wave;                                // this wave's 16-row tile
    char* sb = reinterpret_cast<char*>(&stage[0][0]);
    // ---- operands: everything the block needs from the working matrix is requested BEFORE the elimination (none of it
    //      depends on it), so that the only LDS round trip left behind the dependent chain is M's transpose.  Loads are
    //      unconditional (a tile the wave does not use costs one instruction to load and one to define otherwise) ----
    double a[16];                                      // the micro-block, redundantly in every wave and every 16-lane row
    #pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = PYIPM_TB_LD(L.aoff[c] + k0 * (S8 + D8));
    double bW[4][4];                                   // bW[t'][s] = W[16 t' + i15][k0 + 4 s + q]  (W = B[., K], lower-valid storage)
    double4_tb Cn[4];                                  // tiles (t, t'), t' <= t: C[i = 16 t + i15][j = 16 t' + q + 4 rr]
    #pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        #pragma unroll
        for (int s = 0; s < 4; ++s)
            bW[tp][s] = tp > KB ? PYIPM_TB_LD(L.oRow + 16 * tp * S8 + (k0 + 4 * s) * D8)
                                : PYIPM_TB_LD(L.oCol + (k0 + 4 * s) * S8 + 16 * tp * D8);          // (t' == KB: unused)
        #pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cn[tp][rr] = PYIPM_TB_LD(L.oRowT + (16 * tp + 4 * rr) * D8);  // (above the diagonal: junk)
    }
    double bWt[4];                                     // the wave's own rows of W (B operand of U)
    #pragma unroll
    for (int s = 0; s < 4; ++s)
        bWt[s] = t >= KB ? PYIPM_TB_LD(L.oRowT + (k0 + 4 * s) * D8) : PYIPM_TB_LD(L.oColT + (k0 + 4 * s) * S8);
    double pt4[4];
    #pragma unroll
    for (int s = 0; s < 4; ++s) pt4[s] = bs.ptol[k0 + 4 * s + q];
    double d[16], r, lmax = 0.0, ctr = (double)i15, t_, u_, mk_;
    if (dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PYIPM_TB_STAMP(0) }
    // (the statements clobber "memory": the loads above stay above, the stores below stay below)
    PYIPM_MICROBLOCK_ASM_0(a, d, r, lmax, ctr, t_, u_, mk_);
    PYIPM_MICROBLOCK_ASM_1(a, d, r, lmax, ctr, t_, u_, mk_);
    PYIPM_TB_STAMP(1)
    (void)r;
    // M' and the pivots go through shared memory (lane j's register i15 is what lane i15 needs); every wave writes the
    // same values (same instructions on the same data), so a wave only waits for its own stores
    if (q == 0) {
        #pragma unroll
        for (int c = 0; c < 16; ++c) bs.Ms[i15][c] = a[c];
        if (i15 == 0) {
            #pragma unroll
            for (int c = 0; c < 16; ++c) bs.Ds[c] = d[c];                         // (uniform over the 16 lanes)
        }
    }
    // branch-free from here to the vote: flags are accumulated with bitwise operators (|| made hipcc emit ~60
    // exec-mask branches)
    int bad = (int)!(lmax <= PYIPM_BK_INV_ALPHA);                               // multipliers inside the micro-block
    double rsel[4], dsel[4], aM[4], aMT[4];            // 1 / d[4 s + q], d[4 s + q], M[i15][4 s + q], M[4 s + q][i15]
    #pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 4 * s + q;
        const double v = bs.Ms[i15][c], w = bs.Ms[c][i15];
        dsel[s] = bs.Ds[c];
        aM[s] = fma(v, L.mlt[s], L.meq[s]);            // unit lower triangular: what the lanes hold above the diagonal is
        aMT[s] = fma(w, L.mgt[s], L.meq[s]);           // finite junk of the elimination, times zero
    }
    // (each product: two independent accumulators of two k-steps -- a chain of four dependent MFMAs costs ~85 cycles a link)
    double4_tb U = {0.0, 0.0, 0.0, 0.0}, U2 = {0.0, 0.0, 0.0, 0.0};
    U  = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[0], bWt[0], U, 0, 0, 0);
    U2 = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[2], bWt[2], U2, 0, 0, 0);
    U  = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[1], bWt[1], U, 0, 0, 0);
    U2 = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[3], bWt[3], U2, 0, 0, 0);
    #pragma unroll
    for (int s = 0; s < 4; ++s) {
        rsel[s] = blocked_recip(dsel[s]);              // the elimination's own sequence: the same bits as its r
        bad |= (int)!(fabs(dsel[s]) > pt4[s]) | (int)!(fabs(dsel[s]) <= 1.0e300);          // the 16 pivots, four per 16-lane row
    }
    if (dbg) { asm volatile("" :: "v"(U[0])); PYIPM_TB_STAMP(5) }
    // One instruction stream for all four waves.  The wave that holds the micro-block's own rows (t == KB) puts
    // -inv(P) = -M' D^-1 M where the others put X = Lr M: same MFMA slots, operands chosen by a uniform select.  Its tile
    // updates then work on junk and are not stored; tiles above the diagonal (t' > t) are updated and stored as junk by
    // everyone -- nobody reads above the diagonal -- so that the slowest wave's twelve update MFMAs are simply everybody's
    // (a uniform branch per tile cost a register copy per accumulator and bought no time: the barrier waits for the
    // slowest wave anyway).
    const bool isK = t == KB;
    double Lr[4];
    const double lim = t > KB ? PYIPM_BK_INV_ALPHA : 1.0e300;                    // rows not yet eliminated: the Bunch-Kaufman bound
    #pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        Lr[rr] = (U[rr] + U2[rr]) * rsel[rr];
        bad |= (int)!isK & (int)!(fabs(Lr[rr]) <= lim);
    }
    double xa[4], xb[4];
    #pragma unroll
    for (int s = 0; s < 4; ++s) {
        xa[s] = isK ? aMT[s] * rsel[s] : aMT[s];
        xb[s] = isK ? -aMT[s] : Lr[s];
    }
    double4_tb X = {0.0, 0.0, 0.0, 0.0}, X2 = {0.0, 0.0, 0.0, 0.0};   // t != KB: X[i = i15][c = q + 4 r];  t == KB: -inv(P)[b = i15][a = q + 4 r]
    X  = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[0], xb[0], X, 0, 0, 0);
    X2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2], xb[2], X2, 0, 0, 0);
    X  = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[1], xb[1], X, 0, 0, 0);
    X2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[3], xb[3], X2, 0, 0, 0);
    #pragma unroll
    for (int rr = 0; rr < 4; ++rr) { X[rr] += X2[rr]; bad |= (int)!(fabs(X[rr]) <= 1.0e300); }
    // the tiles of the row, k-step by k-step: consecutive MFMAs belong to different tiles (independent accumulators)
    #pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double xn = -X[s];
        #pragma unroll
        for (int tp = 0; tp < 4; ++tp)
            if (tp != KB) Cn[tp] = __builtin_amdgcn_mfma_f64_16x16x4f64(bW[tp][s], xn, Cn[tp], 0, 0, 0);
    }
    if (dbg) { asm volatile("" :: "v"(X[0]), "v"(Cn[0][0]), "v"(Cn[3][3])); PYIPM_TB_STAMP(2) }
    if (__ballot(bad != 0) != 0ull && lane == 0) bs.fail = 1;
    __syncthreads();                                   // everyone has read W; the verdict is in
    PYIPM_TB_STAMP(6)
    if (bs.fail) return false;                         // (uniform) nothing of this block has been written
    // ---- commit (only the lower triangle of the working matrix is ever read: what lands above the diagonal is junk
    //      nobody looks at, so no store needs a per-lane predicate) ----
    {
        // X: stage[16 t + i15][k0 + q + 4 rr] for t >= KB (t == KB: the diagonal block), stage[k0 + q + 4 rr][16 t + i15] below
        const int xo = t >= KB ? L.oRowT + k0 * D8 : L.oColT + k0 * S8;
        #pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            if (t >= KB) PYIPM_TB_ST(xo + 4 * rr * D8) = X[rr];
            else         PYIPM_TB_ST(xo + 4 * rr * S8) = X[rr];
        }
    }
    if (!isK) {                                        // (the rows of the micro-block receive the other waves' X', nothing else)
        #pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            if (tp == KB) continue;
            #pragma unroll
            for (int rr = 0; rr < 4; ++rr) PYIPM_TB_ST(L.oRowT + (16 * tp + 4 * rr) * D8) = Cn[tp][rr];
        }
    } else if (i15 < 4) {
        const long long n0 = i15 == 0 ? -1ll : 0ll, n1 = i15 == 1 ? -1ll : 0ll, n2 = i15 == 2 ? -1ll : 0ll, n3 = i15 == 3 ? -1ll : 0ll;
        dsave[k0 + 4 * i15 + q] = __longlong_as_double((n0 & __double_as_longlong(dsel[0])) | (n1 & __double_as_longlong(dsel[1])) |
                                                       (n2 & __double_as_longlong(dsel[2])) | (n3 & __double_as_longlong(dsel[3])));
    }
    __syncthreads();
    PYIPM_TB_STAMP(3)
#undef PYIPM_TB_STAMP
#undef PYIPM_TB_LD
#undef PYIPM_TB_ST
    return true;
}

// stage: the working matrix, lower triangle valid (entries [i][j], i >= j).  256 threads; everyone has passed a barrier
// after the last write to stage / bs.ptol / bs.fail = 0.  Returns the number of micro-blocks swept (0..4); on return
// everyone has passed a barrier after the last write.  dsave[p] = pivot p as used.
template <int STRIDE, bool LOOPED = false>
__device__ __forceinline__ int tile_blocked_sweep(double (&stage)[TB][STRIDE], BlockedScratch& bs, double* __restrict__ dsave,
                                                  unsigned long long* __restrict__ dbg = nullptr)    // diagnostics: dbg[8 + 32 kb + 8 wave + phase] = clock
{
    const int tid_ = tile_tid<LOOPED>();
    const int lane = tid_ & 63, i15 = lane & 15, q = lane >> 4;
    const int t = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    constexpr int D8 = (int)sizeof(double), S8 = STRIDE * D8;
    BlockedLane L;
    #pragma unroll
    for (int c = 0; c < 16; ++c) L.aoff[c] = c <= i15 ? i15 * S8 + c * D8 : c * S8 + i15 * D8;
    L.oRow = i15 * S8 + q * D8;  L.oCol = q * S8 + i15 * D8;
    L.oRowT = L.oRow + 16 * t * S8;  L.oColT = L.oCol + 16 * t * D8;
    #pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 4 * s + q;
        L.mlt[s] = c < i15 ? 1.0 : 0.0;  L.mgt[s] = c > i15 ? 1.0 : 0.0;  L.meq[s] = c == i15 ? 1.0 : 0.0;
    }
    if (!tile_blocked_block<0, STRIDE, LOOPED>(stage, bs, dsave, L, dbg)) return 0;
    if (!tile_blocked_block<1, STRIDE, LOOPED>(stage, bs, dsave, L, dbg)) return 1;
    if (!tile_blocked_block<2, STRIDE, LOOPED>(stage, bs, dsave, L, dbg)) return 2;
    if (!tile_blocked_block<3, STRIDE, LOOPED>(stage, bs, dsave, L, dbg)) return 3;
    return 4;
}

}  // namespace pyipm
