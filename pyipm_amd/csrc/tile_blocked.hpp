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

// v_rcp_f64 and two Newton steps, exactly as microblock_asm.inc computes 1/d (and as pivot_recip in kernels_factor.hpp)
__device__ __forceinline__ double blocked_recip(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}

// One block sweep, micro-block KB (compile time: every shared-memory address is then a per-lane base -- computed once
// per tile -- plus an immediate; with the block index in a register the address arithmetic alone was ~200 instructions
// per block, and a single wave issues one instruction per ~5 cycles whatever it is).  Returns false (uniform) when the
// block was NOT committed.
template <int KB, int STRIDE>
__device__ __forceinline__ bool tile_blocked_block(double (&stage)[TB][STRIDE], BlockedScratch& bs, double* __restrict__ dsave,
                                                   const int (&aoff)[16], unsigned long long* __restrict__ dbg)
{
#define PYIPM_TB_STAMP(ph_) if (dbg && lane == 0) dbg[8 + 32 * KB + 8 * wave + (ph_)] = clock64();
    constexpr int k0 = 16 * KB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i15 = lane & 15, q = lane >> 4;
    const int t = wave;                                // this wave's 16-row tile
    const char* sbase = reinterpret_cast<const char*>(&stage[0][0]) + (size_t)k0 * (STRIDE + 1) * sizeof(double);
    // ---- operands: everything the block needs from the working matrix is requested BEFORE the elimination (none of it
    //      depends on it), so that the only LDS round trip left behind the dependent chain is M's transpose ----
    double a[16];                                      // the micro-block: lanes 0..15 of every wave (the other three 16-lane
    #pragma unroll                                     // rows run the elimination on zeros: nothing of theirs is used)
    for (int c = 0; c < 16; ++c) a[c] = 0.0;
    if (q == 0) {
        #pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = *reinterpret_cast<const double*>(sbase + aoff[c]);
    }
    double bW[4][4];                                   // bW[t'][s] = W[16 t' + i15][k0 + 4 s + q]  (W = B[., K], lower-valid storage)
    double4_tb Cn[4];                                  // tiles (t, t'), t' <= t: C[i = 16 t + i15][j = 16 t' + q + 4 rr]
    #pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        #pragma unroll
        for (int s = 0; s < 4; ++s) bW[tp][s] = 0.0;
        Cn[tp] = (double4_tb){0.0, 0.0, 0.0, 0.0};
        if (tp == KB) continue;
        if (tp <= t) {                                 // (uniform)
            #pragma unroll
            for (int s = 0; s < 4; ++s)
                bW[tp][s] = tp > KB ? stage[16 * tp + i15][k0 + 4 * s + q] : stage[k0 + 4 * s + q][16 * tp + i15];
            #pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                Cn[tp][rr] = stage[16 * t + i15][16 * tp + q + 4 * rr];            // (above the diagonal: junk, never read back)
        }
    }
    double bWt[4];                                     // the wave's own rows of W (B operand of U); for t == KB the micro-block: U D^-1 = L
    #pragma unroll
    for (int s = 0; s < 4; ++s) bWt[s] = t >= KB ? stage[16 * t + i15][k0 + 4 * s + q] : stage[k0 + 4 * s + q][16 * t + i15];
    double pt4[4];
    #pragma unroll
    for (int s = 0; s < 4; ++s) pt4[s] = bs.ptol[k0 + 4 * s + q];
    double d[16], r[16], lmax = 0.0, ctr = (double)i15, t_, u_, mk_;
    if (dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PYIPM_TB_STAMP(0) }
    // (the statements clobber "memory": the loads above stay above, the stores below stay below)
    PYIPM_MICROBLOCK_ASM_0(a, d, r, lmax, ctr, t_, u_, mk_);
    PYIPM_MICROBLOCK_ASM_1(a, d, r, lmax, ctr, t_, u_, mk_);
    PYIPM_MICROBLOCK_ASM_2(a, d, r, lmax, ctr, t_, u_, mk_);
    PYIPM_MICROBLOCK_ASM_3(a, d, r, lmax, ctr, t_, u_, mk_);
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
    int bad = (int)(q == 0) & (int)!(lmax <= PYIPM_BK_INV_ALPHA);               // multipliers inside the micro-block
    double rsel[4], dsel[4], aM[4], aMT[4];            // 1 / d[4 s + q], d[4 s + q], M[i15][4 s + q], M[4 s + q][i15]
    #pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 4 * s + q;
        const double v = bs.Ms[i15][c], w = bs.Ms[c][i15];
        dsel[s] = bs.Ds[c];
        aM[s] = c < i15 ? v : (c == i15 ? 1.0 : 0.0);
        aMT[s] = i15 < c ? w : (c == i15 ? 1.0 : 0.0);
    }
    double4_tb U = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll
    for (int s = 0; s < 4; ++s) U = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[s], bWt[s], U, 0, 0, 0);
    #pragma unroll
    for (int s = 0; s < 4; ++s) {
        rsel[s] = blocked_recip(dsel[s]);              // the elimination's own sequence: the same bits as its r
        bad |= (int)!(fabs(dsel[s]) > pt4[s]) | (int)!(fabs(dsel[s]) <= 1.0e300);          // the 16 pivots, four per 16-lane row
    }
    if (dbg) { asm volatile("" :: "v"(U[0])); PYIPM_TB_STAMP(5) }
    double4_tb X = {0.0, 0.0, 0.0, 0.0};               // t != KB: X[i = i15][c = q + 4 r];  t == KB: -inv(P)[b = i15][a = q + 4 r]
    if (t != KB) {
        double Lr[4];
        const double lim = t > KB ? PYIPM_BK_INV_ALPHA : 1.0e300;                // rows not yet eliminated: the Bunch-Kaufman bound
        #pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            Lr[rr] = U[rr] * rsel[rr];
            bad |= (int)!(fabs(Lr[rr]) <= lim);
        }
        #pragma unroll
        for (int s = 0; s < 4; ++s) X = __builtin_amdgcn_mfma_f64_16x16x4f64(aMT[s], Lr[s], X, 0, 0, 0);
        // the tiles of the row, k-step by k-step: consecutive MFMAs belong to different tiles (independent accumulators; a
        // tile at a time is a chain of four dependent ones at ~85 cycles each)
        #pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double xn = -X[s];
            #pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                if (tp == KB) continue;
                if (tp <= t)                           // (uniform)
                    Cn[tp] = __builtin_amdgcn_mfma_f64_16x16x4f64(bW[tp][s], xn, Cn[tp], 0, 0, 0);
            }
        }
    } else {
        // the micro-block's own rows: U D^-1 = L, the multipliers of the elimination (unit lower triangular; above the
        // diagonal: junk) -- nothing to check here that lmax has not seen; -inv(P) = -M' D^-1 M
        #pragma unroll
        for (int s = 0; s < 4; ++s) X = __builtin_amdgcn_mfma_f64_16x16x4f64(aMT[s] * rsel[s], -aMT[s], X, 0, 0, 0);
    }
    #pragma unroll
    for (int rr = 0; rr < 4; ++rr) bad |= (int)!(fabs(X[rr]) <= 1.0e300);
    if (dbg) { asm volatile("" :: "v"(X[0]), "v"(Cn[0][0]), "v"(Cn[3][3])); PYIPM_TB_STAMP(2) }
    if (__ballot(bad != 0) != 0ull && lane == 0) bs.fail = 1;
    __syncthreads();                                   // everyone has read W; the verdict is in
    PYIPM_TB_STAMP(6)
    if (bs.fail) return false;                         // (uniform) nothing of this block has been written
    // ---- commit (only the lower triangle of the working matrix is ever read: what lands above the diagonal of a
    //      diagonal tile is junk nobody looks at, so no store needs a per-lane predicate) ----
    if (t != KB) {
        #pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int c = q + 4 * rr;
            if (t > KB) stage[16 * t + i15][k0 + c] = X[rr];
            else        stage[k0 + c][16 * t + i15] = X[rr];
        }
        #pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            if (tp == KB) continue;
            if (tp <= t) {
                #pragma unroll
                for (int rr = 0; rr < 4; ++rr) stage[16 * t + i15][16 * tp + q + 4 * rr] = Cn[tp][rr];
            }
        }
    } else {
        #pragma unroll
        for (int rr = 0; rr < 4; ++rr) stage[k0 + i15][k0 + q + 4 * rr] = X[rr];
        if (i15 < 4) {
            const long long n0 = i15 == 0 ? -1ll : 0ll, n1 = i15 == 1 ? -1ll : 0ll, n2 = i15 == 2 ? -1ll : 0ll, n3 = i15 == 3 ? -1ll : 0ll;
            dsave[k0 + 4 * i15 + q] = __longlong_as_double((n0 & __double_as_longlong(dsel[0])) | (n1 & __double_as_longlong(dsel[1])) |
                                                           (n2 & __double_as_longlong(dsel[2])) | (n3 & __double_as_longlong(dsel[3])));
        }
    }
    __syncthreads();
    PYIPM_TB_STAMP(3)
#undef PYIPM_TB_STAMP
    return true;
}

// stage: the working matrix, lower triangle valid (entries [i][j], i >= j).  256 threads; everyone has passed a barrier
// after the last write to stage / bs.ptol / bs.fail = 0.  Returns the number of micro-blocks swept (0..4); on return
// everyone has passed a barrier after the last write.  dsave[p] = pivot p as used.
template <int STRIDE>
__device__ __forceinline__ int tile_blocked_sweep(double (&stage)[TB][STRIDE], BlockedScratch& bs, double* __restrict__ dsave,
                                                  unsigned long long* __restrict__ dbg = nullptr)    // diagnostics: dbg[8 + 32 kb + 8 wave + phase] = clock
{
    const int i15 = threadIdx.x & 15;
    int aoff[16];                                      // byte offset of entry (i15, c) of a diagonal micro-block relative to its corner
    #pragma unroll
    for (int c = 0; c < 16; ++c) aoff[c] = (int)sizeof(double) * (c <= i15 ? i15 * STRIDE + c : c * STRIDE + i15);
    if (!tile_blocked_block<0>(stage, bs, dsave, aoff, dbg)) return 0;
    if (!tile_blocked_block<1>(stage, bs, dsave, aoff, dbg)) return 1;
    if (!tile_blocked_block<2>(stage, bs, dsave, aoff, dbg)) return 2;
    if (!tile_blocked_block<3>(stage, bs, dsave, aoff, dbg)) return 3;
    return 4;
}

}  // namespace pyipm
