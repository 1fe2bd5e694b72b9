// tile_blocked8.hpp -- the blocked 64x64 tile inversion of tile_blocked.hpp on EIGHT waves (round 5): four that carry
// the chain of dependent eliminations and four that do everything nothing is waiting for.
//
// A block sweep of tile_blocked.hpp is, per wave, [operands from shared memory] -> elimination of the 16x16 micro-block
// (2200 cycles of dependent DP-ALU instructions, the matrix pipe idle) -> U, X (8 MFMAs) -> 12 update MFMAs -> barrier ->
// commit -> barrier -> next block's operands: 7150 cycles, of which the NEXT elimination only needs 4 of the 12 update
// MFMAs (the next diagonal micro-block) and none of the commit / reload traffic.  Here:
//   * critical waves 0..3 (row tile t = wave): elimination, U, X; the wave that owns the next micro-block's rows updates that
//     ONE 16x16 tile and hands it to the others through `Pnext`; everybody's X goes to `Xbuf`; vote; barrier B1; the next
//     elimination starts from `Pnext` at once.
//   * helper waves 4..7 (row tile t = wave - 4): operands of the 12 updates between B2 and B1 (the working matrix is
//     stable there), after B1 the 12 update MFMAs with X from `Xbuf` and the commit of X and of the updated tiles -- under
//     the critical waves' next elimination, which does not touch the matrix pipe.  Barrier B2 (the critical waves reach it
//     after that elimination) publishes the committed matrix for the next block's U / X operands.
// Every entry sees the operations of tile_blocked_block in the same order with the same operands (the update tiles keep
// their k-step order 0..3; the next micro-block is computed twice, by its critical wave and by its helper, from the same
// operands: the same bits), so the inverse, the pivots and the hand-over state of a block that fails the Bunch-Kaufman
// check are bit for bit those of the four-wave path.
#pragma once
#include "tile_blocked.hpp"

namespace pyipm {

struct Blocked8Scratch {
    double Pnext[16][17];       // the next diagonal micro-block after the current block's update (full 16x16, lower part read)
    double Xbuf[4][4][64];      // X of row tile t, k-step s, lane: exactly the critical wave's registers
};

// critical wave, block KB.  a[]: in = micro-block KB (lane i15 holds row i15, lower-valid completion), out = micro-block KB + 1.
// First block (KB == 0): the rejection thresholds do not exist in shared memory yet (the helpers produce the column maxima
// under the first elimination): they are formed from the four partial maxima.
template <int KB, int STRIDE>
__device__ __forceinline__ bool tile_blocked8_crit(double (&stage)[TB][STRIDE], BlockedScratch& bs, Blocked8Scratch& ex,
                                                   double* __restrict__ dsave, const BlockedLane& L, double (&a)[16],
                                                   const double (&xmax)[4][TB], double pivtol_rel, unsigned long long* __restrict__ dbg)
{
#define PYIPM_TB_STAMP(ph_) if (dbg && lane == 0) dbg[8 + 32 * KB + 8 * wave + (ph_)] = clock64();
#define PYIPM_TB_LD(off_) (*reinterpret_cast<const double*>(sb + (off_)))
    constexpr int k0 = 16 * KB;
    constexpr int D8 = (int)sizeof(double), S8 = STRIDE * D8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i15 = lane & 15, q = lane >> 4;
    const int t = wave;
    char* sb = reinterpret_cast<char*>(&stage[0][0]);
    double d[16], r, lmax = 0.0, ctr = (double)i15, t_, u_, mk_;
    if (dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PYIPM_TB_STAMP(0) }
    PYIPM_MICROBLOCK_ASM_0(a, d, r, lmax, ctr, t_, u_, mk_);
    PYIPM_MICROBLOCK_ASM_1(a, d, r, lmax, ctr, t_, u_, mk_);
    PYIPM_TB_STAMP(1)
    (void)r;
    if (q == 0) {
        #pragma unroll
        for (int c = 0; c < 16; ++c) bs.Ms[i15][c] = a[c];
        if (i15 == 0) {
            #pragma unroll
            for (int c = 0; c < 16; ++c) bs.Ds[c] = d[c];
        }
    }
    // (what only needs the elimination -- M, M', the pivots and their reciprocals -- is formed in front of the barrier)
    int bad = (int)!(lmax <= PYIPM_BK_INV_ALPHA);
    double rsel[4], dsel[4], aM[4], aMT[4];
    #pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 4 * s + q;
        const double v = bs.Ms[i15][c], w = bs.Ms[c][i15];
        dsel[s] = bs.Ds[c];
        aM[s] = fma(v, L.mlt[s], L.meq[s]);
        aMT[s] = fma(w, L.mgt[s], L.meq[s]);
    }
    #pragma unroll
    for (int s = 0; s < 4; ++s) rsel[s] = blocked_recip(dsel[s]);
    __syncthreads();                                   // B2: the helpers have committed the block before (KB == 0: column maxima are out)
    PYIPM_TB_STAMP(4)
    double bWt[4];                                     // the wave's own rows of W (B operand of U; A operand of the next micro-block's update)
    #pragma unroll
    for (int s = 0; s < 4; ++s)
        bWt[s] = t >= KB ? PYIPM_TB_LD(L.oRowT + (k0 + 4 * s) * D8) : PYIPM_TB_LD(L.oColT + (k0 + 4 * s) * S8);
    double4_tb Cd;                                     // tile (KB + 1, KB + 1) of the wave that owns those rows (junk elsewhere)
    #pragma unroll
    for (int rr = 0; rr < 4; ++rr) Cd[rr] = PYIPM_TB_LD(L.oRowT + (16 * (KB < 3 ? KB + 1 : KB) + 4 * rr) * D8);
    double pt4[4];
    #pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = k0 + 4 * s + q;
        if (KB == 0) pt4[s] = pivtol_rel * fmax(fmax(xmax[0][c], xmax[1][c]), fmax(xmax[2][c], xmax[3][c]));
        else         pt4[s] = bs.ptol[c];
    }
    double4_tb U = {0.0, 0.0, 0.0, 0.0}, U2 = {0.0, 0.0, 0.0, 0.0};
    U  = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[0], bWt[0], U, 0, 0, 0);
    U2 = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[2], bWt[2], U2, 0, 0, 0);
    U  = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[1], bWt[1], U, 0, 0, 0);
    U2 = __builtin_amdgcn_mfma_f64_16x16x4f64(aM[3], bWt[3], U2, 0, 0, 0);
    #pragma unroll
    for (int s = 0; s < 4; ++s) bad |= (int)!(fabs(dsel[s]) > pt4[s]) | (int)!(fabs(dsel[s]) <= 1.0e300);
    if (dbg) { asm volatile("" :: "v"(U[0])); PYIPM_TB_STAMP(5) }
    const bool isK = t == KB;
    double Lr[4];
    const double lim = t > KB ? PYIPM_BK_INV_ALPHA : 1.0e300;
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
    double4_tb X = {0.0, 0.0, 0.0, 0.0}, X2 = {0.0, 0.0, 0.0, 0.0};
    X  = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[0], xb[0], X, 0, 0, 0);
    X2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2], xb[2], X2, 0, 0, 0);
    X  = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[1], xb[1], X, 0, 0, 0);
    X2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[3], xb[3], X2, 0, 0, 0);
    #pragma unroll
    for (int rr = 0; rr < 4; ++rr) { X[rr] += X2[rr]; bad |= (int)!(fabs(X[rr]) <= 1.0e300); }
    #pragma unroll
    for (int s = 0; s < 4; ++s) ex.Xbuf[t][s][lane] = X[s];
    if (KB < 3) {
        // the next micro-block: C(KB+1, KB+1) -= X W' with the k-steps in the order of tile_blocked_block
        if (t == KB + 1) {                             // (wave-uniform)
            #pragma unroll
            for (int s = 0; s < 4; ++s) Cd = __builtin_amdgcn_mfma_f64_16x16x4f64(bWt[s], -X[s], Cd, 0, 0, 0);
            #pragma unroll
            for (int rr = 0; rr < 4; ++rr) ex.Pnext[i15][q + 4 * rr] = Cd[rr];
        }
    }
    if (dbg) { asm volatile("" :: "v"(X[0]), "v"(Cd[0])); PYIPM_TB_STAMP(2) }
    if (__ballot(bad != 0) != 0ull && lane == 0) bs.fail = 1;
    if (KB == 0 && t == 0) bs.ptol[lane] = pivtol_rel * fmax(fmax(xmax[0][lane], xmax[1][lane]), fmax(xmax[2][lane], xmax[3][lane]));
    __syncthreads();                                   // B1: X, the next micro-block and the verdict are out
    PYIPM_TB_STAMP(6)
    if (bs.fail) return false;                         // (uniform over all eight waves) nothing of this block has been written
    if (isK && i15 < 4) {
        const long long n0 = i15 == 0 ? -1ll : 0ll, n1 = i15 == 1 ? -1ll : 0ll, n2 = i15 == 2 ? -1ll : 0ll, n3 = i15 == 3 ? -1ll : 0ll;
        dsave[k0 + 4 * i15 + q] = __longlong_as_double((n0 & __double_as_longlong(dsel[0])) | (n1 & __double_as_longlong(dsel[1])) |
                                                       (n2 & __double_as_longlong(dsel[2])) | (n3 & __double_as_longlong(dsel[3])));
    }
    if (KB < 3) {
        #pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = c <= i15 ? ex.Pnext[i15][c] : ex.Pnext[c][i15];
    }
    PYIPM_TB_STAMP(3)
#undef PYIPM_TB_STAMP
#undef PYIPM_TB_LD
    return true;
}

// helper waves, block KB.  The six tiles (t, t'), t >= t', t and t' != KB, that the block updates are spread over the four
// helper waves whatever their rows (every operand comes from shared memory: W rows of t' from column block K, X of row t
// from Xbuf, the tile itself): r0 < r1 < r2 the three row tiles other than KB,
//     helper 0: (r0, r0), (r2, r2)    helper 1: (r1, r0), (r2, r1)    helper 2: (r1, r1)    helper 3: (r2, r0)
// Helper h also commits X of row tile h into column block K (the micro-block's own rows receive -inv(P) there, the rows
// above the block their X' in the transposed position).  Tiles above the diagonal, which the four-wave path updates and
// stores as junk, are not touched (nobody reads them).
template <int KB, int I> struct Tile8Rows { static constexpr int value = I + (I >= KB ? 1 : 0); };     // r_I

template <int KB, int T, int TP, int STRIDE>
struct Tile8Update {
    double bW[4];
    double4_tb Cn;
    static constexpr int k0 = 16 * KB, D8 = (int)sizeof(double), S8 = STRIDE * D8;
    __device__ __forceinline__ void load(const char* sb, const BlockedLane& L) {       // between B2 and B1
        #pragma unroll
        for (int s = 0; s < 4; ++s)
            bW[s] = TP > KB ? *reinterpret_cast<const double*>(sb + L.oRow + 16 * TP * S8 + (k0 + 4 * s) * D8)
                            : *reinterpret_cast<const double*>(sb + L.oCol + (k0 + 4 * s) * S8 + 16 * TP * D8);
        #pragma unroll
        for (int rr = 0; rr < 4; ++rr) Cn[rr] = *reinterpret_cast<const double*>(sb + L.oRow + 16 * T * S8 + (16 * TP + 4 * rr) * D8);
    }
    __device__ __forceinline__ void run(char* sb, const BlockedLane& L, const Blocked8Scratch& ex, int lane) {     // after B1
        #pragma unroll
        for (int s = 0; s < 4; ++s) Cn = __builtin_amdgcn_mfma_f64_16x16x4f64(bW[s], -ex.Xbuf[T][s][lane], Cn, 0, 0, 0);
        #pragma unroll
        for (int rr = 0; rr < 4; ++rr) *reinterpret_cast<double*>(sb + L.oRow + 16 * T * S8 + (16 * TP + 4 * rr) * D8) = Cn[rr];
    }
};

template <int KB, int STRIDE>
__device__ __forceinline__ bool tile_blocked8_help(double (&stage)[TB][STRIDE], BlockedScratch& bs, Blocked8Scratch& ex,
                                                   const BlockedLane& L)
{
    constexpr int k0 = 16 * KB;
    constexpr int D8 = (int)sizeof(double), S8 = STRIDE * D8;
    constexpr int r0 = Tile8Rows<KB, 0>::value, r1 = Tile8Rows<KB, 1>::value, r2 = Tile8Rows<KB, 2>::value;
    const int tid = threadIdx.x, lane = tid & 63;
    const int t = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
    char* sb = reinterpret_cast<char*>(&stage[0][0]);
    Tile8Update<KB, r0, r0, STRIDE> u00; Tile8Update<KB, r2, r2, STRIDE> u22;
    Tile8Update<KB, r1, r0, STRIDE> u10; Tile8Update<KB, r2, r1, STRIDE> u21;
    Tile8Update<KB, r1, r1, STRIDE> u11; Tile8Update<KB, r2, r0, STRIDE> u20;
    __syncthreads();                                   // B2: the working matrix holds the blocks before this one
    if (t == 0)      { u00.load(sb, L); u22.load(sb, L); }
    else if (t == 1) { u10.load(sb, L); u21.load(sb, L); }
    else if (t == 2) { u11.load(sb, L); }
    else             { u20.load(sb, L); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (column block K is overwritten with X by the helpers after B1)
    __syncthreads();                                   // B1
    if (bs.fail) return false;
    if (t == 0)      { u00.run(sb, L, ex, lane); u22.run(sb, L, ex, lane); }
    else if (t == 1) { u10.run(sb, L, ex, lane); u21.run(sb, L, ex, lane); }
    else if (t == 2) { u11.run(sb, L, ex, lane); }
    else             { u20.run(sb, L, ex, lane); }
    {
        const int xo = t >= KB ? L.oRowT + k0 * D8 : L.oColT + k0 * S8;
        #pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const double x = ex.Xbuf[t][rr][lane];
            if (t >= KB) *reinterpret_cast<double*>(sb + xo + 4 * rr * D8) = x;
            else         *reinterpret_cast<double*>(sb + xo + 4 * rr * S8) = x;
        }
    }
    return true;
}

template <int STRIDE>
__device__ __forceinline__ void tile_blocked8_lane(BlockedLane& L, int lane, int t) {
    const int i15 = lane & 15, q = lane >> 4;
    constexpr int D8 = (int)sizeof(double), S8 = STRIDE * D8;
    #pragma unroll
    for (int c = 0; c < 16; ++c) L.aoff[c] = c <= i15 ? i15 * S8 + c * D8 : c * S8 + i15 * D8;
    L.oRow = i15 * S8 + q * D8;  L.oCol = q * S8 + i15 * D8;
    L.oRowT = L.oRow + 16 * t * S8;  L.oColT = L.oCol + 16 * t * D8;
    #pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 4 * s + q;
        L.mlt[s] = c < i15 ? 1.0 : 0.0;  L.mgt[s] = c > i15 ? 1.0 : 0.0;  L.meq[s] = c == i15 ? 1.0 : 0.0;
    }
}

// 512 threads.  On entry everyone has passed a barrier after the last write to stage; bs.fail = 0 and xmax[0..3][.] (the
// four partial column maxima of the tile) are written by the helpers before their first barrier in here.  Returns the
// number of micro-blocks swept, the same value in all eight waves; on return everyone has passed a barrier after the last
// write to stage.
template <int STRIDE>
__device__ __forceinline__ int tile_blocked8_sweep_crit(double (&stage)[TB][STRIDE], BlockedScratch& bs, Blocked8Scratch& ex,
                                                        double* __restrict__ dsave, const double (&xmax)[4][TB], double pivtol_rel,
                                                        unsigned long long* __restrict__ dbg)
{
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    BlockedLane L;
    tile_blocked8_lane<STRIDE>(L, lane, t);
    const char* sb = reinterpret_cast<const char*>(&stage[0][0]);
    double a[16];
    #pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = *reinterpret_cast<const double*>(sb + L.aoff[c]);
    int done = 4;
    if (!tile_blocked8_crit<0>(stage, bs, ex, dsave, L, a, xmax, pivtol_rel, dbg)) done = 0;
    else if (!tile_blocked8_crit<1>(stage, bs, ex, dsave, L, a, xmax, pivtol_rel, dbg)) done = 1;
    else if (!tile_blocked8_crit<2>(stage, bs, ex, dsave, L, a, xmax, pivtol_rel, dbg)) done = 2;
    else if (!tile_blocked8_crit<3>(stage, bs, ex, dsave, L, a, xmax, pivtol_rel, dbg)) done = 3;
    if (done == 4) __syncthreads();                    // the helpers' last commit
    return done;
}

template <int STRIDE>
__device__ __forceinline__ bool tile_blocked8_sweep_help(double (&stage)[TB][STRIDE], BlockedScratch& bs, Blocked8Scratch& ex)
{
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
    BlockedLane L;
    tile_blocked8_lane<STRIDE>(L, lane, t);
    if (!tile_blocked8_help<0>(stage, bs, ex, L)) return false;
    if (!tile_blocked8_help<1>(stage, bs, ex, L)) return false;
    if (!tile_blocked8_help<2>(stage, bs, ex, L)) return false;
    if (!tile_blocked8_help<3>(stage, bs, ex, L)) return false;
    __syncthreads();
    return true;
}

}  // namespace pyipm
