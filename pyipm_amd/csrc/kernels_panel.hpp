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
    for (int it = 0; it < nref; ++it) {
        __syncthreads();
        PYIPM_STAGE_TILE(X, -1.0, Tsave)
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
        PYIPM_STAGE_TILE(X, 1.0, Tinv)
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
template <int KS = 8>                     // k-steps (of 4 columns) whose Wn operands are in flight together
__device__ __forceinline__ void strip_update(double4_t (&c2)[4], const double4_t (&acc)[4], const double* __restrict__ wn,
                                             int64_t ldw, int l15, int l4)
{
    #pragma unroll
    for (int h = 0; h < 16 / KS; ++h) {
        double wa[4][KS];
        #pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            #pragma unroll
            for (int t = 0; t < 4; ++t) wa[t][ks] = wn[(16 * t + l15) + (int64_t)(4 * (KS * h + ks) + l4) * ldw];
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

// ---------------------------------------------------------------------------------------------
// Tile steps [t0, t1) of a diagonal block in ONE launch (round 3, option "chain_persist").  k_tile_step needs one launch per
// tile because block 0 of launch t reads what the other blocks of launch t - 1 wrote; the launch boundary and the global
// round trips behind it are ~5 of the 22.5 us a tile takes, and beside a bulk update EVERY launch waits for a free slot
// (section 4, lessons r3: 130 us instead of 33).  Here the roles are fixed for the whole launch and the hand-overs are
// device flags, as in k_bwd_sweep:
//   * workgroup 0 (critical): for t = t0 .. t1-1: stage t-1 on the 64 rows of tile t (its own product inv(T[t-1]) or the
//     previous launch's), the inversion of tile t, inv[t] raised.  Before stage t-1 it needs row tile t as its OWNER left it
//     (stages <= t-2 applied, -S of column tile t-1 saved): prog[t] >= t.
//   * workgroup b >= 1 owns row tile r = t0 + b for the whole launch and applies the stages tp = max(t0-1, 0) .. min(t1-2,
//     r-2) to it, all column tiles tp+1 .. r of the stage in turn (stage r-1 is the critical workgroup's): it waits for
//     inv[tp] and for the -S rows of the row tiles between tp and r (prog[v] >= tp + 1), and raises prog[r] = tp + 2.
//     (t0 = 0: it first saves -S of column tile 0, prog[r] = 1.)
// The same device functions on the same operands in the same order as k_tile_step: the same bits.  Data crosses workgroups
// through plain stores / loads bracketed by release / acquire fences (buffer_wbl2 / buffer_inv, ~1 us each: the 64 x 64
// operands are too many for per-value atomics); flags are relaxed agent-scope atomics; every poll has a timeout (*err).
__device__ __forceinline__ bool chain_wait(const unsigned* p, unsigned want, unsigned* err, unsigned long long timeout) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if ((unsigned long long)wall_clock64() - t0 > timeout) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
}

// (The arguments travel as ONE struct that every phase of the loop re-reads from the kernel-argument segment through a
// pointer the compiler cannot see through: kept live in SGPRs across tile_invert_dev -- which needs a hundred of its own --
// they spilled into VGPRs, 97 SGPRs and with them 153 VGPRs into scratch.)
// tile_invert_dev behind a real call: inlined into the loop of k_tile_chain its per-lane constants (offsets, operand masks:
// ~100 registers that depend on the lane only) were hoisted out of the loop and kept alive across the iterations -- 154
// VGPRs spilled.  The scratch travels as an LDS pointer so that its accesses stay ds_ instructions.
typedef __attribute__((address_space(3))) TileScratch* TileScratchLds;
__device__ __attribute__((noinline)) void tile_invert_from_stage(
    TileScratchLds smp, const double* A, int64_t ld, int64_t grow0, int64_t lcol0, double* Tinv, double* Tsave, double* Tflag,
    double refine_cond, DevStats* st, int64_t Nreal, double pivtol_rel, const unsigned long long* anorm_bits, int64_t neg_from,
    unsigned long long* dbg, int blocked)
{
    tile_invert_dev(*(TileScratch*)smp, A, ld, grow0, lcol0, Tinv, Tsave, Tflag, refine_cond, st, Nreal, pivtol_rel, anorm_bits, neg_from,
                    dbg, /*from_stage=*/true, blocked != 0);
}

struct ChainArgs {
    double* A; int64_t ld, c0, lc0; int nT, t0, t1, nref, blocked;
    double* W; int64_t ldw; double* Dinv; double* Tsv; double* Tflag;
    double refine_cond, pivtol_rel; DevStats* st; int64_t Nreal, neg_from;
    const unsigned long long* anorm_bits; unsigned long long* dbg;
    unsigned* sync; unsigned* err; unsigned long long timeout;
};
typedef const __attribute__((address_space(4))) ChainArgs* ChainArgsPtr;
#define PYIPM_CHAIN_ARGS(a_)                                                                     \
    ChainArgsPtr a_ = (ChainArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();                       \
    asm volatile("" : "+s"(a_));

__global__ __launch_bounds__(256, 2) void k_tile_chain(ChainArgs args_by_value)
{
    __shared__ TileScratch sm;
    __shared__ int ok_s;
    static_assert(sizeof(TileScratch) >= sizeof(double) * TB * (TB + 2), "X must fit into the tile scratch");
    double (&X)[TB][TB + 2] = *reinterpret_cast<double (*)[TB][TB + 2]>(&sm);
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t TT = (int64_t)TB * TB;
    (void)args_by_value;
    // release: every thread's stores have reached L2, one thread writes the L2 back and raises the flag
#define PYIPM_CHAIN_PUBLISH(ptr_, val_)                                                                         \
    { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();                                          \
      if (tid == 0) { __threadfence(); __hip_atomic_store((ptr_), (unsigned)(val_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
    if (tid == 0) ok_s = 1;
    __syncthreads();
    int t0, t1, nT;
    { PYIPM_CHAIN_ARGS(a) t0 = a->t0; t1 = a->t1; nT = a->nT; }
    const int tp_first = t0 > 0 ? t0 - 1 : 0;                          // the first stage this launch applies
    if (b == 0) {
        for (int t = t0; t < t1; ++t) {
            {
            PYIPM_CHAIN_ARGS(a)
            double* __restrict__ A = a->A; const int64_t ld = a->ld, c0 = a->c0, lc0 = a->lc0, ldw = a->ldw;
            if (t == 0) {
                // the first tile comes from global memory (what tile_invert_dev does without from_stage; one call site below)
                double tmp[TB * TB / 256];
                #pragma unroll
                for (int q = 0; q < TB * TB / 256; ++q) {
                    const int e = tid + 256 * q, ii = e & 63, jj = e >> 6;
                    tmp[q] = (ii >= jj) ? A[(c0 + ii) + (lc0 + jj) * ld] : 0.0;
                }
                #pragma unroll
                for (int q = 0; q < TB * TB / 256; ++q) {
                    const int e = tid + 256 * q, ii = e & 63, jj = e >> 6;
                    if (ii >= jj) sm.stage[ii][jj] = tmp[q];
                }
            } else {
                const int tp = t - 1;
                unsigned* prog = a->sync + nT;
                if (t - 2 >= tp_first) {
                    // the owner of row tile t applied stages <= t-2 in this launch (prog[t] >= t)
                    if (tid == 0 && !chain_wait(prog + t, (unsigned)t, a->err, a->timeout)) ok_s = 0;
                    __syncthreads();
                    if (!ok_s) return;
                    __threadfence();                                        // acquire: what that workgroup wrote
                } else if (t0 == 0 && t == 1) {
                    if (tid == 0 && !chain_wait(prog + 1, 1u, a->err, a->timeout)) ok_s = 0;      // -S of column tile 0 of row tile 1
                    __syncthreads();
                    if (!ok_s) return;
                    __threadfence();
                }
                const int64_t i = c0 + (int64_t)t * TB + wave * 16 + l15;
                int nref = a->nref;
                if (nref > 0 && a->Tflag[tp] == 0.0) nref = 0;
                PYIPM_STAGE_TILE(X, 1.0, a->Dinv + tp * TT)
                double sb[16];
                #pragma unroll
                for (int ks = 0; ks < 16; ++ks) sb[ks] = -a->W[i + (int64_t)(tp * TB + ks * 4 + l4) * ldw];
                __syncthreads();
                double4_t acc[4];
                strip_scale(X, a->Dinv + tp * TT, a->Tsv + tp * TT, nref, sb, tid, l15, l4, acc);
                {
                    double gmax = 0.0;
                    #pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            A[i + (lc0 + tp * TB + tt * 16 + l4 + 4 * r) * ld] = acc[tt][r];
                            gmax = fmax(gmax, fabs(acc[tt][r]));
                        }
                    gmax = wave_max(gmax);
                    if (lane == 0) atomicMax(&a->st->growth_bits, (unsigned long long)__double_as_longlong(gmax));
                }
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
            }
            {
            PYIPM_CHAIN_ARGS(a)
            tile_invert_from_stage((TileScratchLds)&sm, a->A, a->ld, a->c0 + (int64_t)t * TB, a->lc0 + (int64_t)t * TB, a->Dinv + t * TT,
                                   a->Tsv + t * TT, a->Tflag + t, a->refine_cond, a->st, a->Nreal, a->pivtol_rel, a->anorm_bits, a->neg_from,
                                   a->dbg, a->blocked);
            }
            {
            PYIPM_CHAIN_ARGS(a)
            PYIPM_CHAIN_PUBLISH(a->sync + t, 1)
            }
            __syncthreads();
        }
        return;
    }
    // ---- the owner of row tile r ----
    const int r = t0 + b;
    if (r >= nT) return;
    PYIPM_CHAIN_ARGS(a)
    double* __restrict__ A = a->A; double* __restrict__ W = a->W;
    const int64_t ld = a->ld, c0 = a->c0, lc0 = a->lc0, ldw = a->ldw;
    unsigned* inv = a->sync; unsigned* prog = a->sync + nT;
    // was the save of -S of column tile `col` (by any row owner) made inside this launch?
    auto in_launch = [&](int col) { return t0 == 0 || col >= t0; };
    const int64_t i = c0 + (int64_t)r * TB + wave * 16 + l15;            // this lane's (global) row
    if (t0 == 0) {
        double tmp[TB * TB / 256];
        const int64_t r0 = c0 + (int64_t)r * TB;
        #pragma unroll
        for (int q = 0; q < TB * TB / 256; ++q) { const int e = tid + 256 * q; tmp[q] = A[(r0 + (e & 63)) + (lc0 + (e >> 6)) * ld]; }
        #pragma unroll
        for (int q = 0; q < TB * TB / 256; ++q) { const int e = tid + 256 * q; W[(r0 + (e & 63)) + (int64_t)(e >> 6) * ldw] = -tmp[q]; }
        PYIPM_CHAIN_PUBLISH(prog + r, 1)
    }
    int tp_last = t1 - 2; if (tp_last > r - 2) tp_last = r - 2;
    for (int tp = tp_first; tp <= tp_last; ++tp) {
        const int t = tp + 1;
        // inv(T[tp]) (this launch's, if tp >= t0) and the -S rows of the row tiles between the stage and this one
        {
            bool ok = true;
            if (tid == 0 && tp >= t0) ok = chain_wait(inv + tp, 1u, a->err, a->timeout);
            if (in_launch(tp) && tid >= 1 && tid < r - tp && tid < 64) ok = chain_wait(prog + tp + tid, (unsigned)(tp + 1), a->err, a->timeout);   // v = tp + tid in [tp+1, r-1]
            if (!ok) ok_s = 0;
        }
        __syncthreads();
        if (!ok_s) return;
        __threadfence();                                                     // acquire
        int nref = a->nref;
        if (nref > 0 && a->Tflag[tp] == 0.0) nref = 0;
        PYIPM_STAGE_TILE(X, 1.0, a->Dinv + tp * TT)
        double sb[16];
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) sb[ks] = -W[i + (int64_t)(tp * TB + ks * 4 + l4) * ldw];
        __syncthreads();
        double4_t acc[4];
        strip_scale(X, a->Dinv + tp * TT, a->Tsv + tp * TT, nref, sb, tid, l15, l4, acc);
        {
            double gmax = 0.0;
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                #pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    A[i + (lc0 + tp * TB + tt * 16 + l4 + 4 * rr) * ld] = acc[tt][rr];
                    gmax = fmax(gmax, fabs(acc[tt][rr]));
                }
            gmax = wave_max(gmax);
            if (lane == 0) atomicMax(&a->st->growth_bits, (unsigned long long)__double_as_longlong(gmax));
        }
        for (int v = t; v <= r; ++v) {
            double4_t c2[4];
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                #pragma unroll
                for (int rr = 0; rr < 4; ++rr) c2[tt][rr] = A[i + (lc0 + v * TB + tt * 16 + l4 + 4 * rr) * ld];
            strip_update(c2, acc, W + (c0 + (int64_t)v * TB) + (int64_t)(tp * TB) * ldw, ldw, l15, l4);
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                #pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int c = tt * 16 + l4 + 4 * rr;
                    A[i + (lc0 + v * TB + c) * ld] = c2[tt][rr];
                    if (v == t) W[i + (int64_t)(t * TB + c) * ldw] = -c2[tt][rr];
                }
        }
        PYIPM_CHAIN_PUBLISH(prog + r, tp + 2)
        __syncthreads();
    }
#undef PYIPM_CHAIN_PUBLISH
}

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
