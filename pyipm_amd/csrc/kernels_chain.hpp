// kernels_chain.hpp -- consecutive tile steps of a diagonal block in ONE launch (round 6).
//
// k_tile_step (kernels_panel.hpp) is one launch per 64 x 64 diagonal tile: launch t eliminates tile t - 1 from the rows of
// the diagonal block and inverts tile t.  Between two dependent launches of one stream the device spends 4.9 us (HIP events,
// tools/group_trace.py) -- 64 times per step at config 2, 128 times at N = 32768, and on every rank's path across GPUs --
// beside 2 us until the first wave of a launch runs and a cold instruction cache on whatever CU it lands on.
//
// k_tile_chain runs the steps [ta, tb) of a diagonal block of nT tiles as one launch of PERSISTENT workgroups:
//   * workgroup 0 is the chain: for t = ta .. tb - 1 it applies stage t - 1 to the 64 rows of tile t (scaling product + the
//     K = 64 update of the diagonal tile) and inverts tile t out of shared memory -- the critical block of k_tile_step, with
//     the same device functions in the same order;
//   * every other workgroup is a unit (r, y): it OWNS row tile r of the diagonal block for the whole launch and applies the
//     stages tp = max(ta - 1, 0) .. min(r - 2, tb - 2) to the column tiles v = tp + 1 .. r with v mod ny(r) = y (a far row tile
//     has many: ny(r) units share it, each forming the scaling product itself, as the y-blocks of k_tile_step do).  Its
//     column tiles never leave the unit, so they move through plain loads and stores;
//   * (per-panel schedule across GPUs, ChainGeo::nR > nT) EXTRA row tiles right below the block -- the rows the next panel's
//     owner waits for -- have units of their own that apply every stage 0 .. nT - 1 as the chain publishes the tiles, in place of
//     a k_panel_rest launch behind the chain; a word set behind their head (which arrives on another stream) releases them.
// What crosses workgroups inside the launch -- inv(T), T and the refinement flag of a tile (chain -> units), the -S rows W of
// a finished column tile (unit -> units, unit -> chain) and row tile t's diagonal tile (unit -> chain) -- is written and read
// with relaxed agent-scope atomics (sc1: written through, read past the L1) and announced by ONE progress word per
// workgroup, stored after the writer's waves have drained their stores (s_waitcnt vmcnt(0) + barrier): no fence anywhere --
// round 3's fenced k_tile_chain paid a buffer_inv per acquire and was slower than the launches (kernels_panel.hpp).  The
// words carry an epoch (`base`, from the host: no memset between launches).  Every poll has a timeout that sets the sticky
// error word; a workgroup that gives up keeps to the control flow (barriers stay matched) on whatever data it has, so the
// launch always ends and the host reports the error with the factorisation's statistics (factor_end).
//
// Every entry of the diagonal block sees the operations of the launch-per-tile schedule in the same order -- scaling =
// strip_scale (+ refinement), updates = strip_update, MFMA groups of 4 columns ascending, inversion = tile_invert_dev --
// so the factor is bit for bit the same (tests/test_gpu_tile_blocked.py).
#pragma once
#include "kernels_panel.hpp"

namespace pyipm {

// (struct ChainGeo: ctx.hpp)

// units of row tile r: ny = ceil(#column tiles of its first stage / cpy), 1 .. 4.  Row tiles r >= nT are EXTRA rows below the
// diagonal block (per-panel schedule, ChainGeo::nR): their column tiles end at the block's last one.
__host__ __device__ inline int chain_ny(int r, int ta, int cpy, int nT = 1 << 30) {
    const int rc = r < nT ? r : nT - 1;
    const int cols = rc - (ta > 0 ? ta - 1 : 0);
    int ny = (cols + cpy - 1) / cpy;
    return ny < 1 ? 1 : (ny > 4 ? 4 : ny);
}
// first row tile with a unit: ta = 0: row 1 (W of column tile 0 is saved by the units); else row ta + 1 (row ta's only stage left is the chain's)
__host__ __device__ inline int chain_first_row(int ta) { return ta > 0 ? ta + 1 : 1; }
__host__ inline int chain_units(int ta, int nT, int cpy, int nR = 0) {
    int u = 0;
    if (nR < nT) nR = nT;
    for (int r = chain_first_row(ta); r < nR; ++r) u += chain_ny(r, ta, cpy, nT);
    return u;
}

// Wave-wide poll: every lane with `active` waits until its word has reached `want` (progress counts of one launch never
// differ by 2^31).  Returns false (wave-uniform) on timeout or when somebody else gave up; never blocks beyond `timeout`.
__device__ __forceinline__ bool chain_wait(const unsigned* word, bool active, unsigned want, unsigned* err, unsigned long long timeout,
                                           bool nap)
{
    bool ok = !active || (int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0;
    if (__ballot(!ok) == 0ull) return true;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        if (!ok) ok = (int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0;
        if (__ballot(!ok) == 0ull) return true;
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if ((unsigned long long)wall_clock64() - t0 > timeout) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        if (nap) __builtin_amdgcn_s_sleep(2);
    }
}

// What the rows below a diagonal block wait for when its chain is ONE launch (ta = 0): the chain has inverted `crit_need`
// tiles and every unit of the row tiles from `row0` on has completed `unit_need` stages.  One wave on the rows' stream in
// place of an event between two launches of the chain; the kernels behind it start with the acquire of a kernel boundary
// and read what the chain has written through.  *err is the chain's sticky error word.
// (Two uses per sub-panel q of a block: in front of its rows kernels -- tiles up to q's last inverted, the later row tiles one
//  stage short of it: W of q's columns is final -- and behind them, in front of whoever reads panel q as a whole (the forward
//  substitution): every stage of q's tiles applied to every row tile, i.e. L of q's columns complete inside the diagonal block.)
__global__ __launch_bounds__(64) void k_chain_wait(ChainGeo cg, int crit_need, int row0, int unit_need)
{
    const int lane = threadIdx.x;
    bool ok = chain_wait(cg.sync, lane == 0, cg.base + (unsigned)crit_need, cg.err, cg.timeout, true);
    for (int idx = lane; ok && idx < 4 * cg.nT; idx += 64) {
        const int r = idx >> 2, y = idx & 3;
        const bool act = r >= row0 && r >= 1 && r < cg.nT && y < chain_ny(r, 0, cg.cpy);
        ok = chain_wait(cg.sync + 1 + idx, act, cg.base + (unsigned)unit_need, cg.err, cg.timeout, true);
    }
}

// What the chain keeps close between two steps (the units' side is untouched, and so is every bit): inv(T[t-1]) and its
// refinement flag stay in shared memory where the inversion leaves them -- the step starts at its S rows (staging the inverse
// from memory: 1.4 us; with it in place 0.5; step 22.0 -> 21.0 us).
// Measured and dropped after that (tools/r06_chain_mode.sh; the inversion pays for ANYTHING that lives across it or is
// issued inside it -- its micro-blocks then reload constants from scratch, or lose their schedule):
//   * the diagonal tile brought in beside S by global_load_lds: its latency was hidden already;
//   * the next step's S rows requested from inside the inversion, when a wave's columns of the inverse are final: into registers
//     they were spilled across the inversion's tail (14.2 -> 16.7 us); as global_load_lds into shared memory the compiler
//     drained them at the inversion's last barrier (15.2); as asm statements they cost the tail what they saved the next
//     step (eight M0 writes + issues: +0.7 against -0.6 us);
//   * only the LOOK at the next row's unit words from inside the inversion (in front of its last micro-block) and the step's
//     progress word sent from the start of the next step, behind the s_waitcnt its S rows need anyway: wait 0.32 -> 0.14,
//     publish 0.24 -> 0.08, inversion 14.2 -> 14.8 us; config 2 2.22 -> 2.26 ms.
__global__ __launch_bounds__(256, 2) void k_tile_chain(
    double* A, int64_t ld, int64_t c0, int64_t lc0,                          // the diagonal block: first global / local column
    double* W, int64_t ldw,                                                  // its -S buffer: W[row + k * ldw], k < 64 nT
    double* Dinv, double* Tsv, double* Tflag,                                // of the block's first tile
    double refine_cond, int nref, DevStats* __restrict__ st, int64_t Nreal, double pivtol_rel,
    const unsigned long long* __restrict__ anorm_bits, int64_t neg_from, int blocked, ChainGeo cg,
    unsigned long long* tdbg)                                                // the tile inversion's diagnostics (NULL normally; debug_timeline_ptr)
{
    __shared__ TileScratch sm;
    __shared__ double flag_s;                         // the refinement flag of the tile the chain inverted last
    extern __shared__ char chain_pad[];               // (never touched: its SIZE keeps other workgroups off this one's compute unit, option chain_lds_kb)
    static_assert(sizeof(TileScratch) >= sizeof(double) * TB * (TB + 2), "X must fit into the tile scratch");
    double (&X)[TB][TB + 2] = *reinterpret_cast<double (*)[TB][TB + 2]>(&sm);
    __builtin_amdgcn_s_setprio(3);
    const int64_t TT = (int64_t)TB * TB;
    const int ta = cg.ta, tb = cg.tb, nT = cg.nT;
    const int sfirst = ta > 0 ? ta - 1 : -1;          // first "stage" of this launch (-1: saving W of column tile 0)
    const int rfirst = chain_first_row(ta);
    unsigned* const crit_w = cg.sync;

    if (blockIdx.x == 0) {
        // ------------------------------------------------ the chain ------------------------------------------------
        #pragma clang loop unroll(disable)
        for (int t = ta; t < tb; ++t) {
            const int tid = tile_tid<true>(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int l15 = lane & 15, l4 = lane >> 4;
#define PYIPM_CH_STAMP(k_) if (cg.dbg && tid == 0) cg.dbg[8 * t + (k_)] = wall_clock64();
            PYIPM_CH_STAMP(0)
            if (t == 0) {
                tile_invert_dev<false, true, true>(sm, A, ld, c0, lc0, Dinv, Tsv, Tflag, refine_cond, st, Nreal, pivtol_rel, anorm_bits,
                                                   neg_from, tdbg, false, blocked != 0, nullptr, &X[0], &flag_s);
            } else {
                const int tp = t - 1;
                const int64_t i = c0 + (int64_t)t * TB + wave * 16 + l15;        // this lane's (global) row
                // row tile t has every stage before tp: its units' words (lane y polls unit y)
                if (t >= rfirst) {
                    const int need = tp - sfirst;                                  // stages the units of row t have completed by then
                    const int ny = chain_ny(t, ta, cg.cpy);
                    if (need > 0) chain_wait(cg.sync + 1 + 4 * t + (lane & 3), lane < ny, cg.base + (unsigned)need, cg.err, cg.timeout, false);
                }
                asm volatile("" ::: "memory");
                PYIPM_CH_STAMP(1)
                const bool inl = t > ta;                                         // inv(T[tp]) is this launch's: still in X (behind the barrier that ended its step)
                int nr = nref;
                if (inl) {
                    if (nr > 0 && flag_s == 0.0) nr = 0;
                } else {
                    if (nr > 0 && ldg_c<true>(Tflag + tp) == 0.0) nr = 0;
                    PYIPM_STAGE_TILE_C(X, 1.0, Dinv + tp * TT, true)
                }
                double sb[16];
                #pragma unroll
                for (int ks = 0; ks < 16; ++ks) sb[ks] = -ldg_c<true>(W + i + (int64_t)(tp * TB + ks * 4 + l4) * ldw);
                if (!inl) __syncthreads();                                   // X staged by everybody
                PYIPM_CH_STAMP(2)
                double4_t acc[4];
                strip_scale<true, true>(X, Dinv + tp * TT, Tsv + tp * TT, nr, sb, tid, l15, l4, acc);
                if (cg.dbg) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[3][3])); PYIPM_CH_STAMP(3) }
                {
                    double gmax = 0.0;
                    #pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // (L is written through: the forward substitution that trails the factorisation reads panel q's L inside
                            //  the diagonal block from a kernel that starts while this one still runs -- k_chain_wait in factor_block)
                            stg_c<true>(A + i + (lc0 + tp * TB + tt * 16 + l4 + 4 * r) * ld, acc[tt][r]);
                            gmax = fmax(gmax, fabs(acc[tt][r]));
                        }
                    gmax = wave_max(gmax);
                    if (lane == 0) atomicMax(&st->growth_bits, (unsigned long long)__double_as_longlong(gmax));
                }
                // the diagonal tile: Wn = -S of the chain's own 64 rows, from the registers through shared memory (k_tile_step).
                // (Requesting the tile ahead of the scaling product -- it comes from memory, written through by row t's units --
                //  spilled 30 more registers and slowed the product by 1.1 us: 8.6 against 7.3 us for the steps before an inversion.)
                double4_t c2[4];
                #pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    #pragma unroll
                    for (int r = 0; r < 4; ++r) c2[tt][r] = ldg_c<true>(A + i + (lc0 + t * TB + tt * 16 + l4 + 4 * r) * ld);
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
                PYIPM_CH_STAMP(4)
                tile_invert_dev<false, true, true>(sm, A, ld, c0 + (int64_t)t * TB, lc0 + (int64_t)t * TB, Dinv + t * TT, Tsv + t * TT,
                                                   Tflag + t, refine_cond, st, Nreal, pivtol_rel, anorm_bits, neg_from, tdbg,
                                                   /*from_stage=*/true, blocked != 0, nullptr, &X[0], &flag_s);
            }
            // inv(T[t]), T[t] and its flag are out (every wave's stores drained) before the word says so; the barrier also
            // stands between the inversion's last writes to X / flag_s and the next step's reads
            PYIPM_CH_STAMP(5)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(crit_w, cg.base + (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            PYIPM_CH_STAMP(6)
#undef PYIPM_CH_STAMP
        }
        return;
    }

    // ------------------------------------------------ the units ------------------------------------------------
    // Row tiles nT .. nR - 1 (per-panel schedule, cg.nR > nT) are EXTRA rows right below the diagonal block: the rows of the
    // panel that the next panel's owner is waiting for (its slice 1).  Their units apply EVERY stage 0 .. nT - 1 to the block's
    // column tiles -- what k_panel_rest did in a launch of its own behind the chain (34 us + a launch boundary on every panel's
    // path across GPUs) -- and start when *cg.xword has reached cg.xwant: the head of those rows comes from the second slice of
    // the panel before, on another stream, while the chain is already running on the first.
    const int nR = cg.nR > nT ? cg.nR : nT;
    int r = rfirst, y = 0, ny = 1;
    {
        int u = (int)blockIdx.x - 1;
        for (; r < nR; ++r) {
            ny = chain_ny(r, ta, cg.cpy, nT);
            if (u < ny) { y = u; break; }
            u -= ny;
        }
        if (r >= nR) return;
    }
    const bool xr = r >= nT;                                        // an extra row tile
    const int rc = xr ? nT - 1 : r;                                 // its last column tile
    unsigned* const my_w = cg.sync + 1 + 4 * r + y;
    int done = 0;                                                   // stages completed (the saving of W counts as one)
    if (xr && cg.xword) {
        // (every wave for itself: nothing of these rows is read before; they were written by kernels that ENDED before the word
        //  was set -- read past this XCD's caches all the same, below)
        chain_wait(cg.xword, (threadIdx.x & 63) == 0, cg.xwant, cg.err, cg.timeout, true);
        asm volatile("" ::: "memory");
    }
    if (ta == 0) {
        if (y == 0) {                                               // -S of column tile 0 (launch 0 of the stepped schedule)
            const int tid = threadIdx.x;
            double tmp[TB * TB / 256];
            const int64_t r0 = c0 + (int64_t)r * TB;
            #pragma unroll
            for (int q = 0; q < TB * TB / 256; ++q) {
                const int e = tid + 256 * q;
                const double* src = A + (r0 + (e & 63)) + (lc0 + (e >> 6)) * ld;
                tmp[q] = xr ? ldg_c<true>(src) : *src;
            }
            #pragma unroll
            for (int q = 0; q < TB * TB / 256; ++q) {
                const int e = tid + 256 * q;
                stg_c<true>(W + (r0 + (e & 63)) + (int64_t)(e >> 6) * ldw, -tmp[q]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        done = 1;
        if (threadIdx.x == 0) __hip_atomic_store(my_w, cg.base + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int s0 = ta > 0 ? ta - 1 : 0;
    const int s1 = xr ? tb - 1 : ((r - 2 < tb - 2) ? r - 2 : tb - 2);   // last stage of this unit in this launch (an extra row: every tile of the block)
    #pragma clang loop unroll(disable)
    for (int tp = s0; tp <= s1; ++tp) {
        const int tid = tile_tid<true>(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int l15 = lane & 15, l4 = lane >> 4;
        const int64_t i = c0 + (int64_t)r * TB + wave * 16 + l15;
        // ---- what this stage reads from other workgroups ----
        //  inv(T[tp]) (the chain, unless an earlier launch inverted it);  S of column tile tp of this row tile = W(r, tp): the unit
        //  of this row that owns column tile tp, stage tp - 1;  Wn operands W(v, tp), v = tp + 1 .. rc (v < r) with v mod ny = y: the
        //  unit of row v that owns ITS column tile tp, stage tp - 1 (lane j polls for v = tp + 1 + j)
        if (tp >= ta) chain_wait(crit_w, lane == 0, cg.base + (unsigned)(tp + 1), cg.err, cg.timeout, true);
        {
            const int need = tp - sfirst;
            const int v = tp + 1 + lane;                            // (v = r is among them: r - tp - 1 < 64)
            bool act = need > 0 && ((v <= rc && v % ny == y) || v == r);
            int nyv = 1;
            if (act) {
                nyv = chain_ny(v, ta, cg.cpy, nT);
                // (stage tp - 1 of row v = tp + 1 is that row's last one: the owner of column tile tp wrote W(v, tp) then)
                if (v == r && tp % ny == y) act = false;            // my own column tile
            }
            const int yo = (tp >= 0 ? tp : 0) % nyv;
            if (need > 0) chain_wait(cg.sync + 1 + 4 * v + yo, act && v >= rfirst, cg.base + (unsigned)need, cg.err, cg.timeout, true);
        }
        asm volatile("" ::: "memory");
        if (cg.dbg && tid == 0 && r < 32) cg.dbg[256 + 64 * (4 * r + y) + 2 * tp] = wall_clock64();
        int nr = nref;
        if (nr > 0 && ldg_c<true>(Tflag + tp) == 0.0) nr = 0;
        PYIPM_STAGE_TILE_C(X, 1.0, Dinv + tp * TT, true)
        double sb[16];
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) sb[ks] = -ldg_c<true>(W + i + (int64_t)(tp * TB + ks * 4 + l4) * ldw);
        __syncthreads();
        double4_t acc[4];
        strip_scale<true, true>(X, Dinv + tp * TT, Tsv + tp * TT, nr, sb, tid, l15, l4, acc);
        if (y == 0) {
            double gmax = 0.0;
            #pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    stg_c<true>(A + i + (lc0 + tp * TB + tt * 16 + l4 + 4 * q) * ld, acc[tt][q]);      // (written through: see the chain's L)
                    gmax = fmax(gmax, fabs(acc[tt][q]));
                }
            gmax = wave_max(gmax);
            if (lane == 0) atomicMax(&st->growth_bits, (unsigned long long)__double_as_longlong(gmax));
        }
        // my column tiles of this stage: v in (tp, rc], v mod ny = y
        int vb = tp + 1;
        vb += ((y - vb) % ny + ny) % ny;
        for (int v = vb; v <= rc; v += ny) {
            double4_t c2[4];
            if (v == r || (xr && tp == s0)) {   // (written through by this unit, stage by stage: read the same way -- a written-through store
                #pragma unroll                  //  need not refresh the CU's own L1 copy of the line; an extra row's first touch: see above)
                for (int tt = 0; tt < 4; ++tt)
                    #pragma unroll
                    for (int q = 0; q < 4; ++q) c2[tt][q] = ldg_c<true>(A + i + (lc0 + v * TB + tt * 16 + l4 + 4 * q) * ld);
            } else {
                #pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    #pragma unroll
                    for (int q = 0; q < 4; ++q) c2[tt][q] = A[i + (lc0 + v * TB + tt * 16 + l4 + 4 * q) * ld];
            }
            strip_update<8, true>(c2, acc, W + (c0 + (int64_t)v * TB) + (int64_t)(tp * TB) * ldw, ldw, l15, l4);
            // The diagonal tile of the row goes to the chain: written through.  So is the next column tile (now final: as -S to
            // everybody, and in place) -- in the NEXT stage unit (r, 0) or the chain overwrites it with L from another compute
            // unit, possibly another XCD: a plain store would leave a dirty copy in this XCD's L2, and whichever L2 writes its
            // line back last at the end of the launch would win (one launch per tile had a write-back between any two writers).
            if (v == r || v == tp + 1) {
                #pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    #pragma unroll
                    for (int q = 0; q < 4; ++q) stg_c<true>(A + i + (lc0 + v * TB + tt * 16 + l4 + 4 * q) * ld, c2[tt][q]);
            } else {
                #pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    #pragma unroll
                    for (int q = 0; q < 4; ++q) A[i + (lc0 + v * TB + tt * 16 + l4 + 4 * q) * ld] = c2[tt][q];
            }
            if (v == tp + 1) {
                #pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    #pragma unroll
                    for (int q = 0; q < 4; ++q) stg_c<true>(W + i + (int64_t)((tp + 1) * TB + tt * 16 + l4 + 4 * q) * ldw, -c2[tt][q]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++done;
        if (tid == 0) __hip_atomic_store(my_w, cg.base + (unsigned)done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cg.dbg && tid == 0 && r < 32) cg.dbg[256 + 64 * (4 * r + y) + 2 * tp + 1] = wall_clock64();
    }
}

}  // namespace pyipm
