// kernels_factor.hpp — K3 tile sweep-inversion (Bunch-Kaufman in a tile), panel scaling,
// K4 fp64-MFMA symmetric rank-k update.  Replaces the LAPACK factorisation reached through
// scipy.linalg.solve at /root/reference/pyipm.py:18-20,1720 and the eigen-inertia test of
// reghess (pyipm.py:1378-1381).
//
// Factorisation:  A = Lb * blockdiag(T_k) * Lb',  64x64 block pivots T_k,
//   Lb[i,k] = S[i,k] * inv(T_k)   (S = Schur-complemented column block, kept as "W"),
//   trailing update  A22 -= Lb21 * W21'.
#pragma once
#include "ctx.hpp"
#include "tile_blocked.hpp"
#include "tile_blocked8.hpp"

namespace pyipm {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double row16_t __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// K3: invert one 64x64 symmetric tile by symmetric sweeps with Bunch-Kaufman 1x1 / 2x2 pivot
// selection restricted to the tile.  No row/column swaps: after sweeping a set S the unswept
// block is the Schur complement, so pivot choice is ordinary BK on what is left; after all 64
// indices the working matrix is -inv(T).  Inertia from pivot signs.
//
// Layout: 256 threads; thread (lane i, wave w) keeps columns [16w,16w+16) of row i in REGISTERS.
// Per sweep the pivot row (= pivot column, by symmetry) is broadcast through a double-buffered
// 64-double LDS line; every wave then redundantly runs the same wavefront (DPP) reductions, so the
// decision needs no second exchange: ONE barrier per sweep in the common 1x1 case.
// ---------------------------------------------------------------------------------------------
extern "C" __device__ __attribute__((const)) unsigned long long __ockl_wfred_max_u64(unsigned long long);
extern "C" __device__ __attribute__((const)) double __ockl_wfred_max_f64(double);

__device__ __forceinline__ double wave_max(double v) { return __ockl_wfred_max_f64(v); }

// Loads / stores of data that crosses workgroups INSIDE one launch (k_tile_chain, kernels_chain.hpp): relaxed agent-scope
// atomics (global_load / global_store ... sc1: past the CU's L1, written through the XCD's L2) instead of a fence per
// hand-over -- the rule of the one-launch sweeps (kernels_solve.hpp).  COH = false: the plain access of every other kernel.
template <bool COH>
__device__ __forceinline__ double ldg_c(const double* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH>
__device__ __forceinline__ void stg_c(double* p, double v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

#define PYIPM_BK_ALPHA 0.6403882032022076   /* (1+sqrt(17))/8 */

// Timing experiments on the pivot loop (tools/tile_variants.sh builds extra libraries with -DPYIPM_TILE_EXPERIMENT=bits;
// any bit set gives WRONG results -- the product build never defines it).  1: no separate pivot-lane path; 2: no
// reciprocal; 4: the owner wave does not read its pivot row back from LDS; 8: no statistics.
#ifndef PYIPM_TILE_EXPERIMENT
#define PYIPM_TILE_EXPERIMENT 0
#endif
#define PYIPM_TX_PIVLANE (!(PYIPM_TILE_EXPERIMENT & 1))
#define PYIPM_TX_RECIP(d_) ((PYIPM_TILE_EXPERIMENT & 2) ? (d_) : pivot_recip(d_))

// 1/d for a pivot: v_rcp_f64 and two Newton steps (r += r (1 - d r)), four dependent fused multiply-adds behind a
// quarter-rate instruction instead of the ~35-instruction IEEE division sequence -- which sat on the critical path of
// every one of the N sequential pivots of a factorisation.  Error <= 1 ulp for normal d (the division gives 0.5); pivots
// are never denormal or infinite here (rejected pivots are replaced before this is called, non-finite ones flagged).
__device__ __forceinline__ double pivot_recip(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}

// f32 wave max with the DPP operand fused into v_max_f32 (6 instructions; magnitudes only steer the
// pivot choice, the Bunch-Kaufman inequalities are evaluated on the exact f64 values).
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false)));  // row_shr:1
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false)));  // row_shr:2
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false)));  // row_shr:4
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false)));  // row_shr:8
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false)));  // row_bcast:15
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false)));  // row_bcast:31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
    const long long b = __double_as_longlong(x);
    return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(b >> 32), l) << 32) |
                                (unsigned int)__builtin_amdgcn_readlane((int)b, l));
}

// Magnitude of a static pivot: sqrt(eps) x the largest assembled entry (sqrt(eps) when that is unknown).
// anorm_bits[1] != 0 says the assembly that accumulates anorm_bits[0] is still running on another stream (round 4: the first
// group's chain starts while the columns to its right are still being assembled): whoever needs the scale waits for it -- a
// rare path (a pivot that cancelled), bounded by a timeout of 1 s (100 MHz clock) after which the partial maximum is taken.
__device__ __forceinline__ double static_pivot(const unsigned long long* __restrict__ anorm_bits) {
    if (!anorm_bits) return 1.4901161193847656e-08;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(anorm_bits + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 100000000ull) break;
    }
    const double an = __longlong_as_double((long long)__hip_atomic_load(anorm_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return 1.4901161193847656e-08 * ((an > 0.0 && an <= 1.0e300) ? an : 1.0);
}

// One tile's (or one closed-form panel's) contribution to the statistics of a factorisation.  Atomics (round 5): the closed-form
// panels of the slack block run beside the x block's tile chain; every quantity is order-independent (integer sums, minimum and
// maximum of non-negative doubles through their bit patterns), so the statistics are the same whoever comes first.
__device__ __forceinline__ void stats_add(DevStats* __restrict__ st, long long neg, long long zero, long long n2, long long pos,
                                          long long bad, double dmin, double dmax) {
    if (neg)  atomicAdd(reinterpret_cast<unsigned long long*>(&st->n_neg), (unsigned long long)neg);
    if (zero) atomicAdd(reinterpret_cast<unsigned long long*>(&st->n_zero), (unsigned long long)zero);
    if (n2)   atomicAdd(reinterpret_cast<unsigned long long*>(&st->n_2x2), (unsigned long long)n2);
    if (pos)  atomicAdd(reinterpret_cast<unsigned long long*>(&st->n_pos), (unsigned long long)pos);
    if (bad)  atomicAdd(reinterpret_cast<unsigned long long*>(&st->nonfinite), (unsigned long long)bad);
    if (dmin >= 0.0) atomicMin(reinterpret_cast<unsigned long long*>(&st->d_min), (unsigned long long)__double_as_longlong(dmin));
    if (dmax >= 0.0) atomicMax(reinterpret_cast<unsigned long long*>(&st->d_max), (unsigned long long)__double_as_longlong(dmax));
}

__global__ __launch_bounds__(64) void k_init_stats(DevStats* __restrict__ st) {
    if (threadIdx.x == 0) {
        st->n_neg = 0; st->n_zero = 0; st->n_2x2 = 0; st->n_pos = 0; st->nonfinite = 0;
        st->d_min = 1.0e308; st->d_max = 0.0; st->growth_bits = 0ull;
    }
}

// Shared-memory scratch of one tile inversion (a 256-thread workgroup).
struct TileScratch {
    double stage[TB][TB + 1];
    double colbuf[2][2][TB];     // [parity][0: column p, 1: column r][row]  (column == row by symmetry)
    double sh_red[4];
    double dsave[TB];            // the 1x1 pivots as they were used (static ones replaced): statistics are taken at the end
    union {
        double pcol[16][TB];     // block steps: the 16 pre-sweep pivot columns one wave publishes for the others
        struct {                 // before them (never at the same time: barriers in between)
            BlockedScratch bs;   //   the blocked fast path (tile_blocked.hpp)
            double xmax[4][TB];  //   cross-wave exchange of the row maxima (the stage itself stays intact for the fast path)
        } f;
    };
    int pcnt;                    // ... how many are out so far (polled), -1-n once it stopped after n (general path next)
};
static_assert(sizeof(BlockedScratch) + 4 * TB * sizeof(double) <= 16 * TB * sizeof(double), "fast-path scratch must fit into pcol");

// The sweep inversion itself; called by k_tile_invert (one tile of the big factorisation) and by the
// batched small-system kernel (every tile of one problem, one workgroup per problem).
// W8 (k_tile_step8, 512 threads): waves 4..7 are HELPERS -- they save the tile for the refinement, form the column maxima
// and run the updates / commits of the blocked sweep (tile_blocked8.hpp) beside the critical waves' eliminations, and
// leave the kernel when the blocked sweep ends; ex = its exchange scratch.  Everything from the general loop on runs on
// waves 0..3 as before.
// COH: the tile's outputs (inverse, saved tile, refinement flag) are read by other workgroups of the SAME launch: written
// through (stg_c).  LOOPED: called from inside a persistent loop (tile_tid).
template <bool W8 = false, bool COH = false, bool LOOPED = false>
__device__ __forceinline__ void tile_invert_dev(
    TileScratch& sm,
    const double* __restrict__ A, int64_t ld, int64_t grow0, int64_t lcol0,
    double* __restrict__ Tinv, double* __restrict__ Tsave,     // inv(T); the tile T itself (full, symmetric)
    double* __restrict__ Tflag, double refine_cond,            // *Tflag = 1 when the block solves with this tile need refining
    DevStats* __restrict__ st, int64_t Nreal, double pivtol_rel,
    const unsigned long long* __restrict__ anorm_bits,   // bits of max |assembled entry| (0 = unknown): scale of a static pivot
    int64_t neg_from,                          // global index from which pivots are expected negative (n + mi; pyipm.py:1381)
    unsigned long long* __restrict__ dbg,      // diagnostics only (NULL normally)
    bool from_stage = false,                   // the caller has put the tile into sm.stage[i][j] (i >= j at least): no global read
    bool blocked = true,                       // try the blocked fast path first (tile_blocked.hpp); false: the sweeps of rounds 1-2 only
    Blocked8Scratch* ex = nullptr,             // W8 only
    double (*xkeep)[TB + 2] = nullptr,         // != NULL (k_tile_chain, four waves): inv(T) is also LEFT in shared memory, in the layout
    double* fkeep = nullptr)                   //   PYIPM_STAGE_TILE gives it, and the refinement flag in *fkeep: the caller's next step
{                                              //   reads them from there instead of from memory (xkeep may overlay sm.stage)
    double (&stage)[TB][TB + 1] = sm.stage;
    double (&colbuf)[2][2][TB] = sm.colbuf;
    double (&sh_red)[4] = sm.sh_red;
    unsigned long long dbg_c0 = 0, dbg_w0 = 0;
    if (dbg) { dbg_c0 = clock64(); dbg_w0 = wall_clock64(); }

    constexpr int NT = W8 ? 512 : 256;
    const int tid = tile_tid<LOOPED>(), lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 3, cb = wave * 16;
    const bool helper = W8 && tid >= 256;

    if (!from_stage) {
        // coalesced read of the lower triangle: ALL loads of a thread in flight before the first wait (the rolled loop
        // waited out one memory latency per trip, on the critical path of the whole factorisation)
        double tmp[TB * TB / NT];
        #pragma unroll
        for (int q = 0; q < TB * TB / NT; ++q) {
            const int e = tid + NT * q, i = e & 63, j = e >> 6;
            tmp[q] = (i >= j) ? A[(grow0 + i) + (lcol0 + j) * ld] : 0.0;
        }
        #pragma unroll
        for (int q = 0; q < TB * TB / NT; ++q) {
            const int e = tid + NT * q, i = e & 63, j = e >> 6;
            if (i >= j) stage[i][j] = tmp[q];
        }
    }
    __syncthreads();
    row16_t row;                           // native 16-wide vector: a wave-uniform dynamic index becomes ONE
    double amax = 0.0;                     // relative-addressed move (s_set_gpr_idx), not a 16-deep select chain
    int kb_done8 = 0;
    if (!W8 || helper || !blocked) {
        #pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int j = cb + c;
            row[c] = (lane >= j) ? stage[lane][j] : stage[j][lane];
            if (!W8 || helper) {
                amax = fmax(amax, fabs(row[c]));
                stg_c<COH>(Tsave + j * TB + lane, row[c]); // kept for the refinement of the block solves
            }
        }
    }
    // per-row maximum of the loaded tile (== per-column, by symmetry): the reference scale of a pivot.
    // A pivot counts as rejected only when it has shrunk below pivtol_rel x its OWN column's original
    // magnitude (cancellation), so a badly scaled but perfectly regular tile (Sigma entries spanning
    // 1e-8..1e8 late in an interior-point run) is left alone.
    if (!W8 || helper) {
        sm.f.xmax[wave][lane] = amax;
        amax = wave_max(amax);
        if (lane == 0) sh_red[wave] = amax;
    }
    if (W8) {
        // the critical waves are already eliminating the first micro-block while the helpers do the above
        if (helper) {
            if (tid == 256) sm.f.bs.fail = 0;
            if (blocked) {
                if (tile_blocked8_sweep_help(stage, sm.f.bs, *ex)) {
                    // all four blocks swept: the general loop has nothing to do, the inverse goes out on eight waves
                    #pragma unroll
                    for (int c = 8; c < 16; ++c) {
                        const int j = cb + c;
                        Tinv[j * TB + lane] = -((lane >= j) ? stage[lane][j] : stage[j][lane]);
                    }
                }
            } else __syncthreads();
            return;                                    // the helpers are done (a barrier counts the waves that are left)
        }
        if (blocked) {
            if (dbg && tid == 0) dbg[4] = clock64() - dbg_c0;
            kb_done8 = tile_blocked8_sweep_crit(stage, sm.f.bs, *ex, sm.dsave, sm.f.xmax, pivtol_rel, dbg);
            if (dbg && tid == 0) { dbg[5] = clock64() - dbg_c0; dbg[6] = dbg_c0; dbg[7] += (unsigned long long)kb_done8; }
        } else __syncthreads();
    } else {
        __syncthreads();
    }
    const double cmax0 = fmax(fmax(sm.f.xmax[0][lane], sm.f.xmax[1][lane]), fmax(sm.f.xmax[2][lane], sm.f.xmax[3][lane]));
    const double scale = fmax(fmax(sh_red[0], sh_red[1]), fmax(sh_red[2], sh_red[3]));
    const double inv_scale = scale > 0.0 ? 1.0 / scale : 0.0;
    // Static pivot (GESP, as in SuperLU_DIST): a pivot that BK cannot avoid inside the tile and that has cancelled to
    // nothing is REPLACED by +-sqrt(eps)*max|A| with the sign its KKT block calls for, and reported (n_zero).  The
    // factor is then that of a matrix perturbed by ~sqrt(eps)|A| in those diagonal entries: finite, the same inertia
    // as the unperturbed matrix whenever that one is non-singular, and an excellent preconditioner -- the host
    // recovers the unperturbed solution by refinement against the KKT blocks (HipNewtonBackend.direction), or
    // regularises as reghess does when that does not converge (pyipm.py:1379-1403).
    auto pert_f = [&]() { return static_pivot(anorm_bits); };      // read only where a pivot is rejected (it may have to wait, see static_pivot)
    const double ptol = pivtol_rel * cmax0;             // this lane's rejection threshold, should it become the pivot
    const int neg_lim = (int)((neg_from - grow0) < 0 ? 0 : ((neg_from - grow0) > TB ? TB : (neg_from - grow0)));   // pivots >= this: expected negative
    const int nreal = (int)((Nreal - grow0) < 0 ? 0 : ((Nreal - grow0) > TB ? TB : (Nreal - grow0)));   // real rows in this tile

    // ---- blocked fast path: 16 pivots at a time in natural order while Bunch-Kaufman would have accepted them ----
    int kb_done = kb_done8;
    const int c_out = (W8 && kb_done8 == 4) ? 8 : 16;  // (all four blocks swept: the helpers write the other columns of the inverse)
    if (W8 && blocked) {
        #pragma unroll
        for (int c = 0; c < 16; ++c) {                 // the working matrix as it stands, in the sweep layout
            const int j = cb + c;
            if (c < c_out) row[c] = (lane >= j) ? stage[lane][j] : stage[j][lane];
        }
    }
    if (!W8 && blocked) {
        if (wave == 0) sm.f.bs.ptol[lane] = ptol;
        if (tid == 0) sm.f.bs.fail = 0;
        __syncthreads();
        if (dbg && tid == 0) dbg[4] = clock64() - dbg_c0;
        kb_done = tile_blocked_sweep<TB + 1, LOOPED>(stage, sm.f.bs, sm.dsave, dbg);
        if (dbg && tid == 0) { dbg[5] = clock64() - dbg_c0; dbg[6] = dbg_c0; dbg[7] += (unsigned long long)kb_done; }   // (micro-blocks committed, summed over tiles)
        #pragma unroll
        for (int c = 0; c < 16; ++c) {                 // the working matrix as it stands, back in the sweep layout
            const int j = cb + c;
            row[c] = (lane >= j) ? stage[lane][j] : stage[j][lane];
        }
    }
    unsigned long long mask = kb_done >= 4 ? 0ull : (~0ull << (16 * kb_done));   // unswept set (identical in every thread)
    int left = TB - 16 * kb_done, parity = 0;
    int neg = 0, zero = 0, n2 = 0, bad = 0;         // (2x2 pivots count here directly; 1x1 pivots at the end, from dsave)
    double dmin = 1.0e308, dmax = 0.0;
    unsigned long long zmask = 0ull, m2mask = 0ull; // static 1x1 pivots; indices taken in 2x2 pivots
    // (Tried and dropped, r02: "scaled rows" -- every later sweep acts linearly on a swept row, so the pivot lane can keep its
    //  row unscaled, stay out of its own sweep's update and convert at the boundaries (publish entry x 1/d, scale once at the
    //  end): one multiply per pivot instead of sixteen in a divergent branch, 996 -> 883 cycles per pivot.  But the replacement
    //  row p := column p / d is also what RE-SYMMETRISES the working matrix at every pivot; without it the rounding asymmetry
    //  between a row and its column accumulates, and dense tiles of condition 1e11 lose their inertia
    //  (tests/test_gpu_condensed.py::test_ill_conditioned_tiles_keep_inertia).  Robustness kept, 10 % not taken.)

    // publish column `col` of the tile (== row `col`) from the wave that holds it: one full-wave store
#define PYIPM_PUBLISH(col_, dst_)                                                            \
    if (((col_) >> 4) == wave) {                     /* wave-uniform */                       \
        (dst_)[lane] = row[__builtin_amdgcn_readfirstlane((col_) & 15)];                     \
    }

    // One symmetric sweep on the 1x1 pivot pv_ whose (pre-sweep) column is cpi_ (this lane's entry), cpj_ (the
    // entries at this wave's 16 rows) and dpp_ (the diagonal).  Textually shared by both phases below.
#define PYIPM_SWEEP1(pv_, cpi_, cpj_, dpp_)                                                              \
    {                                                                                                    \
        double d = (dpp_);                                                                               \
        /* rejected?  lane pv_ compares its own entry with its own threshold (one vector compare, no broadcasts) */ \
        if (__builtin_expect(__ballot(lane == (pv_) && fabs(cpi_) <= ptol) != 0ull, 0)) {                \
            const double pivtol = readlane_f64(ptol, (pv_));                                             \
            const double pert = pert_f(); const double t = pivtol > pert ? pivtol : pert;                                              \
            d = ((pv_) < neg_lim) ? t : -t;                      /* static pivot, expected sign */       \
            zmask |= 1ull << (pv_);                                                                      \
        }                                                                                                \
        /* sign / magnitude / finiteness of the pivot are counted after the last sweep (10 % of the loop's      \
           instructions were statistics on wave-uniform values); a wave that is not the column's owner notes it */ \
        if (!(PYIPM_TILE_EXPERIMENT & 8) && wave == ((((pv_) >> 4) + 1) & 3) && lane == 0) sm.dsave[(pv_)] = d; \
        const double inv_d = PYIPM_TX_RECIP(d);                                                          \
        const double lpi = (cpi_) * inv_d;                                                               \
        if (PYIPM_TX_PIVLANE && lane == (pv_)) {                 /* one lane: row p <- cp/d */            \
            _Pragma("unroll") for (int c = 0; c < 16; ++c) row[c] = (cpj_)[c] * inv_d;                   \
        } else {                                                                                         \
            _Pragma("unroll") for (int c = 0; c < 16; ++c) row[c] = fma(-lpi, (cpj_)[c], row[c]);        \
        }                                                                                                \
        if (wave == ((pv_) >> 4))                                /* wave-uniform: column p <- lp, pivot <- -1/d */ \
            row[__builtin_amdgcn_readfirstlane((pv_) & 15)] = (lane == (pv_)) ? -inv_d : lpi;            \
        mask &= ~(1ull << (pv_));                                                                        \
        left -= 1;                                                                                       \
    }

    // (Tried and dropped, r02: the block-step pivots in PAIRS -- the second pivot column after the first sweep is one fused
    //  multiply-add per lane, so both columns are published together and one LDS round trip returns both pivot rows, the second
    //  pivot's test / reciprocal / multipliers computed while the reads are in flight.  Bit-identical, and SLOWER: 1436 against
    //  1109 cycles per pivot.  The owner wave's time is its instruction count on one dependent path, not the LDS round trip.)
    // ---- block steps --------------------------------------------------------------------------------------
    // As long as Bunch-Kaufman accepts the diagonal pivots in their natural order (every definite tile; most
    // others), a sweep only needs the pivot's own column.  The wave that holds columns [16k, 16k+16) therefore
    // runs those 16 sweeps on ITS columns with no barrier, publishing each pre-sweep column as it goes; the other
    // waves poll the count and replay each sweep on their columns right behind it.  Same operations on every
    // entry in the same order as the general loop below (bitwise the same inverse) with one barrier per 16
    // pivots instead of one per pivot.  The first pivot that fails the acceptance test ends this phase; the
    // general loop takes over from exactly that state.
    // (the count goes through explicit LDS instructions: a volatile access through the scratch reference would be
    //  compiled to a flat store + vmcnt wait per pivot)
    const unsigned pcnt_lds = (unsigned)(size_t)&sm.pcnt;
    auto pcnt_store = [&](int v) { asm volatile("ds_write_b32 %0, %1" :: "v"(pcnt_lds), "v"(v) : "memory"); };
    auto pcnt_load = [&]() { int v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(pcnt_lds) : "memory");
                             return __builtin_amdgcn_readfirstlane(v); };
    for (int k = kb_done; k < 4 && left > 0; ++k) {
        if (tid == 0) sm.pcnt = 0;
        __syncthreads();
        int n = 16;
        if (wave == k) {
            for (int j = 0; j < 16; ++j) {
                const int pv = 16 * k + j;
                const double cpi = row[j];                            // (j is wave-uniform: relative-addressed read)
                const double dpp = readlane_f64(cpi, pv);
                if (left > 1 && __ballot(lane > pv && fabs(cpi) * PYIPM_BK_ALPHA > fabs(dpp)) != 0ull) { n = j; break; }
                sm.pcol[j][lane] = cpi;
                if (lane == 0) pcnt_store(j + 1);                     // LDS keeps a wave's stores in order
                double cpj[16];                                       // own rows of the column: read back (8 wide reads
                #pragma unroll                                        // beat 32 readlanes)
                for (int c = 0; c < 16; ++c) cpj[c] = (PYIPM_TILE_EXPERIMENT & 4) ? cpi + c : sm.pcol[j][cb + c];
                PYIPM_SWEEP1(pv, cpi, cpj, dpp)
            }
            if (n < 16 && lane == 0) pcnt_store(-1 - n);
        } else {
            for (int j = 0; j < 16; ++j) {
                int c;
                while ((c = pcnt_load()) >= 0 && c <= j) __builtin_amdgcn_s_sleep(1);
                if (c < 0 && -1 - c <= j) { n = -1 - c; break; }
                const int pv = 16 * k + j;
                const double* rp = sm.pcol[j];
                const double cpi = rp[lane];
                double cpj[16];
                #pragma unroll
                for (int cc = 0; cc < 16; ++cc) cpj[cc] = rp[cb + cc];
                const double dpp = readlane_f64(cpi, pv);
                PYIPM_SWEEP1(pv, cpi, cpj, dpp)
            }
        }
        __syncthreads();                                              // everyone is done with pcol
        if (n < 16) break;                                            // (uniform: every wave saw the same stop count)
    }

    // Loop invariant: the BK candidate column p (first unswept index) has been published into
    // colbuf[parity][0] by the previous iteration (or the prologue), so an iteration starts at the barrier.
    int p = left > 0 ? __ffsll(mask) - 1 : 0, sel = 0;
    bool forced = false;                    // pivot already decided (the off-diagonal candidate r): its column
                                            // sits in colbuf[parity][1], already synchronised
    if (left > 0) { PYIPM_PUBLISH(p, colbuf[0][0]) }
    while (left > 0) {
        if (!forced) __syncthreads();
        const double* rp = colbuf[parity][sel];
        const double cpi = rp[lane];        // B[lane][p] (= B[p][lane])
        double cpj[16];                     // this wave's 16 entries of row p
        #pragma unroll
        for (int c = 0; c < 16; ++c) cpj[c] = rp[cb + c];
        const double dpp = readlane_f64(cpi, p);
        int kind = 1, q = -1;
        if (!forced) {
            const bool u = (mask >> lane) & 1ull;
            const double app = fabs(dpp);
            // Bunch-Kaufman acceptance |B[p][p]| >= alpha * max_i |B[i][p]| as ONE wave-wide compare: the
            // maximum itself (a 6-stage DPP reduction + argmax) is only needed when the test fails.
            if (left > 1 && __ballot(u && lane != p && fabs(cpi) * PYIPM_BK_ALPHA > app) != 0ull) {
                // r = argmax_{i unswept, i != p} |B[i][p]| chosen on f32 magnitudes scaled by the tile maximum
                const float magf = (u && lane != p) ? (float)(fabs(cpi) * inv_scale) : -1.0f;
                const float mx = wave_max_f32(magf);
                const int r = __builtin_amdgcn_readfirstlane(__ffsll(__ballot(magf == mx)) - 1);
                const double lam = fabs(readlane_f64(cpi, r));
                PYIPM_PUBLISH(r, colbuf[parity][1])
                __syncthreads();
                const double* rr = colbuf[parity][1];
                const double sigma = wave_max((u && lane != r) ? fabs(rr[lane]) : -1.0);
                const double arr = fabs(rr[r]);
                if (app * sigma >= PYIPM_BK_ALPHA * lam * lam) { /* 1x1 pivot p */ }
                else if (arr >= PYIPM_BK_ALPHA * sigma)        { forced = true; p = r; sel = 1; continue; }   // 1x1 pivot r
                else                                            { kind = 2; q = r; }
            }
        }
        if (kind == 1) {
            PYIPM_SWEEP1(p, cpi, cpj, dpp)
        } else {
            const double* rq = colbuf[parity][1];
            const double a = rp[p], b = rp[q], cc = rq[q];
            double det = a * cc - b * b;                           // < 0 by the BK test
            if (!(fabs(det) <= 1.0e308)) bad = 1;
            const double tr = a + cc, disc = sqrt((a - cc) * (a - cc) + 4.0 * b * b);
            const double e1 = 0.5 * (tr + disc), e2 = 0.5 * (tr - disc);
            const double pivtol = pivtol_rel * fmax(readlane_f64(cmax0, p), readlane_f64(cmax0, q));
            n2++; m2mask |= (1ull << p) | (1ull << q);
            // a 2x2 block pivot has det < 0 by the BK test: one positive, one negative eigenvalue (e1 >= 0 >= e2)
            neg += 1;
            if (fabs(e1) <= pivtol) zero++; else { dmin = fmin(dmin, fabs(e1)); dmax = fmax(dmax, fabs(e1)); }
            if (fabs(e2) <= pivtol) zero++; else { dmin = fmin(dmin, fabs(e2)); dmax = fmax(dmax, fabs(e2)); }
            if (!(fabs(det) > 0.0)) { const double pert = pert_f(); det = -pert * pert; }
            const double ia = cc / det, ib = -b / det, ic = a / det;   // inverse of [[a,b],[b,cc]]
            const double vp = rp[lane], vq = rq[lane];
            const double lpi = vp * ia + vq * ib, lqi = vp * ib + vq * ic;
            const bool isp = lane == p, isq = lane == q;
            #pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int j = cb + c;
                const double cpv = rp[j], cqv = rq[j];
                double v = fma(-lqi, cqv, fma(-lpi, cpv, row[c]));
                if (j == p) v = lpi;
                if (j == q) v = lqi;
                if (isp) v = (j == p) ? -ia : ((j == q) ? -ib : cpv * ia + cqv * ib);
                if (isq) v = (j == p) ? -ib : ((j == q) ? -ic : cpv * ib + cqv * ic);
                row[c] = v;
            }
            mask &= ~((1ull << p) | (1ull << q));
            left -= 2;
        }
        forced = false; sel = 0;
        if (left > 0) {
            p = __ffsll(mask) - 1;          // next BK candidate: first unswept index (wave-uniform, scalar)
            PYIPM_PUBLISH(p, colbuf[parity ^ 1][0])
        }
        parity ^= 1;
    }

#undef PYIPM_PUBLISH
#undef PYIPM_SWEEP1
    #pragma unroll
    for (int c = 0; c < 16; ++c) if (c < c_out) stg_c<COH>(Tinv + (cb + c) * TB + lane, -row[c]);
    if (dbg && tid == 0) { dbg[0] = clock64() - dbg_c0; dbg[1] = wall_clock64() - dbg_w0; dbg[2] += 1; }
    __syncthreads();                                     // dsave complete
    if (wave == 0) {                                     // statistics of the 1x1 pivots: lane p looks at pivot p
        const double d = sm.dsave[lane], ad = fabs(d);
        const bool is1 = ((m2mask >> lane) & 1ull) == 0ull, real = lane < nreal, stat = ((zmask >> lane) & 1ull) != 0ull;
        if (__ballot(is1 && !(ad <= 1.0e308)) != 0ull) bad = 1;                       // NaN or Inf
        neg += __popcll(__ballot(is1 && real && d < 0.0));                           // static pivots count by their sign
        zero += __popcll(__ballot(is1 && real && stat));
        dmin = fmin(dmin, -wave_max((is1 && real && !stat) ? -ad : -1.0e308));
        dmax = fmax(dmax, wave_max((is1 && real && !stat) ? ad : 0.0));
    }
    if (xkeep) {                                         // (behind the barrier: nobody reads the stage any more)
        #pragma unroll
        for (int c = 0; c < 16; ++c) xkeep[cb + c][lane] = -row[c];
    }
    if (tid == 0) {
        // pivot spread of THIS tile ~ cond(T): the explicit inverse is accurate to cond*eps, so only tiles
        // beyond refine_cond (or with 2x2 pivots) pay for refined block solves.  A tile with a rejected pivot
        // is singular to working precision: its "inverse" belongs to a perturbed tile, nothing to refine against.
        const double fl = (zero == 0 && (n2 > 0 || !(dmax <= refine_cond * dmin))) ? 1.0 : 0.0;
        stg_c<COH>(Tflag, fl);
        if (fkeep) *fkeep = fl;
        stats_add(st, neg, zero, n2, nreal - neg, bad, dmin, dmax);                          // static pivots count by their sign
    }
}

__global__ __launch_bounds__(256) void k_tile_invert(
    const double* __restrict__ A, int64_t ld, int64_t grow0, int64_t lcol0,
    double* __restrict__ Tinv, double* __restrict__ Tsave, double* __restrict__ Tflag, double refine_cond,
    DevStats* __restrict__ st, int64_t Nreal, double pivtol_rel,
    const unsigned long long* __restrict__ anorm_bits, int64_t neg_from, unsigned long long* __restrict__ dbg,
    double* __restrict__ Wnext,            // != NULL: -S of the NEXT diagonal tile's rows in this column block (64 x 64,
                                           // column-major, ld 64): the fused scaling + update launch that follows lets
                                           // every strip read it while the strip that owns those rows overwrites them with L
    int blocked)                           // fast path of tile_blocked.hpp on / off
{
    __shared__ TileScratch sm;
    __builtin_amdgcn_s_setprio(3);         // latency-critical chain: win issue arbitration against co-resident bulk waves
    if (Wnext) {
        double tmp[TB * TB / 256];
        #pragma unroll
        for (int q = 0; q < TB * TB / 256; ++q) {
            const int e = threadIdx.x + 256 * q;
            tmp[q] = A[(grow0 + TB + (e & 63)) + (lcol0 + (e >> 6)) * ld];
        }
        #pragma unroll
        for (int q = 0; q < TB * TB / 256; ++q) Wnext[threadIdx.x + 256 * q] = -tmp[q];
    }
    tile_invert_dev(sm, A, ld, grow0, lcol0, Tinv, Tsave, Tflag, refine_cond, st, Nreal, pivtol_rel, anorm_bits, neg_from, dbg,
                    false, blocked != 0);
}

// 64x64 tile (global, row-major [k][c]) -> LDS array dst_[k][c], scaled: the 16 loads of a thread are issued together
// (a rolled loop costs one memory latency per trip -- 16 of them were the whole run time of k_panel_scale).
#define PYIPM_STAGE_TILE(dst_, scale_, src_) PYIPM_STAGE_TILE_C(dst_, scale_, src_, false)
#define PYIPM_STAGE_TILE_C(dst_, scale_, src_, coh_)                                                  \
    {                                                                                                 \
        double stg_[TB * TB / 256];                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < TB * TB / 256; ++q_) stg_[q_] = ldg_c<coh_>((src_) + tid + 256 * q_); \
        _Pragma("unroll") for (int q_ = 0; q_ < TB * TB / 256; ++q_) {                                \
            const int e_ = tid + 256 * q_;                                                            \
            (dst_)[e_ >> 6][e_ & 63] = (scale_) * stg_[q_];                                           \
        }                                                                                             \
    }

// ---------------------------------------------------------------------------------------------
// Panel scaling: for rows below a factored tile, keep the Schur-complemented block as W and
// overwrite it with the block factor L = W * inv(T).  256 threads per 64 rows: wave w produces
// output columns [16w,16w+16); inv(T) in LDS (broadcast reads).
// ---------------------------------------------------------------------------------------------
// The in-panel update of the NEXT 64-column block, fused into the scaling launch of the current one (both are
// row-parallel over the same 64-row strips, and a strip's own L of the current tile is still in its registers):
//   C[i][c] += sum_{k < K + 64} L[i][k] * Wn[c][k],   Wn = -S of the next diagonal tile's rows,
// the same products in the same order as k_inpanel_update (k ascending in MFMA groups of 4): bit-identical.  The last 64
// columns of Wn (this tile's) come from `Wnext`, copied by the tile kernel before this launch overwrites them with L.
struct NextUpd {
    int on;                                // 0: plain scaling
    double* C; int64_t ldc, ccol;          // next column block: C[i + (ccol + c) * ldc]
    const double* Lop; int64_t ldl;        // the panel's columns before this tile: L[i][k] = Lop[i + k*ldl], k < K
    const double* Wop; int64_t ldw, cglob; // Wn[c][k] = Wop[(cglob + c) + k*ldw], k < K (cglob = global row of the next tile)
    const double* Wnext;                   // Wn[c][K + k] = Wnext[c + k*64]
    int K;                                 // 64 t
};

__global__ __launch_bounds__(256) void k_panel_scale(
    double* __restrict__ Aout, int64_t ld_out, int64_t col_out,      // L written at Aout[i + (col_out+c)*ld_out]
    const double* __restrict__ Win, int64_t ld_in, int64_t col_in,   // S read from  Win[i + (col_in+k)*ld_in]
    double* __restrict__ Wcopy, int64_t ld_w, int64_t col_w,         // copy of -S (may be NULL = no copy)
    const double* __restrict__ Tinv, const double* __restrict__ Tsave, const double* __restrict__ Tflag, int nref,
    int64_t row_begin, int64_t hole0, int64_t hole1,   // 64-row blocks inside [hole0, hole1) hold exact zeros: skipped
    unsigned long long* __restrict__ growth_bits,
    double sign,      // owner: Win = S, Wcopy = -S (the update kernel wants -W), sign = +1;
                      // non-owner rebuilding L from a received -S: sign = -1
    NextUpd nu)
{
    // L[i][c] = sign * sum_k S[i][k] X[k][c], X = inv(T), on fp64 MFMA.  D[m][n]: m <- c (A operand = X,
    // symmetric), n <- i (B operand = S, read straight from global: 16 lanes x 8 B contiguous per k).  One
    // wave owns 16 rows x 64 columns (4 accumulator tiles); a 256-thread block covers 64 rows.
    //
    // X is an explicit inverse, so L0 = S X solves L T = S only to cond(T)*eps.  nref steps of
    //     R = S - L T ;  L += R X
    // make each row of L a backward-stable solution (residual ~ eps |L||T|), which is what keeps the Schur
    // complement -- and the inertia read off it -- right for ill-conditioned dense tiles.  The f64 C/D map
    // (row = (lane>>4)+4r of tile t, col = lane&15) puts L[i][16t+4r+(lane>>4)] in accumulator [t][r], which
    // is exactly the B-operand layout of k-step 4t+r: the refinement GEMMs need no data movement.
    __shared__ double X[TB][TB + 2];        // holds sign*inv(T), then -sign*T, then sign*inv(T) again (one array: LDS decides
    // gridDim.y > 1 (a receiver rebuilding L of a whole panel, pyipm_newton_panel_unpack): tile column blockIdx.y of the panel --
    // the tile columns are independent there (L_t = W_t inv(T_t)), one launch instead of nb / 64 in a row (round 4)
    if (gridDim.y > 1) {
        const int64_t t = blockIdx.y;
        col_out += t * TB; col_in += t * TB; col_w += t * TB;
        Tinv += t * (int64_t)(TB * TB); Tsave += t * (int64_t)(TB * TB); Tflag += t;
    }
    {                                       // how many of these short blocks fit into the slot a retiring update block frees)
        const int64_t r0 = row_begin + (int64_t)blockIdx.x * TB;
        if (r0 >= hole0 && r0 + TB <= hole1) return;          // S rows identically zero here (KKT structure): L = 0 already
    }
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    if (nref > 0 && *Tflag == 0.0) nref = 0;            // well-conditioned tile: the plain product is accurate (block-uniform)
    PYIPM_STAGE_TILE(X, sign, Tinv)        // symmetric
    const int64_t i = row_begin + (int64_t)blockIdx.x * TB + wave * 16 + l15;
    double b[16];
    #pragma unroll
    for (int ks = 0; ks < 16; ++ks) b[ks] = Win[i + (col_in + ks * 4 + l4) * ld_in];
    if (Wcopy) {                                         // -S: the owner loaded S (sign = +1), a receiver -S itself (sign = -1; a product by +-1 is exact)
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) Wcopy[i + (col_w + ks * 4 + l4) * ld_w] = -sign * b[ks];
    }
    __syncthreads();
    double4_t acc[4];
    #pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    #pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        #pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double a = X[t * 16 + l15][ks * 4 + l4];          // A[m = c][k] = X[c][k]
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[ks], acc[t], 0, 0, 0);
        }
    }
    for (int it = 0; it < nref; ++it) {
        __syncthreads();
        PYIPM_STAGE_TILE(X, -sign, Tsave)
        __syncthreads();
        // R = S - L T  [D layout]: C operand = S = sign * the loaded b values, in accumulator order
        double4_t res[4];
        #pragma unroll
        for (int t = 0; t < 4; ++t)
            #pragma unroll
            for (int r = 0; r < 4; ++r) res[t][r] = sign * b[4 * t + r];
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double lop = acc[ks >> 2][ks & 3];                 // B[k][n = i] = L[i][k]
            #pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double a = X[t * 16 + l15][ks * 4 + l4];       // -sign*T[c][k], B = sign*L  ->  -T L
                res[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sign * lop, res[t], 0, 0, 0);
            }
        }
        __syncthreads();
        PYIPM_STAGE_TILE(X, sign, Tinv)
        __syncthreads();
        // L += R inv(T)   (A = sign*inv(T), B = sign*R)
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const double rop = sign * res[ks >> 2][ks & 3];
            #pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double a = X[t * 16 + l15][ks * 4 + l4];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rop, acc[t], 0, 0, 0);
            }
        }
    }
    double gmax = 0.0;
    #pragma unroll
    for (int t = 0; t < 4; ++t)
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = t * 16 + l4 + 4 * r;                       // C/D map: row = (lane>>4)+4r -> c, col = lane&15 -> i
            Aout[i + (col_out + c) * ld_out] = acc[t][r];
            gmax = fmax(gmax, fabs(acc[t][r]));
        }
    gmax = wave_max(gmax);
    if (lane == 0 && growth_bits) atomicMax(growth_bits, (unsigned long long)__double_as_longlong(gmax));
    if (!nu.on) return;
    // ---- fused: in-panel update of the next column block for this strip (wave: 16 rows x 64 columns) ----
    double4_t c2[4];
    #pragma unroll
    for (int t = 0; t < 4; ++t)
        #pragma unroll
        for (int r = 0; r < 4; ++r) c2[t][r] = nu.C[i + (nu.ccol + t * 16 + l4 + 4 * r) * nu.ldc];
    {
        const double* wp = nu.Wop + (nu.cglob + l15) + (int64_t)l4 * nu.ldw;      // A operand: m = c (per 16-group t), k = 4 ks + l4
        const double* lp = nu.Lop + i + (int64_t)l4 * nu.ldl;                      // B operand: n = i
        for (int k0 = 0; k0 < nu.K; k0 += TB / 2) {          // 32 columns of K in flight (registers: 2-3 such waves per SIMD)
            double wa[4][8], lb[8];
            #pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                lb[ks] = lp[(int64_t)(k0 + 4 * ks) * nu.ldl];
                #pragma unroll
                for (int t = 0; t < 4; ++t) wa[t][ks] = wp[16 * t + (int64_t)(k0 + 4 * ks) * nu.ldw];
            }
            #pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                #pragma unroll
                for (int t = 0; t < 4; ++t)
                    c2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[t][ks], lb[ks], c2[t], 0, 0, 0);
        }
        // last slab: this tile's columns -- Wn from the tile kernel's copy, L from the accumulators (their C/D map is
        // the B-operand map of k-step 4 t + r, as in the refinement above)
        #pragma unroll
        for (int h = 0; h < 2; ++h) {
            double wa[4][8];
            #pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                #pragma unroll
                for (int t = 0; t < 4; ++t) wa[t][ks] = nu.Wnext[(16 * t + l15) + (4 * (8 * h + ks) + l4) * TB];
            #pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const double lop = acc[(8 * h + ks) >> 2][(8 * h + ks) & 3];
                #pragma unroll
                for (int t = 0; t < 4; ++t)
                    c2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[t][ks], lop, c2[t], 0, 0, 0);
            }
        }
    }
    #pragma unroll
    for (int t = 0; t < 4; ++t)
        #pragma unroll
        for (int r = 0; r < 4; ++r) nu.C[i + (nu.ccol + t * 16 + l4 + 4 * r) * nu.ldc] = c2[t][r];
}

// ---------------------------------------------------------------------------------------------
// Panels that lie inside the slack block.  There the KKT matrix is known in closed form when elimination
// reaches it: the (s,s) block is still diag(Sigma) (the (s,x) block is zero, so the x columns never touched
// it), the only other non-zeros of an s column are the -1 towards its own lambda_i row, and an s column
// updates nothing but that lambda_i's diagonal entry.  One small kernel does for such a panel what the tile
// inversions, scalings and updates would do on 99.9 % zeros -- with the same operations in the same order
// (1/d by division, the refinement recurrences of flagged tiles as fused multiply-adds), so the factor, the
// statistics and the directions are bit for bit those of the dense path.
// One block, 256 threads: wave w takes tiles w, w+4, ...; lane k owns column 64 t + k of the panel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_s_panel(
    double* __restrict__ A, int64_t ld, int64_t c0, int64_t lc0, int nt,       // global / local first column of the panel
    double* __restrict__ Tinv, double* __restrict__ Tsave, double* __restrict__ Tflag,     // of the panel's first tile
    double refine_cond, int nref, DevStats* __restrict__ st, int64_t s0, int64_t i0, double pivtol_rel,
    const unsigned long long* __restrict__ anorm_bits)
{
    __shared__ long long sh_cnt[4][3];      // neg, zero, bad per wave
    __shared__ double sh_mm[4][3];          // dmin, dmax, gmax per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long neg = 0, zero = 0, bad = 0;
    double dmin = 1.0e308, dmax = 0.0, gmax = 0.0;
    const double spert = static_pivot(anorm_bits);         // static pivot of a Sigma entry that is exactly zero (lambda_i = 0),
                                                            // as tile_invert_dev would place it (its tile column is zero)
    for (int t = wave; t < nt; t += 4) {
        const int64_t col = c0 + (int64_t)t * TB + lane;            // global column = row of its diagonal entry
        const int64_t lcol = lc0 + (int64_t)t * TB + lane;          // where this rank stores it
        const int64_t irow = i0 + (col - s0);                        // its lambda_i row
        double d = A[col + lcol * ld];
        const double ad = fabs(d);
        const bool isbad = !(ad <= 1.0e308);
        const bool iszero = ad <= pivtol_rel * ad;                   // the tile-local column maximum is |d| itself
        if (iszero) d = spert;
        const double x = pivot_recip(d);                             // (as the sweep of the dense path computes it)
        // tile statistics (as k_tile_invert keeps them)
        const unsigned long long mz = __ballot(iszero), mb = __ballot(isbad), mn = __ballot(!iszero && d < 0.0);
        const double tmin = -wave_max(iszero ? -1.0e308 : -ad), tmax = wave_max(iszero ? 0.0 : ad);
        const int nz = __popcll(mz);
        const bool flagged = nz == 0 && !(tmax <= refine_cond * tmin);
        zero += nz; bad += __popcll(mb); neg += __popcll(mn);
        dmin = fmin(dmin, tmin); dmax = fmax(dmax, tmax);
        // inv(T) and T as dense 64x64 tiles (the substitutions and a receiver's rebuild read them densely)
        double* Xi = Tinv + (int64_t)t * TB * TB;
        double* Tt = Tsave + (int64_t)t * TB * TB;
        const double dorig = A[col + lcol * ld];
        #pragma unroll 8
        for (int r = 0; r < TB; ++r) {                              // diagonal tiles: row r written by the whole wave (one
            Xi[r * TB + lane] = (r == lane) ? x : 0.0;              // contiguous 512 B store per row; lane-major indexing
            Tt[r * TB + lane] = (r == lane) ? dorig : 0.0;          // scattered every store over 64 cache lines)
        }
        if (lane == 0) Tflag[t] = flagged ? 1.0 : 0.0;
        // L(lambda_i row, this column) = S X with S = -1, refined like k_panel_scale refines a flagged tile
        const double sv = A[irow + lcol * ld];                      // -1 (pyipm.py:838-842)
        double l = fma(x, sv, 0.0);
        if (flagged)
            for (int it = 0; it < nref; ++it) {
                const double r = fma(-A[col + lcol * ld], l, sv);
                l = fma(x, r, l);
            }
        A[irow + lcol * ld] = l;                                    // (its Schur update: k_s_schur, on the update stream)
        gmax = fmax(gmax, fabs(l));
    }
    gmax = wave_max(gmax);
    if (lane == 0) {
        sh_cnt[wave][0] = neg; sh_cnt[wave][1] = zero; sh_cnt[wave][2] = bad;
        sh_mm[wave][0] = dmin; sh_mm[wave][1] = dmax; sh_mm[wave][2] = gmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long n = 0, z = 0, b = 0; double mn = 1.0e308, mx = 0.0, gm = 0.0;
        for (int w = 0; w < 4; ++w) { n += sh_cnt[w][0]; z += sh_cnt[w][1]; b += sh_cnt[w][2];
                                      mn = fmin(mn, sh_mm[w][0]); mx = fmax(mx, sh_mm[w][1]); gm = fmax(gm, sh_mm[w][2]); }
        stats_add(st, n, z, 0, (long long)nt * TB - n, b, mn, mx);                               // static pivots are positive
        atomicMax(&st->growth_bits, (unsigned long long)__double_as_longlong(gm));
    }
}

// The trailing update of slack-block source columns [c0, c0+ncols): column c touches exactly one entry, the
// diagonal of its lambda_i row, C += L * (-S) with -S = 1.  Restricted to target columns [tc0, tc1) so that
// it can stand in for the head / bulk launches of the group (same place in the order of operations).
__global__ __launch_bounds__(256) void k_s_schur(double* __restrict__ A, int64_t ld, int64_t c0, int64_t ncols,
                                                 int64_t s0, int64_t i0, int64_t tc0, int64_t tc1)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= ncols) return;
    const int64_t col = c0 + k, irow = i0 + (col - s0);
    if (irow < tc0 || irow >= tc1) return;
    A[irow + irow * ld] = fma(A[irow + col * ld], 1.0, A[irow + irow * ld]);
}

// Same update when the source columns live on another rank (or no group schedule exists): the one non-zero of
// L is recomputed from the staged s / lambda_i exactly as k_s_panel computes it (d is the assembled
// lambda_i/(s+eps), pyipm.py:498), so no message is needed for a slack-block panel at all.  One wave per
// source tile; a rank touches the diagonal entries of the lambda_i columns it owns inside global panels
// [tp0, tp1).
__global__ __launch_bounds__(64) void k_s_schur_sigma(
    double* __restrict__ A, int64_t ld, int64_t c0, int64_t s0, int64_t i0,
    const double* __restrict__ s, const double* __restrict__ lda_i, double eps,
    double refine_cond, int nref, double pivtol_rel, int nb, int world, int rank, int64_t tp0, int64_t tp1,
    const unsigned long long* __restrict__ anorm_bits)     // (ranks agree on it: DistNewton reduces it after assembly)
{
    const int lane = threadIdx.x;
    const int64_t col = c0 + (int64_t)blockIdx.x * TB + lane, b = col - s0;
    const double d0 = lda_i[b] / (s[b] + eps);
    double d = d0;
    const double ad = fabs(d), spert = static_pivot(anorm_bits);
    const bool iszero = ad <= pivtol_rel * ad;
    if (iszero) d = spert;
    const double x = pivot_recip(d);
    const double tmin = -wave_max(iszero ? -1.0e308 : -ad), tmax = wave_max(iszero ? 0.0 : ad);
    const bool flagged = __popcll(__ballot(iszero)) == 0 && !(tmax <= refine_cond * tmin);
    const double sv = -1.0;
    double l = fma(x, sv, 0.0);
    if (flagged)
        for (int it = 0; it < nref; ++it) { const double r = fma(-d0, l, sv); l = fma(x, r, l); }
    const int64_t irow = i0 + b, q = irow / nb;
    if (q < tp0 || q >= tp1 || (int)(q % world) != rank) return;
    const int64_t lcol = (q / world) * (int64_t)nb + (irow - q * nb);
    A[irow + lcol * ld] = fma(l, 1.0, A[irow + lcol * ld]);
}

// ---------------------------------------------------------------------------------------------
// K4: C[i,j] -= sum_k L[i,k] * W[j,k]  on the lower block-triangle, fp64 MFMA
// (v_mfma_f64_16x16x4_f64).  Column-major everywhere, so i is the contiguous index of C, L and W.
// MFMA orientation: D[m][n] with m <- j (A operand = W), n <- i (B operand = L); the f64 C/D map
// is col = lane&15 (-> i, contiguous), row = (lane>>4) + 4*reg (-> j).
// Block tile 128(i) x BN(j), 4 waves, each wave 64 x (BN/2); K staged through LDS 16 at a time
// with register prefetch.  The W operand buffer holds -W (written by k_panel_scale) so the accumulator
// starts at C and the staged registers go to LDS untouched (no VALU op forces an early vmcnt wait).
// Column tiles are enumerated over LOCALLY owned panels (block-cyclic), rows are global.
// ---------------------------------------------------------------------------------------------
// Shared by host and device: number of 8x8 super-tiles a launch enumerates, and the decode of a
// linear super-tile index into (super-row, super-col).  Only super-tiles that contain at least one
// tile on or below the diagonal are enumerated (column by column).
struct UpdGeo {
    int64_t row_begin, Npad, first_lp, sub0;
    int nb, world, rank, nrt, nct;         // nrt/nct: row / column tiles of this launch
    int prio;                              // != 0: raise wave priority (latency-critical panel-chain launches)
    int64_t a0, a1, b0, b1;                // rows / columns that can be non-zero in BOTH operands: [a0,a1) u [b0,b1); a tile
                                           // outside them would add an exact zero (KKT block structure) and is skipped
    int rt_min0, rt_step;                  // first row tile on/below the diagonal for super-column sJ = rt_min0 + sJ*rt_step (tiles)
    unsigned long long* dbg;               // diagnostics only (NULL in normal operation): per-block timeline
    const unsigned* tiles;                 // swizzled launches: block b works on tile (tiles[b] & 0xffff, tiles[b] >> 16), built on
                                           // the host in the XCD-aware order below with every empty tile left out; NULL = decode here
    unsigned persist = 0;                  // != 0: a persistent launch over a tile list of this many entries (see k_update)
    int head_ct = 0;                       // > 0 (fused head, round 3): column tiles [0, head_ct) are the NEXT group's columns; their tiles
    unsigned* head_counter = nullptr;      // come first in the list, and every one of them bumps *head_counter once its C tile is stored
    int64_t ks_cstride;                    // split-K launches (grid.y = splits, tile-list order only): split y accumulates its K
                                           // columns of the operands into C + y*ks_cstride; 0 = one split (every KKT launch)
};
template <int BN>
__host__ __device__ inline void upd_col(const UpdGeo& u, int64_t ct, int64_t& jglob, int64_t& jloc) {
    const int tpp = u.nb / BN;
    const int64_t c = ct + u.sub0;
    const int64_t lp = u.first_lp + c / tpp;
    const int sub = (int)(c % tpp);
    jglob = (lp * u.world + u.rank) * (int64_t)u.nb + (int64_t)sub * BN;
    jloc = lp * (int64_t)u.nb + (int64_t)sub * BN;
}
constexpr int SUPER = 8;
// First super-row that holds a tile on/below the diagonal, for super-column sJ.  Division-free: the
// host fills (rt_min0, rt_step) -- the first valid row TILE is affine in sJ because a super-column of 8
// column tiles always spans whole panels (launches use the swizzled order only when nb/BN divides 8).
__host__ __device__ inline int upd_super_min_row(const UpdGeo& u, int sJ) {
    int rt_min = u.rt_min0 + sJ * u.rt_step;
    if (rt_min < 0) rt_min = 0;
    return rt_min >> 3;                                  // / SUPER
}
template <int BN>
inline void upd_fill_affine(UpdGeo& u) {
    int64_t jg0, jl0;
    upd_col<BN>(u, 0, jg0, jl0);
    u.rt_min0 = (int)((jg0 - u.row_begin) / BM);         // may be negative (columns left of the row range)
    u.rt_step = SUPER * BN * u.world / BM;               // tiles per super-column step
}
template <int BN>
inline bool upd_swizzle_ok(const UpdGeo& u) { const int tpp = u.nb / BN; return tpp >= 1 && (SUPER % tpp) == 0; }
template <int BN>
inline int64_t upd_super_count(const UpdGeo& u) {
    const int nsr = (u.nrt + SUPER - 1) / SUPER, nsc = (u.nct + SUPER - 1) / SUPER;
    int64_t tot = 0;
    for (int sJ = 0; sJ < nsc; ++sJ) {
        const int mn = upd_super_min_row(u, sJ);
        if (mn < nsr) tot += nsr - mn;
    }
    return tot;
}

// BN = 256 (bulk launches, round 3): a block works on 128 x 256 -- each of its 8 waves on 64 x 64 (16 accumulator tiles, 8
// fragment reads per 16 MFMAs instead of 6 per 8), one block of 107 KB of LDS per CU (two waves per SIMD, 256 registers
// each): operand bytes per flop from L2 down by a quarter, half as many C-tile prologues per flop.  The products reach every
// entry in the same order as with BN = 128: the same bits.
template <int BN, bool SWZ, int NW = 4>
__global__ __launch_bounds__(NW * 64, (BN >= 256 ? 2 : NW / 2)) void k_update(
    double* __restrict__ C, int64_t ldc,
    const double* __restrict__ Lop, int64_t ldl,
    const double* __restrict__ Wop, int64_t ldw,
    int K, UpdGeo u)
{
    constexpr int NT = NW * 64;            // threads; waves are laid out 2 (along i) x NW/2 (along j)
    constexpr int TJ = BN / (NW / 2) / 16; // 16-wide MFMA tiles per wave along j
    constexpr int TI = 4;                  // along i (wave covers 64 rows)
    constexpr int LSTR = BM + 16;          // padded LDS row strides (doubles): rows k and k+1 hit
    constexpr int WSTR = BN + 16;          // disjoint halves of the 64 banks
    __shared__ double Ls[2][BKU][LSTR];     // double-buffered: one barrier per k-stage
    __shared__ double Ws[2][BKU][WSTR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (u.prio) __builtin_amdgcn_s_setprio(3);     // panel-chain launches (side stream)
    const int64_t Npad = u.Npad;
    // One tile of the launch; `ent` = its entry in the tile list (= blockIdx.x of an ordinary launch).
    auto one_tile = [&](const unsigned ent) {
    int64_t rt, ct;
    if (SWZ && u.tiles) {
        // split-K launches rotate the XCD sequences by the split index: a list shorter than 8 real entries
        // per round (few output tiles) would otherwise put every split's blocks on the same XCDs
        const unsigned b = u.ks_cstride ? ((ent & ~7u) | ((ent + blockIdx.y) & 7u)) : ent;
        const unsigned code = u.tiles[b];
        if (code == 0xffffffffu) return;               // padding of a shorter XCD sequence
        rt = code & 0xffffu; ct = code >> 16;
        if (u.ks_cstride) {                            // Gram launches over a long K (kernels_lbfgs.hpp)
            C += (int64_t)blockIdx.y * u.ks_cstride;
            Lop += (int64_t)blockIdx.y * K * ldl;
            Wop += (int64_t)blockIdx.y * K * ldw;
        }
    } else if (SWZ) {
        // XCD-aware order: block b runs on XCD b%8 (observed dispatch); each XCD walks its own
        // sequence of 8x8 super-tiles so the 16 operand tiles of a super-tile are reused from its L2.
        const unsigned b = ent;
        const int xcd = (int)(b & 7u);
        const unsigned slot = b >> 3;
        int sidx = (int)((slot >> 6) * 8u) + xcd;            // / (SUPER*SUPER)
        const int within = (int)(slot & 63u);
        const int nsr = (u.nrt + SUPER - 1) >> 3, nsc = (u.nct + SUPER - 1) >> 3;
        int sJ = 0; int sI = -1;
        for (; sJ < nsc; ++sJ) {
            const int mn = upd_super_min_row(u, sJ);
            const int cnt = mn < nsr ? nsr - mn : 0;
            if (sidx < cnt) { sI = mn + sidx; break; }
            sidx -= cnt;
        }
        if (sI < 0) return;
        rt = sI * SUPER + (within & (SUPER - 1));
        ct = (int64_t)sJ * SUPER + (within >> 3);
        if (rt >= u.nrt || ct >= u.nct) return;
    } else {
        rt = blockIdx.x; ct = blockIdx.y;
    }
    int64_t jglob, jloc;
    upd_col<BN>(u, ct, jglob, jloc);
    if (jglob >= Npad) return;
    const int64_t i0 = u.row_begin + rt * BM;
    if (i0 + BM <= jglob) return;          // tile strictly above the diagonal
    {   // structural zeros: the L rows of this tile or the W rows of its columns are identically zero
        const bool ri = (i0 + BM > u.a0 && i0 < u.a1) || (i0 + BM > u.b0 && i0 < u.b1);
        const bool ci = (jglob + BN > u.a0 && jglob < u.a1) || (jglob + BN > u.b0 && jglob < u.b1);
        if (!(ri && ci)) return;
    }
    const int wi = (wave & 1) * 64, wj = (wave >> 1) * (BN / (NW / 2));
    const int l15 = lane & 15, l4 = lane >> 4;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, cy1 = 0, cy2 = 0;
    if (u.dbg) ts0 = wall_clock64();

    double4_t acc[TJ][TI];
    #pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
        #pragma unroll
        for (int ti = 0; ti < TI; ++ti)
            #pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[tj][ti][r] = C[(i0 + wi + ti * 16 + l15) + (jloc + wj + tj * 16 + l4 + 4 * r) * ldc];

    // staging registers: L tile 16 x 128 doubles = 1024 double2 -> 1024/NT per thread;
    //                    W tile 16 x BN  doubles -> 8 BN / NT per thread
    constexpr int LKPP = NT / 64;                                     // L: k-rows per pass (64 threads x double2 per row)
    constexpr int LPASS = BKU / LKPP;
    constexpr int WTPR = BN / 2;                                      // threads per k-row of W
    constexpr int WKPP = NT / WTPR;                                   // W: k-rows per pass
    constexpr int WPASS = BKU / WKPP;
    double2_t lreg[LPASS], wreg[WPASS];
    const int lk = tid >> 6, li = (tid & 63) * 2;
    const int wk = tid / WTPR, wjj = (tid % WTPR) * 2;

    const double* lsrc = Lop + (i0 + li) + (int64_t)lk * ldl;
    const double* wsrc = Wop + (jglob + wjj) + (int64_t)wk * ldw;
#define PYIPM_LOAD_REGS(k0_)                                                                          \
    {                                                                                                 \
        _Pragma("unroll") for (int ps = 0; ps < LPASS; ++ps)                                          \
            lreg[ps] = *reinterpret_cast<const double2_t*>(lsrc + (int64_t)((k0_) + LKPP * ps) * ldl);  \
        _Pragma("unroll") for (int ps = 0; ps < WPASS; ++ps)                                          \
            wreg[ps] = *reinterpret_cast<const double2_t*>(wsrc + (int64_t)((k0_) + WKPP * ps) * ldw);  \
    }

#define PYIPM_STORE_LDS(buf_)                                                                         \
    {                                                                                                 \
        _Pragma("unroll") for (int ps = 0; ps < LPASS; ++ps)                                          \
            *reinterpret_cast<double2_t*>(&Ls[buf_][lk + LKPP * ps][li]) = lreg[ps];                  \
        _Pragma("unroll") for (int ps = 0; ps < WPASS; ++ps)                                          \
            *reinterpret_cast<double2_t*>(&Ws[buf_][wk + WKPP * ps][wjj]) = wreg[ps];                 \
    }
#define PYIPM_FRAGS(buf_, kk_, a_, b_)                                                                \
    {                                                                                                 \
        _Pragma("unroll") for (int tj = 0; tj < TJ; ++tj) a_[tj] = Ws[buf_][(kk_) + l4][wj + tj * 16 + l15]; \
        _Pragma("unroll") for (int ti = 0; ti < TI; ++ti) b_[ti] = Ls[buf_][(kk_) + l4][wi + ti * 16 + l15]; \
    }
#define PYIPM_MFMAS(a_, b_)                                                                           \
    {                                                                                                 \
        _Pragma("unroll") for (int tj = 0; tj < TJ; ++tj)                                             \
            _Pragma("unroll") for (int ti = 0; ti < TI; ++ti)                                         \
                acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_[tj], b_[ti], acc[tj][ti], 0, 0, 0); \
    }

    // One DS / VMEM instruction is pinned behind each 64-cycle MFMA (sched_group_barrier) and the
    // fragments are prefetched a full sub-step ahead: the two co-resident waves of a SIMD otherwise
    // run in lockstep and wait on LDS latency together (tools/ubench/mfma_structure.hip: 66 -> 76.8 TF/s).
#define PYIPM_ILV(n_, mask_)                                                                          \
    { _Pragma("unroll") for (int q_ = 0; q_ < (n_); ++q_) {                                           \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
          __builtin_amdgcn_sched_group_barrier(mask_, 1, 0); } }

    // prologue: stage 0 into LDS buffer 0, stage 1 into registers
    PYIPM_LOAD_REGS(0)
    PYIPM_STORE_LDS(0)
    { const int k1 = (BKU < K) ? BKU : 0; PYIPM_LOAD_REGS(k1) }
    __syncthreads();
    if (u.dbg) { asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[TJ - 1][TI - 1][3])); ts1 = wall_clock64(); cy1 = clock64(); }
    int cur = 0;
    double a0[TJ], b0[TI], a1[TJ], b1[TI];
    PYIPM_FRAGS(cur, 0, a0, b0)
    for (int k0 = 0; k0 < K; k0 += BKU) {
        // branch-free body: past the end the staging traffic re-reads the last stage (harmless)
        int k2 = k0 + 2 * BKU; if (k2 > K - BKU) k2 = K - BKU;
        PYIPM_FRAGS(cur, 4, a1, b1)
        PYIPM_MFMAS(a0, b0)
        PYIPM_ILV(2 * TJ + 2 * TI > 16 ? 16 : TJ + TI, 0x100)
        PYIPM_FRAGS(cur, 8, a0, b0)
        PYIPM_MFMAS(a1, b1)
        PYIPM_ILV(TJ + TI, 0x100)
        PYIPM_FRAGS(cur, 12, a1, b1)
        PYIPM_STORE_LDS(cur ^ 1)                    // stage k+1: registers -> the other LDS buffer
        PYIPM_LOAD_REGS(k2)   // stage k+2 -> registers (a full stage ahead of its use)
        PYIPM_MFMAS(a0, b0)
        PYIPM_ILV(LPASS + WPASS + LPASS + WPASS < TJ * TI ? LPASS + WPASS + LPASS + WPASS : TJ * TI, 0x0A0)
        __syncthreads();
        PYIPM_FRAGS(cur ^ 1, 0, a0, b0)
        PYIPM_MFMAS(a1, b1)
        PYIPM_ILV(TJ + TI, 0x100)
        cur ^= 1;
    }
#undef PYIPM_ILV
    if (u.dbg) { ts2 = wall_clock64(); cy2 = clock64(); }
#undef PYIPM_STORE_LDS
#undef PYIPM_FRAGS
#undef PYIPM_MFMAS
#undef PYIPM_LOAD_REGS

    #pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
        #pragma unroll
        for (int ti = 0; ti < TI; ++ti)
            #pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(i0 + wi + ti * 16 + l15) + (jloc + wj + tj * 16 + l4 + 4 * r) * ldc] = acc[tj][ti][r];
    if (SWZ && u.head_ct > 0 && ct < u.head_ct) {
        // fused head: this tile belongs to the columns the next group's chain is waiting for (k_wait_counter on its stream).
        // Barrier: every thread's stores are issued; device-scope release by one thread (writes the XCD's L2 back: the
        // reader is a later kernel on any XCD, which invalidates at its start); then the count.
        __syncthreads();
        if (tid == 0) { __threadfence(); atomicAdd(u.head_counter, 1u); }
    }
    if (u.dbg && tid == 0) {
        const unsigned lin = SWZ ? ent : blockIdx.x + gridDim.x * blockIdx.y;
        unsigned long long* d = u.dbg + 8ull * lin;
        const unsigned long long ts3 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // diagnostics: when have this wave's stores drained?
        d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = ts3; 
        d[4] = (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) & 0xffffu)          // HW_ID low 16 bits (size field = 15+1)
               | ((cy2 - cy1) << 16);                                          // ... and the shader cycles of the main loop (clock64; ts* tick at 100 MHz)
        d[5] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);          // XCC_ID
        d[6] = wall_clock64() - ts3; d[7] = ct;
    }
    };   // one_tile
    // Persistent launches (u.persist = length of the tile list; chain-bound phases and the per-panel schedule): fewer
    // blocks than the GPU has slots walk the list with stride gridDim.x -- a multiple of 8, so a block stays on its XCD's
    // sequence -- and the slots they leave empty are there for the panel chain the moment it needs them.  An ordinary
    // launch fills every CU with blocks that all retire together one tile time (~250 us) later: a chain kernel launched in
    // between waits for that (tools/contention_probe.py, tools/ubench/contention.hip), three to six times per group.
    if (SWZ && u.tiles && u.persist) {
        for (unsigned e = blockIdx.x; e < u.persist; e += gridDim.x) {
            one_tile(e);
            __syncthreads();                           // the staging buffers are the next tile's
        }
    } else {
        one_tile(blockIdx.x);
    }
}

// ---------------------------------------------------------------------------------------------
// In-panel left-looking update of one 64-column block with the K = 64 t columns before it (K <= nb - 64):
//   C[i][c] += sum_k L[i][k] * Wn[c][k]        (Wn = -W), rows i >= row_begin
// The big update kernel is built for long K (LDS-staged 16-deep stages, two in flight); at K <= 192 its fixed
// latencies are most of its 8-17 us, and a 128-row tile keeps one CU busy for 3.4 us of MFMA per 64 columns of K
// while, late in the factorisation, most CUs have no tile at all.  Here a block is 32 rows x 64 columns (4x the
// blocks), operands go straight from global memory into the MFMA operand registers (lanes along the contiguous
// index of both, as in k_panel_scale), a whole 64-wide slab of K in flight at a time.  Same products accumulated in
// the same order as k_update<64> (D[m <- c][n <- i], k ascending in MFMA groups of 4): bit-identical results.
// grid = (rows/32, column blocks), block 256: wave w owns columns [16w, 16w+16) x 32 rows (two accumulator tiles).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_inpanel_update(
    double* __restrict__ C, int64_t ldc, int64_t ccol,                 // column block: local column ccol, global row range = cglob..+64
    const double* __restrict__ Lop, int64_t ldl,                       // L[i][k] = Lop[i + k*ldl]
    const double* __restrict__ Wop, int64_t ldw, int64_t cglob,        // Wn[c][k] = Wop[(cglob + c) + k*ldw]
    int K, int64_t row_begin, int64_t Npad,
    int64_t a0, int64_t a1, int64_t b0, int64_t b1, int prio)
{
    if (prio) __builtin_amdgcn_s_setprio(3);
    ccol += (int64_t)blockIdx.y * TB;                                  // grid.y column blocks of 64 (the pending update of a
    cglob += (int64_t)blockIdx.y * TB;                                 // whole panel inside its group uses nb/64 of them)
    const int64_t i0 = row_begin + (int64_t)blockIdx.x * 32;
    if (i0 + 32 <= cglob) return;                                      // wholly above the diagonal block
    {   // structural zeros (see k_update): rows or columns outside the active ranges add exact zeros
        const bool ri = (i0 + 32 > a0 && i0 < a1) || (i0 + 32 > b0 && i0 < b1);
        const bool ci = (cglob + TB > a0 && cglob < a1) || (cglob + TB > b0 && cglob < b1);
        if (!(ri && ci)) return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    double4_t acc[2];
    #pragma unroll
    for (int h = 0; h < 2; ++h)
        #pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[h][r] = C[(i0 + 16 * h + l15) + (ccol + wave * 16 + l4 + 4 * r) * ldc];
    const double* wp = Wop + (cglob + wave * 16 + l15) + (int64_t)l4 * ldw;       // A operand: m = c, k = 4 ks + l4
    const double* lp = Lop + (i0 + l15) + (int64_t)l4 * ldl;                       // B operand: n = i, k = 4 ks + l4
    for (int k0 = 0; k0 < K; k0 += TB) {
        double wa[16], lb0[16], lb1[16];
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            wa[ks] = wp[(int64_t)(k0 + 4 * ks) * ldw];
            lb0[ks] = lp[(int64_t)(k0 + 4 * ks) * ldl];
            lb1[ks] = lp[16 + (int64_t)(k0 + 4 * ks) * ldl];
        }
        #pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[ks], lb0[ks], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[ks], lb1[ks], acc[1], 0, 0, 0);
        }
    }
    #pragma unroll
    for (int h = 0; h < 2; ++h)
        #pragma unroll
        for (int r = 0; r < 4; ++r)
            C[(i0 + 16 * h + l15) + (ccol + wave * 16 + l4 + 4 * r) * ldc] = acc[h][r];
}

// A slice of the two-message protocol in ONE launch: E rows of the panel's W (all nbw columns, leading dimension E in the
// message) and, with slice 1, the panel's tile inverses, tiles and flags behind them (four copies in a row sat on the owners'
// chain path otherwise).
__global__ __launch_bounds__(256) void k_slice_pack(double* __restrict__ buf, int64_t E, int64_t nbw, const double* __restrict__ W,
                                                    int64_t ldw, const double* __restrict__ Dinv, const double* __restrict__ Tsv,
                                                    const double* __restrict__ Tflag, int with_tiles)
{
    const int64_t rows = E * nbw, tb = (nbw / TB) * (int64_t)(TB * TB);
    const int64_t total = rows + (with_tiles ? 2 * tb + nbw / TB : 0);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        double v;
        if (e < rows) v = W[(e % E) + (e / E) * ldw];
        else if (e < rows + tb) v = Dinv[e - rows];
        else if (e < rows + 2 * tb) v = Tsv[e - rows - tb];
        else v = Tflag[e - rows - 2 * tb];
        buf[e] = v;
    }
}

// Register-resident MFMA-only loop for the fp64 matrix peak measurement.  Inline asm keeps the
// eight accumulators in VGPRs (the builtin form made hipcc shuttle them through AGPRs every trip).
__global__ __launch_bounds__(256) void k_mfma_peak(double* out, int iters)
{
    double4_t acc[8];
    #pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            #pragma unroll
            for (int t = 0; t < 8; ++t)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0.0;
    #pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 123.456) out[0] = s;
}

}  // namespace pyipm
