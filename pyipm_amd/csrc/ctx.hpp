// ctx.hpp — handle, geometry and small helpers of the Newton-step core (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <new>
#include <stdexcept>
#include <map>
#include <vector>
#include "../../include/pyipm_newton.h"

namespace pyipm {

constexpr int TB = PYIPM_TILE;      // 64: block-pivot tile
constexpr int PADG = PYIPM_PAD;     // 128: Npad granularity = row tile of the MFMA update
constexpr int BM = 128;             // MFMA update tile rows (i)
constexpr int BKU = 16;             // MFMA update k-stage
constexpr int ROWCHUNK = 2048;      // rows per block in the backward column dots

// Block-cyclic 1D column distribution by panels of width nb.
struct Geo {
    int64_t n, me, mi, N, Npad;
    int nb, world, rank;
    int64_t npanels;        // global panels
    int64_t ncols_local;    // columns stored on this rank
    __host__ __device__ int64_t panel_c0(int64_t p) const { return p * (int64_t)nb; }
    __host__ __device__ int64_t panel_w(int64_t p) const {
        int64_t w = Npad - p * (int64_t)nb;
        return w < nb ? w : nb;
    }
    __host__ __device__ int owner(int64_t p) const { return (int)(p % world); }
    __host__ __device__ int64_t local_c0(int64_t p) const { return (p / world) * (int64_t)nb; }
};

// Panels aggregated per bulk trailing update (K = group * nb).  Grouping needs the group's panels on
// one rank, so it applies to single-rank handles; PYIPM_NEWTON_GROUP overrides (read at create time).
inline int default_group(int world, int nb = 256) {
    if (world > 1) return 1;
    const char* e = getenv("PYIPM_NEWTON_GROUP");
    // 8 panels of 256 since the end of round 3 (K = 2048 per bulk launch while more than tail_cols columns remain, then
    // tail_group = 4): with the 128 x 256 bulk tiles -1.2 % at N = 32768, -2.7 % at N = 131072 (4 before: no difference with
    // 128 x 128 tiles).  A group's diagonal block is one chain of at most 32 tile steps: 8 x nb / 64 <= 32.
    int g = e ? atoi(e) : (nb <= 256 ? 8 : 4);
    if (g < 1) g = 1;
    if (g > 8) g = 8;
    return g;
}

inline Geo make_geo(int64_t n, int64_t me, int64_t mi, int nb, int world, int rank) {
    Geo g;
    g.n = n; g.me = me; g.mi = mi;
    g.N = n + 2 * mi + me;
    g.Npad = ((g.N + PADG - 1) / PADG) * PADG;
    if (g.Npad == 0) g.Npad = PADG;
    g.nb = nb > 0 ? nb : 256;
    g.world = world; g.rank = rank;
    g.npanels = (g.Npad + g.nb - 1) / g.nb;
    int64_t c = 0;
    for (int64_t p = rank; p < g.npanels; p += world) c += g.panel_w(p);
    g.ncols_local = c;
    return g;
}

struct DevStats {          // lives in device memory; tile kernels of one rank run serially
    long long n_neg, n_zero, n_2x2, n_pos, nonfinite;
    double d_min, d_max;
    unsigned long long growth_bits;   // bit pattern of max |L| (monotone for non-negative doubles)
};

// One launch of k_tile_chain (kernels_chain.hpp)
struct ChainGeo {
    int ta, tb, nT;          // steps [ta, tb) of a diagonal block of nT tiles
    int cpy;                 // column tiles per unit and stage a row tile is split for (5: k_tile_step's rule)
    int nR;                  // row tiles of the launch: nT, or more -- EXTRA rows right below the diagonal block whose every stage the
                             // launch applies too (per-panel schedule: the rows the next panel's owner waits for)
    const unsigned* xword; unsigned xwant;   // the extra rows may be touched once *xword has reached xwant (NULL: at once)
    unsigned base;           // epoch of the progress words: word - base = progress of THIS launch (wrap-safe compare)
    unsigned* sync;          // [0]: tiles inverted by the chain (base + t + 1 after tile t);  [1 + 4 r + y]: stages unit (r, y) completed
    unsigned* err;           // sticky: a poll timed out
    unsigned long long timeout;   // 100 MHz ticks
    unsigned long long* dbg;      // diagnostics (NULL normally; tools/chain_clock.py): 100 MHz stamps, [8 t + k] of the chain's step t,
                                  // [256 + 64 (4 r + y) + 2 tp + k] of unit (r, y)'s stage tp (CHAIN_DBG_WORDS per launch)
};
constexpr int CHAIN_DBG_WORDS = 256 + 64 * 4 * 32;

struct Ctx {
    Geo g;
    int batch = 1;                        // problems of a batched small-system handle (kernels_batched.hpp)
    bool batched = false;                 // batched handle: single-system entry points refuse it
    int64_t b_sH = 0, b_sJe = 0, b_sJi = 0;   // batch strides (doubles) of the caller's blocks
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;           // panel lookahead stream (created on first factor)
    hipEvent_t ev_head = nullptr, ev_panel = nullptr, ev_fwd = nullptr;
    hipStream_t fwd = nullptr;            // fused forward-substitution stream
    std::vector<hipEvent_t> ev_done;      // panel q factored
    int fuse_forward = 1;
    bool forward_fused = false;
    bool forward_pending = false;         // factor() already forward-substituted the pending residual into v0
    int lookahead = 2;                    // 0 none, 1 one group (two groups on a dedicated stream measured no faster: removed), 2 = 1 + the
                                          // first group's update panel by panel under its own chain where the slack block follows it (factor_all)
    hipEvent_t ev_early = nullptr;
    int s_fast = 1;                       // panels inside the slack block: closed-form elimination (k_s_panel)
    std::vector<char> grp_fast;           // per group: every panel of it takes that path (built by factor_all)
    std::vector<char> grp_x;              // per group: lies inside the x block (its chain kernels may skip the slack rows)
    struct TileList { unsigned* dev = nullptr; unsigned count = 0; unsigned head_count = 0;
                      hipEvent_t ready = nullptr; hipStream_t on = nullptr; bool seen_done = false; };   // the async upload: its event, its stream
    struct TlArena { unsigned* dev = nullptr; unsigned* host = nullptr; size_t cap = 0, used = 0; };    // device + pinned host memory of the lists
    std::vector<TlArena> tl_arenas;
    std::map<std::vector<int64_t>, TileList> tile_lists;   // compact tile orders of the bulk launches (geometry repeats every step)
    int skip_zeros = 1;                   // trailing updates skip tiles that the KKT block structure makes exact zeros
    int group = 1;                        // panels per bulk trailing update
    bool tail_group_user = false;         // set_option("tail_group") was called (else: 8 for systems of at most 8192 rows)
    int tail_group = 4;                   // group size once at most tail_cols columns remain: there the panel chain outlasts
    int64_t tail_cols = 24576;            // the bulk update, and shorter groups move in-group update work off the chain
    std::vector<int> grp_of, grp_off;     // per panel: group id and offset inside the group (built by factor_all)
    std::vector<int64_t> grp_first;       // per group: first panel (+ one past the last group)
    int bulk_waves = 8;                   // waves per block of the BULK update tiles (8: 128 VGPRs each, 4 waves per SIMD;
                                          // 4: 241 VGPRs, 2 per SIMD: 1.5 % slower); the short side-stream launches keep 4
    int fuse_su = 1;                      // ... fused into the previous tile's scaling launch (k_panel_scale + NextUpd): two
                                          // dependent launches per tile on the chain instead of three; same bits
    double* Wnext = nullptr;              // 64 x 64: -S of the next diagonal tile's rows (handed from the tile kernel to that launch)
    int group_chain = 1;                  // single rank: the panels of a group are chained tile to tile (factor_group), the rows
                                          // below the group's diagonal block follow on their own stream; same bits
    hipStream_t rest = nullptr;           // ... that stream (high priority, created on first use)
    std::vector<hipEvent_t> ev_band;      // panel (by offset in its group): its tiles are inverted and applied inside the diagonal block
    hipEvent_t ev_join = nullptr, ev_main = nullptr, ev_split = nullptr, ev_sfast = nullptr;
    int head_waves = 4;                   // waves per block of a lookahead head launched on the chain's stream (4: k_update<128,true,4>,
                                          // its own line in a kernel trace; 8: the bulk instance)
    int64_t pending_left_rows = 12288;    // group chain: left-looking in-group updates of the rows below the diagonal block while more
                                          // rows than this remain below it (-1: never)
    int asm_tri = 1;                      // K1 launches only the patches on or below the diagonal (single rank, Npad a multiple of 512)
    int keep_zeros = 1;                   // K1 does not store again the zeros nothing can fill in (k_assemble, zeros_in_place)
    bool zeros_clean = false;             // ... which requires that the last writer of those places was a full assembly
    bool storage_exported = false;        // kkt_storage() handed the pointer out: a holder may write into those zeros at any time,
                                          // so every assembly is a full one until set_option("keep_zeros") is called again
    int rest_prio = 1;                    // ctx->rest is a high-priority stream (set before the first factorisation)
    int fast_on_main = 1;                 // groups inside the slack block (closed form) run on the main stream, not through the lookahead
    int head_on_side = 1;                 // the lookahead head runs on the stream of the chain it follows (no stream crossing between
                                          // a group's chain, the head and the next chain); ordered against the main stream by an event
    int bwd_diag4 = 1;                    // in-panel backward substitution on 1024 threads through shared memory (k_bwd_diag4)
    int tile_step = 1;                    // stepped panel schedule (kernels_panel.hpp): one launch per diagonal tile (the rows inside the
                                          // diagonal block), one for the rows below it; panels of at most 4 tiles; same bits
    int64_t head32_rows = 6144;           // ... and the lookahead HEAD update (next group's columns, on the critical path between two
                                          // groups' chains: one 128x128 tile at K = 512 takes 132 us however few tiles there are)
    bool per_panel_mode = false;          // the per-panel phases (pyipm_newton_factor_begin ...) drive this factorisation
    int64_t head32_rows_dist = 16384;     // per-panel (multi-GPU) schedule: single-panel launches (the owner's head update of the
                                          // next panel, always on the critical path there) while at most this many rows remain
    int64_t bulk_bn_rows = 20480;         // 128 x 256 tiles only for launches over more rows than this (and than persist_rows): below it the
                                          // next group's chain is no longer hidden behind the bulk launch, and beside the wide tiles (one
                                          // block per CU, every register) its kernels wait twice as long for a slot -- exposed panel 6.0
                                          // instead of 5.3 ms with the threshold at persist_rows, the step 0.8 % slower
    int sweep_max_blocks = 0;             // test hook: cap on the workgroups of the one-launch sweeps (0 = as many as the GPU holds)
    int occ_fwd_sweep = 0, occ_bwd_sweep = 0;   // resident workgroups per CU of the one-launch sweeps (occupancy query, cached)
    int sweep_persist = 1;                // single rank, one right-hand side: the backward sweep as ONE device-driven launch (k_bwd_sweep)
    int64_t sweep_buf_n = 0;              // ... (allocated for this many rows)
    double* sweep_buf = nullptr;          // ... the near sums as the column owners hand them to workgroup 0 (Npad doubles, NaN = not there yet)
    unsigned* sweep_sync = nullptr;       // ... its flags and counters (3 npanels + 1 words, zeroed before every sweep)
    bool sweep_used = false;              // ... a sweep ran since the error word was last read (solve_info / factor_end look at it)
    int dist_head_split = 1;              // per-panel schedule: the owner's head in two launches -- the next panel's diagonal block (its chain waits for
                                          // that alone), then the rows below it on ctx->rest (round 4)
    int dist_slices = 2;                  // distributed schedule: the two-message protocol (slices ahead of the panel message: the next owner's tile
                                          // chain starts on an nb x nb message); 0 = one message per panel (rounds 1-4); collective
    double dist_timeout_s = 300.0;        // distributed step: bound on the host's wait for the device (dist_impl.hpp:bounded_wait); <= 0: none
    int dist_comm2 = 0;                   // RCCL transport: the slice messages on a second communicator of their own (set before comm_init; dist_impl.hpp:comm2_setup)
    int wide_sub = 256;                   // per-panel schedule: a panel wider than this is factored as a block of sub-panels this wide
                                          // (factor_wide_panel: the single-rank group chain inside one panel); 0 = all stages in one launch
    int64_t pending32_rows = 24576;       // ... and a panel's pending in-group update too while at most this many rows remain
                                          // (128x128 tiles keep one CU busy for 27 us per 256 columns of K, on the chain)
    int reserve_cus = 16;                 // chain-bound phases: bulk updates run as persistent launches that leave this many CUs
    int64_t persist_rows = 12288;         // free for the panel chain -- while at most this many rows remain (on one rank the per-panel schedule with it
    int num_cus = 256;                    // everywhere took 140 instead of 120 ms); 0 = ordinary launches everywhere.  num_cus: of this device
    int bulk_bn = 256;                    // column width of a bulk update tile: 256 = 128 x 256 per block (8 waves x 64 x 64, one block per
                                          // CU): 22 % less L2-miss traffic than 128 x 128; the same step time on a fast box, 1.3 % less on
                                          // the slow ones (boxes differ in memory speed, not in MFMA rate); a chain kernel waits twice as
                                          // long for a slot beside it (tools/contention_probe.py), so the chain-bound phase keeps the
                                          // persistent 128 x 128 launches (bulk_bn_all = 0).  128: 128 x 128 everywhere (rounds 1-2)
    int bulk_bn_min_k = 512;              // ... for launches with at least this K (shorter ones keep 128 x 128: twice the blocks)
    int xcd_swizzle = 1;
    int side_prio = 1;                    // raise wave priority in panel-chain update launches
    unsigned long long* dbg_buf = nullptr;   // diagnostics only
    // condensed KKT option: factor the (n+me)-dimensional system
    //   [[H + delta I + Ji Sigma Ji', Je], [Je', -delta_c I]]  instead of the full (n+2mi+me) one
    int condensed = 0;                    // requested by set_option("condensed", 1); single-rank, mi > 0
    int cond_min_refine = 0;              // refinement steps against the FULL blocks every condensed solve gets at least
                                          // (dli = Sigma ds - b_s amplifies the rounding of ds by Sigma <= cond_sigma_max)
    bool cond_active = false;             // the current assembled / factored matrix is the condensed one
    Geo gc;                               // geometry of the condensed system: (n, me + cond_na, 0), set by assemble
    double cond_sigma_max = 1.0e4;        // inequalities with Sigma above this stay explicit rows (-1/Sigma diagonal)
    int64_t cond_na = 0;                  // |A| of the current condensed system
    int *cond_pos = nullptr, *cond_idx = nullptr, *cond_cnt = nullptr;   // device: position in A / members / count
    double* Jx = nullptr; size_t jx_bytes = 0;     // [Je | Ji[:, A]] (lazily hipMalloc'd)
    double *JT = nullptr, *WT = nullptr;  // Ji' and Sigma Ji' operands of the rank-mi update (lazily hipMalloc'd)
    size_t jt_bytes = 0;
    double *vc = nullptr, *vt = nullptr;  // condensed vector / mi-sized temporary (carved)
    double* fwd_vec = nullptr;            // vector the fused forward substitution runs on
    double t_gram = 0;                    // ms of the Ji Sigma Ji' launch (profile)
    bool provider_only = false;           // pyipm_newton_create_provider: staged blocks + vectors, products and residuals; no factorisation
    bool own_ws = false;
    char* ws = nullptr; size_t ws_bytes = 0;
    // carved from workspace
    double *A = nullptr, *Wbuf = nullptr, *Lbuf = nullptr, *Dinv = nullptr;
    double *Tsv = nullptr;                // the diagonal tiles T_k themselves (refinement of the block solves)
    double *Tflag = nullptr;              // per tile: 1.0 = refine block solves with it (pivot spread beyond refine_cond)
    double *rhs = nullptr, *v0 = nullptr, *v1 = nullptr, *v2 = nullptr, *partial = nullptr;
    double *v3 = nullptr;                 // adaptive refinement: the iterate before the last correction (a step that made it worse is taken back)
    double *df = nullptr, *ce = nullptr, *ci = nullptr, *s = nullptr, *lda = nullptr;
    DevStats* dstats = nullptr;
    unsigned long long* anorm = nullptr;  // device: bits of max |assembled KKT entry| (per problem for a batched handle): scale of a static pivot
    // staged blocks (device pointers; either caller-owned or library staging)
    const double *d2L = nullptr, *Je = nullptr, *Ji = nullptr;
    int64_t ld_d2L = 0, ld_Je = 0, ld_Ji = 0;
    int sharded = 0;                      // the staged blocks hold only the rows of the x-columns this rank owns (local column order)
    struct DistState* dist = nullptr;     // distributed driver (dist_impl.hpp): exchange, streams, message buffers
    double *stg_d2L = nullptr, *stg_Je = nullptr, *stg_Ji = nullptr;   // lazily hipMalloc'd
    size_t stg_d2L_sz = 0, stg_Je_sz = 0, stg_Ji_sz = 0;
    double mu = 0.2, eps = 2.220446049250313e-16;
    double delta = 0.0, delta_c = 0.0;
    bool have_blocks = false, have_vectors = false, have_rhs = false, assembled = false, factored = false;
    bool have_direction = false;          // v2 holds the last sign-flipped direction (for step_lengths)
    // merit-function pieces (kernels_merit.hpp): scratch, the products of the current direction with the blocks (Q dx | Je' dx | Ji' dx)
    double* merit_buf = nullptr; double* ray_buf = nullptr; size_t ray_buf_n = 0;
    const double* ray_for = nullptr; bool ray_valid = false, ray_quad_given = false;
    // last solve (pyipm_newton_solve_info): refinement steps taken, |b - Hc x|/|b| before the first and after the last
    // one (-1 = not measured: a fixed-count solve), 1 = the adaptive loop met its target
    int info_steps = 0, info_converged = 0;
    double info_berr0 = -1.0, info_berr = -1.0;
    int rcond_its[2] = {0, 0};            // power / inverse iterations the last pyipm_newton_rcond took (solve_info reports them)
    double* rc_warm[2] = {nullptr, nullptr};   // warm start of the adaptive condition estimate: the vectors its power / inverse
    int64_t rc_warm_n = 0;                // iterations ended with last time (the next estimate starts from them)
    bool rc_warm_valid[2] = {false, false};
    double refine_target = 1.0e-14;       // adaptive refinement stops at this backward error ...
    int refine_max = 8;                   // ... or after this many steps, or when a step gains less than 4x
    // options
    double pivtol_rel = 1e-14;
    bool tile_blocked_user = false;       // set_option("tile_blocked") was called: a batched handle's set_option("condensed") leaves it alone
    int tile_blocked = 1;                 // tile inversion 16 pivots at a time while Bunch-Kaufman would accept them in natural order
                                          // (tile_blocked.hpp); 0: the single sweeps of rounds 1-2 only
    int tile_waves = 8;                   // k_tile_step on 512 threads (round 5): the critical block = four chain waves + four helper waves
                                          // (diagonal tile prefetched beside the scaling product; the blocked inversion's updates and commits
                                          // beside the next elimination, tile_blocked8.hpp); 4: the 256-thread kernel of rounds 2-4.  Same bits.
    int tile_chain = 1;                   // the tile steps of a diagonal block as ONE launch of persistent workgroups per piece (k_tile_chain,
                                          // kernels_chain.hpp; round 6): 1 where the chain is exposed (first group, at most tile8_rows rows left,
                                          // the per-panel / multi-GPU schedule), 2 everywhere, 0 one launch per tile.  Same bits.
    int chain_whole = 1;                  // ... a group's (wide panel's) whole diagonal block as ONE launch: the rows below wait for the chain's
                                          // progress words on their own stream (k_chain_wait); 0: one launch per sub-panel piece, events between
    int chain_lds_kb = 100;               // ... KB of untouched dynamic shared memory per workgroup (keeps other workgroups off its CU)
    bool chain_lds_set = false;
    int chain_cpy = 5;                    // ... column tiles per unit and stage a row tile is split for
    static constexpr int CHAIN_SLOTS = 8, CHAIN_WORDS = 160;
    unsigned* chain_sync = nullptr;       // ... progress words (CHAIN_SLOTS regions used round robin, epoch-stamped) + the sticky error word
    unsigned chain_epoch = 0;
    ChainGeo chain_last = {};             // ... the last chain launch (factor_block hands it to k_chain_wait)
    unsigned long long* chain_dbg = nullptr; int chain_dbg_launch = 0;   // diagnostics only (option debug_chain_ptr)
    bool chain_used = false;              // ... a chain launch ran since the error word was last read (factor_end)
    int64_t tile8_rows = 12288;           // ... used by the single-rank schedule for the first group and where at most this many rows are left
    int tile_free_cus = 64;               // ... CUs assumed free beside a persistent bulk launch (units per block: 1 while the launch fits)
    int bc_per_problem = 1;               // batched condensed form: the Gram part by one workgroup per problem where n = 64 .. 256 allows it
    int s_early = 1;                      // the slack block's closed-form panels enqueued up front on the rows stream (factor_all)
    int tile8_dist = 0;                   // ... also in the per-panel (multi-GPU) schedule
    int tile_upb = 0;                     // ... its other row tiles: (row tile, y) units per 512-thread block, 1 | 2; 0 = by the size of the launch
    int block_refine = 2;                 // refinement steps of L T = S in the panel scaling and of T z = y in the solves
    double refine_cond = 1.0e3;           // ... applied to tiles whose pivot spread dmax/dmin exceeds this
    int profile = 0;
    int expert = 0;                       // set_option("expert", 1): the expert switches are accepted on this handle (else PYIPM_EXPERT=1)
    // timings of last calls (ms)
    double t_assemble = 0, t_panel = 0, t_trailing = 0, t_solve = 0, t_factor = 0;
    double t_trailing_union = 0; int64_t n_trailing_real = 0;   // time with some update launch running (launches may overlap); launches that did work
    double trailing_flops = 0, trailing_area = 0; int64_t n_trailing = 0;   // area: matrix entries updated, summed over launches
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_trailing;   // reused event pool
    struct TrailTag { int bn; double flops, area, cbytes; };   // (area: the launch's algorithmic bytes -- C tiles once in, once out, operand panels once)
    std::vector<TrailTag> trailing_tag;   // per bulk launch of the last factorisation: which k_update instance ran, its flops / bytes
    double inst_ms[2] = {0, 0}, inst_flops[2] = {0, 0}, inst_area[2] = {0, 0}, inst_cbytes[2] = {0, 0}; int64_t inst_n[2] = {0, 0};   // [0]: 128 x 128 tiles, [1]: 128 x 256
    hipEvent_t ev[8] = {};
    hipEvent_t ev_prov[4] = {}; bool prov_valid[2] = {false, false}; double prov_bytes[2] = {0.0, 0.0};   // provider products
    bool ev_assemble_valid = false, ev_solve_valid = false;
    double setup_lists_ms = 0.0; int setup_lists_n = 0;     // host time spent building tile lists (one-time per geometry; PYIPM_SETUP_TRACE)
    int debug_fault = 0;                  // test hook: 1 / 2 = the next tile-list build throws std::bad_alloc / std::runtime_error;
                                          // 3 = the message of the middle panel of the next distributed factorisation stalls (dist_impl.hpp)
    std::string err;
};

#define PYIPM_HIP(call)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e__);                \
            return PYIPM_E_HIP;                                                           \
        }                                                                                 \
    } while (0)

#define PYIPM_KCHECK()  PYIPM_HIP(hipGetLastError())

// No C++ exception crosses the C-ABI (include/pyipm_newton.h): every extern "C" entry is a function-try-block that
// ends in one of these.  std::bad_alloc (host containers: tile lists, schedules, event pools) -> PYIPM_E_NOMEM,
// anything else -> PYIPM_E_HIP; the message goes to last_error when that itself does not throw.
template <class H>
inline void set_err_noexcept(H* h, const char* what) noexcept {
    if (!h) return;
    try { h->err = what; } catch (...) {}
}
#define PYIPM_CATCH_CORE(seterr_, ret_nomem_, ret_other_)                                              \
    catch (const std::bad_alloc&) { seterr_("out of host memory (std::bad_alloc)"); return ret_nomem_; } \
    catch (const std::exception& e__) { seterr_(e__.what()); return ret_other_; }                        \
    catch (...) { seterr_("unknown C++ exception"); return ret_other_; }
#define PYIPM_SETERR_NONE(msg_) (void)(msg_)
#define PYIPM_CATCH_NOH   PYIPM_CATCH_CORE(PYIPM_SETERR_NONE, PYIPM_E_NOMEM, PYIPM_E_HIP)
#define PYIPM_CATCH_SIZE  PYIPM_CATCH_CORE(PYIPM_SETERR_NONE, 0, 0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
struct Carve { size_t off = 0; size_t take(size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; } };

// Rows of the staged blocks a rank works on (the columns it owns: column j of the lower triangle is row j of
// triu(d2L) | Je | Ji), in local column order; kernels_assemble.hpp.
struct RowMap {
    int64_t nloc;                 // rows this rank works on
    int nb, world, rank, sharded;
    __host__ __device__ int64_t glob(int64_t r) const {
        return world == 1 ? r : ((r / nb) * world + rank) * (int64_t)nb + r % nb;
    }
    __host__ __device__ int64_t brow(int64_t r) const { return sharded ? r : glob(r); }
};
inline RowMap make_rowmap(const Geo& g, int sharded) {
    RowMap m; m.nb = g.nb; m.world = g.world; m.rank = g.rank; m.sharded = sharded;
    int64_t c = 0;
    for (int64_t p = g.rank; p < g.npanels; p += g.world) {
        const int64_t c0 = p * (int64_t)g.nb;
        if (c0 >= g.n) break;
        c += (c0 + g.nb <= g.n) ? g.nb : g.n - c0;
    }
    m.nloc = g.world == 1 ? g.n : c;
    return m;
}


// Run a section of the single-rank machinery on another geometry (the condensed system).
struct GeoSwap {
    Ctx* c; Geo saved;
    GeoSwap(Ctx* c_, const Geo& g) : c(c_), saved(c_->g) { c->g = g; }
    ~GeoSwap() { c->g = saved; }
};

}  // namespace pyipm
