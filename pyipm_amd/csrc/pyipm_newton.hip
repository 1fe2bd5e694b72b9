// pyipm_newton.hip — C-ABI of the MI355X Newton-step core (declared in include/pyipm_newton.h).
// gfx950 only; no CUDA path, no CPU fallback.  Host-side orchestration of the HIP kernels in
// kernels_*.hpp.  Replaces /root/reference/pyipm.py:1717-1725 (see the header for the mapping).
#include "driver.hpp"
#include "kernels_assemble.hpp"
#include "kernels_factor.hpp"
#include "kernels_panel.hpp"
#include "kernels_chain.hpp"
#include "kernels_solve.hpp"
#include "kernels_batched.hpp"
#include "kernels_merit.hpp"

using namespace pyipm;
using namespace pyipm::drv;

namespace pyipm { namespace drv {


// Workspace layout; returns total bytes.  When base != nullptr also sets the pointers.
size_t carve_workspace(Ctx* c, const Geo& g, char* base, bool provider_only = false) {
    Carve cv;
    const size_t D = sizeof(double);
    // (a provider-only handle -- pyipm_newton_create_provider: block products and residuals, no factorisation -- has no KKT
    //  storage, no W buffers, no tile inverses: N^2 x 8 bytes that an L-BFGS run never needs)
    const size_t oA = cv.take(provider_only ? 256 : (size_t)g.Npad * (size_t)(g.ncols_local > 0 ? g.ncols_local : 1) * D);
    const size_t oW = cv.take(provider_only ? 256 : 3 * (size_t)default_group(g.world, g.nb) * (size_t)g.Npad * g.nb * D);   // -W of a panel group, three rotating buffers (group g+1 is written while g is read)
    const size_t oL = cv.take(g.world > 1 ? (size_t)g.Npad * g.nb * D : 256);
    const size_t oD = cv.take(provider_only ? 256 : (size_t)(g.Npad / TB) * TB * TB * D);
    const size_t oT = cv.take(provider_only ? 256 : (size_t)(g.Npad / TB) * TB * TB * D);
    const size_t oTf = cv.take((size_t)(g.Npad / TB + 1) * D);
    const size_t orhs = cv.take((size_t)g.Npad * D);
    const size_t ov0 = cv.take((size_t)g.Npad * D);
    const size_t ov1 = cv.take((size_t)g.Npad * D);
    const size_t ov2 = cv.take((size_t)g.Npad * D);
    const size_t ov3 = cv.take((size_t)g.Npad * D);
    const size_t nchunk = (size_t)((g.Npad + ROWCHUNK - 1) / ROWCHUNK) + 1;
    size_t npart = nchunk * (size_t)g.nb;
    const size_t cd_chunks = 64;                       // column-dot partials for the mat-vec
    size_t maxcol = (size_t)(g.n > g.me ? g.n : g.me); if ((size_t)g.mi > maxcol) maxcol = g.mi;
    if (cd_chunks * maxcol > npart) npart = cd_chunks * maxcol;
    {   // one-pass symmetric product (k_symv_tiles): 64 chunks of column partials + one row partial per column segment
        const size_t nseg = ((size_t)g.n + SYMV_SEG - 1) / SYMV_SEG;
        if (((size_t)SYMV_CHUNKS + nseg) * (size_t)g.n > npart) npart = ((size_t)SYMV_CHUNKS + nseg) * (size_t)g.n;
        // ... and the same for a Jacobian block (k_jac_tiles: [SYMV_CHUNKS][m] column partials + [segments of m][n] row partials)
        const size_t mm = (size_t)(g.me > g.mi ? g.me : g.mi), jseg = (mm + SYMV_SEG - 1) / SYMV_SEG;
        if ((size_t)SYMV_CHUNKS * mm + jseg * (size_t)g.n > npart) npart = (size_t)SYMV_CHUNKS * mm + jseg * (size_t)g.n;
    }
    const size_t op = cv.take(npart * D);
    const size_t odf = cv.take((size_t)(g.n + 1) * D);
    const size_t oce = cv.take((size_t)(g.me + 1) * D);
    const size_t oci = cv.take((size_t)(g.mi + 1) * D);
    const size_t os = cv.take((size_t)(g.mi + 1) * D);
    const size_t ol = cv.take((size_t)(g.me + g.mi + 1) * D);
    const size_t ost = cv.take(sizeof(DevStats));
    const size_t ovc = cv.take((size_t)g.Npad * D);
    const size_t ovt = cv.take((size_t)(g.mi + 16) * D);
    const size_t oan = cv.take(64);
    const size_t own = cv.take((size_t)TB * TB * D);
    if (base) {
        c->Wnext = (double*)(base + own);
        c->vc = (double*)(base + ovc); c->vt = (double*)(base + ovt);
        c->anorm = (unsigned long long*)(base + oan);

        c->A = (double*)(base + oA); c->Wbuf = (double*)(base + oW); c->Lbuf = (double*)(base + oL);
        c->Dinv = (double*)(base + oD); c->Tsv = (double*)(base + oT); c->Tflag = (double*)(base + oTf); c->rhs = (double*)(base + orhs);
        c->v0 = (double*)(base + ov0); c->v1 = (double*)(base + ov1); c->v2 = (double*)(base + ov2);
        c->v3 = (double*)(base + ov3);
        c->partial = (double*)(base + op);
        c->df = (double*)(base + odf); c->ce = (double*)(base + oce); c->ci = (double*)(base + oci);
        c->s = (double*)(base + os); c->lda = (double*)(base + ol);
        c->dstats = (DevStats*)(base + ost);
    }
    return cv.off;
}

// Batched small-system handles: per problem A (Npad^2), tile inverses / tiles / flags, rhs and solution, the
// staged vectors; no W buffer (the upper blocks of A hold it) and no panel machinery.
size_t carve_batched(Ctx* c, const Geo& g, int64_t B, char* base) {
    Carve cv;
    const size_t D = sizeof(double), nt = (size_t)(g.Npad / TB);
    const size_t oA = cv.take((size_t)B * g.Npad * g.Npad * D);
    const size_t oD = cv.take((size_t)B * nt * TB * TB * D);
    const size_t oT = cv.take((size_t)B * nt * TB * TB * D);
    const size_t oTf = cv.take((size_t)B * nt * D);
    const size_t orhs = cv.take((size_t)B * g.Npad * D);
    const size_t ov0 = cv.take((size_t)B * g.Npad * D);
    const size_t ov2 = cv.take((size_t)B * g.N * D);
    const size_t odf = cv.take((size_t)B * (g.n + 1) * D);
    const size_t oce = cv.take((size_t)B * (g.me + 1) * D);
    const size_t oci = cv.take((size_t)B * (g.mi + 1) * D);
    const size_t os = cv.take((size_t)B * (g.mi + 1) * D);
    const size_t ol = cv.take((size_t)B * (g.me + g.mi + 1) * D);
    const size_t ost = cv.take((size_t)B * sizeof(DevStats));
    const size_t oan = cv.take((size_t)B * 2 * sizeof(unsigned long long));
    // condensed form (set_option("condensed", 1)): per problem the active set -- position, members, count -- and the backward errors
    const size_t ocp = cv.take((size_t)B * (size_t)(2 * g.mi + 2) * sizeof(int) + (size_t)B * sizeof(int));
    const size_t obe = cv.take((size_t)B * sizeof(double));
    const size_t osg = cv.take((size_t)B * (size_t)(g.mi + 1) * sizeof(double));
    if (base) {
        c->WT = (double*)(base + osg);                  // (batched handle: Sigma of the eliminated pairs, the Gram operand's scaling)
        c->cond_pos = (int*)(base + ocp); c->cond_idx = c->cond_pos + (size_t)B * (size_t)(g.mi + 1);
        c->cond_cnt = c->cond_idx + (size_t)B * (size_t)(g.mi + 1);
        c->vt = (double*)(base + obe);
        c->anorm = (unsigned long long*)(base + oan);
        c->A = (double*)(base + oA); c->Dinv = (double*)(base + oD); c->Tsv = (double*)(base + oT);
        c->Tflag = (double*)(base + oTf); c->rhs = (double*)(base + orhs); c->v0 = (double*)(base + ov0);
        c->v2 = (double*)(base + ov2);
        c->df = (double*)(base + odf); c->ce = (double*)(base + oce); c->ci = (double*)(base + oci);
        c->s = (double*)(base + os); c->lda = (double*)(base + ol); c->dstats = (DevStats*)(base + ost);
    }
    return cv.off;
}

BatchPtrs batch_ptrs(Ctx* ctx) {
    const Geo& g = ctx->g;
    BatchPtrs bp;
    const int64_t nt = g.Npad / TB;
    bp.A = ctx->A; bp.sA = g.Npad * g.Npad;
    bp.Tinv = ctx->Dinv; bp.Tsave = ctx->Tsv; bp.sT = nt * TB * TB;
    bp.Tflag = ctx->Tflag; bp.sF = nt;
    bp.st = ctx->dstats;
    bp.rhs = ctx->rhs; bp.sol = ctx->v0; bp.sV = g.Npad;
    bp.d2L = ctx->d2L; bp.Je = ctx->Je; bp.Ji = ctx->Ji;
    bp.sH = ctx->b_sH; bp.sJe = ctx->b_sJe; bp.sJi = ctx->b_sJi;
    bp.ldh = ctx->ld_d2L; bp.ldje = ctx->ld_Je; bp.ldji = ctx->ld_Ji;
    bp.df = ctx->df; bp.ce = ctx->ce; bp.ci = ctx->ci; bp.s = ctx->s; bp.lda = ctx->lda;
    bp.anorm = ctx->anorm;
    return bp;
}

BatchCond batch_cond(Ctx* ctx) {
    BatchCond bc;
    bc.pos = ctx->cond_pos; bc.idx = ctx->cond_idx; bc.cnt = ctx->cond_cnt; bc.sig = ctx->WT; bc.sP = ctx->g.mi + 1;
    bc.sigma_max = ctx->cond_sigma_max;
    return bc;
}

int single_only(Ctx* ctx) { ctx->err = "batched handle: only stage_*_batched / stage_vectors / step_batched apply"; return PYIPM_E_BADARG; }

// copy `count` doubles from caller memory (host or device) into library device memory
int put_vec(Ctx* ctx, double* dst, const double* src, size_t count, int memkind) {
    if (count == 0) return 0;
    if (!src) { ctx->err = "null vector pointer"; return PYIPM_E_BADARG; }
    PYIPM_HIP(hipMemcpyAsync(dst, src, count * sizeof(double),
                             memkind == PYIPM_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                             ctx->stream));
    if (memkind == PYIPM_MEM_HOST) PYIPM_HIP(hipStreamSynchronize(ctx->stream));   // host memory not retained
    return 0;
}

int stage_block(Ctx* ctx, const double* src, int64_t rows, int64_t cols, int64_t ld, int memkind,
                double** stg, size_t* stg_sz, const double** out_ptr, int64_t* out_ld) {
    if (rows == 0 || cols == 0) { *out_ptr = nullptr; *out_ld = cols; return 0; }
    if (!src || ld < cols) { ctx->err = "bad block pointer / leading dimension"; return PYIPM_E_BADARG; }
    if (memkind == PYIPM_MEM_DEVICE) { *out_ptr = src; *out_ld = ld; return 0; }
    const size_t need = (size_t)rows * (size_t)cols * sizeof(double);
    if (*stg_sz < need) {
        if (*stg) PYIPM_HIP(hipFree(*stg));
        *stg = nullptr; *stg_sz = 0;
        PYIPM_HIP(hipMalloc((void**)stg, need));
        *stg_sz = need;
    }
    PYIPM_HIP(hipMemcpy2DAsync(*stg, (size_t)cols * sizeof(double), src, (size_t)ld * sizeof(double),
                               (size_t)cols * sizeof(double), (size_t)rows, hipMemcpyHostToDevice, ctx->stream));
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    *out_ptr = *stg; *out_ld = cols;
    return 0;
}


// ---- per-panel building blocks -----------------------------------------------------------------

// Rows (= columns, by symmetry) on which a rank-K update with source columns [cA, cB) can be non-zero.
// KKT order [x | s | lambda_e | lambda_i] (pyipm.py:824-842): the (s,x) block is zero and nothing fills it, so
// while the source columns lie in the x block the s rows of L (and the s rows of W) are exact zeros; while they
// lie in the s block (diagonal Sigma, -I towards lambda_i) only the matching lambda_i rows are touched.
void active_ranges(const Ctx* ctx, int64_t cA, int64_t cB, int64_t* a0, int64_t* a1, int64_t* b0, int64_t* b1) {
    const Geo& g = ctx->g;
    *a0 = 0; *a1 = g.Npad; *b0 = 0; *b1 = 0;
    if (!ctx->skip_zeros || g.mi == 0) return;
    const int64_t s0 = g.n, s1 = g.n + g.mi, i0 = g.n + g.mi + g.me;
    if (cB <= s0)                { *a0 = 0; *a1 = s0; *b0 = s1; *b1 = g.Npad; }              // x-block sources
    else if (cA >= s0 && cB <= s1) { *a0 = i0 + (cA - s0); *a1 = i0 + (cB - s0); *b0 = 0; *b1 = 0; }   // s-block sources
}

// lower-triangle entries (i >= j) with i in rows, j in [j0, j1), both restricted to the active ranges
double active_area(int64_t row_begin, int64_t Npad, int64_t j0, int64_t j1, int64_t a0, int64_t a1, int64_t b0, int64_t b1) {
    double area = 0.0;
    const int64_t rng[2][2] = {{a0, a1}, {b0, b1}};
    for (int cj = 0; cj < 2; ++cj) {
        const int64_t ja = j0 > rng[cj][0] ? j0 : rng[cj][0], jb = j1 < rng[cj][1] ? j1 : rng[cj][1];
        if (jb <= ja) continue;
        for (int ri = 0; ri < 2; ++ri) {
            int64_t ia = rng[ri][0] > row_begin ? rng[ri][0] : row_begin, ib = rng[ri][1] < Npad ? rng[ri][1] : Npad;
            if (ib <= ia) continue;
            // sum over j in [ja,jb) of #{ i in [ia,ib) : i >= j }
            for (int64_t j = ja; j < jb; ++j) { const int64_t lo = ia > j ? ia : j; if (ib > lo) area += (double)(ib - lo); }
        }
    }
    return area;
}

// Compact, XCD-aware tile order of one swizzled launch.  The device decode (k_update, SWZ) is replayed here for
// every block; tiles that are out of range, above the diagonal or structurally zero are dropped, the eight
// per-XCD sequences (block b runs on XCD b % 8) are levelled by moving the tails of long ones to short ones,
// and the result is interleaved back into launch order.  Cached per geometry: it repeats every step.
// The list reaches the device WITHOUT a host synchronisation (round 4): built into pinned host memory, copied with
// hipMemcpyAsync on the stream of its first consumer (a plain hipMemcpy is ordered against the legacy default stream -- with
// torch's default stream as the handle's stream every list of a first factorisation waited for all the work enqueued so
// far: 35 lists, 98 ms of host time, the GPU idle behind the host for 25 ms of the first step of every solve); a consumer on
// another stream waits for the copy's event until it has been seen complete.  Device and host memory come from arenas.
int tile_list(Ctx* ctx, const UpdGeo& u, int64_t nsup, const unsigned** dev, unsigned* count, int bn = 128,
              unsigned* head_count = nullptr, hipStream_t consumer = nullptr) {
    if (ctx->debug_fault == 1 || ctx->debug_fault == 2) {   // test hook (tests/test_gpu_host_abi.py): the containers below can throw
        const int k = ctx->debug_fault; ctx->debug_fault = 0;
        if (k == 1) throw std::bad_alloc();
        throw std::runtime_error("injected fault");
    }
    std::vector<int64_t> key = {u.row_begin, u.Npad, u.first_lp, u.nrt, u.nct, u.a0, u.a1, u.b0, u.b1, u.nb, u.rt_min0, u.rt_step, bn, u.head_ct, u.sub0};
    auto it = ctx->tile_lists.find(key);
    if (it != ctx->tile_lists.end()) {
        Ctx::TileList& t = it->second;
        if (t.ready && !t.seen_done) {
            if (hipEventQuery(t.ready) == hipSuccess) t.seen_done = true;
            else if (consumer != t.on) PYIPM_HIP(hipStreamWaitEvent(consumer, t.ready, 0));
        }
        *dev = t.dev; *count = t.count; if (head_count) *head_count = t.head_count; return 0;
    }
    if (ctx->tile_lists.size() >= 1024) {              // geometries that keep changing (condensed option: |A| varies): start over
        PYIPM_HIP(hipDeviceSynchronize());
        for (auto& kv : ctx->tile_lists) if (kv.second.ready) hipEventDestroy(kv.second.ready);
        ctx->tile_lists.clear();
        for (auto& a : ctx->tl_arenas) { a.used = 0; }
    }
    const auto t_build0 = std::chrono::steady_clock::now();
    struct BuildTimer { Ctx* c; std::chrono::steady_clock::time_point t0;
                        ~BuildTimer() { c->setup_lists_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); c->setup_lists_n++; } } bt_{ctx, t_build0};
    const int64_t rounds = (nsup + 7) / 8, nblocks = rounds * 8 * SUPER * SUPER;
    const int nsr = (u.nrt + SUPER - 1) >> 3, nsc = (u.nct + SUPER - 1) >> 3;
    std::vector<unsigned> seq[8];
    for (int64_t b = 0; b < nblocks; ++b) {
        const int xcd = (int)(b & 7);
        const int64_t slot = b >> 3;
        int sidx = (int)((slot >> 6) * 8) + xcd;
        const int within = (int)(slot & 63);
        int sJ = 0, sI = -1;
        for (; sJ < nsc; ++sJ) {
            const int mn = upd_super_min_row(u, sJ);
            const int cnt = mn < nsr ? nsr - mn : 0;
            if (sidx < cnt) { sI = mn + sidx; break; }
            sidx -= cnt;
        }
        if (sI < 0) continue;
        const int64_t rt = (int64_t)sI * SUPER + (within & (SUPER - 1)), ct = (int64_t)sJ * SUPER + (within >> 3);
        if (rt >= u.nrt || ct >= u.nct) continue;
        int64_t jglob, jloc;
        if (bn == 256) upd_col<256>(u, ct, jglob, jloc); else if (bn == 64) upd_col<64>(u, ct, jglob, jloc); else upd_col<128>(u, ct, jglob, jloc);
        if (jglob >= u.Npad) continue;
        const int64_t i0 = u.row_begin + rt * BM;
        if (i0 + BM <= jglob) continue;
        const bool ri = (i0 + BM > u.a0 && i0 < u.a1) || (i0 + BM > u.b0 && i0 < u.b1);
        const bool ci = (jglob + bn > u.a0 && jglob < u.a1) || (jglob + bn > u.b0 && jglob < u.b1);
        if (!(ri && ci)) continue;
        seq[xcd].push_back((unsigned)rt | ((unsigned)ct << 16));
    }
    unsigned nhead = 0;
    if (u.head_ct > 0)                                  // fused head: the next group's columns first, on every XCD's sequence
        for (auto& v : seq) {
            std::stable_partition(v.begin(), v.end(), [&](unsigned c) { return (int)(c >> 16) < u.head_ct; });
            for (unsigned c : v) nhead += (int)(c >> 16) < u.head_ct;
        }
    size_t total = 0;
    for (auto& v : seq) total += v.size();
    const size_t target = (total + 7) / 8;
    std::vector<unsigned> spare;
    for (auto& v : seq) while (v.size() > target) { spare.push_back(v.back()); v.pop_back(); }
    for (auto& v : seq) while (v.size() < target && !spare.empty()) { v.push_back(spare.back()); spare.pop_back(); }
    if (u.head_ct > 0)                                  // (levelling appends moved tails: keep the head tiles in front)
        for (auto& v : seq) std::stable_partition(v.begin(), v.end(), [&](unsigned c) { return (int)(c >> 16) < u.head_ct; });
    std::vector<unsigned> list(8 * target, 0xffffffffu);
    for (int x = 0; x < 8; ++x) for (size_t j = 0; j < seq[x].size(); ++j) list[8 * j + x] = seq[x][j];
    Ctx::TileList tl;
    tl.count = (unsigned)list.size(); tl.head_count = nhead;
    if (tl.count) {
        // a slot in the arenas (device + pinned host, 1 Mi entries each; a list longer than that gets an arena of its own)
        const size_t need = (list.size() + 63) / 64 * 64;
        Ctx::TlArena* ar = nullptr;
        for (auto& a : ctx->tl_arenas) if (a.cap - a.used >= need) { ar = &a; break; }
        if (!ar) {
            Ctx::TlArena a;
            a.cap = need > ((size_t)1 << 20) ? need : ((size_t)1 << 20);
            PYIPM_HIP(hipMalloc((void**)&a.dev, a.cap * sizeof(unsigned)));
            if (hipHostMalloc((void**)&a.host, a.cap * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) {
                hipFree(a.dev); ctx->err = "no pinned host memory for the tile lists"; return PYIPM_E_NOMEM; }
            ctx->tl_arenas.push_back(a);
            ar = &ctx->tl_arenas.back();
        }
        tl.dev = ar->dev + ar->used;
        unsigned* hsrc = ar->host + ar->used;
        ar->used += need;
        memcpy(hsrc, list.data(), list.size() * sizeof(unsigned));
        PYIPM_HIP(hipMemcpyAsync(tl.dev, hsrc, list.size() * sizeof(unsigned), hipMemcpyHostToDevice, consumer));
        PYIPM_HIP(hipEventCreateWithFlags(&tl.ready, hipEventDisableTiming));
        PYIPM_HIP(hipEventRecord(tl.ready, consumer));
        tl.on = consumer;
    }
    ctx->tile_lists[key] = tl;
    *dev = tl.dev; *count = tl.count; if (head_count) *head_count = nhead;
    return 0;
}

// Rank-K update of `n_lp` locally owned panels starting at local panel `first_lp`.
// ldw / row_end / col_end default to the KKT storage's (the Gram launch of the condensed option narrows them).
int launch_update128(Ctx* ctx, hipStream_t stream, const double* Lop, int64_t ldl, const double* Wop, int K,
                     int64_t row_begin, int64_t first_lp, int64_t n_lp, bool bulk,
                     int64_t ldw, int64_t row_end, int64_t col_end, int64_t src_c0,
                     int ksplit, int64_t ks_cstride, int waves,        // waves: 0 = the handle's bulk_waves
                     int head_ct, unsigned* head_counter, unsigned* head_count, bool list_only,
                     int sub0, int nct_sub,    // nct_sub > 0: only column tiles [sub0, sub0 + nct_sub) of local panel first_lp
                     int* used_bn,           // out: the tile width of the instance that ran (128 / 256)
                     int prio) {                  // >= 0: wave priority flag of the launch whatever `bulk` says
    const Geo& g = ctx->g;                             // (128 wide; the sub-panels of a wide panel, factor_block)
    if (ldw <= 0) ldw = g.Npad;
    if (row_end <= 0) row_end = g.Npad;
    if (col_end <= 0) col_end = g.Npad;
    const int64_t m = row_end - row_begin;
    if (m <= 0 || n_lp <= 0) return 0;
    UpdGeo u;
    u.row_begin = row_begin; u.Npad = col_end; u.first_lp = first_lp; u.sub0 = sub0;
    u.nb = g.nb; u.world = g.world; u.rank = g.rank;
    u.nrt = (int)(m / BM); u.nct = nct_sub > 0 ? nct_sub : (int)(n_lp * (g.nb / 128));
    u.dbg = ctx->dbg_buf;
    u.prio = prio >= 0 ? prio : (bulk ? 0 : ctx->side_prio);
    u.rt_min0 = 0; u.rt_step = 0; u.tiles = nullptr; u.ks_cstride = 0; u.persist = 0;
    u.head_ct = head_ct; u.head_counter = head_counter;
    if (src_c0 >= 0) active_ranges(ctx, src_c0, src_c0 + K, &u.a0, &u.a1, &u.b0, &u.b1);
    else { u.a0 = 0; u.a1 = g.Npad; u.b0 = 0; u.b1 = 0; }
    const int use_waves = waves ? waves : ctx->bulk_waves;
    // (col_end % 256: Npad is a multiple of 128 only -- a 256-wide tile at the last 128 columns would read and rewrite 128
    // columns past the storage, i.e. the first W slot, and W rows past Npad; such shapes keep 128 x 128 tiles)
    if (ctx->xcd_swizzle && bulk && ctx->bulk_bn == 256 && use_waves == 8 && waves == 0 && ksplit == 1 && g.nb % 256 == 0 &&
        col_end % 256 == 0 && K >= ctx->bulk_bn_min_k && head_ct % 2 == 0 && nct_sub == 0 && (ctx->reserve_cus <= 0 || m > (ctx->bulk_bn_rows > ctx->persist_rows ? ctx->bulk_bn_rows : ctx->persist_rows))) {
        // 128 x 256 tiles (the K = 1024 bulk launches of the single-rank schedule, the K = nb launches of the per-panel one)
        u.nct = (int)(n_lp * (g.nb / 256));
        u.head_ct = head_ct / 2;                        // (fused head: the callers count 128-column tiles)
        upd_fill_affine<256>(u);
        const int64_t nsup = upd_super_count<256>(u);
        if (nsup <= 0) return 0;
        unsigned ntiles = 0;
        int rc = tile_list(ctx, u, nsup, &u.tiles, &ntiles, 256, head_count, stream); if (rc) return rc;
        if (used_bn) *used_bn = 256;
        if (list_only || ntiles == 0) return 0;
        hipLaunchKernelGGL((k_update<256, true, 8>), dim3(ntiles), dim3(512), 0, stream, ctx->A, g.Npad, Lop, ldl, Wop, ldw, K, u);
        PYIPM_KCHECK();
        if (used_bn) *used_bn = 256;
        return 0;
    }
    if (ctx->xcd_swizzle && bulk && upd_swizzle_ok<128>(u)) {
        upd_fill_affine<128>(u);
        const int64_t nsup = upd_super_count<128>(u);
        if (nsup <= 0) return 0;
        unsigned ntiles = 0;
        int rc = tile_list(ctx, u, nsup, &u.tiles, &ntiles, 128, head_count, stream); if (rc) return rc;
        if (list_only || ntiles == 0) return 0;
        if (ctx->bulk_bn == 256 && use_waves == 8 && waves == 0 && ksplit == 1 && head_ct == 0 && nct_sub == 0 &&
            ntiles <= (unsigned)(ctx->num_cus * 5 / 8) && upd_swizzle_ok<64>(u)) {
            // A launch that leaves most of the GPU idle (the last groups of a factorisation, the panel-by-panel updates of a small
            // system's first group): 128 x 64 tiles, twice as many blocks of half the length, no scratch.  (Config 2's K = 2048
            // launch of 136 tiles, 272 us: unchanged by itself -- 272 half tiles on 256 CUs end when 136 whole ones do -- but the K = 256
            // pieces of lookahead = 2 gain 2 %.  As the instance of EVERY narrow launch at N = 32768 it lost 1 %: tools/r06_bn64.sh.)
            // Every entry still receives its products k ascending in MFMA groups of 4: the same bits.
            UpdGeo v = u;
            v.nct = (int)(n_lp * (g.nb / 64)); v.tiles = nullptr;
            upd_fill_affine<64>(v);
            const int64_t nsup64 = upd_super_count<64>(v);
            unsigned nt64 = 0;
            if (nsup64 > 0) { rc = tile_list(ctx, v, nsup64, &v.tiles, &nt64, 64, nullptr, stream); if (rc) return rc; }
            if (nt64 > 0) {
                hipLaunchKernelGGL((k_update<64, true, 8>), dim3(nt64), dim3(512), 0, stream, ctx->A, g.Npad, Lop, ldl, Wop, ldw, K, v);
                PYIPM_KCHECK();
                if (used_bn) *used_bn = 64;
                return 0;
            }
        }
        dim3 grid(ntiles, (unsigned)ksplit);             // K is the length of ONE split
        if (ksplit > 1) u.ks_cstride = ks_cstride;
        if (use_waves == 8 && waves == 0 && ksplit == 1 && ctx->reserve_cus > 0 &&
            m <= ctx->persist_rows) {
            // chain-bound phase: leave `reserve_cus` CUs without update blocks (two 8-wave blocks fill one)
            const unsigned slots = (unsigned)(2 * (ctx->num_cus - ctx->reserve_cus)) & ~7u;
            if (slots >= 8 && ntiles > slots) { u.persist = ntiles; grid.x = slots; }
        }
        if (use_waves == 8)
            hipLaunchKernelGGL((k_update<128, true, 8>), grid, dim3(512), 0, stream, ctx->A, g.Npad, Lop, ldl, Wop, ldw, K, u);
        else
            hipLaunchKernelGGL((k_update<128, true>), grid, dim3(256), 0, stream, ctx->A, g.Npad, Lop, ldl, Wop, ldw, K, u);
    } else {
        if (ksplit > 1) { ctx->err = "split-K launches need the tile-list order (xcd_swizzle)"; return PYIPM_E_BADARG; }
        dim3 grid((unsigned)u.nrt, (unsigned)u.nct);
        hipLaunchKernelGGL((k_update<128, false>), grid, dim3(256), 0, stream, ctx->A, g.Npad, Lop, ldl, Wop, ldw, K, u);
    }
    PYIPM_KCHECK();
    return 0;
}

// Per-panel mode (no group schedule: the multi-rank driver): a panel inside the slack block needs no tile chain,
// no message and no update launch (k_s_panel / k_s_schur_sigma).
bool panel_in_s(const Ctx* ctx, int64_t p) {
    const Geo& g = ctx->g;
    if (!ctx->s_fast || !ctx->skip_zeros || g.mi == 0 || !ctx->grp_of.empty() || ctx->cond_active) return false;
    const int64_t c0 = g.panel_c0(p);
    return c0 >= g.n && c0 + g.panel_w(p) <= g.n + g.mi;
}

// Per-panel mode: rows [h0, h1) of a panel that lies inside the x block are whole 128-row tiles of slack rows --
// exact zeros that no update launch with these source columns ever reads (active_ranges).  The owner's chain
// kernels skip them and the panel message leaves them out.  Empty range when nothing can be skipped.
// Rows of exact zeros (whole slack rows of x-block columns) that the chain kernels neither compute nor WRITE into W.  Whole
// HOLEG-row blocks only, HOLEG = the widest column tile of an update launch: a 128 x 256 tile whose columns straddled the edge of
// a 128-aligned hole read W rows nobody had written -- stale memory, silently wrong factors whenever n or n + mi is an odd
// multiple of 128 and the workspace is not fresh (found by tests/test_gpu_configs.py::test_oracle_lu_where_the_headline_runs,
// round 4; the benchmark shapes are 256-aligned and were never affected).
constexpr int64_t HOLEG = 256;
inline void slack_hole(const Geo& g, int64_t* h0, int64_t* h1) {
    *h0 = (g.n + HOLEG - 1) / HOLEG * HOLEG; *h1 = (g.n + g.mi) / HOLEG * HOLEG;
    if (*h1 < *h0) *h1 = *h0;
}

void panel_hole(const Ctx* ctx, int64_t p, int64_t* h0, int64_t* h1) {
    const Geo& g = ctx->g;
    *h0 = 0; *h1 = 0;
    if (!ctx->skip_zeros || g.mi == 0 || !ctx->grp_of.empty() || ctx->cond_active) return;
    if (g.panel_c0(p) + g.panel_w(p) > g.n) return;
    int64_t a, b;
    slack_hole(g, &a, &b);
    if (b > a) { *h0 = a; *h1 = b; }
}

int factor_wide_panel(Ctx* ctx, int64_t p, hipStream_t stream);
int launch_tile_steps(Ctx* ctx, hipStream_t chain, int64_t gc0, int64_t glc0, int nT, int ta, int tb, double* Wg, double* Dv, double* Ts,
                      int nX = 0, const unsigned* xword = nullptr, unsigned xwant = 0);

// Factor panel p on `stream`.  apply_pending: first apply the earlier panels of p's group to p's columns
// (grouped single-rank driver; their bulk update is deferred to the end of the group).
int factor_panel(Ctx* ctx, int64_t p, hipStream_t stream, bool apply_pending) {
    const Geo& g = ctx->g;
    if (p < 0 || p >= g.npanels || g.owner(p) != g.rank) { ctx->err = "factor_panel: not the owner"; return PYIPM_E_BADARG; }
    const int64_t c0 = g.panel_c0(p), lc0 = g.local_c0(p);
    const int nbw = (int)g.panel_w(p);
    const int nt = nbw / TB;
    const int64_t lp = p / g.world;
    if (((size_t)p < ctx->grp_of.size() && !ctx->grp_fast.empty() && ctx->grp_fast[(size_t)ctx->grp_of[p]]) || panel_in_s(ctx, p)) {
        // the whole group (or, per-panel mode, the panel) lies inside the slack block: closed form, no W, no updates
        hipLaunchKernelGGL(k_s_panel, dim3(1), dim3(256), 0, stream, ctx->A, g.Npad, c0, lc0, nt,
                           ctx->Dinv + (c0 / TB) * (int64_t)(TB * TB), ctx->Tsv + (c0 / TB) * (int64_t)(TB * TB),
                           ctx->Tflag + c0 / TB, ctx->refine_cond, ctx->block_refine, ctx->dstats,
                           g.n, g.n + g.mi + g.me, ctx->pivtol_rel, ctx->anorm);
        PYIPM_KCHECK();
        return 0;
    }
    double* W = wbuf(ctx, p);
    const int64_t poff = (size_t)p < ctx->grp_off.size() ? ctx->grp_off[p] : p % ctx->group;
    if (apply_pending && poff != 0) {
        const int64_t p0 = p - poff;
        const int K = (int)((p - p0) * g.nb);
        if (g.Npad - c0 <= ctx->pending32_rows) {
            int64_t pa0, pa1, pb0, pb1;
            active_ranges(ctx, g.panel_c0(p0), g.panel_c0(p0) + K, &pa0, &pa1, &pb0, &pb1);
            hipLaunchKernelGGL(k_inpanel_update, dim3((unsigned)((g.Npad - c0) / 32), (unsigned)(nbw / TB)), dim3(256), 0, stream,
                               ctx->A, g.Npad, lc0, ctx->A + g.local_c0(p0) * g.Npad, g.Npad, wbuf(ctx, p0), g.Npad, c0, K, c0,
                               g.Npad, pa0, pa1, pb0, pb1, ctx->side_prio);
            PYIPM_KCHECK();
        } else {
            int rc = launch_update128(ctx, stream, ctx->A + g.local_c0(p0) * g.Npad, g.Npad, wbuf(ctx, p0),
                                      K, c0, lp, 1, /*bulk=*/false, 0, 0, 0, g.panel_c0(p0));
            if (rc) return rc;
        }
    }
    // Panels inside the x block: the slack rows of the panel are exact zeros (active_ranges), whole 128-row tiles of
    // them are skipped by the in-panel updates and the scalings too (single rank only: a receiver rebuilds L from W).
    int64_t ha0 = 0, ha1 = g.Npad, hb0 = 0, hb1 = 0, hole0 = 0, hole1 = 0;
    // Only when the whole GROUP lies in the x block: a group's bulk update reads every W row the structure of its
    // source columns allows, so a mixed group needs all of them written.
    const bool grp_in_x = (size_t)p < ctx->grp_of.size() && !ctx->grp_x.empty() && ctx->grp_x[(size_t)ctx->grp_of[p]];
    if (ctx->skip_zeros && g.world == 1 && g.mi > 0 && grp_in_x) {
        active_ranges(ctx, c0, c0 + nbw, &ha0, &ha1, &hb0, &hb1);
        slack_hole(g, &hole0, &hole1);
    } else {
        panel_hole(ctx, p, &hole0, &hole1);             // per-panel mode (any number of ranks)
        if (hole1 > hole0) active_ranges(ctx, c0, c0 + nbw, &ha0, &ha1, &hb0, &hb1);
    }
    if (ctx->tile_step && ctx->per_panel_mode && ctx->wide_sub >= 128 && ctx->wide_sub % 128 == 0 &&
        nbw > ctx->wide_sub && nt <= 32 && c0 + nbw < g.Npad)
        return factor_wide_panel(ctx, p, stream);
    if (ctx->tile_step && nt <= 16) {
        // stepped schedule: launch t inverts tile t (after eliminating tile t - 1 from the rows of the diagonal block), one
        // more launch runs all stages for the rows below the diagonal block
        double* Dv = ctx->Dinv + (c0 / TB) * (int64_t)(TB * TB);
        double* Ts = ctx->Tsv + (c0 / TB) * (int64_t)(TB * TB);
        { int r0 = launch_tile_steps(ctx, stream, c0, lc0, nt, 0, nt, W, Dv, Ts); if (r0) return r0; }
        const int64_t rb = c0 + nbw;
        if (g.Npad > rb) {
            hipLaunchKernelGGL(k_panel_rest, dim3((unsigned)((g.Npad - rb) / TB)), dim3(256), 0, stream, ctx->A, g.Npad, c0, lc0, nt, rb,
                               W, g.Npad, Dv, Ts, ctx->Tflag + c0 / TB, ctx->block_refine, hole0, hole1, &ctx->dstats->growth_bits);
            PYIPM_KCHECK();
        }
        return 0;
    }
    const bool fused = ctx->fuse_su != 0;                   // a tile's scaling launch also updates the next column block
    for (int t = 0; t < nt; ++t) {
        const int64_t j0 = c0 + (int64_t)t * TB, lcol = lc0 + (int64_t)t * TB;
        const int64_t below = g.Npad - (j0 + TB);
        if (t > 0 && !fused) {
            // left-looking in-panel update of this tile's column block with the t tiles before it (32-row blocks)
            const int64_t row_begin = (j0 / 32) * 32;
            hipLaunchKernelGGL(k_inpanel_update, dim3((unsigned)((g.Npad - row_begin) / 32)), dim3(256), 0, stream,
                               ctx->A, g.Npad, lcol, ctx->A + lc0 * g.Npad, g.Npad, W, g.Npad, j0, t * TB, row_begin, g.Npad,
                               ha0, ha1, hb0, hb1, ctx->side_prio);
            PYIPM_KCHECK();
        }
        const bool next = fused && t + 1 < nt && below > 0;   // the scaling launch below carries the update of tile t + 1
        hipLaunchKernelGGL(k_tile_invert, dim3(1), dim3(256), 0, stream, ctx->A, g.Npad, j0, lcol,
                           ctx->Dinv + (j0 / TB) * (int64_t)(TB * TB), ctx->Tsv + (j0 / TB) * (int64_t)(TB * TB),
                           ctx->Tflag + j0 / TB, ctx->refine_cond, ctx->dstats, g.N, ctx->pivtol_rel, ctx->anorm,
                           g.n + g.mi, ctx->dbg_buf, next ? ctx->Wnext : (double*)nullptr, ctx->tile_blocked);
        PYIPM_KCHECK();
        if (below > 0) {
            NextUpd nu; memset(&nu, 0, sizeof(nu));
            if (next) {
                nu.on = 1; nu.C = ctx->A; nu.ldc = g.Npad; nu.ccol = lcol + TB;
                nu.Lop = ctx->A + lc0 * g.Npad; nu.ldl = g.Npad;
                nu.Wop = W; nu.ldw = g.Npad; nu.cglob = j0 + TB; nu.Wnext = ctx->Wnext; nu.K = t * TB;
            }
            hipLaunchKernelGGL(k_panel_scale, dim3((unsigned)(below / TB)), dim3(256), 0, stream,
                               ctx->A, g.Npad, lcol, ctx->A, g.Npad, lcol, W, g.Npad, (int64_t)t * TB,
                               ctx->Dinv + (j0 / TB) * (int64_t)(TB * TB), ctx->Tsv + (j0 / TB) * (int64_t)(TB * TB),
                               ctx->Tflag + j0 / TB, ctx->block_refine, j0 + TB, hole0, hole1, &ctx->dstats->growth_bits, 1.0, nu);
            PYIPM_KCHECK();
        }
    }
    return 0;
}

int ensure_rest_stream(Ctx* ctx) {
    if (!ctx->rest) {
        int lo = 0, hi = 0;
        PYIPM_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        PYIPM_HIP(hipStreamCreateWithPriority(&ctx->rest, hipStreamNonBlocking, ctx->rest_prio ? hi : lo));
    }
    return 0;
}

// One block of adjacent sub-panels owned by this rank, chained tile to tile (kernels_panel.hpp): a GROUP of panels of the
// single-rank schedule (factor_group) or ONE wide panel of the per-panel schedule cut into sub-panels (factor_wide_panel).
// The diagonal block of the WHOLE block runs as one sequence of k_tile_step launches on `chain`: a tile waits for nothing
// but the tile before it, and what a per-panel schedule does between two panels (the pending update of the next panel's
// columns, the scaling of all rows below) no longer sits between two tile inversions.  The rows below the diagonal block
// follow on ctx->rest, sub-panel by sub-panel: all stages of the sub-panel (k_panel_rest), then its contribution to the
// later sub-panels of the block (right-looking: the pending update, one source at a time -- the same products in the same
// order).  on_done(id, stream): sub-panel `id` is complete once `stream` reaches this point.
// Tile steps [ta, tb) of a diagonal block of nT tiles on `chain`: one k_tile_step launch per tile.
// does the diagonal block at global column gc0 (nT tiles) take k_tile_chain (kernels_chain.hpp)?
static bool chain_applies(const Ctx* ctx, int64_t gc0, int nT) {
    const Geo& g = ctx->g;
    const bool exposed = ctx->per_panel_mode || gc0 == 0 || g.Npad - gc0 <= ctx->tile8_rows;   // (where the chain is what the step waits for)
    return ctx->tile_chain && nT >= 2 && nT <= 32 && (ctx->tile_chain >= 2 || exposed);
}
// nX > 0 (k_tile_chain only, ta = 0, tb = nT): nX row tiles right below the diagonal block take every stage in the same launch,
// once *xword has reached xwant (chain_extra_ok says whether a caller may ask for it).
int launch_tile_steps(Ctx* ctx, hipStream_t chain, int64_t gc0, int64_t glc0, int nT, int ta, int tb, double* Wg, double* Dv, double* Ts,
                      int nX, const unsigned* xword, unsigned xwant) {
    const Geo& g = ctx->g;
    if (tb <= ta) return 0;
    if (tb - ta >= 2 && chain_applies(ctx, gc0, nT)) {
        // the steps [ta, tb) as ONE launch of persistent workgroups (kernels_chain.hpp): no launch boundary between two tiles
        if (!ctx->chain_sync) {
            PYIPM_HIP(hipMalloc((void**)&ctx->chain_sync, (size_t)(Ctx::CHAIN_SLOTS * Ctx::CHAIN_WORDS + 1) * sizeof(unsigned)));
            PYIPM_HIP(hipMemset(ctx->chain_sync, 0, (size_t)(Ctx::CHAIN_SLOTS * Ctx::CHAIN_WORDS + 1) * sizeof(unsigned)));
            // (the fill is ordered on the NULL stream; k_chain_wait polls these words from a non-blocking stream: it met the words of
            //  an earlier handle's life in this memory -- larger epochs: "done" -- and let the rows kernels run ahead of the chain, on
            //  the first step of a handle only: tools/chain_stress.py.  Once per handle: wait for the fill.)
            PYIPM_HIP(hipDeviceSynchronize());
            ctx->chain_epoch = 0;
        }
        if (ctx->chain_epoch >= (1u << 24)) {            // the words' epochs must stay ordered: start over (every 16 M launches)
            PYIPM_HIP(hipDeviceSynchronize());
            PYIPM_HIP(hipMemset(ctx->chain_sync, 0, (size_t)(Ctx::CHAIN_SLOTS * Ctx::CHAIN_WORDS) * sizeof(unsigned)));
            ctx->chain_epoch = 0;
        }
        ChainGeo cg;
        cg.ta = ta; cg.tb = tb; cg.nT = nT; cg.cpy = ctx->chain_cpy > 0 ? ctx->chain_cpy : 5;
        cg.nR = nT + ((ta == 0 && tb == nT) ? nX : 0); cg.xword = xword; cg.xwant = xwant;
        ctx->chain_epoch += 1;
        cg.base = ctx->chain_epoch * 64u;
        cg.sync = ctx->chain_sync + (size_t)(ctx->chain_epoch % Ctx::CHAIN_SLOTS) * Ctx::CHAIN_WORDS;
        cg.err = ctx->chain_sync + (size_t)Ctx::CHAIN_SLOTS * Ctx::CHAIN_WORDS;
        cg.timeout = (unsigned long long)2.0e8;          // 2 s (100 MHz clock)
        cg.dbg = nullptr;
        if (ctx->chain_dbg && ctx->chain_dbg_launch < 64) cg.dbg = ctx->chain_dbg + (size_t)(ctx->chain_dbg_launch++) * CHAIN_DBG_WORDS;
        const unsigned nblk = 1u + (unsigned)chain_units(ta, nT, cg.cpy, cg.nR);
        // dynamic shared memory nobody touches: with it a workgroup of the chain has its compute unit to itself (single-rank
        // schedule only: beside the bulk updates of the per-panel schedule it would wait for a whole CU to drain)
        size_t pad = (ctx->chain_lds_kb > 0 && !ctx->per_panel_mode) ? (size_t)ctx->chain_lds_kb * 1024 : 0;
        if (pad && !ctx->chain_lds_set) {
            if (hipFuncSetAttribute((const void*)k_tile_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024) != hipSuccess) { (void)hipGetLastError(); ctx->chain_lds_kb = 0; pad = 0; }
            ctx->chain_lds_set = true;
        }
        hipLaunchKernelGGL(k_tile_chain, dim3(nblk), dim3(256), pad, chain, ctx->A, g.Npad, gc0, glc0, Wg, g.Npad, Dv, Ts,
                           ctx->Tflag + gc0 / TB, ctx->refine_cond, ctx->block_refine, ctx->dstats, g.N, ctx->pivtol_rel, ctx->anorm,
                           g.n + g.mi, ctx->tile_blocked, cg, ctx->dbg_buf);
        PYIPM_KCHECK();
        ctx->chain_used = true;
        ctx->chain_last = cg;
        return 0;
    }
    if (nX > 0) { ctx->err = "launch_tile_steps: extra rows need the one-launch chain (chain_extra_ok)"; return PYIPM_E_BADARG; }
    for (int j = ta; j < tb; ++j) {
        int ny = (nT - j + 4) / 5; if (ny < 1) ny = 1; if (ny > 4) ny = 4;     // <= ~5 column tiles per block
        // the critical block on eight waves (chain + helpers) where a whole CU is to be had: a 512-thread block owns the
        // registers of its CU, so beside an ordinary bulk launch it would wait for BOTH resident bulk blocks of a CU to retire
        // (all launches: exposed panel 7.1 -> 10.9 ms at N = 32768; the per-panel schedule, whose ranks run their bulk updates
        // beside every chain: owners' chain path 31.0 -> 32.1 ms in the replay -- it keeps the 256-thread kernel)
        const int64_t m_left = g.Npad - gc0;
        if (ctx->tile_waves == 8 && (ctx->per_panel_mode ? ctx->tile8_dist != 0 : (gc0 == 0 || m_left <= ctx->tile8_rows))) {
            const int units = (nT - j - 1) * ny;
            const int free_cus = (gc0 == 0 || ctx->reserve_cus <= 0 || ctx->per_panel_mode) ? ctx->num_cus : ctx->tile_free_cus;
            const int upb = ctx->tile_upb > 0 ? ctx->tile_upb : (units + 1 <= free_cus ? 1 : 2);
            const unsigned nblk = 1u + (unsigned)((units + upb - 1) / upb);
            hipLaunchKernelGGL(k_tile_step8, dim3(nblk), dim3(512), 0, chain, ctx->A, g.Npad, gc0, glc0, j, nT, ny, upb,
                               Wg, g.Npad, Dv, Ts, ctx->Tflag + gc0 / TB, ctx->refine_cond, ctx->block_refine, ctx->dstats, g.N,
                               ctx->pivtol_rel, ctx->anorm, g.n + g.mi, ctx->dbg_buf, ctx->tile_blocked);
            PYIPM_KCHECK();
            continue;
        }
        hipLaunchKernelGGL(k_tile_step, dim3((unsigned)(nT - j), (unsigned)ny), dim3(256), 0, chain, ctx->A, g.Npad, gc0, glc0, j,
                           Wg, g.Npad, Dv, Ts, ctx->Tflag + gc0 / TB, ctx->refine_cond, ctx->block_refine, ctx->dstats, g.N,
                           ctx->pivtol_rel, ctx->anorm, g.n + g.mi, ctx->dbg_buf, ctx->tile_blocked);
        PYIPM_KCHECK();
    }
    return 0;
}

struct SubPanel { int64_t c0, lc0; int nt; double* W; int64_t id; };
struct BlockDesc {
    std::vector<SubPanel> sp;              // adjacent in global AND local columns; W adjacent too (sp[k].W = sp[0].W + (c0_k - c0_0) * Npad)
    bool wide = false;                     // sub-panels of local panel `lp` (column tiles addressed inside it) instead of whole panels
    int64_t lp = 0;                        // wide: the local panel index; else: sp[k].id is the (local = global) panel index
    int64_t hole0 = 0, hole1 = 0;          // rows of exact zeros the rows kernels skip (KKT structure)
};
int factor_block(Ctx* ctx, const BlockDesc& bd, hipStream_t chain, const std::function<int(int64_t, hipStream_t)>& on_done) {
    const Geo& g = ctx->g;
    const int64_t n0 = (int64_t)bd.sp.size();
    std::vector<int> toff((size_t)n0 + 1, 0);           // first tile of each sub-panel inside the block (the matrix's last panel may be narrower)
    for (int64_t k = 0; k < n0; ++k) toff[(size_t)k + 1] = toff[(size_t)k] + bd.sp[(size_t)k].nt;
    const int nT = toff[(size_t)n0];
    const int64_t gc0 = bd.sp[0].c0, glc0 = bd.sp[0].lc0, gend = gc0 + (int64_t)nT * TB;
    const int64_t TT = (int64_t)TB * TB;
    { int r0 = ensure_rest_stream(ctx); if (r0) return r0; }
    if (!ctx->ev_join) PYIPM_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    while ((int64_t)ctx->ev_band.size() < n0) { hipEvent_t e; PYIPM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->ev_band.push_back(e); }
    double* Wg = bd.sp[0].W;
    double* Dv = ctx->Dinv + (gc0 / TB) * TT;
    double* Ts = ctx->Tsv + (gc0 / TB) * TT;
    // update of the block's own columns, rows from `row_begin`: target sub-panels [k1, k1 + cnt)
    auto upd = [&](hipStream_t st, const double* Lop, const double* Wop, int K, int64_t row_begin, int64_t k1, int64_t cnt,
                   int64_t src_c0) -> int {
        if (!bd.wide)
            return launch_update128(ctx, st, Lop, g.Npad, Wop, K, row_begin, bd.sp[(size_t)k1].id, cnt, /*bulk=*/false, 0, 0, 0, src_c0);
        const int s0 = (int)((bd.sp[(size_t)k1].c0 - g.panel_c0(bd.lp * g.world + g.rank)) / 128);
        const int nc = (int)((int64_t)(toff[(size_t)(k1 + cnt)] - toff[(size_t)k1]) * TB / 128);
        return launch_update128(ctx, st, Lop, g.Npad, Wop, K, row_begin, bd.lp, 1, /*bulk=*/false, 0, 0, 0, src_c0,
                                1, 0, 0, 0, nullptr, nullptr, false, s0, nc);
    };
    // sub-panel q's columns are final inside the diagonal block once the first tile of sub-panel q + 1 has applied its last
    // stage: the tile steps run in pieces that END with that step, an event after each
    // (k_tile_chain: the whole diagonal block is ONE launch; the rows of sub-panel q wait on their own stream for the chain's
    //  progress words instead -- k_chain_wait below)
    const bool one_launch = ctx->chain_whole && chain_applies(ctx, gc0, nT);
    ChainGeo cgw; memset(&cgw, 0, sizeof(cgw));
    if (one_launch) {
        // (the rows stream polls the chain's progress with a bounded wait: it must not start polling before the chain can start --
        //  behind a Gram launch of seconds, the condensed form at n = 65536, the bound ran out while the chain's stream was still busy)
        PYIPM_HIP(hipEventRecord(ctx->ev_band[0], chain));
        PYIPM_HIP(hipStreamWaitEvent(ctx->rest, ctx->ev_band[0], 0));
        int r0 = launch_tile_steps(ctx, chain, gc0, glc0, nT, 0, nT, Wg, Dv, Ts); if (r0) return r0;
        cgw = ctx->chain_last;
    }
    for (int kp = 1, ta = 0; kp <= n0 && !one_launch; ++kp) {
        const int tb = kp < n0 ? toff[(size_t)kp] + 1 : nT;
        int r0 = launch_tile_steps(ctx, chain, gc0, glc0, nT, ta, tb, Wg, Dv, Ts); if (r0) return r0;
        if (kp < n0) PYIPM_HIP(hipEventRecord(ctx->ev_band[(size_t)(kp - 1)], chain));
        ta = tb;
    }
    for (int64_t k = 0; k < n0; ++k) {
        const SubPanel& q = bd.sp[(size_t)k];
        const int64_t c0 = q.c0, lc0 = q.lc0;
        const int nt = q.nt;
        // the last sub-panel's rows are what the next block waits for: they stay on the chain's stream (a dependency across
        // streams costs 10-30 us when the waiting side is idle), behind the other sub-panels' work on ctx->rest
        hipStream_t rs = ctx->rest;
        if (k + 1 == n0) {
            PYIPM_HIP(hipEventRecord(ctx->ev_join, ctx->rest));
            PYIPM_HIP(hipStreamWaitEvent(chain, ctx->ev_join, 0));
            rs = chain;
        } else if (one_launch) {
            // sub-panel k's tiles are inverted (the chain has passed tile toff[k + 1] - 1) and every later row tile has the stages
            // up to the sub-panel's last tile but one: W of the sub-panel's columns is final inside the diagonal block
            hipLaunchKernelGGL(k_chain_wait, dim3(1), dim3(64), 0, ctx->rest, cgw, toff[(size_t)k + 1], toff[(size_t)k + 1], toff[(size_t)k + 1]);
            PYIPM_KCHECK();
        } else {
            PYIPM_HIP(hipStreamWaitEvent(ctx->rest, ctx->ev_band[(size_t)k], 0));
        }
        // Where the bulk update is the bound (many rows below), the in-block updates of the rows below the diagonal block
        // are applied LEFT-looking -- all earlier sub-panels at once (K up to 768), right before the sub-panel's own
        // stages -- instead of one by one: half the read-modify-write passes over those rows and longer K per launch.
        // In the tail the one-by-one form keeps only the last sub-panel's stages behind the chain.  Same bits either way.
        const bool left = ctx->pending_left_rows >= 0 && g.Npad - gend > ctx->pending_left_rows;
        if (g.Npad > gend && left && k > 0) {
            const int K = toff[(size_t)k] * TB;
            int rc = upd(rs, ctx->A + glc0 * g.Npad, Wg, K, gend, k, 1, gc0);
            if (rc) return rc;
        }
        if (g.Npad > gend) {
            hipLaunchKernelGGL(k_panel_rest, dim3((unsigned)((g.Npad - gend) / TB)), dim3(256), 0, rs, ctx->A, g.Npad, c0, lc0, nt,
                               gend, q.W, g.Npad, ctx->Dinv + (c0 / TB) * TT, ctx->Tsv + (c0 / TB) * TT, ctx->Tflag + c0 / TB,
                               ctx->block_refine, bd.hole0, bd.hole1, &ctx->dstats->growth_bits);
            PYIPM_KCHECK();
            if (k + 1 < n0 && !left) {
                // this sub-panel's contribution to the later ones of the block, rows below the diagonal block
                const int K = nt * TB;
                const SubPanel& nx = bd.sp[(size_t)k + 1];
                const int64_t tc0 = nx.c0, ncols = (int64_t)(nT - toff[(size_t)k + 1]) * TB;
                if (g.Npad - tc0 <= ctx->pending32_rows) {
                    int64_t pa0, pa1, pb0, pb1;
                    active_ranges(ctx, c0, c0 + K, &pa0, &pa1, &pb0, &pb1);
                    hipLaunchKernelGGL(k_inpanel_update, dim3((unsigned)((g.Npad - gend) / 32), (unsigned)(ncols / TB)), dim3(256), 0,
                                       rs, ctx->A, g.Npad, nx.lc0, ctx->A + lc0 * g.Npad, g.Npad, q.W, g.Npad,
                                       tc0, K, gend, g.Npad, pa0, pa1, pb0, pb1, ctx->side_prio);
                    PYIPM_KCHECK();
                } else {
                    int rc = upd(rs, ctx->A + lc0 * g.Npad, q.W, K, gend, k + 1, n0 - 1 - k, c0);
                    if (rc) return rc;
                }
            }
        }
        // One launch for the whole chain: the rows stream ran ahead of it on progress words that say what the ROWS kernels read (W,
        // inv(T)) is there.  Whoever reads panel q as a whole (the forward substitution that trails the factorisation) needs L of its
        // columns inside the diagonal block too, i.e. the state after the tile step of the next sub-panel's first tile (what an event
        // behind that launch used to say): a second wait, behind the rows kernels.  L, like W and inv(T), is written through.
        if (one_launch && k + 1 < n0) {
            const int tq = toff[(size_t)k + 1] + 1;
            hipLaunchKernelGGL(k_chain_wait, dim3(1), dim3(64), 0, ctx->rest, cgw, tq, tq, tq);
            PYIPM_KCHECK();
        }
        int rc = on_done(q.id, rs); if (rc) return rc;
    }
    return 0;
}

// A group of panels of the single-rank schedule: panels [p0, p0 + n0).
int factor_group(Ctx* ctx, int64_t p0, int64_t n0, hipStream_t chain, const std::function<int(int64_t, hipStream_t)>& on_done) {
    const Geo& g = ctx->g;
    BlockDesc bd;
    for (int64_t k = 0; k < n0; ++k) {
        const int64_t q = p0 + k;
        bd.sp.push_back(SubPanel{g.panel_c0(q), g.local_c0(q), (int)(g.panel_w(q) / TB), wbuf(ctx, q), q});
    }
    const size_t gi = (size_t)ctx->grp_of[(size_t)p0];
    const bool grp_in_x = !ctx->grp_x.empty() && ctx->grp_x[gi];
    if (ctx->skip_zeros && g.mi > 0 && grp_in_x) {
        slack_hole(g, &bd.hole0, &bd.hole1);
    }
    return factor_block(ctx, bd, chain, on_done);
}

// Per-panel schedule, a panel wider than `wide_sub` columns (nb = 512 / 1024): the owner factors it like the single-rank
// schedule factors a group -- one tile chain over the panel's diagonal block, the rows below it sub-panel by sub-panel on
// ctx->rest with the in-panel updates as MFMA launches of K up to nb - wide_sub (k_panel_rest with all nt stages in one
// launch is right-looking 64 columns at a time through L2: 120 tile passes per strip at nt = 16).  The same products in the
// same order: the same bits.  What the other ranks see does not change (one message, one rank-nb update per panel), only
// that the message is 4 x larger and 4 x rarer than at nb = 256 and the owner's chain is the single-rank one.
int factor_wide_panel(Ctx* ctx, int64_t p, hipStream_t stream) {
    const Geo& g = ctx->g;
    const int64_t c0 = g.panel_c0(p), lc0 = g.local_c0(p);
    const int nt = (int)(g.panel_w(p) / TB), snt = ctx->wide_sub / TB;
    BlockDesc bd;
    bd.wide = true; bd.lp = p / g.world;
    double* W = wbuf(ctx, p);
    for (int t = 0; t < nt; t += snt) {
        const int w = nt - t < snt ? nt - t : snt;
        bd.sp.push_back(SubPanel{c0 + (int64_t)t * TB, lc0 + (int64_t)t * TB, w, W + (int64_t)t * TB * g.Npad, p});
    }
    panel_hole(ctx, p, &bd.hole0, &bd.hole1);
    return factor_block(ctx, bd, stream, [](int64_t, hipStream_t) { return 0; });
}

// ---- per-panel schedule, a panel in pieces (round 5: the two-message protocol of the distributed driver, dist_impl.hpp) ------
// The owner of panel p + 1 starts that panel's tile chain as soon as the rows of panel p that meet ITS diagonal block have
// arrived (a slice message, nb x nb); the rows below follow when their operands do.  A panel's rows below its diagonal block
// do not interact, so factoring it in pieces -- the tile chain, then the stages of any row range -- runs the same kernels on
// every entry in the same order as factor_panel does in one go: the same bits (tests/test_gpu_dist.py, tools/rank_replay.py).
bool panel_piecewise_ok(const Ctx* ctx, int64_t p) {
    const Geo& g = ctx->g;
    if (!ctx->tile_step || panel_in_s(ctx, p)) return false;
    const int nbw = (int)g.panel_w(p), nt = nbw / TB;
    const bool wide = ctx->wide_sub >= 128 && ctx->wide_sub % 128 == 0 && nbw > ctx->wide_sub;
    return wide ? (nt <= 32) : (nt <= 16);
}
// the tile steps of panel p's diagonal block (all that needs is the diagonal block up to date)
int panel_chain(Ctx* ctx, int64_t p, hipStream_t stream, int64_t xrows, const unsigned* xword, unsigned xwant) {
    const Geo& g = ctx->g;
    const int64_t c0 = g.panel_c0(p), lc0 = g.local_c0(p);
    const int nt = (int)(g.panel_w(p) / TB);
    return launch_tile_steps(ctx, stream, c0, lc0, nt, 0, nt, wbuf(ctx, p), ctx->Dinv + (c0 / TB) * (int64_t)(TB * TB),
                             ctx->Tsv + (c0 / TB) * (int64_t)(TB * TB), (int)(xrows / TB), xword, xwant);
}
// may panel p's chain take the `xrows` rows right below its diagonal block along (every stage of theirs in the chain's launch
// instead of panel_rows behind it)?  The one-launch chain, whole tiles -- at most 16 of them: a wide panel's (nb = 1024) rows for
// the next owner are 16 row tiles of 16 stages each, units that keep the chain's pace like the block's own far rows --, none of
// them inside the panel's hole of structural zeros (k_panel_rest leaves those alone), progress words enough.
bool chain_extra_ok(const Ctx* ctx, int64_t p, int64_t xrows) {
    const Geo& g = ctx->g;
    const int64_t c0 = g.panel_c0(p);
    const int nbw = (int)g.panel_w(p), nt = nbw / TB;
    if (xrows <= 0 || xrows % TB != 0 || xrows / TB > 16 || c0 + nbw + xrows > g.Npad) return false;
    if (!(nt >= 2 && chain_applies(ctx, c0, nt))) return false;
    if (1 + 4 * (nt + (int)(xrows / TB)) > Ctx::CHAIN_WORDS) return false;
    int64_t h0 = 0, h1 = 0;
    panel_hole(ctx, p, &h0, &h1);
    const int64_t r0 = c0 + nbw, r1 = r0 + xrows;
    return !(h1 > h0 && r1 > h0 && r0 < h1);
}
// every stage of panel p for the rows [r0, r1) below its diagonal block (128-aligned), the chain complete on `stream` before
int panel_rows(Ctx* ctx, int64_t p, int64_t r0, int64_t r1, hipStream_t stream) {
    const Geo& g = ctx->g;
    if (r1 > g.Npad) r1 = g.Npad;
    if (r1 <= r0) return 0;
    const int64_t c0 = g.panel_c0(p), lc0 = g.local_c0(p), TT = (int64_t)TB * TB;
    const int nbw = (int)g.panel_w(p), nt = nbw / TB;
    double* W = wbuf(ctx, p);
    int64_t hole0 = 0, hole1 = 0;
    panel_hole(ctx, p, &hole0, &hole1);
    const bool wide = ctx->wide_sub >= 128 && ctx->wide_sub % 128 == 0 && nbw > ctx->wide_sub;
    if (!wide) {
        hipLaunchKernelGGL(k_panel_rest, dim3((unsigned)((r1 - r0) / TB)), dim3(256), 0, stream, ctx->A, g.Npad, c0, lc0, nt, r0,
                           W, g.Npad, ctx->Dinv + (c0 / TB) * TT, ctx->Tsv + (c0 / TB) * TT, ctx->Tflag + c0 / TB,
                           ctx->block_refine, hole0, hole1, &ctx->dstats->growth_bits);
        PYIPM_KCHECK();
        return 0;
    }
    // a wide panel: sub-panel by sub-panel as factor_block does below a diagonal block (left-looking MFMA in-panel updates while
    // many rows remain below the panel, right-looking one source at a time in the tail: decided by the PANEL, so every row range
    // of it takes the same kernels)
    const int snt = ctx->wide_sub / TB;
    const int64_t gend = c0 + nbw, lp = p / g.world;
    const bool left = ctx->pending_left_rows >= 0 && g.Npad - gend > ctx->pending_left_rows;
    for (int t = 0; t < nt; t += snt) {
        const int w = nt - t < snt ? nt - t : snt;
        const int64_t sc0 = c0 + (int64_t)t * TB, slc0 = lc0 + (int64_t)t * TB;
        double* sW = W + (int64_t)t * TB * g.Npad;
        if (left && t > 0) {
            int rc = launch_update128(ctx, stream, ctx->A + lc0 * g.Npad, g.Npad, W, t * TB, r0, lp, 1, /*bulk=*/false, 0, r1, 0, c0,
                                      1, 0, 0, 0, nullptr, nullptr, false, (t * TB) / 128, (w * TB) / 128);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_panel_rest, dim3((unsigned)((r1 - r0) / TB)), dim3(256), 0, stream, ctx->A, g.Npad, sc0, slc0, w, r0,
                           sW, g.Npad, ctx->Dinv + (sc0 / TB) * TT, ctx->Tsv + (sc0 / TB) * TT, ctx->Tflag + sc0 / TB,
                           ctx->block_refine, hole0, hole1, &ctx->dstats->growth_bits);
        PYIPM_KCHECK();
        if (!left && t + w < nt) {
            const int K = w * TB;
            const int64_t tc0 = sc0 + K, ncols = (int64_t)(nt - t - w) * TB;
            if (g.Npad - tc0 <= ctx->pending32_rows) {
                int64_t pa0, pa1, pb0, pb1;
                active_ranges(ctx, sc0, sc0 + K, &pa0, &pa1, &pb0, &pb1);
                hipLaunchKernelGGL(k_inpanel_update, dim3((unsigned)((r1 - r0) / 32), (unsigned)(ncols / TB)), dim3(256), 0, stream,
                                   ctx->A, g.Npad, slc0 + K, ctx->A + slc0 * g.Npad, g.Npad, sW, g.Npad, tc0, K, r0, g.Npad,
                                   pa0, pa1, pb0, pb1, ctx->side_prio);
                PYIPM_KCHECK();
            } else {
                int rc = launch_update128(ctx, stream, ctx->A + slc0 * g.Npad, g.Npad, sW, K, r0, lp, 1, /*bulk=*/false, 0, r1, 0, sc0,
                                          1, 0, 0, 0, nullptr, nullptr, false, ((t + w) * TB) / 128, (int)(ncols / 128));
                if (rc) return rc;
            }
        }
    }
    return 0;
}

// Slice j (1 or 2) of panel p: the rows of panel p + j, i.e. W rows [c0(p + j), c0(p + j) + w(p + j)) of all the panel's columns
// (leading dimension = the slice's rows); slice 1 also carries the panel's tile inverses, tiles and flags.
int64_t slice_rows(const Geo& g, int64_t p, int j, int64_t* r0) {
    if (p + j >= g.npanels) { *r0 = g.Npad; return 0; }
    *r0 = g.panel_c0(p + j);
    return g.panel_w(p + j);
}
size_t slice_numel(const Geo& g, int64_t p, int j) {
    int64_t r0; const int64_t E = slice_rows(g, p, j, &r0), nbw = g.panel_w(p);
    if (E <= 0) return 0;
    return (size_t)(E * nbw + (j == 1 ? 2 * (nbw / TB) * TB * TB + nbw / TB : 0));
}
int pack_slice(Ctx* ctx, int64_t p, int j, double* buf, hipStream_t st) {
    const Geo& g = ctx->g;
    int64_t r0; const int64_t E = slice_rows(g, p, j, &r0), nbw = g.panel_w(p), c0 = g.panel_c0(p);
    if (E <= 0) return 0;
    const int64_t total = (int64_t)slice_numel(g, p, j);
    int64_t blocks = (total + 1023) / 1024; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_slice_pack, dim3((unsigned)blocks), dim3(256), 0, st, buf, E, nbw, wbuf(ctx, p) + r0, g.Npad,
                       ctx->Dinv + (c0 / TB) * (int64_t)(TB * TB), ctx->Tsv + (c0 / TB) * (int64_t)(TB * TB), ctx->Tflag + c0 / TB,
                       j == 1 ? 1 : 0);
    PYIPM_KCHECK();
    return 0;
}
// receiver: L = W inv(T) of the slice's rows into `EL` (E x nbw, leading dimension E), in ONE launch straight from the message:
// the tile inverses are read where slice 1 carries them (`tiles` = the tail of the slice-1 buffer), and slice 1's W rows --
// the columns of the panel this rank factors next: the W operand of every update of those columns -- are copied to where the
// panel message would put them by the same launch.  (Slice 2's W rows are the column positions of a panel this rank does not
// own: nobody here reads them.)
int unpack_slice(Ctx* ctx, int64_t p, int j, const double* buf, const double* tiles, double* EL, hipStream_t st) {
    const Geo& g = ctx->g;
    int64_t r0; const int64_t E = slice_rows(g, p, j, &r0), nbw = g.panel_w(p);
    if (E <= 0) return 0;
    const int64_t tb = (nbw / TB) * (int64_t)(TB * TB);
    int64_t h0, h1;
    panel_hole(ctx, p, &h0, &h1);
    NextUpd nu_off; memset(&nu_off, 0, sizeof(nu_off));
    hipLaunchKernelGGL(k_panel_scale, dim3((unsigned)(E / TB), (unsigned)(nbw / TB)), dim3(256), 0, st,
                       EL - r0, E, (int64_t)0, buf - r0, E, (int64_t)0, j == 1 ? wbuf(ctx, p) : (double*)nullptr, g.Npad, (int64_t)0,
                       tiles, tiles + tb, tiles + 2 * tb, ctx->block_refine, r0, h0, h1, (unsigned long long*)nullptr, -1.0, nu_off);
    PYIPM_KCHECK();
    return 0;
}
// ... and the panel's tile inverses, tiles and flags from the slice-1 buffer into the handle's arrays (what the rest of the
// panel's unpacking and nothing on the chain path reads)
int unpack_slice_tiles(Ctx* ctx, int64_t p, const double* tiles, hipStream_t st) {
    const Geo& g = ctx->g;
    const int64_t nbw = g.panel_w(p), c0 = g.panel_c0(p);
    const size_t tb = (size_t)(nbw / TB) * TB * TB;
    PYIPM_HIP(hipMemcpyAsync(ctx->Dinv + (c0 / TB) * (int64_t)(TB * TB), tiles, tb * sizeof(double), hipMemcpyDeviceToDevice, st));
    PYIPM_HIP(hipMemcpyAsync(ctx->Tsv + (c0 / TB) * (int64_t)(TB * TB), tiles + tb, tb * sizeof(double), hipMemcpyDeviceToDevice, st));
    PYIPM_HIP(hipMemcpyAsync(ctx->Tflag + c0 / TB, tiles + 2 * tb, (size_t)(nbw / TB) * sizeof(double), hipMemcpyDeviceToDevice, st));
    return 0;
}
// The panel message of p on a receiver, rows from `row_from` on only (the rows before it came as slices), tiles optional.
int unpack_panel_from(Ctx* ctx, int64_t p, const double* buf, int64_t row_from, bool with_tiles, hipStream_t st) {
    const Geo& g = ctx->g;
    int64_t h0, h1;
    panel_hole(ctx, p, &h0, &h1);
    const int64_t nbw = g.panel_w(p), c0 = g.panel_c0(p), c1 = c0 + nbw, m = g.Npad - c1 - (h1 - h0);
    double* dinv = ctx->Dinv + (c0 / TB) * (int64_t)(TB * TB);
    double* tsv = ctx->Tsv + (c0 / TB) * (int64_t)(TB * TB);
    if (with_tiles) {
        const size_t tbytes = (size_t)(nbw / TB) * TB * TB * sizeof(double);
        PYIPM_HIP(hipMemcpyAsync(dinv, buf + m * nbw, tbytes, hipMemcpyDeviceToDevice, st));
        PYIPM_HIP(hipMemcpyAsync(tsv, buf + m * nbw + (nbw / TB) * TB * TB, tbytes, hipMemcpyDeviceToDevice, st));
        PYIPM_HIP(hipMemcpyAsync(ctx->Tflag + c0 / TB, buf + m * nbw + 2 * (nbw / TB) * TB * TB, (size_t)(nbw / TB) * sizeof(double),
                                 hipMemcpyDeviceToDevice, st));
    }
    const int64_t rf = row_from > c1 ? row_from : c1;
    if (m <= 0 || rf >= g.Npad) return 0;
    // rows [c1, Npad) without the hole [h0, h1) sit in the message with leading dimension m.  One launch per segment at or beyond
    // rf reads -S straight from the message, rebuilds L = S inv(T) into Lbuf and leaves -S where the update launches read it
    // (round 6; until then: a strided copy of the segment into the W slot, then the product from there -- the panel's 8 nb bytes per
    // row read twice and written twice on every receiver, 57-70 ms per rank of the N = 131072 replay).  The tile columns of a
    // received panel are independent (L_t = W_t inv(T_t)): grid.y walks them.
    const int64_t segs[2][2] = {{c1, (h1 > h0) ? h0 : g.Npad}, {(h1 > h0) ? h1 : g.Npad, g.Npad}};
    NextUpd nu_off; memset(&nu_off, 0, sizeof(nu_off));
    for (int k = 0; k < 2; ++k) {
        const int64_t a = segs[k][0] > rf ? segs[k][0] : rf, b = segs[k][1];
        if (b <= a) continue;
        const int64_t mrow = a - c1 - ((h1 > h0 && a >= h1) ? (h1 - h0) : 0);      // the message row of matrix row a
        hipLaunchKernelGGL(k_panel_scale, dim3((unsigned)((b - a) / TB), (unsigned)(nbw / TB)), dim3(256), 0, st,
                           ctx->Lbuf, g.Npad, (int64_t)0, buf + mrow - a, m, (int64_t)0,
                           wbuf(ctx, p), g.Npad, (int64_t)0, dinv, tsv, ctx->Tflag + c0 / TB, ctx->block_refine, a, (int64_t)0, (int64_t)0,
                           (unsigned long long*)nullptr, -1.0, nu_off);
        PYIPM_KCHECK();
    }
    return 0;
}

// One bulk k_update<128> launch: panels [p0, p0+np) (contiguous, same rank) applied with K = their total
// width to local panels [first_lp, first_lp+n_lp); timed with HIP events on the handle's stream.
int timed_update(Ctx* ctx, int64_t p0, int64_t np, int64_t first_lp, int64_t n_lp, hipStream_t stream,
                 int head_ct, unsigned* head_counter, unsigned* head_count, bool list_only,
                 bool as_bulk) {             // as_bulk: a bulk launch although it runs on another stream
    const Geo& g = ctx->g;
    if (!stream) stream = ctx->stream;
    if (n_lp <= 0) return 0;
    int K = 0;
    for (int64_t q = p0; q < p0 + np; ++q) K += (int)g.panel_w(q);
    if ((size_t)p0 < ctx->grp_of.size() && !ctx->grp_fast.empty() && ctx->grp_fast[(size_t)ctx->grp_of[p0]]) {
        // slack-block sources: each column updates one diagonal entry (k_s_schur), here for the target columns
        // of this launch -- on the update stream, where the dense launch would have run
        const int64_t tq0 = first_lp, tq1 = first_lp + n_lp;                       // single rank: local = global panels
        const int64_t tc0 = g.panel_c0(tq0), tc1 = tq1 >= g.npanels ? g.Npad : g.panel_c0(tq1);
        hipLaunchKernelGGL(k_s_schur, grid1(K), dim3(256), 0, stream, ctx->A, g.Npad, g.panel_c0(p0), (int64_t)K,
                           g.n, g.n + g.mi + g.me, tc0, tc1);
        PYIPM_KCHECK();
        return 0;
    }
    if (np == 1 && panel_in_s(ctx, p0)) {
        // per-panel mode, slack-block source panel: every rank updates the diagonal entries it owns, from s / lambda
        const int64_t tp0 = first_lp * g.world + g.rank, tp1 = (first_lp + n_lp - 1) * g.world + g.rank + 1;
        hipLaunchKernelGGL(k_s_schur_sigma, dim3((unsigned)(K / TB)), dim3(64), 0, stream, ctx->A, g.Npad, g.panel_c0(p0),
                           g.n, g.n + g.mi + g.me, ctx->s, ctx->lda + g.me, ctx->eps, ctx->refine_cond, ctx->block_refine,
                           ctx->pivtol_rel, g.nb, g.world, g.rank, tp0, tp1, ctx->anorm);
        PYIPM_KCHECK();
        return 0;
    }
    const bool mine = g.owner(p0) == g.rank;
    const double* Lop = mine ? ctx->A + g.local_c0(p0) * g.Npad : ctx->Lbuf;
    const int64_t q0 = first_lp * g.world + g.rank;           // first global panel updated
    const int64_t row_begin = g.panel_c0(q0);
    if (row_begin >= g.Npad) return 0;
    // A launch on another stream than the handle's is a lookahead head riding the chain's stream: it overlaps the bulk launch
    // on the main stream, so its duration says nothing about the kernel's rate -- it is not part of the "trailing" figures
    // (time, flops, launches), and it uses the 4-wave instance of the kernel so that a kernel trace keeps the two apart.
    const bool chain_side = stream != ctx->stream && !ctx->per_panel_mode && !as_bulk;
    if (list_only)                                      // fused head: only build / look up the tile list (how many head tiles?)
        return launch_update128(ctx, stream, Lop, g.Npad, wbuf(ctx, p0), K, row_begin, first_lp, n_lp, true, 0, 0, 0,
                                g.panel_c0(p0), 1, 0, 0, head_ct, head_counter, head_count, true);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int used_bn = 128;
    if (ctx->profile && !chain_side) {
        if ((size_t)ctx->n_trailing >= ctx->ev_trailing.size()) {
            hipEvent_t a, b;
            PYIPM_HIP(hipEventCreate(&a)); PYIPM_HIP(hipEventCreate(&b));
            ctx->ev_trailing.push_back({a, b});
        }
        e0 = ctx->ev_trailing[ctx->n_trailing].first; e1 = ctx->ev_trailing[ctx->n_trailing].second;
        PYIPM_HIP(hipEventRecord(e0, stream));
    }
    if (ctx->per_panel_mode && n_lp == 1 && g.Npad - row_begin <= ctx->head32_rows_dist) {
        // one panel of columns (the head of the per-panel schedule): 32 x 64 blocks instead of a handful of 128 x 128 tiles
        int64_t pa0, pa1, pb0, pb1;
        active_ranges(ctx, g.panel_c0(p0), g.panel_c0(p0) + K, &pa0, &pa1, &pb0, &pb1);
        hipLaunchKernelGGL(k_inpanel_update, dim3((unsigned)((g.Npad - row_begin) / 32), (unsigned)(g.panel_w(q0) / TB)), dim3(256), 0,
                           stream, ctx->A, g.Npad, first_lp * (int64_t)g.nb, Lop, g.Npad, wbuf(ctx, p0), g.Npad, row_begin, K,
                           row_begin, g.Npad, pa0, pa1, pb0, pb1, ctx->side_prio);
        PYIPM_KCHECK();
    } else {
        int rc = launch_update128(ctx, stream, Lop, g.Npad, wbuf(ctx, p0), K, row_begin, first_lp, n_lp, true, 0, 0, 0,
                                  g.panel_c0(p0), 1, 0, chain_side ? ctx->head_waves : 0, head_ct, head_counter, head_count, false, 0, 0,
                                  &used_bn);
        if (rc) return rc;
    }
    if (chain_side) return 0;
    if (ctx->profile) PYIPM_HIP(hipEventRecord(e1, stream));
    // algorithmic flops of this launch: 2*K per lower-triangle entry of the updated local columns
    // (entries the KKT block structure leaves at zero are not counted: the launch skips their tiles)
    double fl = 0.0;
    int64_t a0, a1, b0, b1;
    active_ranges(ctx, g.panel_c0(p0), g.panel_c0(p0) + K, &a0, &a1, &b0, &b1);
    for (int64_t k = 0; k < n_lp; ++k) {
        const int64_t qq = (first_lp + k) * g.world + g.rank;
        if (qq >= g.npanels) break;
        fl += 2.0 * K * active_area(g.panel_c0(qq), g.Npad, g.panel_c0(qq), g.panel_c0(qq) + g.panel_w(qq), a0, a1, b0, b1);
    }
    ctx->trailing_flops += fl;
    ctx->trailing_area += fl / (2.0 * K);
    if ((size_t)ctx->n_trailing >= ctx->trailing_tag.size()) ctx->trailing_tag.resize((size_t)ctx->n_trailing + 64);
    // algorithmic bytes of the launch: the C tiles read and written once (16 B per updated entry) + the two operand panels
    // read once (K columns over the rows below the first updated column)
    ctx->trailing_tag[(size_t)ctx->n_trailing] = {used_bn, fl, 16.0 * (fl / (2.0 * K)) + 16.0 * (double)K * (double)(g.Npad - row_begin),
                                                  16.0 * (fl / (2.0 * K))};
    ctx->n_trailing++;
    return 0;
}

// Rank-nb update of every locally owned panel to the right of panel p.
int trailing_update(Ctx* ctx, int64_t p) {
    const Geo& g = ctx->g;
    if (g.panel_c0(p) + g.panel_w(p) >= g.Npad) return 0;
    int64_t q = p + 1;
    while (q < g.npanels && g.owner(q) != g.rank) ++q;
    if (q >= g.npanels) return 0;
    const int64_t local_panels = (g.ncols_local + g.nb - 1) / g.nb;
    return timed_update(ctx, p, 1, q / g.world, local_panels - q / g.world);
}

// Reset of the device-side statistics, on `st` (the stream the first tile kernels of this factorisation run on; default: the
// handle's).  A kernel, not a copy from a stack temporary followed by a host synchronisation (rounds 1-3): the host must be
// able to enqueue the first group's chain while the assembly is still running.
int factor_begin(Ctx* ctx, hipStream_t st) {
    if (!st) st = ctx->stream;
    hipLaunchKernelGGL(k_init_stats, dim3(1), dim3(64), 0, st, ctx->dstats);
    PYIPM_KCHECK();
    ctx->n_trailing = 0; ctx->trailing_flops = 0.0; ctx->trailing_area = 0.0;
    return 0;
}

int factor_end(Ctx* ctx, pyipm_factor_stats* stats) {
    DevStats z;
    PYIPM_HIP(hipMemcpyAsync(&z, ctx->dstats, sizeof(z), hipMemcpyDeviceToHost, ctx->stream));
    int sweep_err = 0;
    if (ctx->sweep_used && ctx->sweep_sync)
        PYIPM_HIP(hipMemcpyAsync(&sweep_err, ctx->sweep_sync + 3 * 4096, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    int chain_err = 0;
    unsigned* const chain_errw = ctx->chain_sync ? ctx->chain_sync + (size_t)Ctx::CHAIN_SLOTS * Ctx::CHAIN_WORDS : nullptr;
    if (ctx->chain_used && chain_errw)
        PYIPM_HIP(hipMemcpyAsync(&chain_err, chain_errw, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    ctx->chain_used = false;
    if (chain_err) {
        PYIPM_HIP(hipMemsetAsync(chain_errw, 0, sizeof(unsigned), ctx->stream));
        ctx->zeros_clean = false;
        ctx->err = "tile chain (k_tile_chain): a poll timed out -- a workgroup of the chain did not become resident; this factorisation is invalid "
                   "(set_option(\"tile_chain\", 0) runs one launch per tile)";
        return PYIPM_E_HIP;
    }
    if (sweep_err) {
        PYIPM_HIP(hipMemsetAsync(ctx->sweep_sync + 3 * 4096, 0, sizeof(unsigned), ctx->stream));
        ctx->err = "backward sweep (k_bwd_sweep): a poll timed out in an earlier solve; its result was NaN"; return PYIPM_E_HIP;
    }
    ctx->sweep_used = false;
    if (ctx->profile) {
        // time during which SOME update launch ran: the launches of the main stream are serial, a lookahead head on the
        // side stream may overlap the bulk update that follows it -- union of the intervals, not their sum
        std::vector<std::pair<float, float>> iv;
        for (int64_t i = 0; i < ctx->n_trailing; ++i) {
            float a = 0.f, d = 0.f;
            PYIPM_HIP(hipEventElapsedTime(&a, ctx->ev_trailing[0].first, ctx->ev_trailing[i].first));
            PYIPM_HIP(hipEventElapsedTime(&d, ctx->ev_trailing[i].first, ctx->ev_trailing[i].second));
            iv.push_back({a, a + d});
        }
        std::sort(iv.begin(), iv.end());
        double uni = 0.0, sum = 0.0; float hi = -1.0e30f;
        int64_t nreal = 0;
        for (auto& x : iv) {
            sum += x.second - x.first;
            if (x.second - x.first > 0.03f) ++nreal;           // (a launch whose tile list came out empty leaves two events a few
                                                               //  microseconds apart; the shortest real launch takes 70 us)
            if (x.second <= hi) continue;
            uni += x.second - (x.first > hi ? x.first : hi);
            hi = x.second;
        }
        for (int k = 0; k < 2; ++k) { ctx->inst_ms[k] = 0.0; ctx->inst_flops[k] = 0.0; ctx->inst_area[k] = 0.0; ctx->inst_cbytes[k] = 0.0; ctx->inst_n[k] = 0; }
        for (int64_t i = 0; i < ctx->n_trailing && (size_t)i < ctx->trailing_tag.size(); ++i) {
            float d = 0.f;
            PYIPM_HIP(hipEventElapsedTime(&d, ctx->ev_trailing[i].first, ctx->ev_trailing[i].second));
            if (d <= 0.03f) continue;                            // (empty tile list: see above)
            const int k = ctx->trailing_tag[(size_t)i].bn == 256 ? 1 : 0;
            ctx->inst_ms[k] += d; ctx->inst_flops[k] += ctx->trailing_tag[(size_t)i].flops; ctx->inst_area[k] += ctx->trailing_tag[(size_t)i].area;
            ctx->inst_cbytes[k] += ctx->trailing_tag[(size_t)i].cbytes;
            ctx->inst_n[k]++;
        }
        ctx->t_trailing = sum;                               // what a kernel trace adds up for the same launches
        ctx->t_trailing_union = uni;
        ctx->n_trailing_real = nreal;
    }
    ctx->factored = true;
    if (stats) {
        stats->n_neg = z.n_neg; stats->n_zero = z.n_zero; stats->n_2x2 = z.n_2x2; stats->n_pos = z.n_pos;
        stats->d_min = z.d_min; stats->d_max = z.d_max;
        long long gb = (long long)z.growth_bits; double gr; memcpy(&gr, &gb, sizeof(gr));
        stats->growth = gr; stats->nonfinite = z.nonfinite;
    }
    if (z.nonfinite) { ctx->zeros_clean = false; ctx->err = "NaN/Inf met during factorisation"; return PYIPM_E_NONFINITE; }
    return 0;
}

// nrhs right-hand sides at stride vstride (doubles) share every launch (last grid dimension); the single-vector
// callers use the defaults.
// A panel wider than wide_sub columns is swept in sub-panels of that width (the in-panel kernels are one workgroup with nt - 1
// dependent steps: at nb = 1024 the sweeps took 5.9 ms instead of 2.1): the launches of the same matrix at nb = wide_sub.
int fwd_panel(Ctx* ctx, int64_t p, double* v, hipStream_t stream, int nrhs, int64_t vstride) {
    if (!stream) stream = ctx->stream;
    const Geo& g = ctx->g;
    const int64_t pc0 = g.panel_c0(p), plc0 = g.local_c0(p);
    const int pw = (int)g.panel_w(p);
    const int sw = (ctx->wide_sub >= TB && ctx->wide_sub % TB == 0 && pw > ctx->wide_sub) ? ctx->wide_sub : pw;
    for (int off = 0; off < pw; off += sw) {
    const int64_t c0 = pc0 + off, lc0 = plc0 + off;
    const int nbw = pw - off < sw ? pw - off : sw;
    hipLaunchKernelGGL(k_fwd_diag, dim3(1, nrhs), dim3(nbw), nbw * sizeof(double), stream, ctx->A, g.Npad, lc0, c0, nbw, v, vstride);
    PYIPM_KCHECK();
    const int64_t below = g.Npad - (c0 + nbw);
    if (below > 0) {
        int64_t a0, a1, b0, b1;
        active_ranges(ctx, c0, c0 + nbw, &a0, &a1, &b0, &b1);
        hipLaunchKernelGGL(k_fwd_gemv, dim3(grid1(below).x, nrhs), dim3(256), nbw * sizeof(double), stream, ctx->A, g.Npad, lc0, c0,
                           nbw, c0 + nbw, g.Npad, v, vstride, a0, a1, b0, b1);
        PYIPM_KCHECK();
    }
    }
    return 0;
}

int diag_panel(Ctx* ctx, int64_t p, double* v, hipStream_t stream, int nrhs, int64_t vstride) {
    if (!stream) stream = ctx->stream;
    const Geo& g = ctx->g;
    const int64_t c0 = g.panel_c0(p);
    const int nbw = (int)g.panel_w(p);
    hipLaunchKernelGGL(k_diag_apply, dim3(nbw / TB, nrhs), dim3(64), 0, stream, ctx->Dinv, ctx->Tsv, ctx->Tflag, ctx->block_refine,
                       c0 / TB, c0, v, vstride);
    PYIPM_KCHECK();
    return 0;
}

// part / pstride: partial-sum buffer for several right-hand sides (>= nchunk*nb doubles each); default = the handle's own
int bwd_panel(Ctx* ctx, int64_t p, double* v, int nrhs, int64_t vstride, double* part, int64_t pstride) {
    const Geo& g = ctx->g;
    if (!part) part = ctx->partial;
    const int64_t pc0 = g.panel_c0(p), plc0 = g.local_c0(p);
    const int pw = (int)g.panel_w(p);
    const int sw = (ctx->wide_sub >= TB && ctx->wide_sub % TB == 0 && pw > ctx->wide_sub) ? ctx->wide_sub : pw;
    for (int off = ((pw - 1) / sw) * sw; off >= 0; off -= sw) {        // sub-panels of a wide panel, last first (see fwd_panel)
    const int64_t c0 = pc0 + off, lc0 = plc0 + off;
    const int nbw = pw - off < sw ? pw - off : sw;
    const int64_t below = g.Npad - (c0 + nbw);
    int nchunk = 0;
    if (below > 0) {
        nchunk = (int)((below + ROWCHUNK - 1) / ROWCHUNK);
        int64_t a0, a1, b0, b1;
        active_ranges(ctx, c0, c0 + nbw, &a0, &a1, &b0, &b1);
        hipLaunchKernelGGL(k_bwd_dot, dim3(nbw, nchunk, nrhs), dim3(256), 0, ctx->stream, ctx->A, g.Npad, lc0, g.nb,
                           c0 + nbw, g.Npad, v, part, vstride, pstride, a0, a1, b0, b1);
        PYIPM_KCHECK();
    }
    if (ctx->bwd_diag4 && nbw <= 4 * TB)
        hipLaunchKernelGGL(k_bwd_diag4, dim3(1, nrhs), dim3(1024), 0, ctx->stream, ctx->A, g.Npad, lc0, c0, nbw,
                           g.nb, part, nchunk, v, vstride, pstride);
    else
        hipLaunchKernelGGL(k_bwd_diag, dim3(1, nrhs), dim3(nbw), nbw * sizeof(double), ctx->stream, ctx->A, g.Npad, lc0, c0, nbw,
                           g.nb, part, nchunk, v, vstride, pstride);
    PYIPM_KCHECK();
    }
    return 0;
}

// x := M^{-1} x in place for the factored matrix of the current geometry (single-rank path)
int solve_plain(Ctx* ctx, double* v, bool forward_done, int nrhs, int64_t vstride, double* part,
                int64_t pstride) {
    const Geo& g = ctx->g;
    const bool one_launch = ctx->sweep_persist && nrhs == 1 && g.world == 1 && g.nb <= 4 * TB && g.nb % TB == 0 && g.npanels >= 2 &&
                            g.npanels <= 4096 && g.Npad % 8 == 0 && g.Npad / TB <= 8192;
    if (one_launch) {
        if (ctx->sweep_buf_n < g.Npad) {
            if (ctx->sweep_buf) { PYIPM_HIP(hipStreamSynchronize(ctx->stream)); PYIPM_HIP(hipFree(ctx->sweep_buf)); ctx->sweep_buf = nullptr; }
            PYIPM_HIP(hipMalloc((void**)&ctx->sweep_buf, (size_t)g.Npad * sizeof(double)));
            ctx->sweep_buf_n = g.Npad;
        }
        if (!ctx->sweep_sync) {
            PYIPM_HIP(hipMalloc((void**)&ctx->sweep_sync, (3 * 4096 + 1) * sizeof(unsigned)));
            PYIPM_HIP(hipMemsetAsync(ctx->sweep_sync + 3 * 4096, 0, sizeof(unsigned), ctx->stream));      // the error word: sticky
        }
    }
    SweepGeo sg;
    sg.Npad = g.Npad; sg.ld = g.Npad; sg.n = g.n; sg.mi = g.mi; sg.me = g.me; sg.nb = g.nb; sg.npanels = (int)g.npanels;
    sg.skip = (ctx->skip_zeros && g.mi > 0) ? 1 : 0;
    if (!forward_done && one_launch) {
        // forward sweep in one launch (k_fwd_sweep: workgroup 0 on the diagonal blocks, the others on their 64-row chunks),
        // then inv(T) y for all tiles at once
        const int P = (int)g.npanels;
        PYIPM_HIP(hipMemsetAsync(ctx->sweep_sync, 0, (size_t)(P + g.Npad / TB) * sizeof(unsigned), ctx->stream));
        int64_t blocks = 1 + (g.Npad / TB - g.nb / TB);
        // every workgroup of a sweep must be resident (they wait for each other): the grid comes from an occupancy query of
        // THIS build of the kernel, not from an assumption about its registers / shared memory
        if (ctx->occ_fwd_sweep <= 0) {
            int occ = 0;
            PYIPM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_fwd_sweep, 512, 0));
            ctx->occ_fwd_sweep = occ < 1 ? 1 : (occ > 2 ? 2 : occ);
        }
        if (blocks > (int64_t)ctx->occ_fwd_sweep * ctx->num_cus) blocks = (int64_t)ctx->occ_fwd_sweep * ctx->num_cus;
        if (ctx->sweep_max_blocks > 0 && blocks > ctx->sweep_max_blocks) blocks = ctx->sweep_max_blocks;      // (test hook: several chunks per owner)
        if (blocks < 2) blocks = 2;
        hipLaunchKernelGGL(k_fwd_sweep, dim3((unsigned)blocks), dim3(512), 0, ctx->stream, ctx->A, sg, v, ctx->sweep_sync,
                           ctx->sweep_sync + 3 * 4096, (unsigned long long)2.0e8);
        PYIPM_KCHECK();
        hipLaunchKernelGGL(k_diag_apply, dim3((unsigned)(g.Npad / TB), 1), dim3(64), 0, ctx->stream, ctx->Dinv, ctx->Tsv, ctx->Tflag,
                           ctx->block_refine, (int64_t)0, (int64_t)0, v, (int64_t)0);
        PYIPM_KCHECK();
        ctx->sweep_used = true;
    } else if (!forward_done) {
        for (int64_t p = 0; p < g.npanels; ++p) { int rc = fwd_panel(ctx, p, v, nullptr, nrhs, vstride); if (rc) return rc; }
        for (int64_t p = 0; p < g.npanels; ++p) { int rc = diag_panel(ctx, p, v, nullptr, nrhs, vstride); if (rc) return rc; }
    }
    if (one_launch) {
        // the whole backward sweep in one launch (k_bwd_sweep): workgroup 0 on the diagonal blocks, every other wave on its columns
        const int P = (int)g.npanels;
        PYIPM_HIP(hipMemsetAsync(ctx->sweep_sync, 0, (size_t)(2 * P + 4096) * sizeof(unsigned), ctx->stream));     // flags, near counts, a progress word per owner wave
        PYIPM_HIP(hipMemsetAsync(ctx->sweep_buf, 0xFF, (size_t)g.Npad * sizeof(double), ctx->stream));      // NaN: "not there yet"
        const int64_t groups = g.Npad / 8;
        const int64_t nearb = (g.nb / 8 + 15) / 16;                  // workgroups that only do the next panel's columns
        int64_t blocks = 1 + nearb + (groups + 15) / 16;
        if (ctx->occ_bwd_sweep <= 0) {
            int occ = 0;
            PYIPM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bwd_sweep, 1024, 0));
            if (occ < 1) { ctx->err = "k_bwd_sweep does not fit a compute unit"; return PYIPM_E_HIP; }
            ctx->occ_bwd_sweep = 1;                                  // (one workgroup per CU by design, whatever would fit)
        }
        if (blocks > ctx->num_cus) blocks = ctx->num_cus;
        // a progress word per owner WAVE, (blocks - 1 - nearb) x 16 of them, inside the 4096 words zeroed above (on a part
        // with more than 257 CUs the words beyond them would keep last sweep's counts and workgroup 0 would stop waiting)
        if ((blocks - 1 - nearb) * 16 > 4096) blocks = 1 + nearb + 4096 / 16;
        if (ctx->sweep_max_blocks > 0 && blocks > ctx->sweep_max_blocks) blocks = ctx->sweep_max_blocks;
        if (blocks < 2 + nearb) blocks = 2 + nearb;
        hipLaunchKernelGGL(k_bwd_sweep, dim3((unsigned)blocks), dim3(1024), 0, ctx->stream, ctx->A, sg, v, ctx->sweep_sync,
                           ctx->sweep_sync + 3 * 4096, (unsigned long long)2.0e8, ctx->sweep_buf, ctx->dbg_buf);                       // polls give up after 2 s (100 MHz clock)
        PYIPM_KCHECK();
        ctx->sweep_used = true;
        return 0;
    }
    for (int64_t p = g.npanels - 1; p >= 0; --p) { int rc = bwd_panel(ctx, p, v, nrhs, vstride, part, pstride); if (rc) return rc; }
    return 0;
}

// ---- condensed option: right-hand side reduction and solution expansion --------------------------
// vc <- [ b_x + Ji (Sigma b_i + b_s) ; b_e ; 0 ]   from a full-order right-hand side b
// (every rank of a distributed run does both over ALL rows: the vectors are replicated and the blocks fully staged)
static RowMap all_rows(const Geo& g) { Geo g1 = g; g1.world = 1; g1.rank = 0; return make_rowmap(g1, 0); }

int cond_reduce(Ctx* ctx, const double* b, double* vc) {
    const Geo& g = ctx->g;
    hipLaunchKernelGGL(k_cond_t, grid1(g.mi), dim3(256), 0, ctx->stream, ctx->vt, b, g, ctx->s, ctx->lda + g.me, ctx->eps,
                       ctx->cond_pos);
    PYIPM_KCHECK();
    hipLaunchKernelGGL(k_rowdot2, grid1(g.n, 4), dim3(256), 0, ctx->stream, vc, b, g.n,
                       (const double*)nullptr, (int64_t)0, (const double*)nullptr, (int64_t)0,
                       ctx->Ji, ctx->ld_Ji, ctx->vt, g.mi, 0, 0, all_rows(g));
    PYIPM_KCHECK();
    if (ctx->gc.Npad > g.n) {
        hipLaunchKernelGGL(k_cond_gather, grid1(ctx->gc.Npad - g.n), dim3(256), 0, ctx->stream, vc, b, g, ctx->gc.Npad,
                           ctx->cond_idx, ctx->cond_na, ctx->s, ctx->lda + g.me, ctx->eps);
        PYIPM_KCHECK();
    }
    return 0;
}

// v (holding the full right-hand side) <- [dx ; ds ; dle ; dli] from the condensed solution vc
int cond_expand(Ctx* ctx, const double* vc, double* v) {
    const Geo& g = ctx->g;
    const int nchunk = 64;
    const int64_t rpc = (g.n + nchunk - 1) / nchunk;
    hipLaunchKernelGGL(k_coldot_partial, dim3((unsigned)((g.mi + 255) / 256), nchunk), dim3(256), 0, ctx->stream,
                       ctx->partial, ctx->Ji, ctx->ld_Ji, g.n, g.mi, vc, rpc, 0, all_rows(g));
    PYIPM_KCHECK();
    hipLaunchKernelGGL(k_coldot_reduce, grid1(g.mi), dim3(256), 0, ctx->stream, ctx->vt, ctx->partial, g.mi, nchunk, 0);
    PYIPM_KCHECK();
    hipLaunchKernelGGL(k_cond_expand, grid1(g.Npad), dim3(256), 0, ctx->stream, v, vc, ctx->vt, g, ctx->s, ctx->lda + g.me, ctx->eps,
                       ctx->cond_pos);
    PYIPM_KCHECK();
    return 0;
}

// x := Hc^{-1} x  in place on a full-order Npad device vector (single-rank path).  With the condensed
// factor: reduce, solve the (n+me) system, expand.  forward_done: the fused pass already ran (on v, or on
// ctx->vc for the condensed factor, in which case v still holds the right-hand side).
int solve_inplace(Ctx* ctx, double* v, bool forward_done) {
    if (!ctx->cond_active) return solve_plain(ctx, v, forward_done);
    int rc;
    if (!forward_done) { rc = cond_reduce(ctx, v, ctx->vc); if (rc) return rc; }
    { GeoSwap sw(ctx, ctx->gc); rc = solve_plain(ctx, ctx->vc, forward_done); }
    if (rc) return rc;
    return cond_expand(ctx, ctx->vc, v);
}

// y = Hc v from the staged blocks (Npad vectors on the device).  With several ranks every rank adds the terms of the
// KKT columns it owns (row j of triu(d2L) | Je | Ji is column j of the lower triangle), rank 0 the element-wise s /
// multiplier part; the caller sums y over the ranks (dist_impl.hpp).

// y[0:n) = sym(triu(d2L)) v + delta v for the rows / columns this rank works on: one pass over the upper triangle
// (k_symv_tiles + k_symv_finish; ctx->partial holds the partials)
int symv_dev(Ctx* ctx, const double* v, double* y, double delta, const RowMap& rm) {
    const Geo& g = ctx->g;
    const int64_t n = g.n;
    const int nchunk = SYMV_CHUNKS;
    int64_t rpc = ((rm.nloc + nchunk - 1) / nchunk + 63) / 64 * 64; if (rpc < 64) rpc = 64;
    const int nch = (int)((rm.nloc + rpc - 1) / rpc);
    const int nseg = (int)((n + SYMV_SEG - 1) / SYMV_SEG);
    double* part_col = ctx->partial;
    double* part_row = ctx->partial + (size_t)nchunk * (size_t)n;
    if (nch > 0) {
        hipLaunchKernelGGL(k_symv_tiles, dim3((unsigned)nseg, (unsigned)nch), dim3(256), 0, ctx->stream, part_row, part_col,
                           ctx->d2L, ctx->ld_d2L, n, v, rpc, rm);
        PYIPM_KCHECK();
    }
    hipLaunchKernelGGL(k_symv_finish, grid1(n), dim3(256), 0, ctx->stream, y, part_row, part_col, n, v, delta, nseg, nch, rm);
    PYIPM_KCHECK();
    return 0;
}
int kkt_matvec_dev(Ctx* ctx, const double* v, double* y) {
    const Geo& g = ctx->g;
    const int64_t n = g.n, me = g.me, mi = g.mi;
    if (!ctx->have_blocks || !ctx->have_vectors) { ctx->err = "matvec: stage blocks and vectors first"; return PYIPM_E_BADARG; }
    const RowMap rm = make_rowmap(g, ctx->sharded);
    const int64_t nl = rm.nloc;
    if (g.world > 1) { hipLaunchKernelGGL(k_fill, grid1(g.Npad), dim3(256), 0, ctx->stream, y, 0.0, g.Npad); PYIPM_KCHECK(); }
    if (nl > 0) {
        // x block: sym(triu(d2L)) v + delta v, one pass over the upper triangle (row sums and mirrored column sums together)
        { int r2 = symv_dev(ctx, v, y, ctx->delta, rm); if (r2) return r2; }
    }
    // s, lambda_e, lambda_i, pad: element-wise parts (one rank)
    if (g.Npad > n && g.rank == 0) {                   // (nothing beyond the x block when me = mi = 0 and n is a multiple of 128)
        hipLaunchKernelGGL(k_matvec_tail, grid1(g.Npad - n), dim3(256), 0, ctx->stream, y, v, g, ctx->s, ctx->lda,
                           ctx->eps, ctx->delta_c);
        PYIPM_KCHECK();
    }
    // + Je v_e, Ji v_i into the x rows and + Je' v_x, Ji' v_x into the multiplier rows: each Jacobian block is passed over ONCE
    // (round 4: k_jac_tiles -- row sums and column sums from the same loads, as for the triangle of d2L; was k_rowdot2 + k_coldot_*,
    // two passes)
    if (nl > 0) {
        int64_t rpc = ((nl + SYMV_CHUNKS - 1) / SYMV_CHUNKS + 63) / 64 * 64; if (rpc < 64) rpc = 64;
        const int nch = (int)((nl + rpc - 1) / rpc);
        for (int blk = 0; blk < 2; ++blk) {
            const int64_t m = blk == 0 ? me : mi;
            if (m <= 0) continue;
            const double* M = blk == 0 ? ctx->Je : ctx->Ji;
            const int64_t ldm = blk == 0 ? ctx->ld_Je : ctx->ld_Ji;
            const int64_t off = n + mi + (blk == 0 ? 0 : me);                 // the multiplier rows of this block
            const int nseg = (int)((m + SYMV_SEG - 1) / SYMV_SEG);
            double* part_col = ctx->partial;
            double* part_row = ctx->partial + (size_t)SYMV_CHUNKS * (size_t)m;
            hipLaunchKernelGGL(k_jac_tiles, dim3((unsigned)nseg, (unsigned)nch), dim3(256), 0, ctx->stream, part_row, part_col,
                               M, ldm, m, v + off, v, rpc, rm);
            PYIPM_KCHECK();
            hipLaunchKernelGGL(k_jac_finish_rows, grid1(nl), dim3(256), 0, ctx->stream, y, part_row, nseg, rm);
            PYIPM_KCHECK();
            hipLaunchKernelGGL(k_coldot_reduce, grid1(m), dim3(256), 0, ctx->stream, y + off, part_col, m, nch, 1);
            PYIPM_KCHECK();
        }
    }
    return 0;
}

// g = -grad into ctx->rhs.  With several ranks: the x rows this rank owns and (rank 0) the s / multiplier rows, zero
// elsewhere; the distributed driver sums it over the ranks.
int residual_dev(Ctx* ctx) {
    const Geo& g = ctx->g;
    if (!ctx->have_blocks || !ctx->have_vectors) { ctx->err = "residual: stage blocks and vectors first"; return PYIPM_E_BADARG; }
    const RowMap rm = make_rowmap(g, ctx->sharded);
    if (g.world > 1) { hipLaunchKernelGGL(k_fill, grid1(g.Npad), dim3(256), 0, ctx->stream, ctx->rhs, 0.0, g.Npad); PYIPM_KCHECK(); }
    if (rm.nloc > 0) {
        hipLaunchKernelGGL(k_rowdot2, grid1(rm.nloc, 4), dim3(256), 0, ctx->stream, ctx->rhs, ctx->df, rm.nloc,
                           ctx->Je, ctx->ld_Je, ctx->lda, g.me, ctx->Ji, ctx->ld_Ji, ctx->lda + g.me, g.mi, 0, 1, rm);
        PYIPM_KCHECK();
    }
    if (g.Npad > g.n && g.rank == 0) {
        hipLaunchKernelGGL(k_residual_tail, grid1(g.Npad - g.n), dim3(256), 0, ctx->stream, ctx->rhs, g, ctx->ce, ctx->ci,
                           ctx->s, ctx->lda, ctx->mu, ctx->eps);
        PYIPM_KCHECK();
    }
    ctx->have_rhs = true;
    ctx->forward_pending = false;
    return 0;
}

// Condensed assembly.  Split the inequalities by Sigma (k_active_scan), then
//   A (ld = gc.Npad) <- tril [[H + delta I, Jx], [Jx', D]]   by K1 on the condensed geometry, Jx = [Je | Ji_A],
//                        D = diag(-delta_c I, -1/Sigma_A);
//   += Ji_I Sigma_I Ji_I'   as ONE rank-mi launch of the MFMA update kernel (C += JT * WT').
void asm_grid(const Ctx* ctx, const Geo& g, dim3* grid, int* tri);
int assemble_condensed(Ctx* ctx, double delta, double delta_c) {
    const Geo& g = ctx->g;
    const int64_t nx = (g.n + BM - 1) / BM * BM;                 // rows / columns of the x-x block the Gram launch touches
    const int64_t mi_pad = (g.mi + BKU - 1) / BKU * BKU, ldt = nx;
    const size_t need = 2 * (size_t)ldt * (size_t)mi_pad * sizeof(double);
    if (ctx->jt_bytes < need) {
        if (ctx->JT) PYIPM_HIP(hipFree(ctx->JT));
        ctx->JT = nullptr; ctx->jt_bytes = 0;
        if (hipMalloc((void**)&ctx->JT, need) != hipSuccess) { ctx->err = "condensed: no memory for the Ji' operands"; return PYIPM_E_NOMEM; }
        ctx->jt_bytes = need;
    }
    if (!ctx->cond_pos) {
        if (hipMalloc((void**)&ctx->cond_pos, (size_t)(2 * g.mi + 4) * sizeof(int)) != hipSuccess) { ctx->err = "condensed: no memory"; return PYIPM_E_NOMEM; }
        ctx->cond_idx = ctx->cond_pos + g.mi;
        ctx->cond_cnt = ctx->cond_idx + g.mi;
    }
    // active rows: Sigma_k > cond_sigma_max  (one scalar comes back to size the system)
    hipLaunchKernelGGL(k_active_scan, dim3(1), dim3(1024), 0, ctx->stream, ctx->cond_pos, ctx->cond_idx, ctx->cond_cnt,
                       ctx->s, ctx->lda + g.me, ctx->eps, ctx->cond_sigma_max, g.mi);
    PYIPM_KCHECK();
    int na = 0;
    PYIPM_HIP(hipMemcpyAsync(&na, ctx->cond_cnt, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    ctx->cond_na = na;
    ctx->gc = make_geo(g.n, g.me + na, 0, g.nb, g.world, g.rank);     // (several ranks: the same 1-D block-cyclic map, fewer panels)
    const Geo& gc = ctx->gc;
    const int64_t mx = g.me + na;
    ctx->WT = ctx->JT + (size_t)ldt * (size_t)mi_pad;
    const double* Jx = ctx->Je; int64_t ldx = ctx->ld_Je;
    if (na > 0) {
        const size_t jneed = (size_t)g.n * (size_t)(g.me + g.mi) * sizeof(double);
        if (ctx->jx_bytes < jneed) {
            if (ctx->Jx) PYIPM_HIP(hipFree(ctx->Jx));
            ctx->Jx = nullptr; ctx->jx_bytes = 0;
            if (hipMalloc((void**)&ctx->Jx, jneed) != hipSuccess) { ctx->err = "condensed: no memory for [Je | Ji_A]"; return PYIPM_E_NOMEM; }
            ctx->jx_bytes = jneed;
        }
        dim3 grid((unsigned)g.n, (unsigned)((mx + 255) / 256));
        hipLaunchKernelGGL(k_cond_gather_J, grid, dim3(256), 0, ctx->stream, ctx->Jx, mx, ctx->Je, ctx->ld_Je, g.me,
                           ctx->Ji, ctx->ld_Ji, ctx->cond_idx, (int64_t)na);
        PYIPM_KCHECK();
        Jx = ctx->Jx; ldx = mx;
    }
    {
        dim3 grid; int tri = 0;
        asm_grid(ctx, gc, &grid, &tri);
        PYIPM_HIP(hipMemsetAsync(ctx->anorm, 0, sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(k_assemble, grid, dim3(256), 0, ctx->stream, ctx->A, gc.Npad, gc, ctx->d2L, ctx->ld_d2L,
                           Jx, ldx, (const double*)nullptr, (int64_t)0, ctx->s, ctx->lda, ctx->eps, delta, delta_c, ctx->anorm, 0, 0, 0, (int64_t)0, tri);
        PYIPM_KCHECK();
        if (na > 0) {
            hipLaunchKernelGGL(k_cond_fix_diag, grid1(na), dim3(256), 0, ctx->stream, ctx->A, gc.Npad, g.n + g.me,
                               ctx->cond_idx, (int64_t)na, ctx->s, ctx->lda + g.me, ctx->eps, gc);
            PYIPM_KCHECK();
        }
    }
    {
        dim3 grid((unsigned)((ldt + 31) / 32), (unsigned)((mi_pad + 31) / 32));
        hipLaunchKernelGGL(k_transpose_scale, grid, dim3(256), 0, ctx->stream, ctx->JT, ctx->WT, ldt, ctx->Ji, ctx->ld_Ji,
                           g.n, g.mi, mi_pad, ctx->s, ctx->lda + g.me, ctx->eps, ctx->cond_pos);
        PYIPM_KCHECK();
    }
    if (ctx->profile) PYIPM_HIP(hipEventRecord(ctx->ev[6], ctx->stream));
    {
        GeoSwap sw(ctx, gc);
        // Gram update of the x-x block: the panels below column nx, of which a rank holds every world-th (its local panels
        // 0 .. ceil(#panels / world) - 1; a local panel past the block is cut off by col_end)
        const int64_t npx = (nx + gc.nb - 1) / gc.nb;
        int rc = launch_update128(ctx, ctx->stream, ctx->JT, ldt, ctx->WT, (int)mi_pad, 0, 0, (npx + gc.world - 1) / gc.world,
                                  true, ldt, nx, nx);
        if (rc) return rc;
    }
    if (ctx->profile) PYIPM_HIP(hipEventRecord(ctx->ev[7], ctx->stream));
    return 0;
}

// Grid of a whole-matrix k_assemble launch.  Single rank, Npad a multiple of 512: a 1-D grid over the 512 x 16 patches on or
// below the diagonal only (round 4: the 2-D grid started as many workgroups above the diagonal that returned at once --
// 65000 of them at N = 32768 cost 0.5 of K1's 1.5 ms; with them gone K1 moves its algorithmic bytes at 6.5 TB/s).
void asm_grid(const Ctx* ctx, const Geo& g, dim3* grid, int* tri) {
    *grid = dim3((unsigned)((g.Npad + 511) / 512), (unsigned)((g.ncols_local + 15) / 16));
    *tri = 0;
    if (ctx->asm_tri && g.world == 1 && g.Npad % 512 == 0) {
        const int64_t nbx = g.Npad / 512, ngrp = (g.ncols_local / 16 + 31) / 32;          // column patches in groups of 32
        int64_t tot = 0;
        for (int64_t G = 0; G < ngrp; ++G) tot += 32 * (nbx - G);
        if (ngrp <= nbx && tot < (int64_t)1 << 31) { *grid = dim3((unsigned)tot, 1); *tri = (int)nbx; }
    }
}

// panels of the first group of the single-rank schedule (the rule of factor_all)
int64_t first_group_panels(const Ctx* ctx) {
    const Geo& g = ctx->g;
    const int tg = (!ctx->tail_group_user && g.Npad <= 8192) ? 8 : ctx->tail_group;
    int64_t G = (tg > 0 && g.Npad <= ctx->tail_cols) ? tg : ctx->group;
    if (G > ctx->group) G = ctx->group;
    if (G > g.npanels) G = g.npanels;
    return G;
}

int assemble_dev(Ctx* ctx, double delta, double delta_c) {
    const Geo& g = ctx->g;
    if (ctx->provider_only) { ctx->err = "a provider-only handle has no KKT storage: block products and residuals only"; return PYIPM_E_BADARG; }
    if (!ctx->have_blocks || !ctx->have_vectors) { ctx->err = "assemble: stage blocks and vectors first"; return PYIPM_E_BADARG; }
    if (delta != ctx->delta || delta_c != ctx->delta_c) ctx->rc_warm_valid[0] = ctx->rc_warm_valid[1] = false;   // (a shifted matrix: the condition estimate starts cold)
    ctx->delta = delta; ctx->delta_c = delta_c;
    if (ctx->condensed && g.mi > 0 && g.world > 1 && ctx->sharded) {
        ctx->err = "condensed option across ranks needs the full blocks on every rank (stage_blocks, not stage_blocks_owned): "
                   "a column of Ji Sigma Ji' takes every row of Ji";
        return PYIPM_E_BADARG;
    }
    if (ctx->condensed && g.mi > 0) {
        ctx->zeros_clean = false;                       // the condensed system lives in the same storage with another layout
        int rc = assemble_condensed(ctx, delta, delta_c); if (rc) return rc;
        ctx->cond_active = true;
        ctx->assembled = true; ctx->factored = false; ctx->forward_pending = false;
        return 0;
    }
    ctx->cond_active = false;
    PYIPM_HIP(hipMemsetAsync(ctx->anorm, 0, sizeof(unsigned long long), ctx->stream));
    if (g.ncols_local > 0) {
        const int zip = (ctx->keep_zeros && !ctx->storage_exported && ctx->zeros_clean && g.world == 1 && g.mi > 0) ? 1 : 0;
        dim3 grid; int tri = 0;
        asm_grid(ctx, g, &grid, &tri);
        hipLaunchKernelGGL(k_assemble, grid, dim3(256), 0, ctx->stream, ctx->A, g.Npad, g, ctx->d2L, ctx->ld_d2L,
                           ctx->Je, ctx->ld_Je, ctx->Ji, ctx->ld_Ji, ctx->s, ctx->lda, ctx->eps, delta, delta_c, ctx->anorm, 0, ctx->sharded,
                           zip, (int64_t)0, tri);
        PYIPM_KCHECK();
    }
    // the zeros of this assembly survive a factorisation of finite numbers (every update that reaches them adds an exact zero);
    // whatever else may write into the storage clears the flag (zeros_dirty)
    ctx->zeros_clean = g.world == 1;
    ctx->assembled = true; ctx->factored = false; ctx->forward_pending = false;
    return 0;
}

// Single-rank factorisation.  Three-level hierarchy: 64-wide block pivots inside nb-wide panels inside
// groups of `group` panels.  Panels of a group are factored left-looking against the group's earlier
// panels; the trailing matrix is updated ONCE per group with K = group*nb, which divides the C-tile
// read-modify-write traffic (the HBM bound of the rank-nb update) by `group`.  One-group lookahead: while
// the bulk update of group g runs on the main stream, group g+1 (already updated by a head launch) is
// factored on a second stream.
int factor_all(Ctx* ctx, pyipm_factor_stats* stats, bool fuse_forward = false) {
    const Geo& g = ctx->g;
    if (g.world != 1) { ctx->err = "factor(): single-rank entry point; use the per-panel phases when world > 1"; return PYIPM_E_BADARG; }
    if (!ctx->assembled) { ctx->err = "factor: assemble first"; return PYIPM_E_BADARG; }
    ctx->per_panel_mode = false;
    const auto t_host0 = std::chrono::steady_clock::now();
    int rc = 0;
    if (!ctx->side) {
        // panel kernels are latency-critical and tiny: highest dispatch priority, so they take the first
        // CU slot a retiring bulk-update block frees instead of queueing behind the whole bulk grid
        int lo = 0, hi = 0;
        PYIPM_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        PYIPM_HIP(hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, hi));
    }
    PYIPM_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    // Group schedule: `group` panels per bulk update while the bulk update outlasts the panel chain; once at most
    // tail_cols columns remain the chain is the critical path and shorter groups (less in-group update work on
    // the chain, more in the cheap bulk launches) end the factorisation sooner.
    const int64_t np = g.npanels;
    ctx->grp_of.assign((size_t)np, 0); ctx->grp_off.assign((size_t)np, 0); ctx->grp_first.clear();
    {
        int64_t p = 0; int gid = 0;
        while (p < np) {
            const int64_t remaining = g.Npad - g.panel_c0(p);
            // (round 6: systems of at most 8192 rows take groups of 8 in the tail too -- with the chain of a group as ONE launch a group
            //  boundary costs more than the in-group updates of the longer group: config 2 2.25 against 2.29 ms; N = 32768: 104.9
            //  against 102.4 ms, config 3 185.4 against 183.2: 4 stays there)
            const int tg = (!ctx->tail_group_user && g.Npad <= 8192) ? 8 : ctx->tail_group;
            int64_t G = (tg > 0 && remaining <= ctx->tail_cols) ? tg : ctx->group;
            if (G > ctx->group) G = ctx->group;
            if (p + G > np) G = np - p;
            // (round 5) the panels inside the slack block as ONE group: their kernels run up front (s_early) and what is left of
            // a slack group in the loop below is a head, a k_s_schur launch and four stream hops -- 56 us per group of four
            // panels between the x block's last bulk update and the multiplier block's first chain (N = 32768: six groups)
            if (ctx->s_early && ctx->s_fast && ctx->skip_zeros && g.mi > 0 && g.world == 1 && ctx->lookahead && p > 0 &&
                g.panel_c0(p) >= g.n && g.panel_c0(p + G - 1) + g.panel_w(p + G - 1) <= g.n + g.mi) {
                while (p + G < np && g.panel_c0(p + G) + g.panel_w(p + G) <= g.n + g.mi) ++G;
            }
            ctx->grp_first.push_back(p);
            for (int64_t q = 0; q < G; ++q) { ctx->grp_of[(size_t)(p + q)] = gid; ctx->grp_off[(size_t)(p + q)] = (int)q; }
            p += G; ++gid;
        }
        ctx->grp_first.push_back(np);
        ctx->grp_fast.assign(ctx->grp_first.size() - 1, 0);
        ctx->grp_x.assign(ctx->grp_first.size() - 1, 0);
        for (size_t gi = 0; gi + 1 < ctx->grp_first.size(); ++gi) {
            const int64_t pb = ctx->grp_first[gi + 1];
            ctx->grp_x[gi] = (g.panel_c0(pb - 1) + g.panel_w(pb - 1) <= g.n) ? 1 : 0;
        }
        if (ctx->s_fast && ctx->skip_zeros && g.mi > 0 && g.world == 1)
            for (size_t gi = 0; gi + 1 < ctx->grp_first.size(); ++gi) {
                const int64_t pa = ctx->grp_first[gi], pb = ctx->grp_first[gi + 1];
                const int64_t ca = g.panel_c0(pa), cb = g.panel_c0(pb - 1) + g.panel_w(pb - 1);
                ctx->grp_fast[gi] = (ca >= g.n && cb <= g.n + g.mi) ? 1 : 0;
            }
    }
    const int64_t ngroups = (int64_t)ctx->grp_first.size() - 1;
    auto gsize = [&](int64_t grp) { return ctx->grp_first[(size_t)grp + 1] - ctx->grp_first[(size_t)grp]; };
    // diagnostics (PYIPM_GROUP_TRACE=1): where each group's chain, head and bulk update begin and end on the device, without a
    // tracer's per-call cost on the host (tools/group_trace.py)
    static const bool group_trace = getenv("PYIPM_GROUP_TRACE") != nullptr;
    std::vector<std::pair<std::string, hipEvent_t>> marks;
    auto mark = [&](const char* what, int64_t grp, hipStream_t st) {
        if (!group_trace) return;
        hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return;
        hipEventRecord(e, st);
        char buf[64]; snprintf(buf, sizeof(buf), "%s g%lld", what, (long long)grp);
        marks.push_back({buf, e});
    };
    // Fused forward substitution: y_p only needs panel p factored, so the forward pass of the step's
    // right-hand side (already in v0) trails the factorisation on its own stream.
    ctx->forward_fused = false;
    if (fuse_forward) {
        if (!ctx->fwd) PYIPM_HIP(hipStreamCreateWithFlags(&ctx->fwd, hipStreamNonBlocking));
        while ((int64_t)ctx->ev_done.size() < np) { hipEvent_t e; PYIPM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->ev_done.push_back(e); }
        PYIPM_HIP(hipEventRecord(ctx->ev_head, ctx->stream));          // v0 = rhs copy was enqueued on the main stream
        PYIPM_HIP(hipStreamWaitEvent(ctx->fwd, ctx->ev_head, 0));
    }
    std::vector<char> s_done_early((size_t)np, 0);     // closed-form panels of the slack block already enqueued (s_early, below)
    auto after_panel = [&](int64_t q, hipStream_t used) -> int {
        if (!fuse_forward) return 0;
        if (!s_done_early[(size_t)q]) PYIPM_HIP(hipEventRecord(ctx->ev_done[q], used));      // (else: recorded behind its kernel)
        PYIPM_HIP(hipStreamWaitEvent(ctx->fwd, ctx->ev_done[q], 0));
        int r2 = fwd_panel(ctx, q, ctx->fwd_vec, ctx->fwd); if (r2) return r2;
        return diag_panel(ctx, q, ctx->fwd_vec, ctx->fwd);
    };
    // does group grp run as a tile chain (factor_group)?
    auto chain_group = [&](int64_t grp) -> bool {
        const bool fast = !ctx->grp_fast.empty() && ctx->grp_fast[(size_t)grp];
        return ctx->group_chain && !fast && gsize(grp) * (g.nb / TB) <= 32 && g.nb % 128 == 0;
    };
    // Group 0 right-looking panel by panel (round 6, lookahead = 2): a system whose x block is ONE group followed by the slack
    // block (config 2).  Nothing runs beside the first group's chain, the slack group is no work, and the x group's bulk
    // update -- a launch of a hundred-odd tiles -- is what the multiplier block's chain waits for in full (272 us of 2.22 ms).
    // Here every panel of group 0 goes to the columns beyond the group as soon as it is complete: a K = nb launch on the side
    // stream (idle until the next chain) behind the panel's event, under the chain of the panels that follow; neither a head
    // nor a bulk update is left for group 0.  Every entry receives the same products in the same order.  Measured
    // (tools/r06_early.sh): config 2 2.225 -> 2.165 ms -- the chain itself slows from 880 to 970 us beside the pieces (its
    // round trips to memory share the fabric with their operands), the next chain starts 140 us earlier.  Where the next group
    // is an ordinary one the one-group lookahead already hides the bulk update under that group's chain and the pieces lose
    // (n, me, mi = 3072, 512, 1024: 2.99 -> 3.08 ms): not applied there.  Fewer, persistent blocks per piece (64 ... 192
    // of them) only made the last piece longer.
    const bool early0 = ctx->lookahead >= 2 && ngroups > 2 && chain_group(0) && !ctx->grp_fast.empty() && !ctx->grp_fast[0] &&
                        ctx->grp_fast[1] && ctx->s_early && ctx->head_on_side;
    if (early0 && !ctx->ev_early) PYIPM_HIP(hipEventCreateWithFlags(&ctx->ev_early, hipEventDisableTiming));
    auto early_piece = [&](int64_t q, hipStream_t used) -> int {
        const int64_t p1 = ctx->grp_first[1];
        PYIPM_HIP(hipEventRecord(ctx->ev_early, used));
        PYIPM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_early, 0));
        return timed_update(ctx, q, 1, p1, np - p1, ctx->side, 0, nullptr, nullptr, false, /*as_bulk=*/true);
    };
    // all panels of one group on stream S
    auto run_group = [&](int64_t grp, hipStream_t S) -> int {
        const int64_t pA = ctx->grp_first[(size_t)grp], nA = gsize(grp);
        auto done = [&](int64_t q, hipStream_t used) -> int {
            int r2 = after_panel(q, used); if (r2) return r2;
            return (early0 && grp == 0) ? early_piece(q, used) : 0;
        };
        if (chain_group(grp)) return factor_group(ctx, pA, nA, S, done);
        for (int64_t q = pA; q < pA + nA; ++q) {
            if (!s_done_early[(size_t)q]) { int r2 = factor_panel(ctx, q, S, true); if (r2) return r2; }
            int r2 = done(q, S); if (r2) return r2;
        }
        return 0;
    };
    // s_early (round 5): the closed-form panels of the slack block depend on nothing but the assembly -- no x-block update
    // reaches their columns -- and used to sit, one 7 us launch after the other, between the x block's last chain and the
    // multiplier block's first (config 2: 0.15 of 2.7 ms, plus the last x group's bulk update queued behind them on the same
    // stream).  They are enqueued up front on the rows stream; what READS them (k_s_schur on the update stream, the forward
    // sweep) keeps its place and waits for one event.  Statistics are atomic sums / minima / maxima: the same numbers.
    bool s_early = false;
    auto enqueue_s_early = [&](hipStream_t after) -> int {
        if (!(ctx->s_early && ctx->lookahead && g.world == 1 && ngroups > 1) || ctx->grp_fast.empty()) return 0;
        bool any = false;
        for (int64_t gi = 1; gi < ngroups; ++gi) any = any || ctx->grp_fast[(size_t)gi];
        if (!any || (ctx->grp_fast[0])) return 0;
        { int r0 = ensure_rest_stream(ctx); if (r0) return r0; }
        if (!ctx->ev_sfast) PYIPM_HIP(hipEventCreateWithFlags(&ctx->ev_sfast, hipEventDisableTiming));
        PYIPM_HIP(hipEventRecord(ctx->ev_sfast, after));                  // the assembly and the reset of the statistics
        PYIPM_HIP(hipStreamWaitEvent(ctx->rest, ctx->ev_sfast, 0));
        for (int64_t gi = 1; gi < ngroups; ++gi) {
            if (!ctx->grp_fast[(size_t)gi]) continue;
            for (int64_t q = ctx->grp_first[(size_t)gi]; q < ctx->grp_first[(size_t)gi + 1]; ++q) {
                int r2 = factor_panel(ctx, q, ctx->rest, true); if (r2) return r2;
                if (fuse_forward) PYIPM_HIP(hipEventRecord(ctx->ev_done[q], ctx->rest));
                s_done_early[(size_t)q] = 1;
            }
        }
        PYIPM_HIP(hipEventRecord(ctx->ev_sfast, ctx->rest));              // every closed-form panel is done
        s_early = true;
        return 0;
    };
    rc = factor_begin(ctx); if (rc) return rc;
    rc = enqueue_s_early(ctx->stream); if (rc) return rc;
    mark("chain+rows begin", 0, ctx->stream);
    rc = run_group(0, ctx->stream); if (rc) return rc;
    mark("chain+rows end", 0, ctx->stream);
    bool across_prev = false;
    if (group_trace) fprintf(stderr, "[pyipm group trace] %lld groups\n", (long long)ngroups);
    for (int64_t grp = 0; grp + 1 < ngroups; ++grp) {
        const int64_t p0 = ctx->grp_first[(size_t)grp], n0 = gsize(grp), p1 = p0 + n0, n1 = gsize(grp + 1);
        if (ctx->lookahead) {
            // head: next group's columns (the next chain waits for it).  Sources: the panels of this group whose
            // contribution an early head has not applied yet (see below).
            const int64_t hc0 = g.panel_c0(p1);
            const bool fast_src = !ctx->grp_fast.empty() && ctx->grp_fast[(size_t)grp];
            auto head_from = [&](int64_t q0, int64_t nq, int64_t tp, int64_t tn, hipStream_t hs, bool split = false) -> int {
                const int64_t tc0 = g.panel_c0(tp);
                if (split && !fast_src && g.Npad - tc0 <= ctx->head32_rows) {   // (where the chain is the bound)
                    // The chain of the target group needs the head only inside that group's diagonal block (rows [tc0, tend));
                    // the rows below it are first read by the group's rows stream.  Two launches: the block on the chain's
                    // stream, the rest on ctx->rest behind it -- the same entries, the same operations.
                    int K = 0; int64_t cols = 0;
                    for (int64_t q = q0; q < q0 + nq; ++q) K += (int)g.panel_w(q);
                    for (int64_t q = tp; q < tp + tn; ++q) cols += g.panel_w(q);
                    const int64_t tend = tc0 + cols;
                    int64_t pa0, pa1, pb0, pb1;
                    active_ranges(ctx, g.panel_c0(q0), g.panel_c0(q0) + K, &pa0, &pa1, &pb0, &pb1);
                    if (tend < g.Npad) {
                        int r2 = ensure_rest_stream(ctx); if (r2) return r2;
                        if (!ctx->ev_split) PYIPM_HIP(hipEventCreateWithFlags(&ctx->ev_split, hipEventDisableTiming));
                        PYIPM_HIP(hipEventRecord(ctx->ev_split, hs));          // the source group is complete, the main stream's
                        PYIPM_HIP(hipStreamWaitEvent(ctx->rest, ctx->ev_split, 0));   // earlier updates of these columns are ordered
                    }
                    hipLaunchKernelGGL(k_inpanel_update, dim3((unsigned)(cols / 32), (unsigned)(cols / TB)), dim3(256), 0,
                                       hs, ctx->A, g.Npad, g.local_c0(tp), ctx->A + g.local_c0(q0) * g.Npad, g.Npad,
                                       wbuf(ctx, q0), g.Npad, tc0, K, tc0, g.Npad, pa0, pa1, pb0, pb1, ctx->side_prio);
                    PYIPM_KCHECK();
                    if (tend < g.Npad) {
                        if (g.Npad - tend <= ctx->head32_rows) {
                            hipLaunchKernelGGL(k_inpanel_update, dim3((unsigned)((g.Npad - tend) / 32), (unsigned)(cols / TB)), dim3(256), 0,
                                               ctx->rest, ctx->A, g.Npad, g.local_c0(tp), ctx->A + g.local_c0(q0) * g.Npad, g.Npad,
                                               wbuf(ctx, q0), g.Npad, tc0, K, tend, g.Npad, pa0, pa1, pb0, pb1, ctx->side_prio);
                            PYIPM_KCHECK();
                        } else {
                            int r2 = launch_update128(ctx, ctx->rest, ctx->A + g.local_c0(q0) * g.Npad, g.Npad, wbuf(ctx, q0), K, tend, tp, tn,
                                                      /*bulk=*/true, 0, 0, 0, g.panel_c0(q0), 1, 0, ctx->head_waves);
                            if (r2) return r2;
                        }
                    }
                    return 0;
                }
                if (!fast_src && g.Npad - tc0 <= ctx->head32_rows) {
                    int K = 0; int64_t cols = 0;
                    for (int64_t q = q0; q < q0 + nq; ++q) K += (int)g.panel_w(q);
                    for (int64_t q = tp; q < tp + tn; ++q) cols += g.panel_w(q);
                    int64_t pa0, pa1, pb0, pb1;
                    active_ranges(ctx, g.panel_c0(q0), g.panel_c0(q0) + K, &pa0, &pa1, &pb0, &pb1);
                    // W of a panel inside its group's buffer: wbuf(q0) addresses it (column offset of q0 in the group)
                    hipLaunchKernelGGL(k_inpanel_update, dim3((unsigned)((g.Npad - tc0) / 32), (unsigned)(cols / TB)), dim3(256), 0,
                                       hs, ctx->A, g.Npad, g.local_c0(tp), ctx->A + g.local_c0(q0) * g.Npad, g.Npad,
                                       wbuf(ctx, q0), g.Npad, tc0, K, tc0, g.Npad, pa0, pa1, pb0, pb1, ctx->side_prio);
                    PYIPM_KCHECK();
                    return 0;
                }
                return timed_update(ctx, q0, nq, tp, tn, hs);
            };
            // The head follows the chain of this group on ITS stream when that is the side stream (every group but the first):
            // chain -> head -> next chain then never cross streams.  It touches the columns the bulk update of the group before
            // (and the early heads) touched on the main stream: ordered by one event, normally long complete.
            // A next group inside the slack block is a handful of microsecond launches (closed form): it runs on the main stream,
            // head included -- sending it through the side stream cost two stream crossings per group, 50-240 us each time for
            // 30 us of work (config 2: 0.6 of 3.6 ms).
            const bool nxt_fast = !ctx->grp_fast.empty() && ctx->grp_fast[(size_t)(grp + 1)];
            hipStream_t cs = (nxt_fast && ctx->fast_on_main) ? ctx->stream : ctx->side;      // where the next group runs
            hipStream_t hs = ctx->stream;
            // (the FIRST head stays on the main stream also when group 0 ran on the chain's stream: nothing runs beside it either
            //  way, and as a main-stream launch it takes the bulk instance and is part of the trailing figures, as in rounds 1-3)
            // (round 5) Lookahead ACROSS the slack block.  With its panels run up front the slack group between the last x group
            // and the first multiplier group is no work at all -- but that multiplier group's columns used to get this group's
            // contribution from its BULK update, and its chain started behind the whole launch and the slack group's hops
            // (N = 32768: 1.6 + 0.3 ms with nothing else on the chain's path; config 2: 0.14 ms).  The group after the slack group
            // is treated as the lookahead target: a head (split: its diagonal block on the chain's stream, the rows below on the
            // rows stream) applies this group's contribution to its columns, the bulk update starts beyond them.  The same
            // products in the same order per entry (x groups in order, then the slack block's diagonal update).
            const bool across = s_early && nxt_fast && !fast_src && grp > 0 && grp + 2 < ngroups &&
                                !ctx->grp_fast[(size_t)(grp + 2)] && chain_group(grp + 2) && ctx->head_on_side;
            const int64_t pT = across ? ctx->grp_first[(size_t)(grp + 2)] : 0, nT = across ? gsize(grp + 2) : 0;
            if (grp > 0 && ctx->head_on_side && (cs == ctx->side || across)) {
                // (recorded at the end of the iteration before.  Behind an `across` iteration the source is the slack group: its
                //  head is the diagonal update of the target's columns, ordered behind the x group's head on this stream, and the
                //  x group's bulk update -- what the event would wait for -- no longer touches those columns)
                if (!across_prev) PYIPM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_main, 0));
                hs = ctx->side;
                if (s_early && fast_src) PYIPM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_sfast, 0));   // (k_s_schur reads the slack columns)
            }
            const bool pieces = early0 && grp == 0;        // group 0's contribution is on the side stream already, panel by panel
            if (pieces) {
                PYIPM_HIP(hipEventRecord(ctx->ev_early, ctx->side));          // the last piece
                PYIPM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_early, 0)); // (later updates of those columns; cs, if it is the main stream)
                if (ctx->rest) PYIPM_HIP(hipStreamWaitEvent(ctx->rest, ctx->ev_early, 0));
            }
            if (!pieces) {
                const int64_t q = p0;
                const bool split = chain_group(grp + 1) && cs != ctx->stream;
                mark("head begin", grp, hs);
                // (a head INTO the slack block is structurally empty: no x column reaches an s column, and a slack column's only
                //  update is a diagonal entry of the multiplier block -- no launch, 10-15 us of the chain's path each)
                const bool empty_head = ctx->s_early && nxt_fast;
                if (across) { rc = head_from(p0, n0, pT, nT, hs, true); if (rc) return rc; }
                else if (q < p0 + n0 && !empty_head) { rc = head_from(q, p0 + n0 - q, p1, n1, hs, split); if (rc) return rc; }
                mark("head end (its first launch's stream)", grp, hs);
            }
            (void)hc0;
            if (hs == ctx->stream && cs == ctx->side) {
                PYIPM_HIP(hipEventRecord(ctx->ev_head, ctx->stream));
                PYIPM_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_head, 0));
            }
            mark("chain+rows begin", grp + 1, cs);
            if (s_early && nxt_fast) {                                   // its kernels ran up front: whoever follows on cs (and the
                PYIPM_HIP(hipStreamWaitEvent(cs, ctx->ev_sfast, 0));      // update stream, for k_s_schur) is ordered behind them
                if (cs != ctx->stream) PYIPM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_sfast, 0));
            }
            rc = run_group(grp + 1, cs); if (rc) return rc;
            mark("chain+rows end", grp + 1, cs);
            PYIPM_HIP(hipEventRecord(ctx->ev_panel, cs));
            mark("bulk begin", grp, ctx->stream);
            across_prev = across;
            if (pieces) { /* applied panel by panel */ }
            else if (across) { rc = timed_update(ctx, p0, n0, pT + nT, np - (pT + nT)); if (rc) return rc; }   // (the target's columns: the head above)
            else
            { rc = timed_update(ctx, p0, n0, p1 + n1, np - (p1 + n1)); if (rc) return rc; }   // bulk (overlaps the side stream)
            mark("bulk end", grp, ctx->stream);
            // what the next head must not overtake on the main stream: this group's bulk update -- recorded
            // BEFORE the main stream starts waiting for the side stream (the head would otherwise wait for its own stream,
            // two stream crossings for nothing)
            if (!ctx->ev_main) PYIPM_HIP(hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
            PYIPM_HIP(hipEventRecord(ctx->ev_main, ctx->stream));
            PYIPM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_panel, 0));
        } else {
            rc = timed_update(ctx, p0, n0, p1, np - p1); if (rc) return rc;
            rc = run_group(grp + 1, ctx->stream); if (rc) return rc;
        }
    }
    if (fuse_forward) {                     // join: the main stream continues after the forward pass
        PYIPM_HIP(hipEventRecord(ctx->ev_fwd, ctx->fwd));
        PYIPM_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_fwd, 0));
        ctx->forward_fused = true;
    }
    PYIPM_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    ctx->assembled = false;                 // storage now holds the factor
    if (getenv("PYIPM_SETUP_TRACE")) {
        const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
        fprintf(stderr, "[pyipm] factor_all: host enqueue %.2f ms (tile lists built so far: %d in %.2f ms)\n", host_ms,
                ctx->setup_lists_n, ctx->setup_lists_ms);
    }
    rc = factor_end(ctx, stats);
    float ms = 0.f;
    PYIPM_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    if (group_trace) {
        hipEventSynchronize(ctx->ev[1]);
        for (auto& m : marks) {
            float t = 0.f;
            if (hipEventSynchronize(m.second) == hipSuccess && hipEventElapsedTime(&t, ctx->ev[0], m.second) == hipSuccess)
                fprintf(stderr, "[pyipm group trace] %9.1f us  %s\n", 1e3 * t, m.first.c_str());
            hipEventDestroy(m.second);
        }
        fprintf(stderr, "[pyipm group trace] %9.1f us  factorisation end\n", 1e3 * ms);
    }
    ctx->t_factor = ms;
    ctx->t_panel = ctx->profile ? (ms - ctx->t_trailing_union) : 0.0;    // exposed panel time: no update launch running
    return rc;
}

// factor_all on whichever system assemble() built; inertia reported for the FULL KKT matrix either way:
// every eliminated (s_k, lambda_i_k) pair [[Sigma_k, -1], [-1, 0]] has determinant -1, i.e. one positive
// and one negative eigenvalue whatever the sign of Sigma_k.
int factor_dispatch(Ctx* ctx, pyipm_factor_stats* stats, bool fuse_forward) {
    if (!ctx->cond_active) { ctx->fwd_vec = ctx->v0; return factor_all(ctx, stats, fuse_forward); }
    pyipm_factor_stats local; if (!stats) stats = &local;
    int rc;
    ctx->fwd_vec = ctx->vc;
    { GeoSwap sw(ctx, ctx->gc); rc = factor_all(ctx, stats, fuse_forward); }
    // eliminated pairs (I) carry one negative and one positive eigenvalue each; an active row kept in the
    // system already counts its negative one there, the positive one is its eliminated Sigma_k pivot
    stats->n_neg += ctx->g.mi - ctx->cond_na; stats->n_pos += ctx->g.mi;
    if (ctx->profile) { float ms = 0.f; PYIPM_HIP(hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7])); ctx->t_gram = ms; }
    return rc;
}

void dist_free(Ctx* ctx);        // dist_impl.hpp

// a failed create releases whatever the handle already owns (events, a library-owned workspace) the way destroy does
int dist_set_option(Ctx* ctx, const char* name, double value, bool* handled);      // dist_impl.hpp
int create_fail(Ctx* ctx, int code) {
    pyipm_newton_destroy(reinterpret_cast<pyipm_newton_ctx*>(ctx));
    return code;
}

// kernels of this translation unit on behalf of the others (a kernel is launched from the unit that defines it)
int launch_axpby(Ctx* ctx, hipStream_t st, double* out, const double* a, const double* b, double alpha, double beta, int64_t n) {
    hipLaunchKernelGGL(k_axpby, grid1(n), dim3(256), 0, st, out, a, b, alpha, beta, n); PYIPM_KCHECK(); return 0; }
int launch_fill(Ctx* ctx, hipStream_t st, double* out, double v, int64_t n) {
    hipLaunchKernelGGL(k_fill, grid1(n), dim3(256), 0, st, out, v, n); PYIPM_KCHECK(); return 0; }
int launch_copy_flip(Ctx* ctx, hipStream_t st, double* out, const double* in, int flip) {
    hipLaunchKernelGGL(k_copy_flip, grid1(ctx->g.N), dim3(256), 0, st, out, in, ctx->g.N, ctx->g.n + ctx->g.mi, flip); PYIPM_KCHECK(); return 0; }
int launch_mask_owned(Ctx* ctx, hipStream_t st, double* v, const double* b) {
    hipLaunchKernelGGL(k_mask_owned, grid1(ctx->g.Npad), dim3(256), 0, st, v, b, ctx->g); PYIPM_KCHECK(); return 0; }
int launch_sumsq2(Ctx* ctx, hipStream_t st, double* out, const double* a, const double* b, int64_t n) {
    hipLaunchKernelGGL(k_sumsq2, dim3(1), dim3(1024), 0, st, out, a, b, n); PYIPM_KCHECK(); return 0; }
int launch_inpanel_update(Ctx* ctx, hipStream_t st, dim3 grid, double* Cm, int64_t ldc, int64_t ccol, const double* Lop, int64_t ldl,
                          const double* Wop, int64_t ldw, int64_t cglob, int K, int64_t row_begin, int64_t row_end,
                          int64_t a0, int64_t a1, int64_t b0, int64_t b1, int prio) {
    hipLaunchKernelGGL(k_inpanel_update, grid, dim3(256), 0, st, Cm, ldc, ccol, Lop, ldl, Wop, ldw, cglob, K, row_begin, row_end, a0, a1, b0, b1, prio);
    PYIPM_KCHECK(); return 0; }

} }  // namespace pyipm::drv

// =================================================================================================
#pragma GCC visibility push(default)
extern "C" {

size_t pyipm_newton_workspace_bytes(int64_t n, int64_t me, int64_t mi, int nb, int world, int rank) try {
    if (n <= 0 || me < 0 || mi < 0 || world < 1 || rank < 0 || rank >= world) return 0;
    if (nb == 0) nb = 256;
    if (nb % 128 != 0 || nb > 1024) return 0;
    Geo g = make_geo(n, me, mi, nb, world, rank);
    return carve_workspace(nullptr, g, nullptr);
} PYIPM_CATCH_SIZE

static int create_impl(pyipm_newton_ctx** out, int64_t n, int64_t me, int64_t mi, int nb, int device,
                       int world, int rank, void* workspace, size_t workspace_bytes, void* stream, bool provider_only);

int pyipm_newton_create(pyipm_newton_ctx** out, int64_t n, int64_t me, int64_t mi, int nb, int device,
                        int world, int rank, void* workspace, size_t workspace_bytes, void* stream) try {
    return create_impl(out, n, me, mi, nb, device, world, rank, workspace, workspace_bytes, stream, false);
} PYIPM_CATCH_NOH

size_t pyipm_newton_workspace_bytes_provider(int64_t n, int64_t me, int64_t mi) try {
    if (n <= 0 || me < 0 || mi < 0) return 0;
    Geo g = make_geo(n, me, mi, 256, 1, 0);
    return carve_workspace(nullptr, g, nullptr, true);
} PYIPM_CATCH_SIZE

int pyipm_newton_create_provider(pyipm_newton_ctx** out, int64_t n, int64_t me, int64_t mi, int device,
                                 void* workspace, size_t workspace_bytes, void* stream) try {
    return create_impl(out, n, me, mi, 256, device, 1, 0, workspace, workspace_bytes, stream, true);
} PYIPM_CATCH_NOH

static int create_impl(pyipm_newton_ctx** out, int64_t n, int64_t me, int64_t mi, int nb, int device,
                       int world, int rank, void* workspace, size_t workspace_bytes, void* stream, bool provider_only) {
    if (!out) return PYIPM_E_BADARG;
    *out = nullptr;
    if (n <= 0 || me < 0 || mi < 0 || world < 1 || rank < 0 || rank >= world) return PYIPM_E_BADARG;
    if (nb == 0) nb = 256;
    if (nb % 128 != 0 || nb > 1024) return PYIPM_E_BADARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PYIPM_E_NODEVICE;
    Ctx* ctx = new Ctx();
    ctx->g = make_geo(n, me, mi, nb, world, rank);
    ctx->gc = make_geo(n, me, 0, nb, 1, 0);          // re-derived by every condensed assemble
    ctx->group = default_group(world, nb);
    ctx->device = device;
    ctx->stream = (hipStream_t)stream;
    if (hipSetDevice(device) != hipSuccess) return create_fail(ctx, PYIPM_E_NODEVICE);
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) ctx->num_cus = ncu; }
    ctx->provider_only = provider_only;
    const size_t need = carve_workspace(nullptr, ctx->g, nullptr, provider_only);
    if (workspace) {
        if (workspace_bytes < need) return create_fail(ctx, PYIPM_E_NOMEM);
        ctx->ws = (char*)workspace; ctx->own_ws = false;
    } else {
        if (hipMalloc((void**)&ctx->ws, need) != hipSuccess) return create_fail(ctx, PYIPM_E_NOMEM);
        if (getenv("PYIPM_POISON_WORKSPACE")) hipMemset(ctx->ws, 0xFF, need);      // test hook: NaN wherever nothing is written before it is read
        ctx->own_ws = true;
    }
    ctx->ws_bytes = need;
    carve_workspace(ctx, ctx->g, ctx->ws, provider_only);
    if (hipMemset(ctx->anorm, 0, 2 * sizeof(unsigned long long)) != hipSuccess) return create_fail(ctx, PYIPM_E_HIP);      // [0] max |entry|, [1] "assembly pending"
    for (int i = 0; i < 8; ++i) if (hipEventCreate(&ctx->ev[i]) != hipSuccess) return create_fail(ctx, PYIPM_E_HIP);
    for (int i = 0; i < 4; ++i) if (hipEventCreate(&ctx->ev_prov[i]) != hipSuccess) return create_fail(ctx, PYIPM_E_HIP);
    if (hipEventCreateWithFlags(&ctx->ev_fwd, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_head, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_panel, hipEventDisableTiming) != hipSuccess) return create_fail(ctx, PYIPM_E_HIP);
    *out = reinterpret_cast<pyipm_newton_ctx*>(ctx);
    return PYIPM_OK;
}

size_t pyipm_newton_workspace_bytes_batched(int64_t n, int64_t me, int64_t mi, int batch) try {
    if (n <= 0 || me < 0 || mi < 0 || batch < 1) return 0;
    Geo g = make_geo(n, me, mi, 128, 1, 0);
    if (g.Npad > 1024) return 0;
    return carve_batched(nullptr, g, batch, nullptr);
} PYIPM_CATCH_SIZE

int pyipm_newton_create_batched(pyipm_newton_ctx** out, int64_t n, int64_t me, int64_t mi, int batch, int device,
                                void* workspace, size_t workspace_bytes, void* stream) try {
    if (!out) return PYIPM_E_BADARG;
    *out = nullptr;
    if (n <= 0 || me < 0 || mi < 0 || batch < 1) return PYIPM_E_BADARG;
    Geo g = make_geo(n, me, mi, 128, 1, 0);
    if (g.Npad > 1024) return PYIPM_E_BADARG;          // one workgroup per problem: Npad threads in the substitutions
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PYIPM_E_NODEVICE;
    Ctx* ctx = new Ctx();
    ctx->g = g; ctx->gc = g;
    ctx->batch = batch;
    ctx->group = 1;
    ctx->device = device;
    ctx->stream = (hipStream_t)stream;
    if (hipSetDevice(device) != hipSuccess) return create_fail(ctx, PYIPM_E_NODEVICE);
    const size_t need = carve_batched(nullptr, g, batch, nullptr);
    if (workspace) {
        if (workspace_bytes < need) return create_fail(ctx, PYIPM_E_NOMEM);
        ctx->ws = (char*)workspace; ctx->own_ws = false;
    } else {
        if (hipMalloc((void**)&ctx->ws, need) != hipSuccess) return create_fail(ctx, PYIPM_E_NOMEM);
        if (getenv("PYIPM_POISON_WORKSPACE")) hipMemset(ctx->ws, 0xFF, need);      // test hook: NaN wherever nothing is written before it is read
        ctx->own_ws = true;
    }
    ctx->ws_bytes = need;
    carve_batched(ctx, g, batch, ctx->ws);
    ctx->batched = true;
    ctx->tile_blocked = 1;               // the blocked inversion in both forms since round 6 (512 x N = 768, full form: 3.24 against 3.44 ms;
                                          // condensed: 0.99 against 1.10; round 3 measured the opposite for the full form, before the per-problem
                                          // Gram kernel and the eight-wave-free register budget of today's k_b_factor)
    if (hipMemset(ctx->anorm, 0, (size_t)batch * 2 * sizeof(unsigned long long)) != hipSuccess) return create_fail(ctx, PYIPM_E_HIP);
    for (int i = 0; i < 8; ++i) if (hipEventCreate(&ctx->ev[i]) != hipSuccess) return create_fail(ctx, PYIPM_E_HIP);
    *out = reinterpret_cast<pyipm_newton_ctx*>(ctx);
    return PYIPM_OK;
} PYIPM_CATCH_NOH

int pyipm_newton_stage_blocks_batched(pyipm_newton_ctx* h, const double* d2L, int64_t ld_d2L, int64_t stride_d2L,
                                      const double* Je, int64_t ld_Je, int64_t stride_Je,
                                      const double* Ji, int64_t ld_Ji, int64_t stride_Ji) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (!ctx->batched) { ctx->err = "stage_blocks_batched: not a batched handle"; return PYIPM_E_BADARG; }
    if (!d2L || ld_d2L < g.n || (g.me && (!Je || ld_Je < g.me)) || (g.mi && (!Ji || ld_Ji < g.mi))) {
        ctx->err = "stage_blocks_batched: bad block pointer / leading dimension (device pointers only)"; return PYIPM_E_BADARG; }
    ctx->d2L = d2L; ctx->ld_d2L = ld_d2L; ctx->b_sH = stride_d2L;
    ctx->Je = g.me ? Je : nullptr; ctx->ld_Je = ld_Je; ctx->b_sJe = stride_Je;
    ctx->Ji = g.mi ? Ji : nullptr; ctx->ld_Ji = ld_Ji; ctx->b_sJi = stride_Ji;
    ctx->have_blocks = true;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_step_batched(pyipm_newton_ctx* h, double delta, double delta_c, double* dz,
                              pyipm_factor_stats* stats, int memkind) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (!ctx->batched) { ctx->err = "step_batched: not a batched handle"; return PYIPM_E_BADARG; }
    if (!ctx->have_blocks || !ctx->have_vectors) { ctx->err = "step_batched: stage blocks and vectors first"; return PYIPM_E_BADARG; }
    if (!dz) { ctx->err = "step_batched: null output"; return PYIPM_E_BADARG; }
    const int B = ctx->batch;
    BatchPtrs bp = batch_ptrs(ctx);
    BatchCond bc = batch_cond(ctx);
    const int cond = (ctx->condensed && g.mi > 0) ? 1 : 0;
    ctx->delta = delta; ctx->delta_c = delta_c;
    ctx->cond_active = cond != 0;
    PYIPM_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    PYIPM_HIP(hipMemsetAsync(ctx->anorm, 0, (size_t)B * 2 * sizeof(unsigned long long), ctx->stream));
    if (cond) {
        // condensed form: per problem n + me + |A| columns instead of n + 2 mi + me (config 5: 256 instead of 768) -- the
        // (s, lambda_i) pairs with Sigma <= condensed_sigma_max eliminated analytically (pyipm.py:824-842's block structure)
        hipLaunchKernelGGL(k_bc_prep, dim3(B), dim3(256), 0, ctx->stream, bp, g, ctx->mu, ctx->eps, bc);
        PYIPM_KCHECK();
        const int nt0 = (int)((g.n + g.me + TB - 1) / TB);
        const int nx = (g.n % TB == 0 && g.n >= TB && g.n <= 4 * TB && g.mi % 32 == 0 && ctx->bc_per_problem) ? (int)(g.n / TB) : 0;
        // one workgroup per problem while the Jacobian chunk of all n columns fits shared memory twice (n <= 256), else a tile each
        if (nx == 4)      hipLaunchKernelGGL((k_bc_assemble_p<4>), dim3((unsigned)B), dim3(512), 0, ctx->stream, bp, g, ctx->eps, delta, delta_c, bc);
        else if (nx == 3) hipLaunchKernelGGL((k_bc_assemble_p<3>), dim3((unsigned)B), dim3(384), 0, ctx->stream, bp, g, ctx->eps, delta, delta_c, bc);
        else if (nx == 2) hipLaunchKernelGGL((k_bc_assemble_p<2>), dim3((unsigned)B), dim3(256), 0, ctx->stream, bp, g, ctx->eps, delta, delta_c, bc);
        else if (nx == 1) hipLaunchKernelGGL((k_bc_assemble_p<1>), dim3((unsigned)B), dim3(128), 0, ctx->stream, bp, g, ctx->eps, delta, delta_c, bc);
        else
            hipLaunchKernelGGL(k_bc_assemble, dim3((unsigned)(nt0 * (nt0 + 1) / 2 + 1), (unsigned)B), dim3(256), 0, ctx->stream, bp, g,
                               ctx->eps, delta, delta_c, bc, nt0);
        PYIPM_KCHECK();
    } else {
        hipLaunchKernelGGL(k_b_residual, dim3(B), dim3(256), 0, ctx->stream, bp, g, ctx->mu, ctx->eps);
        PYIPM_KCHECK();
        dim3 grid((unsigned)((g.Npad + 511) / 512), (unsigned)((g.Npad + 15) / 16), (unsigned)B);
        hipLaunchKernelGGL(k_b_assemble, grid, dim3(256), 0, ctx->stream, bp, g, ctx->eps, delta, delta_c);
        PYIPM_KCHECK();
    }
    PYIPM_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
    hipLaunchKernelGGL(k_b_factor, dim3(B), dim3(256), 0, ctx->stream, bp, g, ctx->refine_cond, ctx->block_refine, ctx->pivtol_rel, ctx->tile_blocked,
                       cond ? (const int*)ctx->cond_cnt : (const int*)nullptr);
    PYIPM_KCHECK();
    PYIPM_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
    double* out_dev = (memkind == PYIPM_MEM_DEVICE) ? dz : ctx->v2;
    hipLaunchKernelGGL(k_b_solve, dim3(B), dim3((unsigned)g.Npad), 2 * g.Npad * sizeof(double), ctx->stream, bp, g,
                       ctx->block_refine, (g.me + g.mi) > 0 ? 1 : 0, out_dev, bc, cond, ctx->eps);
    PYIPM_KCHECK();
    PYIPM_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    ctx->have_rhs = true;                 // g = -grad of every problem is in rhs (backward_error_batched reads it)
    ctx->ev_assemble_valid = true;
    if (memkind == PYIPM_MEM_HOST)
        PYIPM_HIP(hipMemcpyAsync(dz, ctx->v2, (size_t)B * g.N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    int rc = PYIPM_OK;
    if (stats) {
        std::vector<DevStats> z((size_t)B);
        PYIPM_HIP(hipMemcpyAsync(z.data(), ctx->dstats, (size_t)B * sizeof(DevStats), hipMemcpyDeviceToHost, ctx->stream));
        PYIPM_HIP(hipStreamSynchronize(ctx->stream));
        for (int b = 0; b < B; ++b) {
            stats[b].n_neg = z[b].n_neg; stats[b].n_zero = z[b].n_zero; stats[b].n_2x2 = z[b].n_2x2; stats[b].n_pos = z[b].n_pos;
            stats[b].d_min = z[b].d_min; stats[b].d_max = z[b].d_max;
            long long gb = (long long)z[b].growth_bits; double gr; memcpy(&gr, &gb, sizeof(gr));
            stats[b].growth = gr; stats[b].nonfinite = z[b].nonfinite;
            if (z[b].nonfinite) { ctx->err = "NaN/Inf met during factorisation"; rc = PYIPM_E_NONFINITE; }
        }
        float ms = 0.f;
        PYIPM_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        ctx->t_factor = ms;
    } else if (memkind == PYIPM_MEM_HOST) {
        PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return rc;
} PYIPM_CATCH_H(h)

// The per-problem statistics of the last step_batched, fetched when the caller wants them: a step called with stats = NULL
// returns as soon as its five launches are enqueued (round 6: the copy of 512 statistics records and the synchronisation in
// front of it were 0.3 of the 1.4 ms a batch step took on the host's clock).
int pyipm_newton_stats_batched(pyipm_newton_ctx* h, pyipm_factor_stats* stats) try {
    if (check_ctx(h) || !stats) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (!ctx->batched) { ctx->err = "stats_batched: not a batched handle"; return PYIPM_E_BADARG; }
    if (!ctx->have_rhs) { ctx->err = "stats_batched: step_batched first"; return PYIPM_E_BADARG; }
    const int B = ctx->batch;
    std::vector<DevStats> z((size_t)B);
    PYIPM_HIP(hipMemcpyAsync(z.data(), ctx->dstats, (size_t)B * sizeof(DevStats), hipMemcpyDeviceToHost, ctx->stream));
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    int rc = PYIPM_OK;
    for (int b = 0; b < B; ++b) {
        stats[b].n_neg = z[b].n_neg; stats[b].n_zero = z[b].n_zero; stats[b].n_2x2 = z[b].n_2x2; stats[b].n_pos = z[b].n_pos;
        stats[b].d_min = z[b].d_min; stats[b].d_max = z[b].d_max;
        long long gb = (long long)z[b].growth_bits; double gr; memcpy(&gr, &gb, sizeof(gr));
        stats[b].growth = gr; stats[b].nonfinite = z[b].nonfinite;
        if (z[b].nonfinite) { ctx->err = "NaN/Inf met during factorisation"; rc = PYIPM_E_NONFINITE; }
    }
    float ms = 0.f;
    PYIPM_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->t_factor = ms;
    return rc;
} PYIPM_CATCH_H(h)

// out[b] = |g - Hc raw_b| / |g| of every problem of the last step_batched, Hc applied from the blocks (k_b_berr)
int pyipm_newton_backward_error_batched(pyipm_newton_ctx* h, const double* dz, double* out, int memkind) try {
    if (check_ctx(h) || !dz || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (!ctx->batched) { ctx->err = "backward_error_batched: not a batched handle"; return PYIPM_E_BADARG; }
    if (!ctx->have_blocks || !ctx->have_vectors || !ctx->have_rhs) { ctx->err = "backward_error_batched: step_batched first"; return PYIPM_E_BADARG; }
    BatchPtrs bp = batch_ptrs(ctx);
    double* dev = (memkind == PYIPM_MEM_DEVICE) ? out : ctx->vt;
    hipLaunchKernelGGL(k_b_berr, dim3(ctx->batch), dim3(256), 0, ctx->stream, bp, g, dz, ctx->eps, ctx->delta, ctx->delta_c, dev);
    PYIPM_KCHECK();
    if (memkind == PYIPM_MEM_HOST) {
        PYIPM_HIP(hipMemcpyAsync(out, ctx->vt, (size_t)ctx->batch * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_destroy(pyipm_newton_ctx* h) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 8; ++i) if (ctx->ev[i]) hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < 4; ++i) if (ctx->ev_prov[i]) hipEventDestroy(ctx->ev_prov[i]);
    for (auto& pr : ctx->ev_trailing) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    if (ctx->ev_head) hipEventDestroy(ctx->ev_head);
    if (ctx->ev_panel) hipEventDestroy(ctx->ev_panel);
    dist_free(ctx);
    if (ctx->side) { hipStreamSynchronize(ctx->side); hipStreamDestroy(ctx->side); }
    if (ctx->fwd) { hipStreamSynchronize(ctx->fwd); hipStreamDestroy(ctx->fwd); }
    if (ctx->rest) { hipStreamSynchronize(ctx->rest); hipStreamDestroy(ctx->rest); }
    if (ctx->ev_fwd) hipEventDestroy(ctx->ev_fwd);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    if (ctx->ev_split) hipEventDestroy(ctx->ev_split);
    if (ctx->ev_early) hipEventDestroy(ctx->ev_early);
    if (ctx->ev_sfast) hipEventDestroy(ctx->ev_sfast);
    if (ctx->ev_main) hipEventDestroy(ctx->ev_main);
    for (auto e : ctx->ev_band) hipEventDestroy(e);
    for (auto e : ctx->ev_done) hipEventDestroy(e);
    if (ctx->stg_d2L) hipFree(ctx->stg_d2L);
    if (ctx->stg_Je) hipFree(ctx->stg_Je);
    if (ctx->stg_Ji) hipFree(ctx->stg_Ji);
    for (auto& kv : ctx->tile_lists) if (kv.second.ready) hipEventDestroy(kv.second.ready);
    for (auto& a : ctx->tl_arenas) { if (a.dev) hipFree(a.dev); if (a.host) hipHostFree(a.host); }
    if (ctx->JT) hipFree(ctx->JT);
    if (ctx->Jx) hipFree(ctx->Jx);
    if (ctx->cond_pos && !ctx->batched) hipFree(ctx->cond_pos);      // (a batched handle's lives in its workspace)
    if (ctx->sweep_sync) hipFree(ctx->sweep_sync);
    if (ctx->chain_sync) hipFree(ctx->chain_sync);
    if (ctx->sweep_buf) hipFree(ctx->sweep_buf);
    if (ctx->merit_buf) hipFree(ctx->merit_buf);
    if (ctx->rc_warm[0]) hipFree(ctx->rc_warm[0]);
    if (ctx->ray_buf) hipFree(ctx->ray_buf);
    if (ctx->own_ws && ctx->ws) hipFree(ctx->ws);
    delete ctx;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_set_stream(pyipm_newton_ctx* h, void* stream) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    C(h)->stream = (hipStream_t)stream;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

const char* pyipm_newton_last_error(pyipm_newton_ctx* h) {
    if (!h) return "null handle";
    return C(h)->err.c_str();
}

int pyipm_newton_geometry(pyipm_newton_ctx* h, int64_t out[8]) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    const Geo& g = C(h)->g;
    out[0] = g.N; out[1] = g.Npad; out[2] = g.nb; out[3] = g.npanels; out[4] = g.ncols_local;
    out[5] = g.world; out[6] = g.rank; out[7] = (int64_t)C(h)->ws_bytes;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_stage_blocks(pyipm_newton_ctx* h, const double* d2L, int64_t ld_d2L, const double* Je,
                              int64_t ld_Je, const double* Ji, int64_t ld_Ji, int memkind) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    int rc;
    if (!d2L && ctx->provider_only) { ctx->d2L = nullptr; ctx->ld_d2L = g.n; }     // (a factored Hessian model: only the Jacobian products)
    else { rc = stage_block(ctx, d2L, g.n, g.n, ld_d2L, memkind, &ctx->stg_d2L, &ctx->stg_d2L_sz, &ctx->d2L, &ctx->ld_d2L); if (rc) return rc; }
    rc = stage_block(ctx, Je, g.me ? g.n : 0, g.me, ld_Je, memkind, &ctx->stg_Je, &ctx->stg_Je_sz, &ctx->Je, &ctx->ld_Je); if (rc) return rc;
    rc = stage_block(ctx, Ji, g.mi ? g.n : 0, g.mi, ld_Ji, memkind, &ctx->stg_Ji, &ctx->stg_Ji_sz, &ctx->Ji, &ctx->ld_Ji); if (rc) return rc;
    ctx->have_blocks = true;
    ctx->ray_valid = false;
    ctx->rc_warm_valid[0] = ctx->rc_warm_valid[1] = false;      // (another matrix: the condition estimate starts cold)
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_stage_vectors(pyipm_newton_ctx* h, const double* df, const double* ce, const double* ci,
                               const double* s, const double* lda, double mu, double eps, int memkind) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    PYIPM_HIP(hipSetDevice(ctx->device));
    int rc;
    const size_t B = (size_t)ctx->batch;          // batched handles: every vector is [batch][len], contiguous
    rc = put_vec(ctx, ctx->df, df, B * g.n, memkind); if (rc) return rc;
    rc = put_vec(ctx, ctx->ce, ce, B * g.me, memkind); if (rc) return rc;
    rc = put_vec(ctx, ctx->ci, ci, B * g.mi, memkind); if (rc) return rc;
    rc = put_vec(ctx, ctx->s, s, B * g.mi, memkind); if (rc) return rc;
    rc = put_vec(ctx, ctx->lda, lda, B * (g.me + g.mi), memkind); if (rc) return rc;
    ctx->mu = mu; ctx->eps = eps;
    ctx->have_vectors = true;
    ctx->have_rhs = false;                        // (g = -grad belongs to the vectors staged before)
    ctx->ray_valid = false;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

}  // extern "C"
namespace pyipm { namespace drv {
int copy_out(Ctx* ctx, double* dst, const double* src_dev, size_t count, int memkind) {
    if (!dst || count == 0) return 0;
    PYIPM_HIP(hipMemcpyAsync(dst, src_dev, count * sizeof(double),
                             memkind == PYIPM_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
    if (memkind == PYIPM_MEM_HOST) PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
} }
extern "C" {

int pyipm_newton_residual(pyipm_newton_ctx* h, double* g_out, int memkind) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    int rc = residual_dev(ctx); if (rc) return rc;
    return copy_out(ctx, g_out, ctx->rhs, ctx->g.N, memkind);
} PYIPM_CATCH_H(h)

int pyipm_newton_assemble(pyipm_newton_ctx* h, double delta, double delta_c) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    PYIPM_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
    int rc = assemble_dev(ctx, delta, delta_c); if (rc) return rc;
    PYIPM_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->ev_assemble_valid = true;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)


int pyipm_newton_factor(pyipm_newton_ctx* h, pyipm_factor_stats* stats) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    // a residual is pending (pyipm.py:1717 precedes :1718): let its forward substitution trail the
    // factorisation; solve(rhs = NULL) then only runs the block-diagonal and backward parts
    const bool fuse = ctx->fuse_forward && ctx->have_rhs && ctx->g.world == 1 && ctx->assembled;
    if (fuse) { int rc = solve_prepare(ctx, nullptr, PYIPM_MEM_DEVICE, true); if (rc) return rc; }
    int rc = factor_dispatch(ctx, stats, fuse);
    if (rc) ctx->zeros_clean = false;
    ctx->forward_pending = (rc == 0 || rc == PYIPM_E_NONFINITE) ? (fuse && ctx->forward_fused) : false;
    return rc;
} PYIPM_CATCH_H(h)

// load the right-hand side into v1 (kept for refinement) and v0 (solved in place)
}  // extern "C"
namespace pyipm { namespace drv {
int solve_prepare(Ctx* ctx, const double* rhs, int memkind, bool for_fused_forward) {
    const Geo& g = ctx->g;
    if (rhs) {
        hipLaunchKernelGGL(k_fill, grid1(g.Npad), dim3(256), 0, ctx->stream, ctx->v1, 0.0, g.Npad); PYIPM_KCHECK();
        int rc = put_vec(ctx, ctx->v1, rhs, g.N, memkind); if (rc) return rc;
    } else {
        if (!ctx->have_rhs) { ctx->err = "solve: no right-hand side (call residual or pass rhs)"; return PYIPM_E_BADARG; }
        PYIPM_HIP(hipMemcpyAsync(ctx->v1, ctx->rhs, g.Npad * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    PYIPM_HIP(hipMemcpyAsync(ctx->v0, ctx->v1, g.Npad * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (for_fused_forward && ctx->cond_active) return cond_reduce(ctx, ctx->v0, ctx->vc);   // the fused pass runs on vc
    return 0;
}
} }
extern "C" {

static int solve_finish(Ctx* ctx, double* dz, int flip, int refine, int memkind, bool forward_done) {
    const Geo& g = ctx->g;
    int rc = solve_inplace(ctx, ctx->v0, forward_done); if (rc) return rc;
    if (ctx->cond_active && refine >= 0 && refine < ctx->cond_min_refine) refine = ctx->cond_min_refine;
    ctx->info_steps = 0; ctx->info_converged = 0; ctx->info_berr0 = -1.0; ctx->info_berr = -1.0;
    // refine >= 0: that many steps of   r = b - Hc x ;  x += Hc^{-1} r   (Hc applied from the blocks, not from the factor).
    // refine <  0: adaptive -- measure |r|/|b| before every step and stop at refine_target, after refine_max steps or
    //              when a step gains less than 4x.  This is what turns the factor of a statically pivoted (perturbed)
    //              matrix into the solution of the UNperturbed system; the host reads the outcome with solve_info.
    const bool adaptive = refine < 0;
    const int maxit = adaptive ? ctx->refine_max : refine;
    double prev = -1.0;
    for (int it = 0; it <= maxit; ++it) {
        if (!adaptive && it == maxit) break;
        rc = kkt_matvec_dev(ctx, ctx->v0, ctx->v2); if (rc) return rc;
        hipLaunchKernelGGL(k_axpby, grid1(g.Npad), dim3(256), 0, ctx->stream, ctx->v2, ctx->v1, ctx->v2, 1.0, -1.0, g.Npad);
        PYIPM_KCHECK();
        if (adaptive) {
            double ss[2];
            hipLaunchKernelGGL(k_sumsq2, dim3(1), dim3(1024), 0, ctx->stream, ctx->partial, ctx->v2, ctx->v1, g.N);
            PYIPM_KCHECK();
            PYIPM_HIP(hipMemcpyAsync(ss, ctx->partial, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            PYIPM_HIP(hipStreamSynchronize(ctx->stream));
            const double berr = ss[1] > 0.0 ? sqrt(ss[0] / ss[1]) : sqrt(ss[0]);
            if (it == 0) ctx->info_berr0 = berr;
            ctx->info_berr = berr;
            if (prev >= 0.0 && !(berr <= prev)) {                             // the last step made it worse (or NaN): take it back
                PYIPM_HIP(hipMemcpyAsync(ctx->v0, ctx->v3, g.Npad * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                ctx->info_berr = prev; ctx->info_steps = it - 1;
                break;
            }
            if (!(berr <= 1.0e300)) break;                                    // NaN / Inf: nothing to refine
            if (berr <= ctx->refine_target) { ctx->info_converged = 1; break; }
            if (it == maxit || (prev >= 0.0 && berr > 0.25 * prev)) break;    // out of budget / stagnating
            prev = berr;
            PYIPM_HIP(hipMemcpyAsync(ctx->v3, ctx->v0, g.Npad * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));   // the iterate this error belongs to
        }
        rc = solve_inplace(ctx, ctx->v2); if (rc) return rc;
        hipLaunchKernelGGL(k_axpby, grid1(g.Npad), dim3(256), 0, ctx->stream, ctx->v0, ctx->v0, ctx->v2, 1.0, 1.0, g.Npad);
        PYIPM_KCHECK();
        ctx->info_steps = it + 1;
    }
    // flip + copy out (device staging through v2 so host copies stay contiguous)
    hipLaunchKernelGGL(k_copy_flip, grid1(g.N), dim3(256), 0, ctx->stream, ctx->v2, ctx->v0, g.N, g.n + g.mi,
                       (flip && (g.me + g.mi) > 0) ? 1 : 0);
    PYIPM_KCHECK();
    PYIPM_HIP(hipEventRecord(ctx->ev[5], ctx->stream));
    rc = copy_out(ctx, dz, ctx->v2, g.N, memkind); if (rc) return rc;
    ctx->ev_solve_valid = true;
    ctx->have_direction = (flip != 0) || (g.me + g.mi == 0);     // v2 = dz with the reference's sign convention
    ctx->ray_valid = false;                                      // (merit_ray: the products of the OLD direction)
    return PYIPM_OK;
}

int pyipm_newton_solve(pyipm_newton_ctx* h, const double* rhs, double* dz, int flip, int refine, int memkind) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (g.world != 1) { ctx->err = "solve(): single-rank entry point"; return PYIPM_E_BADARG; }
    if (!ctx->factored) { ctx->err = "solve: factor first"; return PYIPM_E_BADARG; }
    if (!dz) { ctx->err = "solve: null output"; return PYIPM_E_BADARG; }
    PYIPM_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
    const bool pending = ctx->forward_pending && rhs == nullptr;      // v0 already holds the forward-substituted residual
    ctx->forward_pending = false;                                     // consumed (or overwritten) either way
    if (!pending) { int rc = solve_prepare(ctx, rhs, memkind); if (rc) return rc; }
    return solve_finish(ctx, dz, flip, refine, memkind, pending);
} PYIPM_CATCH_H(h)

int pyipm_newton_kkt_matvec(pyipm_newton_ctx* h, const double* v, double* y, int memkind) try {
    if (check_ctx(h) || !v || !y) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    ctx->forward_pending = false;                     // v1 (the saved right-hand side) is about to be reused
    // the product goes through vc (free outside a condensed solve), so v2 -- the direction of the last
    // solve, which step_lengths() reads -- survives a backward-error check
    hipLaunchKernelGGL(k_fill, grid1(g.Npad), dim3(256), 0, ctx->stream, ctx->v1, 0.0, g.Npad); PYIPM_KCHECK();
    int rc = put_vec(ctx, ctx->v1, v, g.N, memkind); if (rc) return rc;
    rc = kkt_matvec_dev(ctx, ctx->v1, ctx->vc); if (rc) return rc;
    return copy_out(ctx, y, ctx->vc, g.N, memkind);
} PYIPM_CATCH_H(h)

int pyipm_newton_solve_info(pyipm_newton_ctx* h, double out[4]) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    out[0] = (double)ctx->info_steps; out[1] = ctx->info_berr0; out[2] = ctx->info_berr; out[3] = (double)ctx->info_converged;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

// Restates the quantity reghess tests, rcond = min|w| / max|w| over the eigenvalues of Hc (pyipm.py:1379-1381), without
// an eigendecomposition: max|w| by power iteration on Hc applied from the blocks, min|w| by inverse iteration with
// the factor (each step one substitution sweep).  Few iterations suffice for a threshold test at eps: a singular
// direction dominates inv(Hc) by many orders of magnitude.
int pyipm_newton_rcond(pyipm_newton_ctx* h, int it_inv, int it_pow, double out[4]) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (g.world != 1) { ctx->err = "rcond(): single-rank entry point"; return PYIPM_E_BADARG; }
    if (!ctx->factored) { ctx->err = "rcond: factor first"; return PYIPM_E_BADARG; }
    // it_inv / it_pow >= 1: that many iterations of each; 0 = the defaults of rounds 2-3 (3 and 6).  < 0 = ADAPTIVE (round 4): an estimate for a
    // THRESHOLD test does not need the converged value.  Power iteration stops when two successive estimates agree to 10 %
    // (at most 6); inverse iteration stops when converged likewise (at most 3) -- or as soon as even a pessimistic correction
    // of the running estimate cannot reach the threshold: from a random start one step already finds 1 / |w|_min to within
    // the start vector's component along that eigenvector (~1/sqrt(N)), so min|w| >= 1 / (64 sqrt(N) est_1) is safe; if
    // that is still far above eps max|w| (x 100), the matrix is not singular to working precision and the other two
    // substitution sweeps (2.6 ms each at N = 32768) change nothing.  Late in an interior-point run this estimate is taken at
    // every iterate (Sigma spreads the pivots by itself): 18 -> ~7 ms.
    const bool adaptive_inv = it_inv < 0, adaptive_pow = it_pow < 0;
    if (it_inv < 1) it_inv = 3;
    if (it_pow < 1) it_pow = 6;
    // work vectors: v3 (the adaptive refinement's spare iterate) and vc (free outside a condensed solve), so that v0 / v1 -- a
    // right-hand side already forward-substituted under the factorisation -- and v2, the last direction, survive: the solve
    // that follows starts at the backward sweep (1.3 ms at N = 32768, at every suspect iterate).  With the condensed system
    // the substitution itself goes through vc: v0 / v1 as before, the pending forward pass is redone.
    double* wa = ctx->cond_active ? ctx->v0 : ctx->v3;
    double* wb = ctx->cond_active ? ctx->v1 : ctx->vc;
    if (ctx->cond_active) { ctx->forward_pending = false; ctx->have_direction = false; ctx->ray_valid = false; }
    double ss[2], lmax = 0.0, linv = 0.0;
    int used_pow = 0, used_inv = 0;
    const bool trace = getenv("PYIPM_RCOND_TRACE") != nullptr;
    auto norm_of = [&](const double* v, double* nrm) -> int {
        hipLaunchKernelGGL(k_sumsq2, dim3(1), dim3(1024), 0, ctx->stream, ctx->partial, v, v, g.N);
        PYIPM_KCHECK();
        PYIPM_HIP(hipMemcpyAsync(ss, ctx->partial, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PYIPM_HIP(hipStreamSynchronize(ctx->stream));
        *nrm = sqrt(ss[0]);
        return 0;
    };
    // Warm start (round 5, adaptive mode only): late in an interior-point run this estimate is taken at EVERY iterate, on a matrix
    // that changes little from one iterate to the next -- and so do its extreme eigenvectors.  Starting each iteration from the
    // vector the previous estimate ended with (instead of a hashed one) the first two estimates already agree: two substitution
    // sweeps instead of three, two products instead of up to six (-5 ms per late iterate at N = 32768).  Any start vector with a
    // component along the extreme eigenvector is valid; the stopping rules are unchanged; a hashed start is taken whenever the
    // stored vector is missing, of another size, or not finite.  stage_blocks forgets it.
    const bool warm_ok = (adaptive_inv || adaptive_pow) && !ctx->cond_active && getenv("PYIPM_RCOND_COLD") == nullptr;
    if (warm_ok && ctx->rc_warm_n != g.Npad) {
        if (ctx->rc_warm[0]) { PYIPM_HIP(hipStreamSynchronize(ctx->stream)); PYIPM_HIP(hipFree(ctx->rc_warm[0])); ctx->rc_warm[0] = nullptr; }
        if (hipMalloc((void**)&ctx->rc_warm[0], 2 * (size_t)g.Npad * sizeof(double)) != hipSuccess) { ctx->rc_warm[0] = nullptr; ctx->rc_warm_n = 0; }
        else { ctx->rc_warm[1] = ctx->rc_warm[0] + g.Npad; ctx->rc_warm_n = g.Npad; }
        ctx->rc_warm_valid[0] = ctx->rc_warm_valid[1] = false;
    }
    for (int phase = 0; phase < 2; ++phase) {
        const bool adaptive_phase = phase == 0 ? adaptive_pow : adaptive_inv;
        const bool warm = warm_ok && adaptive_phase && ctx->rc_warm[0] && ctx->rc_warm_valid[phase];
        // The start vector is hashed; a warm start ADDS the stored iterate to it, both normalised: the stored vector
        // has been through inverse (power) iterations on every earlier call, so along a direction that has only just become
        // (nearly) singular -- or a Sigma entry that has only just become dominant -- it holds rounding noise, and the stopping
        // rules below (two estimates agree; the pessimistic bound from the start vector's ~1/sqrt(N) component along every
        // eigenvector) would read the OLD extreme eigenvalue off it.  The hashed half keeps that component for every direction.
        hipLaunchKernelGGL(k_hash_vector, grid1(g.Npad), dim3(256), 0, ctx->stream, wb, g.N, g.Npad, (unsigned long long)(17 + phase));
        PYIPM_KCHECK();
        double nrm = 0.0, est = 0.0, prev_est = 0.0;
        int rc = norm_of(wb, &nrm); if (rc) return rc;
        if (warm && nrm > 0.0) {
            double wn = 0.0;
            rc = norm_of(ctx->rc_warm[phase], &wn); if (rc) return rc;
            if (wn > 0.0 && wn <= 1.0e300) {                             // (a stored vector that went bad is ignored)
                hipLaunchKernelGGL(k_axpby, grid1(g.Npad), dim3(256), 0, ctx->stream, wb, wb, ctx->rc_warm[phase], 1.0 / nrm, 1.0 / wn, g.Npad);
                PYIPM_KCHECK();
                rc = norm_of(wb, &nrm); if (rc) return rc;
            }
        }
        const int its = phase == 0 ? it_pow : it_inv;
        for (int it = 0; it < its; ++it) {
            if (!(nrm > 0.0) || !(nrm <= 1.0e300)) break;
            if (phase == 0) {
                hipLaunchKernelGGL(k_scale_copy, grid1(g.Npad), dim3(256), 0, ctx->stream, wa, wb, 1.0 / nrm, g.Npad); PYIPM_KCHECK();
                rc = kkt_matvec_dev(ctx, wa, wb); if (rc) return rc;          // v1 = Hc (v / |v|)
            } else {
                hipLaunchKernelGGL(k_scale_copy, grid1(g.Npad), dim3(256), 0, ctx->stream, wa, wb, 1.0 / nrm, g.Npad); PYIPM_KCHECK();
                rc = solve_inplace(ctx, wa); if (rc) return rc;                     // v0 = inv(factored) (v / |v|)
                PYIPM_HIP(hipMemcpyAsync(wb, wa, g.Npad * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            }
            rc = norm_of(wb, &nrm); if (rc) return rc;
            prev_est = est; est = nrm;
            (phase == 0 ? used_pow : used_inv) = it + 1;
            if (trace) fprintf(stderr, "[pyipm rcond] %s iteration %d: %.6e\n", phase == 0 ? "power" : "inverse", it + 1, est);
            if (phase == 0 && adaptive_pow && it >= 1 && est <= 1.1 * prev_est) break;
            if (phase == 1 && adaptive_inv) {
                if (it >= 1 && est <= 1.1 * prev_est) break;
                // 1 / (pessimistic min|w|) = 64 sqrt(N) est; not singular by a factor 100:  that < 1 / (100 eps max|w|)
                const double pess = 64.0 * sqrt((double)g.N) * est;
                if (lmax > 0.0 && pess * (100.0 * 2.220446049250313e-16 * lmax) < 1.0) break;
            }
        }
        if (phase == 0) lmax = est; else linv = est;
        if (warm_ok && adaptive_phase && ctx->rc_warm[0]) {
            // keep the iterate (unnormalised: the next call normalises it) -- if it is finite
            const bool good = nrm > 0.0 && nrm <= 1.0e300;
            if (good) PYIPM_HIP(hipMemcpyAsync(ctx->rc_warm[phase], wb, g.Npad * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            ctx->rc_warm_valid[phase] = good;
        }
    }
    ctx->rcond_its[0] = used_pow; ctx->rcond_its[1] = used_inv;
    const double lmin = linv > 0.0 ? 1.0 / linv : 0.0;
    out[0] = lmin; out[1] = lmax; out[2] = (lmax > 0.0) ? lmin / lmax : 0.0;
    double an = 0.0;
    PYIPM_HIP(hipMemcpy(&an, ctx->anorm, sizeof(double), hipMemcpyDeviceToHost));
    out[3] = 1.4901161193847656e-08 * ((an > 0.0 && an <= 1.0e300) ? an : 1.0);      // magnitude of a static pivot
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_anorm(pyipm_newton_ctx* h, double** dev_ptr) try {
    if (check_ctx(h) || !dev_ptr) return PYIPM_E_BADARG;
    *dev_ptr = reinterpret_cast<double*>(C(h)->anorm);        // bits of a non-negative double ARE that double
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

// ---- QP family: the derivative provider's products on the device ------------------------------
// The reference evaluates df = Q x + c, ce = A x - b, ci = G x - h and the Jacobian-transpose products of the KKT report
// through compiled Aesara functions on the host (pyipm.py:855-954).  For a QP the blocks are constant and already staged,
// so these are passes over the staged d2L (its UPPER triangle, as everywhere), Je and Ji.
static int block_products_dev(Ctx* ctx, const double* v, double* Qv, double* JeTv, double* JiTv);
int pyipm_newton_block_products(pyipm_newton_ctx* h, const double* v, double* Qv, double* JeTv, double* JiTv) try {
    if (check_ctx(h) || !v) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    return block_products_dev(ctx, v, Qv, JeTv, JiTv);
} PYIPM_CATCH_H(h)

static int block_products_dev(Ctx* ctx, const double* v, double* Qv, double* JeTv, double* JiTv) {
    const Geo& g = ctx->g;
    if (g.world != 1 || ctx->sharded) { ctx->err = "block_products: single-rank handles with fully staged blocks"; return PYIPM_E_BADARG; }
    if (!ctx->have_blocks) { ctx->err = "block_products: stage blocks first"; return PYIPM_E_BADARG; }
    if (Qv && !ctx->d2L) { ctx->err = "block_products: no d2L block staged on this provider-only handle (pass Qv = NULL)"; return PYIPM_E_BADARG; }
    const RowMap rm = make_rowmap(g, 0);
    const int64_t n = g.n, me = g.me, mi = g.mi;
    const int nchunk = 64;
    const int64_t rpc = (n + nchunk - 1) / nchunk;
    if (ctx->profile) PYIPM_HIP(hipEventRecord(ctx->ev_prov[0], ctx->stream));
    double bytes = 0.0;
    if (Qv) {            // sym(triu(d2L)) v: ONE pass over the upper triangle (round 4: was a row pass + a mirrored pass)
        int r2 = symv_dev(ctx, v, Qv, 0.0, rm); if (r2) return r2;
        bytes += 4.0 * (double)n * (double)n;
    }
    if (JeTv && me > 0) {
        hipLaunchKernelGGL(k_coldot_partial, dim3((unsigned)((me + 255) / 256), nchunk), dim3(256), 0, ctx->stream,
                           ctx->partial, ctx->Je, ctx->ld_Je, n, me, v, rpc, 0, rm);
        PYIPM_KCHECK();
        hipLaunchKernelGGL(k_coldot_reduce, grid1(me), dim3(256), 0, ctx->stream, JeTv, ctx->partial, me, nchunk, 0);
        PYIPM_KCHECK();
        bytes += 8.0 * (double)n * (double)me;
    }
    if (JiTv && mi > 0) {
        hipLaunchKernelGGL(k_coldot_partial, dim3((unsigned)((mi + 255) / 256), nchunk), dim3(256), 0, ctx->stream,
                           ctx->partial, ctx->Ji, ctx->ld_Ji, n, mi, v, rpc, 0, rm);
        PYIPM_KCHECK();
        hipLaunchKernelGGL(k_coldot_reduce, grid1(mi), dim3(256), 0, ctx->stream, JiTv, ctx->partial, mi, nchunk, 0);
        PYIPM_KCHECK();
        bytes += 8.0 * (double)n * (double)mi;
    }
    if (ctx->profile) { PYIPM_HIP(hipEventRecord(ctx->ev_prov[1], ctx->stream)); ctx->prov_valid[0] = true; }
    ctx->prov_bytes[0] = bytes;
    return PYIPM_OK;
}

// out (n) = Je le + Ji li  (either may be NULL): the Jacobian terms of dL/dx (pyipm.py:655-668)
int pyipm_newton_block_products_t(pyipm_newton_ctx* h, const double* le, const double* li, double* out) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (g.world != 1 || ctx->sharded) { ctx->err = "block_products_t: single-rank handles with fully staged blocks"; return PYIPM_E_BADARG; }
    if (!ctx->have_blocks) { ctx->err = "block_products_t: stage blocks first"; return PYIPM_E_BADARG; }
    const int64_t me = le ? g.me : 0, mi = li ? g.mi : 0;
    if (ctx->profile) PYIPM_HIP(hipEventRecord(ctx->ev_prov[2], ctx->stream));
    hipLaunchKernelGGL(k_rowdot2, grid1(g.n, 4), dim3(256), 0, ctx->stream, out, (const double*)nullptr, g.n,
                       ctx->Je, ctx->ld_Je, le, me, ctx->Ji, ctx->ld_Ji, li, mi, 2, 0, make_rowmap(g, 0));
    PYIPM_KCHECK();
    if (ctx->profile) { PYIPM_HIP(hipEventRecord(ctx->ev_prov[3], ctx->stream)); ctx->prov_valid[1] = true; }
    ctx->prov_bytes[1] = 8.0 * (double)g.n * (double)(me + mi);
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

// out[0] = ms of the last block_products call ("profile" = 1), out[1] = the bytes of the blocks it passed over,
// out[2], out[3] = the same for block_products_t
int pyipm_newton_provider_stats(pyipm_newton_ctx* h, double out[4]) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    PYIPM_HIP(hipSetDevice(ctx->device));
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 2; ++k) {
        float ms = 0.f;
        if (ctx->prov_valid[k]) PYIPM_HIP(hipEventElapsedTime(&ms, ctx->ev_prov[2 * k], ctx->ev_prov[2 * k + 1]));
        out[2 * k] = ms; out[2 * k + 1] = ctx->prov_bytes[k];
    }
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_step(pyipm_newton_ctx* h, double delta, double delta_c, int refine, double* dz,
                      pyipm_factor_stats* stats, int memkind) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (!dz) { ctx->err = "step: null output"; return PYIPM_E_BADARG; }
    int rc = residual_dev(ctx); if (rc) return rc;
    rc = pyipm_newton_assemble(h, delta, delta_c); if (rc) return rc;
    const bool fuse = ctx->fuse_forward != 0;
    if (fuse) { rc = solve_prepare(ctx, nullptr, memkind, true); if (rc) return rc; }     // v0 = v1 = g before factoring
    rc = factor_dispatch(ctx, stats, fuse); if (rc) { ctx->zeros_clean = false; return rc; }
    ctx->forward_pending = false;
    PYIPM_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
    if (!fuse) { rc = solve_prepare(ctx, nullptr, memkind); if (rc) return rc; }
    return solve_finish(ctx, dz, 1, refine, memkind, fuse && ctx->forward_fused);
} PYIPM_CATCH_H(h)

int pyipm_newton_step_lengths(pyipm_newton_ctx* h, double tau, const double* dz_in, double* alpha_s, double* alpha_l) try {
    if (check_ctx(h) || !alpha_s || !alpha_l) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    *alpha_s = 1.0; *alpha_l = 1.0;
    if (g.mi == 0) return PYIPM_OK;
    if (!ctx->have_vectors || (!dz_in && !ctx->have_direction)) { ctx->err = "step_lengths: stage vectors and solve first (or pass dz)"; return PYIPM_E_BADARG; }
    // v2 holds the last direction in the reference's order with the multiplier block already sign-flipped
    const double* dz = dz_in ? dz_in : ctx->v2;
    hipLaunchKernelGGL(k_step_lengths, dim3(1), dim3(256), 0, ctx->stream, ctx->partial, ctx->s, ctx->lda + g.me,
                       dz + g.n, dz + g.n + g.mi + g.me, g.mi, tau);
    PYIPM_KCHECK();
    double out[2];
    PYIPM_HIP(hipMemcpyAsync(out, ctx->partial, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    *alpha_s = out[0]; *alpha_l = out[1];
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

// ---- merit-function pieces of the line search (kernels_merit.hpp) -----------------------------
static int merit_scratch(Ctx* ctx) {
    if (ctx->merit_buf) return 0;
    // [partials | info out (MERIT_NQ) | gq (2) | dots (8) | alphas (MERIT_MAXK) | ray out (MERIT_MAXK)]
    PYIPM_HIP(hipMalloc((void**)&ctx->merit_buf, (size_t)(MERIT_NB * MERIT_NQ + MERIT_NQ + 2 + 8 + 2 * 1024) * sizeof(double)));
    return 0;
}
static const double* merit_direction(Ctx* ctx, const double* dz, const char* who) {
    if (dz) return dz;
    if (!ctx->have_direction) { ctx->err = std::string(who) + ": no direction (solve with flip first, or pass dz)"; return nullptr; }
    return ctx->v2;
}

// out[16] for the STAGED point (stage_vectors) and the direction dz (device, reference order, multipliers flipped; NULL = the
// last solve's; a handle without one: the direction entries come back as NaN):
//   [0] ||ce||_1  [1] ||ci - s||_1  [2] df.dx  [3] sum ds/(s+eps)  [4] sum log s
//   [5..8] the KKT report |dL/dx|, |s (lda_i - mu/(s+eps))|, |ce|, |ci - s| (2-norms; from g = -grad: NaN before residual())
//   [9] sum s lda_i  [10] min s lda_i  [11] |dx|_2  [12] |ds|_2
int pyipm_newton_merit_info(pyipm_newton_ctx* h, const double* dz, double* out) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (g.world != 1) { ctx->err = "merit_info: single-rank handles"; return PYIPM_E_BADARG; }
    if (!ctx->have_vectors) { ctx->err = "merit_info: stage vectors first"; return PYIPM_E_BADARG; }
    int rc = merit_scratch(ctx); if (rc) return rc;
    const double* d = dz ? dz : (ctx->have_direction ? ctx->v2 : nullptr);
    const double* gres = ctx->have_rhs ? ctx->rhs : nullptr;
    double* part = ctx->merit_buf; double* res = part + MERIT_NB * MERIT_NQ;
    hipLaunchKernelGGL(k_merit_info, dim3(MERIT_NB), dim3(256), 0, ctx->stream, part, g, ctx->df, ctx->ce, ctx->ci, ctx->s,
                       ctx->lda, d, gres, ctx->eps);
    PYIPM_KCHECK();
    hipLaunchKernelGGL(k_merit_info_final, dim3(1), dim3(64), 0, ctx->stream, res, part, MERIT_NB);
    PYIPM_KCHECK();
    double q[MERIT_NQ];
    PYIPM_HIP(hipMemcpyAsync(q, res, sizeof(q), hipMemcpyDeviceToHost, ctx->stream));
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    const double nan = __builtin_nan("");
    for (int k = 0; k < MERIT_NQ; ++k) out[k] = q[k];
    for (int k = 5; k <= 8; ++k) out[k] = sqrt(q[k]);
    out[11] = sqrt(q[11]); out[12] = sqrt(q[12]);
    if (!d) { out[2] = out[3] = out[11] = out[12] = nan; }
    if (!gres) { out[5] = out[6] = nan; }
    if (g.mi == 0) out[10] = nan;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

// out[p] = a_p . b_p, count <= 8 pairs of device vectors (host array of device pointers); fixed summation order.
int pyipm_newton_dots(pyipm_newton_ctx* h, int count, const double* const* a, const double* const* b, const int64_t* len,
                      double* out) try {
    if (check_ctx(h) || !a || !b || !len || !out || count < 1 || count > 8) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    PYIPM_HIP(hipSetDevice(ctx->device));
    int rc = merit_scratch(ctx); if (rc) return rc;
    DotPairs dp; memset(&dp, 0, sizeof(dp));
    for (int p = 0; p < count; ++p) { dp.a[p] = a[p]; dp.b[p] = b[p]; dp.len[p] = len[p]; if ((!a[p] || !b[p]) && len[p] > 0) return PYIPM_E_BADARG; }
    double* dev = ctx->merit_buf + MERIT_NB * MERIT_NQ + MERIT_NQ + 2;
    hipLaunchKernelGGL(k_dots, dim3((unsigned)count), dim3(1024), 0, ctx->stream, dev, dp);
    PYIPM_KCHECK();
    PYIPM_HIP(hipMemcpyAsync(out, dev, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

// QP family (the staged blocks are the constant Q, A', G'): phi(x + a dx, s + a ds) - phi(x, s) for K candidates a at once
// (k_merit_ray).  The products Q dx, Je' dx, Ji' dx and the two dot products are formed once per direction and kept until the
// next solve / stage_vectors.  quad: NULL = dx' d2L dx from the staged block, else *quad (a caller whose Q is not staged).
int pyipm_newton_merit_ray(pyipm_newton_ctx* h, const double* dz, double nu, double mu, const double* quad,
                           const double* alphas, int K, double* out) try {
    if (check_ctx(h) || !alphas || !out || K < 1 || K > 1024) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (g.world != 1 || ctx->sharded) { ctx->err = "merit_ray: single-rank handles with fully staged blocks"; return PYIPM_E_BADARG; }
    if (!ctx->have_vectors || !ctx->have_blocks) { ctx->err = "merit_ray: stage blocks and vectors first"; return PYIPM_E_BADARG; }
    const double* d = merit_direction(ctx, dz, "merit_ray"); if (!d) return PYIPM_E_BADARG;
    if (!quad && !ctx->d2L) { ctx->err = "merit_ray: no d2L block staged (pass quad = dx' Q dx)"; return PYIPM_E_BADARG; }
    int rc = merit_scratch(ctx); if (rc) return rc;
    const size_t need = (size_t)(g.n + g.me + g.mi + 8);
    if (ctx->ray_buf_n < need) {
        if (ctx->ray_buf) { PYIPM_HIP(hipStreamSynchronize(ctx->stream)); PYIPM_HIP(hipFree(ctx->ray_buf)); ctx->ray_buf = nullptr; ctx->ray_buf_n = 0; }
        PYIPM_HIP(hipMalloc((void**)&ctx->ray_buf, need * sizeof(double)));
        ctx->ray_buf_n = need; ctx->ray_for = nullptr;
    }
    double* qd = ctx->ray_buf; double* dce = qd + g.n; double* dci = dce + g.me;
    double* gq = ctx->merit_buf + MERIT_NB * MERIT_NQ + MERIT_NQ;
    double* dal = gq + 2 + 8; double* dout = dal + 1024;
    if (ctx->ray_for != d || !ctx->ray_valid) {
        rc = block_products_dev(ctx, d, quad ? nullptr : qd, g.me ? dce : nullptr, g.mi ? dci : nullptr); if (rc) return rc;
        DotPairs dp; memset(&dp, 0, sizeof(dp));
        dp.a[0] = ctx->df; dp.b[0] = d; dp.len[0] = g.n;
        dp.a[1] = d; dp.b[1] = qd; dp.len[1] = quad ? 0 : g.n;
        hipLaunchKernelGGL(k_dots, dim3(2), dim3(1024), 0, ctx->stream, gq, dp);
        PYIPM_KCHECK();
        ctx->ray_for = d; ctx->ray_valid = true; ctx->ray_quad_given = quad != nullptr;
    }
    if (quad) PYIPM_HIP(hipMemcpyAsync(gq + 1, quad, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    else if (ctx->ray_quad_given) { ctx->ray_valid = false; ctx->err = "merit_ray: quad given for this direction before, missing now"; return PYIPM_E_BADARG; }
    PYIPM_HIP(hipMemcpyAsync(dal, alphas, (size_t)K * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_merit_ray, dim3((unsigned)K), dim3(256), 0, ctx->stream, dout, dal, g, gq, ctx->ce, dce, ctx->ci, dci,
                       ctx->s, d + g.n, nu, mu);
    PYIPM_KCHECK();
    PYIPM_HIP(hipMemcpyAsync(out, dout, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

// ---- per-panel phases -----------------------------------------------------------------------------
int pyipm_newton_factor_begin(pyipm_newton_ctx* h) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); PYIPM_HIP(hipSetDevice(ctx->device));
    if (ctx->batched) return single_only(ctx);
    if (!ctx->assembled) { ctx->err = "factor_begin: assemble first"; return PYIPM_E_BADARG; }
    if (ctx->cond_active) { ctx->err = "per-panel phases do not apply to the condensed system; use factor()"; return PYIPM_E_BADARG; }
    ctx->grp_of.clear(); ctx->grp_off.clear(); ctx->grp_fast.clear(); ctx->grp_x.clear();      // per-panel phases: uniform group map, dense panels
    ctx->per_panel_mode = true;
    ctx->zeros_clean = false;                           // (the caller drives the panels: no promise about what gets written)
    return factor_begin(ctx);
} PYIPM_CATCH_H(h)
int pyipm_newton_factor_end(pyipm_newton_ctx* h, pyipm_factor_stats* stats) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); PYIPM_HIP(hipSetDevice(ctx->device));
    if (ctx->batched) return single_only(ctx);
    ctx->assembled = false;
    return factor_end(ctx, stats);
} PYIPM_CATCH_H(h)
int pyipm_newton_factor_panel(pyipm_newton_ctx* h, int64_t p) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); PYIPM_HIP(hipSetDevice(ctx->device));
    if (ctx->batched) return single_only(ctx);
    return factor_panel(ctx, p, ctx->stream, false);
} PYIPM_CATCH_H(h)
int pyipm_newton_trailing_update(pyipm_newton_ctx* h, int64_t p) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); PYIPM_HIP(hipSetDevice(ctx->device));
    if (ctx->batched) return single_only(ctx);
    if (p < 0 || p >= ctx->g.npanels) return PYIPM_E_BADARG;
    return trailing_update(ctx, p);
} PYIPM_CATCH_H(h)

int pyipm_newton_trailing_update_range(pyipm_newton_ctx* h, int64_t p, int64_t first, int64_t count) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); PYIPM_HIP(hipSetDevice(ctx->device));
    if (ctx->batched) return single_only(ctx);
    const Geo& g = ctx->g;
    if (p < 0 || p >= g.npanels || first <= p || count < 0) return PYIPM_E_BADARG;
    if (g.panel_c0(p) + g.panel_w(p) >= g.Npad) return PYIPM_OK;
    // locally owned panels inside [first, first+count)
    int64_t q = first;
    while (q < g.npanels && g.owner(q) != g.rank) ++q;
    int64_t last = first + count; if (last > g.npanels) last = g.npanels;
    if (q >= last) return PYIPM_OK;
    int64_t n_lp = 0;
    for (int64_t qq = q; qq < last; qq += g.world) ++n_lp;
    return timed_update(ctx, p, 1, q / g.world, n_lp);
} PYIPM_CATCH_H(h)

// message = [ W rows below the panel (m x nbw, column-major, ld = m) | nbw/64 tile inverses | nbw/64 tiles | nbw/64 flags ]
size_t pyipm_newton_panel_msg_bytes(pyipm_newton_ctx* h, int64_t p) try {
    if (check_ctx(h)) return 0;
    const Geo& g = C(h)->g;
    if (p < 0 || p >= g.npanels) return 0;
    if (panel_in_s(C(h), p)) return 0;                    // slack-block panel: every rank derives what it needs locally
    int64_t h0, h1;
    panel_hole(C(h), p, &h0, &h1);
    const int64_t nbw = g.panel_w(p), m = g.Npad - (g.panel_c0(p) + nbw) - (h1 - h0);    // slack rows of an x panel stay home
    return (size_t)(m * nbw + 2 * (nbw / TB) * TB * TB + nbw / TB) * sizeof(double);
} PYIPM_CATCH_SIZE

int pyipm_newton_panel_pack(pyipm_newton_ctx* h, int64_t p, double* buf) try {
    if (check_ctx(h) || !buf) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (p < 0 || p >= g.npanels || g.owner(p) != g.rank) { ctx->err = "panel_pack: not the owner"; return PYIPM_E_BADARG; }
    int64_t h0, h1;
    panel_hole(ctx, p, &h0, &h1);
    const int64_t nbw = g.panel_w(p), c1 = g.panel_c0(p) + nbw, m = g.Npad - c1 - (h1 - h0);
    // rows [c1, Npad) without [h0, h1): one or two strided copies into a message of leading dimension m
    const int64_t seg0 = (h1 > h0) ? h0 - c1 : g.Npad - c1, seg1 = (h1 > h0) ? g.Npad - h1 : 0;
    if (seg0 > 0)
        PYIPM_HIP(hipMemcpy2DAsync(buf, (size_t)m * sizeof(double), wbuf(ctx, p) + c1, (size_t)g.Npad * sizeof(double),
                                   (size_t)seg0 * sizeof(double), (size_t)nbw, hipMemcpyDeviceToDevice, ctx->stream));
    if (seg1 > 0)
        PYIPM_HIP(hipMemcpy2DAsync(buf + seg0, (size_t)m * sizeof(double), wbuf(ctx, p) + h1, (size_t)g.Npad * sizeof(double),
                                   (size_t)seg1 * sizeof(double), (size_t)nbw, hipMemcpyDeviceToDevice, ctx->stream));
    const size_t tbytes = (size_t)(nbw / TB) * TB * TB * sizeof(double);
    PYIPM_HIP(hipMemcpyAsync(buf + m * nbw, ctx->Dinv + (g.panel_c0(p) / TB) * (int64_t)(TB * TB), tbytes,
                             hipMemcpyDeviceToDevice, ctx->stream));
    PYIPM_HIP(hipMemcpyAsync(buf + m * nbw + (nbw / TB) * TB * TB, ctx->Tsv + (g.panel_c0(p) / TB) * (int64_t)(TB * TB),
                             tbytes, hipMemcpyDeviceToDevice, ctx->stream));
    PYIPM_HIP(hipMemcpyAsync(buf + m * nbw + 2 * (nbw / TB) * TB * TB, ctx->Tflag + g.panel_c0(p) / TB,
                             (size_t)(nbw / TB) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_panel_unpack(pyipm_newton_ctx* h, int64_t p, const double* buf) try {
    if (check_ctx(h) || !buf) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); const Geo& g = ctx->g;
    if (ctx->batched) return single_only(ctx);
    PYIPM_HIP(hipSetDevice(ctx->device));
    if (p < 0 || p >= g.npanels || g.owner(p) == g.rank) { ctx->err = "panel_unpack: owner does not unpack"; return PYIPM_E_BADARG; }
    { int rc = unpack_panel_from(ctx, p, buf, 0, /*with_tiles=*/true, ctx->stream); if (rc) return rc; }
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_fwd_panel(pyipm_newton_ctx* h, int64_t p, double* v) try {
    if (check_ctx(h) || !v) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); PYIPM_HIP(hipSetDevice(ctx->device));
    if (ctx->batched) return single_only(ctx);
    if (p < 0 || p >= ctx->g.npanels || ctx->g.owner(p) != ctx->g.rank) return PYIPM_E_BADARG;
    return fwd_panel(ctx, p, v);
} PYIPM_CATCH_H(h)
int pyipm_newton_diag_panel(pyipm_newton_ctx* h, int64_t p, double* v) try {
    if (check_ctx(h) || !v) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); PYIPM_HIP(hipSetDevice(ctx->device));
    if (ctx->batched) return single_only(ctx);
    if (p < 0 || p >= ctx->g.npanels) return PYIPM_E_BADARG;
    return diag_panel(ctx, p, v);
} PYIPM_CATCH_H(h)
int pyipm_newton_bwd_panel(pyipm_newton_ctx* h, int64_t p, double* v) try {
    if (check_ctx(h) || !v) return PYIPM_E_BADARG;
    Ctx* ctx = C(h); PYIPM_HIP(hipSetDevice(ctx->device));
    if (ctx->batched) return single_only(ctx);
    if (p < 0 || p >= ctx->g.npanels || ctx->g.owner(p) != ctx->g.rank) return PYIPM_E_BADARG;
    return bwd_panel(ctx, p, v);
} PYIPM_CATCH_H(h)

// ---- introspection --------------------------------------------------------------------------------
int pyipm_newton_kkt_storage(pyipm_newton_ctx* h, double** ptr, int64_t* ld, int64_t* ncols) try {
    if (check_ctx(h)) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (ptr) { *ptr = ctx->A; ctx->zeros_clean = false; ctx->storage_exported = true; }   // the caller may write through it, now or
                                                                // later: every assembly is a full one until "keep_zeros" is set again
    if (ld) *ld = ctx->cond_active ? ctx->gc.Npad : ctx->g.Npad;
    if (ncols) *ncols = ctx->cond_active ? ctx->gc.ncols_local : ctx->g.ncols_local;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_last_timings(pyipm_newton_ctx* h, double out[8]) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    PYIPM_HIP(hipSetDevice(ctx->device));
    PYIPM_HIP(hipStreamSynchronize(ctx->stream));          // timers are resolved lazily, never inside a step
    if (ctx->batched) {                                    // step_batched: [0] residual + assembly, [6] factorisation, [3] substitutions, [1] whole step
        for (int k = 0; k < 8; ++k) out[k] = 0.0;
        if (ctx->ev_assemble_valid) {
            float a = 0.f, f = 0.f, sv = 0.f, all = 0.f;
            PYIPM_HIP(hipEventElapsedTime(&a, ctx->ev[0], ctx->ev[2])); PYIPM_HIP(hipEventElapsedTime(&f, ctx->ev[2], ctx->ev[3]));
            PYIPM_HIP(hipEventElapsedTime(&sv, ctx->ev[3], ctx->ev[1])); PYIPM_HIP(hipEventElapsedTime(&all, ctx->ev[0], ctx->ev[1]));
            out[0] = a; out[6] = f; out[3] = sv; out[1] = all;
        }
        return PYIPM_OK;
    }
    if (ctx->ev_assemble_valid) { float ms = 0.f; PYIPM_HIP(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->t_assemble = ms; }
    if (ctx->ev_solve_valid) { float ms = 0.f; PYIPM_HIP(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5])); ctx->t_solve = ms; }
    out[0] = ctx->t_assemble; out[1] = ctx->t_panel; out[2] = ctx->t_trailing; out[3] = ctx->t_solve;
    out[4] = (double)(ctx->n_trailing_real > 0 ? ctx->n_trailing_real : ctx->n_trailing); out[5] = ctx->trailing_flops; out[6] = ctx->t_factor;
    out[7] = ctx->cond_active ? ctx->t_gram : ctx->trailing_area;   // full system: entries updated by the trailing launches
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_trailing_instances(pyipm_newton_ctx* h, double out[8]) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    for (int k = 0; k < 2; ++k) {
        out[4 * k + 0] = (double)ctx->inst_n[k]; out[4 * k + 1] = ctx->inst_ms[k];
        out[4 * k + 2] = ctx->inst_flops[k]; out[4 * k + 3] = ctx->inst_area[k];
    }
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

// The bulk launches' algorithmic bytes in both definitions (rounds must stay comparable):
// out[k] = C tiles only (16 B per updated entry: read once, written once), out[2 + k] = C tiles + the two operand panels read once
// (what trailing_instances reports); k = 0: 128 x 128 tiles, k = 1: 128 x 256.  Sums over the launches of the last factorisation.
int pyipm_newton_trailing_bytes(pyipm_newton_ctx* h, double out[4]) try {
    if (check_ctx(h) || !out) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    for (int k = 0; k < 2; ++k) { out[k] = ctx->inst_cbytes[k]; out[2 + k] = ctx->inst_area[k]; }
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_newton_set_option(pyipm_newton_ctx* h, const char* name, double value) try {
    if (check_ctx(h) || !name) return PYIPM_E_BADARG;
    Ctx* ctx = C(h);
    if (!strcmp(name, "expert")) { ctx->expert = (int)value != 0; return PYIPM_OK; }
    {   // expert switches (include/pyipm_newton.h): measurement knobs, test hooks, parked experiments
        static const char* const kExpert[] = {
            "tail_group", "group_chain", "tile_step", "tile_waves", "bc_per_problem", "chain_cpy", "chain_whole", "chain_lds_kb",
            "sweep_max_blocks", "debug_fault", "debug_timeline_ptr", "debug_chain_ptr"};
        bool is_expert = false;
        for (const char* e : kExpert) if (!strcmp(name, e)) { is_expert = true; break; }
        if (is_expert && !ctx->expert) {
            const char* env = getenv("PYIPM_EXPERT");
            if (!(env && env[0] && env[0] != '0')) {
                ctx->err = std::string("option ") + name + " is an expert switch: set PYIPM_EXPERT=1 or set_option(\"expert\", 1) first";
                return PYIPM_E_BADARG;
            }
        }
    }
    if (!strcmp(name, "pivtol_rel")) { ctx->pivtol_rel = value; return PYIPM_OK; }
    if (!strcmp(name, "reserve_cus")) { ctx->reserve_cus = (int)value; return PYIPM_OK; }
    if (!strcmp(name, "persist_rows")) { ctx->persist_rows = (int64_t)value; return PYIPM_OK; }
    if (!strcmp(name, "bulk_bn")) { ctx->bulk_bn = (int)value == 256 ? 256 : 128; return PYIPM_OK; }
    if (!strcmp(name, "tile_blocked")) { ctx->tile_blocked = (int)value != 0; ctx->tile_blocked_user = true; return PYIPM_OK; }
    if (!strcmp(name, "tile_waves")) {              // 4 | 8 (eight waves where a whole CU is to be had: first group, last 12288 rows) | 9 (eight waves everywhere: tests)
        const int v = (int)value; ctx->tile_waves = v >= 8 ? 8 : 4; ctx->tile8_rows = v == 9 ? ((int64_t)1 << 40) : 12288; return PYIPM_OK; }
    if (!strcmp(name, "bc_per_problem")) { ctx->bc_per_problem = (int)value != 0; return PYIPM_OK; }
    if (!strcmp(name, "tile_chain")) { int v = (int)value; ctx->tile_chain = v < 0 ? 0 : (v > 2 ? 2 : v); return PYIPM_OK; }
    if (!strcmp(name, "debug_chain_ptr")) {         // diagnostics (tools/chain_clock.py): device buffer of 64 x CHAIN_DBG_WORDS u64, one region per chain launch from now on
        ctx->chain_dbg = (unsigned long long*)(uintptr_t)(unsigned long long)value; ctx->chain_dbg_launch = 0; return PYIPM_OK; }
    if (!strcmp(name, "chain_lds_kb")) { int v = (int)value; ctx->chain_lds_kb = v < 0 ? 0 : (v > 110 ? 110 : v); return PYIPM_OK; }
    if (!strcmp(name, "chain_whole")) { ctx->chain_whole = (int)value != 0; return PYIPM_OK; }
    if (!strcmp(name, "chain_cpy")) { int v = (int)value; ctx->chain_cpy = v < 1 ? 1 : (v > 32 ? 32 : v); return PYIPM_OK; }
    if (!strcmp(name, "profile")) { ctx->profile = (int)value; return PYIPM_OK; }
    if (!strcmp(name, "debug_fault")) { ctx->debug_fault = (int)value; return PYIPM_OK; }
    if (!strcmp(name, "refine_target")) { ctx->refine_target = value; return PYIPM_OK; }
    if (!strcmp(name, "refine_max")) { ctx->refine_max = (int)value < 0 ? 0 : (int)value; return PYIPM_OK; }
    if (!strcmp(name, "refine_cond")) { ctx->refine_cond = value; return PYIPM_OK; }
    if (!strcmp(name, "block_refine")) { int v = (int)value; ctx->block_refine = v < 0 ? 0 : (v > 3 ? 3 : v); return PYIPM_OK; }
    if (!strcmp(name, "condensed")) {
        ctx->condensed = (int)value;                          // (several ranks: full blocks on every rank, see assemble)
        // (until round 6 this also rewrote a batched handle's tile_blocked -- 1 with the condensed form, 0 with the full one -- and so
        //  discarded an explicit choice; the blocked inversion is the default of both forms now)
        return PYIPM_OK; }
    if (!strcmp(name, "condensed_sigma_max")) { ctx->cond_sigma_max = value; return PYIPM_OK; }
    if (!strcmp(name, "condensed_refine")) { ctx->cond_min_refine = (int)value < 0 ? 0 : (int)value; return PYIPM_OK; }
    if (!strcmp(name, "lookahead")) { ctx->lookahead = (int)value < 0 ? 0 : ((int)value > 2 ? 2 : (int)value); return PYIPM_OK; }
    if (!strcmp(name, "group")) {           // may only shrink below the create-time value (workspace is sized for it)
        int v = (int)value; if (v < 1 || v > default_group(ctx->g.world, ctx->g.nb)) { ctx->err = "group out of range"; return PYIPM_E_BADARG; }
        ctx->group = v; return PYIPM_OK; }
    if (!strcmp(name, "skip_zeros")) { ctx->skip_zeros = (int)value != 0; return PYIPM_OK; }
    if (!strcmp(name, "tail_group")) { ctx->tail_group = (int)value; ctx->tail_group_user = true; return PYIPM_OK; }
    { bool handled = false; int rc = dist_set_option(ctx, name, value, &handled); if (handled) return rc; }
    if (!strcmp(name, "tile_step")) { ctx->tile_step = (int)value != 0; return PYIPM_OK; }
    if (!strcmp(name, "group_chain")) { ctx->group_chain = (int)value != 0; return PYIPM_OK; }
    if (!strcmp(name, "keep_zeros")) { ctx->keep_zeros = (int)value != 0; ctx->zeros_clean = false; ctx->storage_exported = false; return PYIPM_OK; }
    if (!strcmp(name, "wide_sub")) { ctx->wide_sub = (int)value; return PYIPM_OK; }
    if (!strcmp(name, "sweep_persist")) { ctx->sweep_persist = (int)value; return PYIPM_OK; }
    if (!strcmp(name, "sweep_max_blocks")) { ctx->sweep_max_blocks = (int)value; return PYIPM_OK; }
    if (!strcmp(name, "fuse_forward")) { ctx->fuse_forward = (int)value; return PYIPM_OK; }
    if (!strcmp(name, "debug_timeline_ptr")) {      // diagnostics: device buffer of 8 x u64 per block (pointer passed as double-encoded integer halves is lossy; use set via low 52 bits)
        ctx->dbg_buf = (unsigned long long*)(uintptr_t)(unsigned long long)value; return PYIPM_OK; }
    ctx->err = std::string("unknown option ") + name;
    return PYIPM_E_BADARG;
} PYIPM_CATCH_H(h)

int pyipm_newton_abi_version(void) { return PYIPM_NEWTON_ABI_VERSION; }

int pyipm_mfma_f64_peak(int device, int iters, double* tflops) try {
    if (!tflops || iters <= 0) return PYIPM_E_BADARG;
    if (hipSetDevice(device) != hipSuccess) return PYIPM_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PYIPM_E_HIP;
    double* d = nullptr;
    if (hipMalloc((void**)&d, 64) != hipSuccess) return PYIPM_E_NOMEM;
    const int blocks = prop.multiProcessorCount * 2;        // 8 waves per CU = 2 per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, 0, d, 16);   // warm-up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4.0 /*waves*/ * (double)iters * 32.0 * 2.0 * 16 * 16 * 4;
    *tflops = flops / (ms * 1e-3) / 1e12;
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d);
    return hipGetLastError() == hipSuccess ? PYIPM_OK : PYIPM_E_HIP;
} PYIPM_CATCH_NOH

}  // extern "C"

#pragma GCC visibility pop

// (The distributed driver -- pyipm_dist.hip / dist_impl.hpp -- and the L-BFGS direction -- pyipm_lbfgs.hip / lbfgs_impl.hpp -- are
//  translation units of their own since round 6: what they share with this file is declared in driver.hpp.)
