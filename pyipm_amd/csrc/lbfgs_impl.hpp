// lbfgs_impl.hpp — host orchestration + C-ABI of the L-BFGS direction (include/pyipm_lbfgs.h).
// Included at the end of pyipm_newton.hip: the Gram launch and the factorisation / substitutions of G are
// the Newton core's own (launch_update128, factor_dispatch, solve_plain) on an internal handle of order p.
#pragma once
#include "../../include/pyipm_lbfgs.h"
#include "kernels_lbfgs.hpp"

namespace {

struct LbCtx {
    int64_t n = 0, me = 0, mi = 0, p = 0, N = 0;
    int cap = 0, rrmax = 0, device = 0, nb = 256;
    hipStream_t stream = nullptr;
    Ctx* gcx = nullptr;                 // internal handle: symmetric system of order p (geometry (p, 0, 0))
    int64_t p_pad = 0, n_pad = 0;
    char* ws = nullptr; size_t ws_bytes = 0;
    double *JT = nullptr;               // p_pad x n_pad, column-major (see kernels_lbfgs.hpp)
    int ksplit = 1;                     // split-K factor of the Gram launch (fixed by the shape)
    double *Cs = nullptr;               // ksplit partial Gram matrices, p_pad^2 each (ksplit > 1)
    double *Gc = nullptr;               // J'J of the staged Jacobians (p_pad^2): reused until they are staged again
    bool gram_valid = false;
    double *g = nullptr, *s = nullptr, *lda = nullptr, *sig = nullptr, *dz = nullptr;
    double *S = nullptr, *Y = nullptr;  // staging of host S, Y (n x cap, row-major)
    double *V = nullptr;                // n x rrmax row-major: [g_x | S-part | Y-part]
    double *Ju = nullptr, *u = nullptr; // J u (n) and u = y + X00 v11 (p_pad)
    double *P = nullptr, *R = nullptr;  // p_pad x rrmax column-major: J'V (kept) and the solved right-hand sides
    double *part = nullptr;             // J'V partial sums: nsplit x rrmax x p_pad
    double *spart = nullptr; int64_t spstride = 0;   // backward-substitution partial sums, one slab per right-hand side
    double *gpart = nullptr, *Hs = nullptr, *Ha = nullptr, *Hb = nullptr, *M2 = nullptr, *v11 = nullptr, *info = nullptr;
    int nsplit = 1;
    pyipm_lbfgs_allreduce_fn allreduce = nullptr; void* allreduce_user = nullptr;   // row-sharded use
    bool have_J = false;
    long long n_gram = 0;               // Gram launches so far
    hipEvent_t ev[9] = {};
    bool ev_valid = false;
    double gram_flops = 0;
    bool did[8] = {};
    std::string err;
};

#define LB_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            lb->err = std::string(#call) + ": " + hipGetErrorString(e__);                 \
            return PYIPM_E_HIP;                                                           \
        }                                                                                 \
    } while (0)
#define LB_KCHECK() LB_HIP(hipGetLastError())

inline LbCtx* LB(pyipm_lbfgs_ctx* h) { return reinterpret_cast<LbCtx*>(h); }

// sum over the ranks of a row-sharded problem (no-op without a callback)
int lb_allreduce(LbCtx* lb, double* buf, int64_t count) {
    if (!lb->allreduce || count <= 0) return 0;
    if (lb->allreduce(lb->allreduce_user, buf, count, (void*)lb->stream) != 0) {
        lb->err = "the all-reduce callback failed"; return PYIPM_E_COMM;
    }
    return 0;
}

// Split-K factor of the Gram launch.  J'J has only T = nt(nt+1)/2 output tiles (nt = p_pad/128) against 512 block
// slots (2 per CU), while K = n is long: splitting K fills the machine and shortens the ragged last round.
// Model (seconds): rounds of 512 blocks at the in-situ tile rate + clearing and summing the partial matrices at
// HBM speed; every split at least 1024 deep, scratch <= 1 GiB.
int lb_ksplit(int64_t p_pad, int64_t n_pad) {
    const int64_t nt = p_pad / 128, T = nt * (nt + 1) / 2;
    const double cnt = (double)p_pad * (double)p_pad;
    int best = 1; double bc = 1e300;
    for (int ks = 1; ks <= 512; ks *= 2) {
        if (ks > 1 && (n_pad / ks < 1024 || ks * cnt * 8.0 > 1073741824.0)) break;
        const double t_tile = 2.0 * 128.0 * 128.0 * (double)(n_pad / ks) / (62.0e12 / 512.0);
        const double c = (double)((T * ks + 511) / 512) * t_tile + (ks > 1 ? (2.0 * ks + 1.0) * cnt * 8.0 / 3.0e12 + 5.0e-6 : 0.0);
        if (c < bc) { bc = c; best = ks; }
    }
    return best;
}
constexpr int64_t LB_KPAD = 8192;       // n_pad granularity: BKU * the largest split factor

// Splits of the reduction over n in the J'V pass (one thread per constraint column, so a narrow J needs many):
// aim at >= 512k threads, keep >= 256 rows per split and the partial sums under 256 MB.
int lb_nsplit(int64_t n, int64_t p_pad, int rrmax) {
    if (p_pad <= 0) return 1;
    int64_t s = (524288 + p_pad - 1) / p_pad;
    const int64_t by_rows = (n + 255) / 256, by_mem = 268435456 / ((int64_t)rrmax * p_pad * 8);
    if (s > by_rows) s = by_rows;
    if (s > by_mem) s = by_mem;
    if (s < 1) s = 1;
    return (int)s;
}

// Device layout of a handle; returns total bytes, sets pointers when base != nullptr.
size_t lb_carve(LbCtx* c, int64_t n, int64_t me, int64_t mi, int cap, int64_t p_pad, int64_t n_pad, char* base) {
    Carve cv;
    const size_t D = sizeof(double);
    const int64_t p = me + mi, N = n + 2 * mi + me;
    const int rr = 2 * cap + 1;
    const int nsplit = lb_nsplit(n, p_pad, rr);
    const int ks = p > 0 ? lb_ksplit(p_pad, n_pad) : 1;
    const size_t oJT = cv.take(p > 0 ? (size_t)p_pad * (size_t)n_pad * D : 256);
    const size_t oCs = cv.take(ks > 1 ? (size_t)ks * (size_t)p_pad * (size_t)p_pad * D : 256);
    const size_t oGc = cv.take(p > 0 ? (size_t)p_pad * (size_t)p_pad * D : 256);
    const size_t og = cv.take((size_t)(N + 1) * D);
    const size_t os = cv.take((size_t)(mi + 1) * D);
    const size_t ol = cv.take((size_t)(p + 1) * D);
    const size_t osg = cv.take((size_t)(mi + 1) * D);
    const size_t odz = cv.take((size_t)(N + 1) * D);
    const size_t oS = cv.take((size_t)n * cap * D);
    const size_t oY = cv.take((size_t)n * cap * D);
    const size_t oV = cv.take((size_t)n * rr * D);
    const size_t oJu = cv.take((size_t)(n + 1) * D);
    const size_t ou = cv.take((size_t)(p_pad + 1) * D);
    const size_t oP = cv.take(p > 0 ? (size_t)p_pad * rr * D : 256);
    const size_t oR = cv.take(p > 0 ? (size_t)p_pad * rr * D : 256);
    const size_t opart = cv.take(p > 0 ? (size_t)nsplit * rr * (size_t)p_pad * D : 256);
    const int64_t spstride = ((p_pad + ROWCHUNK - 1) / ROWCHUNK + 2) * 1024;          // nchunk * nb, nb <= 1024
    const size_t osp = cv.take(p > 0 ? (size_t)rr * (size_t)spstride * D : 256);
    const size_t ogp = cv.take((size_t)LB_GBLK * (size_t)(2 * cap) * rr * D);
    const size_t oHs = cv.take((size_t)(2 * cap) * rr * D);
    const size_t oHa = cv.take((size_t)(2 * cap) * rr * D);
    const size_t oHb = cv.take((size_t)(2 * cap) * rr * D);
    const size_t oM2 = cv.take((size_t)(2 * cap) * (2 * cap) * D);
    const size_t ov = cv.take((size_t)(2 * cap + 8) * D);
    const size_t oi = cv.take(64);
    if (base) {
        c->JT = (double*)(base + oJT); c->Cs = (double*)(base + oCs); c->Gc = (double*)(base + oGc); c->ksplit = ks; c->g = (double*)(base + og); c->s = (double*)(base + os);
        c->lda = (double*)(base + ol); c->sig = (double*)(base + osg); c->dz = (double*)(base + odz);
        c->S = (double*)(base + oS); c->Y = (double*)(base + oY); c->V = (double*)(base + oV);
        c->Ju = (double*)(base + oJu); c->u = (double*)(base + ou); c->P = (double*)(base + oP); c->R = (double*)(base + oR);
        c->part = (double*)(base + opart);
        c->spart = (double*)(base + osp); c->spstride = spstride;
        c->gpart = (double*)(base + ogp); c->Hs = (double*)(base + oHs); c->Ha = (double*)(base + oHa); c->Hb = (double*)(base + oHb); c->M2 = (double*)(base + oM2);
        c->v11 = (double*)(base + ov); c->info = (double*)(base + oi);
    }
    return cv.off;
}

int lb_put(LbCtx* lb, double* dst, const double* src, size_t count, int memkind) {
    if (count == 0) return 0;
    if (!src) { lb->err = "null vector pointer"; return PYIPM_E_BADARG; }
    LB_HIP(hipMemcpyAsync(dst, src, count * sizeof(double),
                          memkind == PYIPM_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, lb->stream));
    return 0;
}

// rows x cols block (row-major, ld) -> dst with pitch dst_ld (doubles)
int lb_put2d(LbCtx* lb, double* dst, int64_t dst_ld, const double* src, int64_t ld, int64_t rows, int64_t cols, int memkind) {
    if (rows == 0 || cols == 0) return 0;
    if (!src || ld < cols) { lb->err = "bad block pointer / leading dimension"; return PYIPM_E_BADARG; }
    LB_HIP(hipMemcpy2DAsync(dst, (size_t)dst_ld * sizeof(double), src, (size_t)ld * sizeof(double),
                            (size_t)cols * sizeof(double), (size_t)rows,
                            memkind == PYIPM_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, lb->stream));
    return 0;
}

// zeta*G = J'J + diag(0, zeta/Sigma) (+ zeta*reg on the equality block) into the internal handle, then factor.
int lb_factor_G(LbCtx* lb, double zeta, double reg_e, pyipm_factor_stats* st, bool timed) {
    Ctx* gc = lb->gcx;
    const Geo& g = gc->g;
    gc->stream = lb->stream;
    const int ks = gc->xcd_swizzle ? lb->ksplit : 1;
    const int64_t cnt = g.Npad * g.Npad;
    int rc;
    if (timed) LB_HIP(hipEventRecord(lb->ev[1], lb->stream));
    if (!lb->gram_valid) {
        // J'J of the staged Jacobians -> Gc.  It depends on neither zeta nor Sigma (the factor is of zeta*G), so it is
        // computed once per stage_jacobian: every direction for linear constraints, and the regularised retry, reuse it.
        double* keep = gc->A;
        if (ks > 1) {
            // split y of ONE launch accumulates columns [y K/ks, (y+1) K/ks) of JT into its own matrix Cs[y]; Gc = sum
            LB_HIP(hipMemsetAsync(lb->Cs, 0, (size_t)ks * (size_t)cnt * sizeof(double), lb->stream));
            gc->A = lb->Cs;
            rc = launch_update128(gc, lb->stream, lb->JT, lb->p_pad, lb->JT, (int)(lb->n_pad / ks), 0, 0,
                                  (g.Npad + g.nb - 1) / g.nb, true, lb->p_pad, g.Npad, g.Npad, -1, ks, cnt);
            gc->A = keep;
            if (rc) { lb->err = gc->err; return rc; }
            hipLaunchKernelGGL(k_lb_ksum, grid1(cnt), dim3(256), 0, lb->stream, lb->Gc, lb->Cs, cnt, ks);
            LB_KCHECK();
        } else {
            LB_HIP(hipMemsetAsync(lb->Gc, 0, (size_t)cnt * sizeof(double), lb->stream));
            gc->A = lb->Gc;
            rc = launch_update128(gc, lb->stream, lb->JT, lb->p_pad, lb->JT, (int)lb->n_pad, 0, 0,
                                  (g.Npad + g.nb - 1) / g.nb, true, lb->p_pad, g.Npad, g.Npad);
            gc->A = keep;
            if (rc) { lb->err = gc->err; return rc; }
        }
        rc = lb_allreduce(lb, lb->Gc, cnt); if (rc) return rc;         // row-sharded: J'J = sum of the ranks' J_r'J_r
        lb->gram_valid = true;
        lb->n_gram++;
    }
    LB_HIP(hipMemcpyAsync(gc->A, lb->Gc, (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice, lb->stream));
    if (timed) LB_HIP(hipEventRecord(lb->ev[2], lb->stream));
    hipLaunchKernelGGL(k_lb_gram_diag, grid1(g.Npad), dim3(256), 0, lb->stream, gc->A, g.Npad, lb->p, lb->me, lb->sig,
                       zeta, reg_e);
    LB_KCHECK();
    gc->assembled = true; gc->factored = false; gc->have_rhs = false; gc->forward_pending = false; gc->cond_active = false;
    rc = factor_dispatch(gc, st, false);
    if (rc && rc != PYIPM_E_NONFINITE) { lb->err = gc->err; return rc; }
    gc->factored = true;
    return rc;
}

}  // namespace

#undef PYIPM_SETERR_NEWTON
#undef PYIPM_CATCH_H
#define PYIPM_SETERR_LBFGS(msg_) set_err_noexcept(reinterpret_cast<LbCtx*>(h), (msg_))
#define PYIPM_CATCH_H(h_)  PYIPM_CATCH_CORE(PYIPM_SETERR_LBFGS, PYIPM_E_NOMEM, PYIPM_E_HIP)

extern "C" {

size_t pyipm_lbfgs_workspace_bytes(int64_t n, int64_t me, int64_t mi, int max_pairs, int nb) try {
    if (n <= 0 || me < 0 || mi < 0 || max_pairs < 1 || max_pairs > 32) return 0;
    if (nb == 0) nb = 256;
    if (nb % 128 != 0 || nb > 1024) return 0;
    const int64_t p = me + mi;
    const int64_t p_pad = p > 0 ? (p + PADG - 1) / PADG * PADG : 0, n_pad = (n + LB_KPAD - 1) / LB_KPAD * LB_KPAD;
    size_t total = lb_carve(nullptr, n, me, mi, max_pairs, p_pad, n_pad, nullptr);
    if (p > 0) total += pyipm_newton_workspace_bytes(p, 0, 0, nb, 1, 0);
    return total;
} PYIPM_CATCH_SIZE

int pyipm_lbfgs_create(pyipm_lbfgs_ctx** out, int64_t n, int64_t me, int64_t mi, int max_pairs, int nb,
                       int device, void* stream) try {
    if (!out) return PYIPM_E_BADARG;
    *out = nullptr;
    if (n <= 0 || me < 0 || mi < 0 || max_pairs < 1 || max_pairs > 32) return PYIPM_E_BADARG;
    if (nb == 0) nb = 256;
    if (nb % 128 != 0 || nb > 1024) return PYIPM_E_BADARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return PYIPM_E_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return PYIPM_E_NODEVICE;
    LbCtx* lb = new LbCtx();
    lb->n = n; lb->me = me; lb->mi = mi; lb->p = me + mi; lb->N = n + 2 * mi + me;
    lb->cap = max_pairs; lb->rrmax = 2 * max_pairs + 1; lb->device = device; lb->nb = nb;
    lb->stream = (hipStream_t)stream;
    lb->n_pad = (n + LB_KPAD - 1) / LB_KPAD * LB_KPAD;
    if (lb->p > 0) {
        pyipm_newton_ctx* gh = nullptr;
        int rc = pyipm_newton_create(&gh, lb->p, 0, 0, nb, device, 1, 0, nullptr, 0, stream);
        if (rc) { delete lb; return rc; }
        lb->gcx = C(gh);
        lb->p_pad = lb->gcx->g.Npad;
    }
    lb->nsplit = lb_nsplit(n, lb->p_pad, lb->rrmax);
    lb->ws_bytes = lb_carve(nullptr, n, me, mi, max_pairs, lb->p_pad, lb->n_pad, nullptr);
    if (hipMalloc((void**)&lb->ws, lb->ws_bytes) != hipSuccess) {
        if (lb->gcx) pyipm_newton_destroy(reinterpret_cast<pyipm_newton_ctx*>(lb->gcx));
        delete lb; return PYIPM_E_NOMEM;
    }
    if (getenv("PYIPM_POISON_WORKSPACE")) hipMemset(lb->ws, 0xFF, lb->ws_bytes);   // test hook (tests/conftest.py): NaN wherever nothing is written first
    lb_carve(lb, n, me, mi, max_pairs, lb->p_pad, lb->n_pad, lb->ws);
    bool ok = true;
    if (lb->p > 0) ok = hipMemsetAsync(lb->JT, 0, (size_t)lb->p_pad * (size_t)lb->n_pad * sizeof(double), lb->stream) == hipSuccess;
    for (int i = 0; i < 9 && ok; ++i) ok = hipEventCreate(&lb->ev[i]) == hipSuccess;
    if (!ok) { pyipm_lbfgs_destroy(reinterpret_cast<pyipm_lbfgs_ctx*>(lb)); return PYIPM_E_HIP; }
    *out = reinterpret_cast<pyipm_lbfgs_ctx*>(lb);
    return PYIPM_OK;
} PYIPM_CATCH_NOH

int pyipm_lbfgs_destroy(pyipm_lbfgs_ctx* h) try {
    if (!h) return PYIPM_E_BADARG;
    LbCtx* lb = LB(h);
    hipSetDevice(lb->device);
    hipStreamSynchronize(lb->stream);
    for (int i = 0; i < 9; ++i) if (lb->ev[i]) hipEventDestroy(lb->ev[i]);
    if (lb->gcx) pyipm_newton_destroy(reinterpret_cast<pyipm_newton_ctx*>(lb->gcx));
    if (lb->ws) hipFree(lb->ws);
    delete lb;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

const char* pyipm_lbfgs_last_error(pyipm_lbfgs_ctx* h) {
    if (!h) return "null handle";
    return LB(h)->err.c_str();
}

int pyipm_lbfgs_set_stream(pyipm_lbfgs_ctx* h, void* stream) try {
    if (!h) return PYIPM_E_BADARG;
    LB(h)->stream = (hipStream_t)stream;
    if (LB(h)->gcx) LB(h)->gcx->stream = (hipStream_t)stream;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_lbfgs_set_allreduce(pyipm_lbfgs_ctx* h, pyipm_lbfgs_allreduce_fn fn, void* user) try {
    if (!h) return PYIPM_E_BADARG;
    LB(h)->allreduce = fn; LB(h)->allreduce_user = user;
    LB(h)->gram_valid = false;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_lbfgs_set_option(pyipm_lbfgs_ctx* h, const char* name, double value) try {
    if (!h) return PYIPM_E_BADARG;
    LbCtx* lb = LB(h);
    if (!lb->gcx) return PYIPM_OK;
    int rc = pyipm_newton_set_option(reinterpret_cast<pyipm_newton_ctx*>(lb->gcx), name, value);
    if (rc) lb->err = lb->gcx->err;
    return rc;
} PYIPM_CATCH_H(h)

int pyipm_lbfgs_stage_jacobian(pyipm_lbfgs_ctx* h, const double* Je, int64_t ld_Je, const double* Ji,
                               int64_t ld_Ji, int memkind) try {
    if (!h) return PYIPM_E_BADARG;
    LbCtx* lb = LB(h);
    if (lb->p == 0) return PYIPM_OK;
    LB_HIP(hipSetDevice(lb->device));
    int rc = lb_put2d(lb, lb->JT, lb->p_pad, Je, ld_Je, lb->n, lb->me, memkind); if (rc) return rc;
    rc = lb_put2d(lb, lb->JT + lb->me, lb->p_pad, Ji, ld_Ji, lb->n, lb->mi, memkind); if (rc) return rc;
    if (memkind == PYIPM_MEM_HOST) LB_HIP(hipStreamSynchronize(lb->stream));     // host memory is not retained
    lb->have_J = true;
    lb->gram_valid = false;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_lbfgs_direction(pyipm_lbfgs_ctx* h, const double* g, const double* s, const double* lda, double zeta,
                          int m, const double* S, int64_t ld_S, const double* Y, int64_t ld_Y,
                          const double* SS, const double* L, const double* D, double reg, double eps,
                          double* dz, int flip, int memkind, pyipm_lbfgs_stats* stats) try {
    if (!h) return PYIPM_E_BADARG;
    LbCtx* lb = LB(h);
    const int64_t n = lb->n, me = lb->me, mi = lb->mi, p = lb->p, N = lb->N;
    if (!g || !dz) { lb->err = "direction: null g / dz"; return PYIPM_E_BADARG; }
    if (m < 0 || m > lb->cap) { lb->err = "direction: m exceeds max_pairs of the handle"; return PYIPM_E_BADARG; }
    if (m > 0 && (!S || !Y || !SS || !L || !D || ld_S < m || ld_Y < m)) { lb->err = "direction: bad storage pointers"; return PYIPM_E_BADARG; }
    if (!(zeta > 0.0)) { lb->err = "direction: zeta must be positive"; return PYIPM_E_BADARG; }
    if (p > 0 && !lb->have_J) { lb->err = "direction: stage the Jacobians first"; return PYIPM_E_BADARG; }
    if (p > 0 && (!lda || (mi > 0 && !s))) { lb->err = "direction: null s / lda"; return PYIPM_E_BADARG; }
    LB_HIP(hipSetDevice(lb->device));
    hipStream_t st = lb->stream;
    const int r = 2 * m, rr = r + 1;
    pyipm_lbfgs_stats out; memset(&out, 0, sizeof(out));
    out.m = m; out.small_pivot_min = 1.0e308;
    for (int i = 0; i < 8; ++i) lb->did[i] = false;
    LB_HIP(hipEventRecord(lb->ev[0], st));

    // ---- inputs to the device; V = [g_x | S-part | Y-part]
    int rc = lb_put(lb, lb->g, g, (size_t)N, memkind); if (rc) return rc;
    if (p > 0) {
        rc = lb_put(lb, lb->lda, lda, (size_t)p, memkind); if (rc) return rc;
        rc = lb_put(lb, lb->s, s, (size_t)mi, memkind); if (rc) return rc;
    }
    const double *dS = S, *dY = Y; int64_t ldS = ld_S, ldY = ld_Y;
    if (m > 0 && memkind == PYIPM_MEM_HOST) {
        rc = lb_put2d(lb, lb->S, m, S, ld_S, n, m, memkind); if (rc) return rc;
        rc = lb_put2d(lb, lb->Y, m, Y, ld_Y, n, m, memkind); if (rc) return rc;
        dS = lb->S; dY = lb->Y; ldS = ldY = m;
    }
    // constrained: W = [zeta S, Y] scales the Hessian (:1127); unconstrained: W = [S, zeta Y] the inverse (:1167)
    const double cS = p > 0 ? zeta : 1.0, cY = p > 0 ? 1.0 : zeta;
    hipLaunchKernelGGL(k_lb_pack, grid1(n * rr), dim3(256), 0, st, lb->V, rr, lb->g, dS, ldS, dY, ldY, n, m, cS, cY);
    LB_KCHECK();
    if (m > 0) {                       // the O(m^2) bookkeeping of lbfgs_update -> the small system's constant part
        std::vector<double> M2((size_t)r * r, 0.0);
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) {
                const double ss = SS[a * m + b], l = L[a * m + b], lt = L[b * m + a], d = D[a * m + b];
                if (p > 0) {           // Minv = [[zeta SS, L], [L', -D]]   (:1140-1143)
                    M2[(size_t)a * r + b] = zeta * ss;      M2[(size_t)a * r + m + b] = l;
                    M2[(size_t)(m + a) * r + b] = lt;       M2[(size_t)(m + a) * r + m + b] = -d;
                } else {               // [[0, L], [L', D + zeta SS]]       (:1169-1171 as one system)
                    M2[(size_t)a * r + m + b] = l;
                    M2[(size_t)(m + a) * r + b] = lt;       M2[(size_t)(m + a) * r + m + b] = d + zeta * ss;
                }
            }
        LB_HIP(hipMemcpyAsync(lb->M2, M2.data(), M2.size() * sizeof(double), hipMemcpyHostToDevice, st));
        LB_HIP(hipStreamSynchronize(st));          // M2 is a local
    }

    if (p == 0) {
        // ---------------- unconstrained: dz = zeta g - [S, zeta Y] c
        if (m > 0) {
            hipLaunchKernelGGL(k_small_gram, dim3(LB_GBLK), dim3(256), (size_t)LB_GCH * (r + 1) * sizeof(double), st,
                               lb->gpart, lb->V, (int64_t)rr, (int64_t)1, 1, lb->V, (int64_t)rr, (int64_t)1, 1, r, n);
            LB_KCHECK();
            hipLaunchKernelGGL(k_small_gram_reduce, grid1(r), dim3(256), 0, st, lb->Hs, lb->gpart, r, LB_GBLK);
            LB_KCHECK();
            rc = lb_allreduce(lb, lb->Hs, r); if (rc) return rc;
            hipLaunchKernelGGL(k_small_solve, dim3(1), dim3(64), 0, st, lb->v11, lb->info, (const double*)nullptr, 0, 0,
                               lb->M2, 1.0, lb->Hs, 1, r);
            LB_KCHECK();
        }
        hipLaunchKernelGGL(k_lb_comb_x, grid1(n), dim3(256), 0, st, lb->dz, lb->V, rr, n, lb->v11, r, zeta, (const double*)nullptr);
        LB_KCHECK();
    } else {
        const int64_t ldp = lb->p_pad;
        Ctx* gc = lb->gcx;
        if (mi > 0) {
            hipLaunchKernelGGL(k_lb_sigma, grid1(mi), dim3(256), 0, st, lb->sig, lb->s, lb->lda + me, eps, mi);
            LB_KCHECK();
        }
        // ---- G: Gram launch + factorisation (+ one regularised retry, :1106-1113)
        pyipm_factor_stats fs;
        rc = lb_factor_G(lb, zeta, 0.0, &fs, true);
        if (rc && rc != PYIPM_E_NONFINITE) return rc;
        out.n_factor = 1;
        bool singular = fs.n_zero > 0 || fs.n_neg > 0 || fs.nonfinite > 0 ||
                        (mi == 0 && fs.d_max > 0 && fs.d_min / fs.d_max <= eps);
        if (singular && me > 0 && reg > 0.0) {
            rc = lb_factor_G(lb, zeta, reg, &fs, false);
            if (rc && rc != PYIPM_E_NONFINITE) return rc;
            out.regularised = 1; out.n_factor = 2;
        }
        LB_HIP(hipEventRecord(lb->ev[3], st));
        out.n_neg = fs.n_neg; out.n_zero = fs.n_zero; out.d_min = fs.d_min; out.d_max = fs.d_max;
        lb->gram_flops = 2.0 * (double)lb->n_pad * 0.5 * (double)ldp * (double)(ldp + 128);
        // ---- pass 1 over J:  P = J' V
        {
            const int64_t kper = (n + lb->nsplit - 1) / lb->nsplit;
            dim3 grid((unsigned)((ldp + 255) / 256), (unsigned)lb->nsplit, (unsigned)((rr + LB_CC - 1) / LB_CC));
            hipLaunchKernelGGL(k_tall_tn, grid, dim3(256), 0, st, lb->part, ldp, lb->JT, ldp, lb->V, rr, n, kper);
            LB_KCHECK();
            hipLaunchKernelGGL(k_tall_tn_reduce, dim3((unsigned)((ldp + 255) / 256), (unsigned)rr), dim3(256), 0, st,
                               lb->P, lb->part, ldp, rr, lb->nsplit);
            LB_KCHECK();
            rc = lb_allreduce(lb, lb->P, ldp * rr); if (rc) return rc;
            hipLaunchKernelGGL(k_lb_rhs, dim3((unsigned)((ldp + 255) / 256), (unsigned)rr), dim3(256), 0, st, lb->R, lb->P, ldp, rr,
                               p, me, lb->g, n, mi, lb->sig, zeta);
            LB_KCHECK();
        }
        LB_HIP(hipEventRecord(lb->ev[4], st));
        // ---- 2m+1 substitutions with the factor of zeta*G: column 0 -> y, column c -> X00_c
        rc = solve_plain(gc, lb->R, false, rr, ldp, lb->spart, lb->spstride);
        if (rc) { lb->err = gc->err; return rc; }
        LB_HIP(hipEventRecord(lb->ev[5], st));
        // ---- small system (W'X01 - Minv) v11 = W'Zg_x without touching J again:
        //        W'X01_x = (W'W + P_w'X00) / zeta ,  W'Zg_x = (W'g_x - P_w'y) / zeta ,  P_w = J'W (columns 1.. of P)
        if (m > 0) {
            hipLaunchKernelGGL(k_small_gram, dim3(LB_GBLK), dim3(256), (size_t)LB_GCH * (r + rr) * sizeof(double), st,
                               lb->gpart, lb->V, (int64_t)rr, (int64_t)1, 1, lb->V, (int64_t)rr, (int64_t)1, rr, r, n);
            LB_KCHECK();
            hipLaunchKernelGGL(k_small_gram_reduce, grid1(r * rr), dim3(256), 0, st, lb->Ha, lb->gpart, r * rr, LB_GBLK);
            LB_KCHECK();
            rc = lb_allreduce(lb, lb->Ha, (int64_t)r * rr); if (rc) return rc;
            hipLaunchKernelGGL(k_small_gram, dim3(LB_GBLK), dim3(256), (size_t)LB_GCH * (r + rr) * sizeof(double), st,
                               lb->gpart, lb->P, (int64_t)1, ldp, 1, lb->R, (int64_t)1, ldp, rr, r, p);
            LB_KCHECK();
            hipLaunchKernelGGL(k_small_gram_reduce, grid1(r * rr), dim3(256), 0, st, lb->Hb, lb->gpart, r * rr, LB_GBLK);
            LB_KCHECK();
            hipLaunchKernelGGL(k_lb_hs, grid1(r * rr), dim3(256), 0, st, lb->Hs, lb->Ha, lb->Hb, r, rr, zeta);
            LB_KCHECK();
            hipLaunchKernelGGL(k_small_solve, dim3(1), dim3(64), 0, st, lb->v11, lb->info, lb->Hs, rr, 1, lb->M2, -1.0,
                               lb->Hs, rr, r);
            LB_KCHECK();
        }
        LB_HIP(hipEventRecord(lb->ev[8], st));
        // ---- u = y + X00 v11 ;  dz_lambda = u ;  dz_s = (g_s + u_i) / Sigma
        hipLaunchKernelGGL(k_lb_comb_ls, grid1(p), dim3(256), 0, st, lb->dz, lb->u, lb->R, ldp, p, me, n, mi, lb->g, lb->sig,
                           lb->v11, r, flip ? -1.0 : 1.0);
        LB_KCHECK();
        // ---- pass 2 over J: ONE vector.  dz_x = (g_x - W v11 - J u) / zeta
        {
            int64_t nblk = (n + 15) / 16; if (nblk > 16384) nblk = 16384;
            hipLaunchKernelGGL(k_jvec, dim3((unsigned)nblk), dim3(256), 0, st, lb->Ju, lb->JT, ldp, lb->u, p, n);
            LB_KCHECK();
        }
        LB_HIP(hipEventRecord(lb->ev[6], st));
        hipLaunchKernelGGL(k_lb_comb_x, grid1(n), dim3(256), 0, st, lb->dz, lb->V, rr, n, lb->v11, r, 1.0 / zeta,
                           (const double*)lb->Ju);
        LB_KCHECK();
    }
    LB_HIP(hipEventRecord(lb->ev[7], st));
    LB_HIP(hipMemcpyAsync(dz, lb->dz, (size_t)N * sizeof(double),
                          memkind == PYIPM_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
    double info = 1.0e308;
    if (m > 0) LB_HIP(hipMemcpyAsync(&info, lb->info, sizeof(double), hipMemcpyDeviceToHost, st));
    LB_HIP(hipStreamSynchronize(st));
    out.small_pivot_min = info;
    lb->ev_valid = true;
    lb->did[0] = p > 0;
    if (stats) *stats = out;
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

int pyipm_lbfgs_last_timings(pyipm_lbfgs_ctx* h, double out[8]) try {
    if (!h || !out) return PYIPM_E_BADARG;
    LbCtx* lb = LB(h);
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    if (!lb->ev_valid) return PYIPM_OK;
    float ms = 0.f;
    LB_HIP(hipEventElapsedTime(&ms, lb->ev[0], lb->ev[7])); out[0] = ms;
    if (lb->did[0]) {
        LB_HIP(hipEventElapsedTime(&ms, lb->ev[1], lb->ev[2])); out[1] = ms;
        LB_HIP(hipEventElapsedTime(&ms, lb->ev[2], lb->ev[3])); out[2] = ms;
        LB_HIP(hipEventElapsedTime(&ms, lb->ev[4], lb->ev[5])); out[3] = ms;
        float a = 0.f, b = 0.f;
        LB_HIP(hipEventElapsedTime(&a, lb->ev[3], lb->ev[4]));
        LB_HIP(hipEventElapsedTime(&b, lb->ev[8], lb->ev[6]));
        out[4] = a + b;
        LB_HIP(hipEventElapsedTime(&a, lb->ev[5], lb->ev[8]));
        LB_HIP(hipEventElapsedTime(&b, lb->ev[6], lb->ev[7]));
        out[5] = a + b;
        out[6] = lb->gram_flops;
        out[7] = (double)lb->n_gram;
    }
    return PYIPM_OK;
} PYIPM_CATCH_H(h)

}  // extern "C"
