// driver.hpp -- what the translation units of the library share (round 6: pyipm_newton.hip = the single-rank driver and the
// C-ABI of include/pyipm_newton.h, pyipm_dist.hip = the distributed driver, pyipm_lbfgs.hip = include/pyipm_lbfgs.h).  Host
// functions only: a kernel is launched from the unit that defines it (launch_* wrappers below).
#pragma once
#include "ctx.hpp"
#include <algorithm>
#include <functional>
#include <chrono>

namespace pyipm { namespace drv {

// An exception unwinds out of the middle of a schedule: kernels may still be running on the helper streams against
// storage the next call will overwrite.  Drain them and drop the half-done state before reporting.
inline void quiesce_noexcept(Ctx* c) noexcept {
    if (!c) return;
    try {
        if (c->side) hipStreamSynchronize(c->side);
        if (c->fwd) hipStreamSynchronize(c->fwd);
        if (c->rest) hipStreamSynchronize(c->rest);
        if (c->dist) hipDeviceSynchronize();            // the distributed driver's own streams
        if (c->stream) hipStreamSynchronize(c->stream); else hipDeviceSynchronize();
        c->factored = false; c->forward_pending = false; c->forward_fused = false; c->zeros_clean = false;
    } catch (...) {}
}
#define PYIPM_SETERR_NEWTON(msg_) (quiesce_noexcept(reinterpret_cast<Ctx*>(h)), set_err_noexcept(reinterpret_cast<Ctx*>(h), (msg_)))
#define PYIPM_CATCH_H(h_)  PYIPM_CATCH_CORE(PYIPM_SETERR_NEWTON, PYIPM_E_NOMEM, PYIPM_E_HIP)


inline int check_ctx(pyipm_newton_ctx* h) { return h ? 0 : PYIPM_E_BADARG; }
inline Ctx* C(pyipm_newton_ctx* h) { return reinterpret_cast<Ctx*>(h); }
inline dim3 grid1(int64_t n, int b = 256) { return dim3((unsigned)((n + b - 1) / b)); }
// -W columns of panel p: the buffer of its group (parity-alternating) + its offset inside the group
inline double* wbuf(Ctx* ctx, int64_t p) {
    // group id / offset: uniform groups unless factor_all built a variable schedule (short groups in the tail)
    const int64_t G = ctx->group;
    const int64_t grp = (size_t)p < ctx->grp_of.size() ? ctx->grp_of[p] : p / G;
    const int64_t off = (size_t)p < ctx->grp_off.size() ? ctx->grp_off[p] : p % G;
    return ctx->Wbuf + ((grp % 3) * G + off) * ctx->g.Npad * (int64_t)ctx->g.nb;
}

int single_only(Ctx* ctx);
int put_vec(Ctx* ctx, double* dst, const double* src, size_t count, int memkind);
int copy_out(Ctx* ctx, double* dst, const double* src_dev, size_t count, int memkind);
int stage_block(Ctx* ctx, const double* src, int64_t rows, int64_t cols, int64_t ld, int memkind,
                double** stg, size_t* stg_sz, const double** out_ptr, int64_t* out_ld);
void active_ranges(const Ctx* ctx, int64_t cA, int64_t cB, int64_t* a0, int64_t* a1, int64_t* b0, int64_t* b1);
bool panel_in_s(const Ctx* ctx, int64_t p);
void panel_hole(const Ctx* ctx, int64_t p, int64_t* h0, int64_t* h1);
int launch_update128(Ctx* ctx, hipStream_t stream, const double* Lop, int64_t ldl, const double* Wop, int K,
                     int64_t row_begin, int64_t first_lp, int64_t n_lp, bool bulk = true,
                     int64_t ldw = 0, int64_t row_end = 0, int64_t col_end = 0, int64_t src_c0 = -1,
                     int ksplit = 1, int64_t ks_cstride = 0, int waves = 0,        // waves: 0 = the handle's bulk_waves
                     int head_ct = 0, unsigned* head_counter = nullptr, unsigned* head_count = nullptr, bool list_only = false,
                     int sub0 = 0, int nct_sub = 0,    // nct_sub > 0: only column tiles [sub0, sub0 + nct_sub) of local panel first_lp
                     int* used_bn = nullptr,           // out: the tile width of the instance that ran (128 / 256)
                     int prio = -1);
int timed_update(Ctx* ctx, int64_t p0, int64_t np, int64_t first_lp, int64_t n_lp, hipStream_t stream = nullptr,
                 int head_ct = 0, unsigned* head_counter = nullptr, unsigned* head_count = nullptr, bool list_only = false,
                 bool as_bulk = false);
int factor_panel(Ctx* ctx, int64_t p, hipStream_t stream, bool apply_pending = false);
int fwd_panel(Ctx* ctx, int64_t p, double* v, hipStream_t stream = nullptr, int nrhs = 1, int64_t vstride = 0);
int diag_panel(Ctx* ctx, int64_t p, double* v, hipStream_t stream = nullptr, int nrhs = 1, int64_t vstride = 0);
int bwd_panel(Ctx* ctx, int64_t p, double* v, int nrhs = 1, int64_t vstride = 0, double* part = nullptr, int64_t pstride = 0);
int factor_begin(Ctx* ctx, hipStream_t st = nullptr);
int solve_inplace(Ctx* ctx, double* v, bool forward_done = false);
int solve_plain(Ctx* ctx, double* v, bool forward_done, int nrhs = 1, int64_t vstride = 0, double* part = nullptr,
                int64_t pstride = 0);
int ensure_rest_stream(Ctx* ctx);
bool panel_piecewise_ok(const Ctx* ctx, int64_t p);
int panel_chain(Ctx* ctx, int64_t p, hipStream_t stream, int64_t xrows = 0, const unsigned* xword = nullptr, unsigned xwant = 0);
bool chain_extra_ok(const Ctx* ctx, int64_t p, int64_t xrows);
int panel_rows(Ctx* ctx, int64_t p, int64_t r0, int64_t r1, hipStream_t stream);
size_t slice_numel(const Geo& g, int64_t p, int j);
int pack_slice(Ctx* ctx, int64_t p, int j, double* buf, hipStream_t st);
int unpack_slice(Ctx* ctx, int64_t p, int j, const double* buf, const double* tiles, double* EL, hipStream_t st);
int unpack_slice_tiles(Ctx* ctx, int64_t p, const double* tiles, hipStream_t st);
int unpack_panel_from(Ctx* ctx, int64_t p, const double* buf, int64_t row_from, bool with_tiles, hipStream_t st);
int factor_end(Ctx* ctx, pyipm_factor_stats* stats);
int factor_dispatch(Ctx* ctx, pyipm_factor_stats* stats, bool fuse_forward);
int cond_reduce(Ctx* ctx, const double* b, double* vc);
int cond_expand(Ctx* ctx, const double* vc, double* v);
int kkt_matvec_dev(Ctx* ctx, const double* v, double* y);
int residual_dev(Ctx* ctx);
int solve_prepare(Ctx* ctx, const double* rhs, int memkind, bool for_fused_forward = false);
// pyipm_dist.hip
void dist_free(Ctx* ctx);
int dist_set_option(Ctx* ctx, const char* name, double value, bool* handled);
// kernels of pyipm_newton.hip on behalf of the other units
int launch_axpby(Ctx* ctx, hipStream_t st, double* out, const double* a, const double* b, double alpha, double beta, int64_t n);
int launch_fill(Ctx* ctx, hipStream_t st, double* out, double v, int64_t n);
int launch_mask_owned(Ctx* ctx, hipStream_t st, double* v, const double* b);
int launch_copy_flip(Ctx* ctx, hipStream_t st, double* out, const double* in, int flip);
int launch_sumsq2(Ctx* ctx, hipStream_t st, double* out, const double* a, const double* b, int64_t n);
int launch_inpanel_update(Ctx* ctx, hipStream_t st, dim3 grid, double* Cm, int64_t ldc, int64_t ccol, const double* Lop, int64_t ldl,
                          const double* Wop, int64_t ldw, int64_t cglob, int K, int64_t row_begin, int64_t row_end,
                          int64_t a0, int64_t a1, int64_t b0, int64_t b1, int prio);

} }  // namespace pyipm::drv
