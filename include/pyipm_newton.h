/*
 * pyipm_newton.h — C-ABI of the MI355X-native Newton-step core.
 *
 * The reference (jkaardal/pyipm) has no FFI: the seam this library replaces is the
 * Python method boundary inside IPM.solve(), /root/reference/pyipm.py:1717-1725:
 *
 *     g  = -self.grad(x, s, lda)                                   # :1717
 *     Hc = self.reghess(self.hess(x, s, lda))                      # :1718
 *     dz = self.sym_solve_cmp(Hc, g.reshape((g.size,1))).reshape() # :1720-1721
 *     dz[nvar+nineq:] = -dz[nvar+nineq:]                           # :1723-1725
 *
 * Each entry point below names the reference function it replaces.  Plain C types
 * only; no torch / C++ types cross this boundary; no exceptions cross it (every entry
 * point catches: std::bad_alloc -> PYIPM_E_NOMEM, anything else -> PYIPM_E_HIP).  Every
 * function returns 0 on success or a negative PYIPM_E_* code;
 * pyipm_newton_last_error() gives the message.  A handle is not thread-safe.
 *
 * Conventions
 *   n = nvar, me = neq, mi = nineq, N = n + 2*mi + me   (pyipm.py:824-825)
 *   block order [x | s | lambda_e | lambda_i]            (pyipm.py:824-842)
 *   all data fp64.  d2L is n x n ROW-major, only its UPPER triangle is read
 *   (pyipm.py:826-827); Je is n x me row-major (= dce), Ji is n x mi row-major
 *   (= dci) (pyipm.py:486-487, 500-501).
 *   Device KKT storage: column-major, lower triangle referenced, leading dimension
 *   Npad = roundup(N,128); rows/cols N..Npad-1 are an identity pad.  Read as a
 *   row-major array this is exactly triu(H) of the reference.
 *
 * Streams and synchronisation
 *   Work is enqueued on the handle's stream (create / set_stream; NULL = the default stream; the
 *   caller passes torch's current stream).  The library also runs panel work and the fused forward
 *   substitution on internal streams, always joined back into the handle's stream before an entry
 *   point's results are visible on it.  A call whose results go to DEVICE memory (PYIPM_MEM_DEVICE)
 *   returns after enqueue: it is asynchronous with respect to the host and ordered on the stream.  A
 *   call that returns data to the HOST — a PYIPM_MEM_HOST output, pyipm_factor_stats, step lengths,
 *   timings — synchronises the stream before returning, and host INPUT buffers are never read after
 *   the call returns (the library stages them).  Device input blocks passed to stage_blocks are NOT
 *   copied: their pointers are retained until the next stage_blocks / destroy.
 */
#ifndef PYIPM_NEWTON_H
#define PYIPM_NEWTON_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYIPM_OK              0
#define PYIPM_E_BADARG       -1   /* invalid argument / call order            */
#define PYIPM_E_HIP          -2   /* a HIP runtime call failed                 */
#define PYIPM_E_NOMEM        -3   /* workspace missing or too small, or host memory exhausted */
#define PYIPM_E_NONFINITE    -4   /* NaN/Inf met during factorisation          */
#define PYIPM_E_NODEVICE     -5   /* no usable HIP device                      */
#define PYIPM_E_COMM         -6   /* the exchange failed (RCCL / a caller-supplied callback) or a distributed step timed out */

#define PYIPM_MEM_DEVICE      0   /* pointer is device memory (e.g. torch.Tensor.data_ptr()) */
#define PYIPM_MEM_HOST        1   /* pointer is host memory; the library stages it           */

#define PYIPM_TILE           64   /* block-pivot (diagonal tile) size          */
#define PYIPM_PAD           128   /* Npad granularity                          */

typedef struct pyipm_newton_ctx pyipm_newton_ctx;   /* opaque handle */

/* What the factorisation reports instead of the reference's eigen-inertia test
 * (reghess, pyipm.py:1378-1381, 1399): inertia from the signs of the block pivots. */
typedef struct pyipm_factor_stats {
    int64_t n_neg;     /* negative pivots  (must equal me+mi for correct inertia, pyipm.py:1381) */
    int64_t n_zero;    /* STATIC pivots: |d| <= pivtol_rel * max|its tile column|, i.e. a pivot Bunch-Kaufman could not
                          avoid INSIDE its 64x64 tile had cancelled to nothing.  It was replaced by sqrt(eps)*max|Hc| with
                          the sign its KKT block calls for (+ for x and s rows, - for multiplier rows) and is ALSO
                          counted in n_neg / n_pos by that sign.  The factor is then that of a slightly perturbed
                          matrix: solve with refine < 0 and read pyipm_newton_solve_info (see there)          */
    int64_t n_2x2;     /* 2x2 Bunch-Kaufman pivots taken inside tiles                              */
    int64_t n_pos;     /* positive pivots among the N real rows (n_neg + n_pos == N)                 */
    double  d_min;     /* min |pivot| over accepted real pivots                                     */
    double  d_max;     /* max |pivot|                                                               */
    double  growth;    /* max |entry| of the block factor L (growth monitor)                        */
    int64_t nonfinite; /* != 0 if a NaN/Inf was met                                                  */
} pyipm_factor_stats;

/* ---- lifetime ----------------------------------------------------------------------- */

/* Bytes of device workspace the handle needs for (n,me,mi) when the KKT columns are
 * distributed block-cyclically (panel width nb) over `world` ranks and this is `rank`.
 * nb must be a multiple of 128, at most 1024 (0 = default 256). */
size_t pyipm_newton_workspace_bytes(int64_t n, int64_t me, int64_t mi, int nb, int world, int rank);

/* Create a handle on HIP device `device`.  `workspace` is caller-owned device memory
 * (a torch tensor) of at least pyipm_newton_workspace_bytes(); pass NULL to let the
 * library hipMalloc it.  `stream` is a hipStream_t (NULL = default stream). */
int pyipm_newton_create(pyipm_newton_ctx** out, int64_t n, int64_t me, int64_t mi, int nb,
                        int device, int world, int rank, void* workspace, size_t workspace_bytes,
                        void* stream);
int pyipm_newton_destroy(pyipm_newton_ctx* ctx);
int pyipm_newton_set_stream(pyipm_newton_ctx* ctx, void* stream);
const char* pyipm_newton_last_error(pyipm_newton_ctx* ctx);
/* Geometry: out[0]=N, [1]=Npad, [2]=nb, [3]=npanels, [4]=local columns, [5]=world, [6]=rank. */
int pyipm_newton_geometry(pyipm_newton_ctx* ctx, int64_t out[8]);

/* ---- staging (host evaluates derivatives; pyipm.py:474-509 stays host-side) ---------- */

/* Constant-per-iteration blocks: Hessian of the Lagrangian and the constraint Jacobians.
 * Device pointers are retained (caller keeps them alive until re-staged); host pointers are
 * copied into library-owned staging and not retained.  Je/Ji may be NULL when me/mi == 0. */
int pyipm_newton_stage_blocks(pyipm_newton_ctx* ctx, const double* d2L, int64_t ld_d2L,
                              const double* Je, int64_t ld_Je, const double* Ji, int64_t ld_Ji,
                              int memkind);
/* Per-step vectors: df(n), ce(me), ci(mi), s(mi), lda(me+mi) and the scalars mu, eps. */
int pyipm_newton_stage_vectors(pyipm_newton_ctx* ctx, const double* df, const double* ce,
                               const double* ci, const double* s, const double* lda,
                               double mu, double eps, int memkind);

/* ---- the hot path ---------------------------------------------------------------------- */

/* replaces self.grad (pyipm.py:655-668) negated as at :1717:
 *   g = -[ df - Je.lda_e - Ji.lda_i ; lda_i - mu/(s+eps) ; ce ; ci - s ]
 * g is kept on the device as the right-hand side; g_out (N doubles, may be NULL) receives it. */
int pyipm_newton_residual(pyipm_newton_ctx* ctx, double* g_out, int memkind);

/* replaces self.hess (pyipm.py:816-844) plus the diagonal shifts reghess applies
 * (pyipm.py:1383-1397): +delta on the x block, -delta_c on the lambda_e block. */
int pyipm_newton_assemble(pyipm_newton_ctx* ctx, double delta, double delta_c);

/* replaces the factorisation inside scipy.linalg.solve (pyipm.py:18-20, 1720) and the
 * eigen-inertia test of reghess (pyipm.py:1378-1381): block LDL' with 64x64 Bunch-Kaufman
 * block pivots; stats carries the inertia.  Inertia mismatch is NOT an error. */
int pyipm_newton_factor(pyipm_newton_ctx* ctx, pyipm_factor_stats* stats);

/* replaces the substitution inside sym_solve_cmp (pyipm.py:911-914, 1720-1721) and, when
 * flip != 0, the multiplier sign flip (pyipm.py:1723-1725).  rhs == NULL uses the residual
 * kept by pyipm_newton_residual.  refine > 0 adds that many steps of fp64 iterative
 * refinement against the KKT blocks.  refine < 0 = ADAPTIVE refinement: |rhs - Hc dz| / |rhs| (Hc from the
 * blocks) is measured before every step; it stops at "refine_target" (option, default 1e-14), after
 * "refine_max" steps (default 8) or when a step gains less than 4x.  This is how the solution of the
 * UNPERTURBED system is recovered from a factor with static pivots (stats.n_zero > 0): the reference's LU
 * pivots across the whole matrix (pyipm.py:18-20), the tile-local pivot search here cannot, static pivots +
 * refinement (GESP) give the same dz with the same "no shift" decision of reghess whenever Hc is
 * non-singular -- and no convergence (solve_info) when it is not, which the host treats like the
 * reference's rcond test (pyipm.py:1379-1381). */
int pyipm_newton_solve(pyipm_newton_ctx* ctx, const double* rhs, double* dz, int flip,
                       int refine, int memkind);
/* Outcome of the last solve()/step(): out[0] = refinement steps taken, out[1] = |rhs - Hc dz|/|rhs| before the
 * first step, out[2] = after the last, out[3] = 1 when the adaptive loop met its target (out[1..2] = -1 for a
 * fixed-count solve, which does not measure). */
int pyipm_newton_solve_info(pyipm_newton_ctx* ctx, double out[4]);
/* Restates the quantity reghess tests (pyipm.py:1379-1381: rcond = min|w| / max|w| over the eigenvalues w of Hc, "singular"
 * when rcond <= eps) without the eigendecomposition: it_pow power iterations on Hc applied from the blocks give max|w|,
 * it_inv inverse iterations through the factor (one substitution sweep each) give min|w| (0 = defaults 6 / 3; negative =
 * ADAPTIVE, for the threshold test only: either iteration stops once two successive estimates agree to 10 %, and the
 * inverse iteration also as soon as even a pessimistic correction of its running estimate -- x 64 sqrt(N): a random start
 * vector's component along the extreme eigenvector -- leaves rcond a factor 100 above eps; the estimate returned is then
 * the running one, an over-estimate of rcond that decides "not singular" correctly).
 * out[0] = min|w| estimate (of the FACTORED matrix: with static pivots that is the perturbed one, whose smallest
 * eigenvalue sits at the perturbation level out[3] exactly when Hc itself is singular), out[1] = max|w| estimate,
 * out[2] = their ratio, out[3] = magnitude of a static pivot (sqrt(eps) max|Hc|).  Call between factor() and solve();
 * a forward substitution fused into factor() is redone by the next solve(). */
int pyipm_newton_rcond(pyipm_newton_ctx* ctx, int it_inv, int it_pow, double out[4]);
/* Device address of max |assembled entry| (one double, valid after assemble): the scale of a static pivot.  Ranks of
 * a distributed factorisation reduce it (MAX) so that every rank perturbs alike. */
int pyipm_newton_anorm(pyipm_newton_ctx* ctx, double** dev_ptr);

/* y = Hc * v with Hc applied from the staged blocks (never from the factor): used by the
 * refinement step and by the parity tests (backward error). */
int pyipm_newton_kkt_matvec(pyipm_newton_ctx* ctx, const double* v, double* y, int memkind);

/* Fused convenience: residual + assemble + factor + solve + flip = pyipm.py:1717-1725
 * without regularisation retries (the host loop re-issues assemble/factor with new shifts). */
int pyipm_newton_step(pyipm_newton_ctx* ctx, double delta, double delta_c, int refine,
                      double* dz, pyipm_factor_stats* stats, int memkind);

/* SURVEY.md section 8(f) rank 3 -- the derivative provider of the QP family on the device.  For
 *   min 1/2 x'Qx + c'x  s.t.  Ax = b,  Gx - h >= 0
 * the reference's compiled provider functions (pyipm.py:855-954: df, ce, ci; the J lambda terms of self.grad,
 * :655-668) are products with the constant blocks stage_blocks already holds: d2L = Q (UPPER triangle read, as
 * everywhere), Je = A', Ji = G'.  All pointers DEVICE memory; any output may be NULL.
 *   block_products   : Qv = sym(triu(d2L)) v (n),  JeTv = Je' v (me),  JiTv = Ji' v (mi)     -- df, ce, ci, merit ray
 *   block_products_t : out = Je le + Ji li (n; le / li may be NULL)                           -- dL/dx
 *   provider_stats   : ms ("profile" = 1) and block bytes of the last call of each */
int pyipm_newton_block_products(pyipm_newton_ctx* ctx, const double* v, double* Qv, double* JeTv, double* JiTv);
int pyipm_newton_block_products_t(pyipm_newton_ctx* ctx, const double* le, const double* li, double* out);
int pyipm_newton_provider_stats(pyipm_newton_ctx* ctx, double out[4]);
/* A provider-only handle (round 3): the same staging calls, block products, residual and kkt_matvec, but no KKT storage,
 * no W buffers and no tile inverses -- O(N) workspace instead of O(N^2) -- for callers that never factor the KKT matrix
 * (the limited-memory mode, pyipm.py:1007-1246: QPDeviceIPM(lbfgs=m) forms df, ce, ci and the J lambda terms through it).
 * stage_blocks may pass d2L = NULL (a factored Hessian model: Jacobian products only; block_products then needs
 * Qv = NULL).  assemble / factor / solve / step return PYIPM_E_BADARG. */
size_t pyipm_newton_workspace_bytes_provider(int64_t n, int64_t me, int64_t mi);
int pyipm_newton_create_provider(pyipm_newton_ctx** ctx, int64_t n, int64_t me, int64_t mi, int device,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* SURVEY.md section 8(f) rank 1 — replaces the two IPM.step calls of the inner loop (pyipm.py:1408-1436,
 * 1737-1742): the largest alpha in [0,1] with v + alpha*dv >= (1-tau)*v, for v = s (dv = ds) and
 * v = lda_i (dv = dlda_i), from the staged s / lda and the direction of the last solve()/step() kept on
 * the device.  Closed form min(1, min_{dv_i<0} -tau*v_i/dv_i); the reference's golden-section search
 * converges to the same value from below (to Xtol = eps).  With mi == 0 both are 1.
 * dz: NULL = the direction of the last solve()/step(); else a DEVICE pointer to N doubles in the reference's order with the
 * multiplier block already sign-flipped (as merit_info / merit_ray take it) -- a host loop that steps along another direction
 * than the handle's last solve (an earlier, less shifted one kept by its regularisation loop) passes that one. */
int pyipm_newton_step_lengths(pyipm_newton_ctx* ctx, double tau, const double* dz, double* alpha_s, double* alpha_l);

/* SURVEY.md section 8(f) rank 1, second half — the merit-function pieces of the line search as device reductions over the
 * STAGED point (stage_vectors: df, ce, ci, s, lda) and a direction dz (device pointer, N doubles, reference order with the
 * multiplier block already sign-flipped; NULL = the direction of the last solve()/step(), kept on the device).  Replaces
 * the host passes of phi / dphi (pyipm.py:670-721, used by search :1438-1565), the merit-parameter update (:1727-1735),
 * the KKT report (:958-991) and the barrier update (:1804-1814).  One small D2H per call; sums are taken over a fixed
 * partition in a fixed order (bit-reproducible, independent of the launch).
 * merit_info: out[16] =
 *   [0] ||ce||_1   [1] ||ci - s||_1   [2] df . dx   [3] sum ds_i / (s_i + eps)   [4] sum log s_i
 *   [5] |dL/dx|_2  [6] |s * (lda_i - mu/(s + eps))|_2   [7] |ce|_2   [8] |ci - s|_2      (5, 6: from g = -grad, i.e. after residual(); else NaN)
 *   [9] sum s_i lda_i   [10] min s_i lda_i   [11] |dx|_2   [12] |ds|_2   [13..15] 0
 *   so that  phi = f + nu ([0] + [1]) - mu [4],  dphi = [2] - nu ([0] + [1]) - mu [3],  nu threshold = ([2] - mu [3]) / ((1 - rho)([0] + [1])).
 *   Entries that need a direction are NaN when there is none. */
int pyipm_newton_merit_info(pyipm_newton_ctx* ctx, const double* dz, double* out16);
/* out[p] = a_p . b_p for count <= 8 pairs of DEVICE vectors (a, b, len: host arrays) -- e.g. f(x) = x'(Qx)/2 + c'x from the
 * provider's product; fixed summation order; works on any handle. */
int pyipm_newton_dots(pyipm_newton_ctx* ctx, int count, const double* const* a, const double* const* b, const int64_t* len,
                      double* out);
/* QP family (the staged blocks are the constant Q, A', G' of min x'Qx/2 + c'x, Ax = b, Gx - h >= 0): the merit function
 * ALONG THE RAY as a difference,  out[k] = phi(x + a_k dx, s + a_k ds) - phi(x, s)  for K <= 1024 candidates a_k at once
 * (host arrays alphas / out) -- one launch evaluates every backtracking candidate alpha * tau^k of a search
 * (pyipm.py:1534-1548 shrinks alpha by tau = 0.995 per trial).  f is quadratic and the constraints affine, so with
 * g1 = df . dx, g2 = dx' Q dx, dce = Je' dx, dci = Ji' dx (formed once per direction from the staged blocks and kept):
 *   a g1 + a^2 g2 / 2 + nu sum(|ce + a dce| - |ce|) + nu sum(|ci - s + a (dci - ds)| - |ci - s|) - mu sum log1p(a ds / s),
 * every term a difference formed element by element (error relative to the CHANGE of phi).  quad: NULL = g2 from the staged
 * d2L block, else *quad is taken for g2 (handles whose Q is not a staged dense block). */
int pyipm_newton_merit_ray(pyipm_newton_ctx* ctx, const double* dz, double nu, double mu, const double* quad,
                           const double* alphas, int K, double* out);

/* ---- per-panel phases (multi-GPU host orchestration; single-rank factor() loops these) --- */

/* Factor panel p on its owner: left-looking in-panel updates, tile inversions, scaling. */
int pyipm_newton_factor_panel(pyipm_newton_ctx* ctx, int64_t p);
/* Owner: pack rows below panel p of W plus its tile inverses into `buf` for broadcast.
 * Non-owner: unpack a received buffer and rebuild the block column L = W * inv(T).
 * pyipm_newton_panel_msg_bytes gives the buffer size. */
size_t pyipm_newton_panel_msg_bytes(pyipm_newton_ctx* ctx, int64_t p);
int pyipm_newton_panel_pack(pyipm_newton_ctx* ctx, int64_t p, double* buf);
int pyipm_newton_panel_unpack(pyipm_newton_ctx* ctx, int64_t p, const double* buf);
/* Rank-nb trailing update of every locally owned column to the right of panel p. */
int pyipm_newton_trailing_update(pyipm_newton_ctx* ctx, int64_t p);
/* Same, restricted to the locally owned panels q with first <= q < first+count (lookahead:
 * the next owner updates and factors its panel before the bulk of the update). */
int pyipm_newton_trailing_update_range(pyipm_newton_ctx* ctx, int64_t p, int64_t first, int64_t count);
/* Finish: fetch statistics accumulated by this rank's tile kernels. */
int pyipm_newton_factor_begin(pyipm_newton_ctx* ctx);
int pyipm_newton_factor_end(pyipm_newton_ctx* ctx, pyipm_factor_stats* stats);
/* Panel-wise substitution phases (vector `v` is Npad doubles on the device). */
int pyipm_newton_fwd_panel(pyipm_newton_ctx* ctx, int64_t p, double* v);
int pyipm_newton_diag_panel(pyipm_newton_ctx* ctx, int64_t p, double* v);
int pyipm_newton_bwd_panel(pyipm_newton_ctx* ctx, int64_t p, double* v);

/* ---- distributed driver: one call per phase, the per-panel schedule runs inside the library -------------------
 * (SURVEY.md section 8e / 8b "Multi-GPU").  One process per GPU, handle created with (world, rank).  The per-panel
 * phases above remain for callers that orchestrate themselves; these entry points run the whole one-panel-lookahead
 * schedule (owner: head update -> factor -> pack -> broadcast on helper streams; everyone: unpack -> bulk update on
 * the handle's stream) and the sweeps in C, so no interpreter sits between two panels.
 *
 * Exchange, variant 1 -- caller-supplied collectives (like pyipm_lbfgs_set_allreduce).  A callback must leave its
 * result ordered on `stream` (a hipStream_t): enqueue on it, or complete before returning.  bcast: `bytes` bytes at
 * dev_buf from rank `root` to every rank, in place.  allreduce: `count` doubles in place, op 0 = sum, 1 = max.
 * Non-zero return = failure (PYIPM_E_COMM).
 * STREAMS: the callbacks are invoked with DIFFERENT streams -- the collective stream for panel broadcasts, the stream of
 * the trailing forward substitution for its segment sums, the handle's stream for statistics and the static-pivot scale --
 * and operations on those streams may be in flight at the same time.  A transport whose communicator must not run two
 * operations concurrently (NCCL / RCCL: one communicator, one stream at a time) has to serialise them itself, e.g. by
 * hopping every call through one stream of its own with a pair of events (what the library does for the communicator it
 * owns, variant 2), or by completing each call before it returns. */
typedef int (*pyipm_bcast_fn)(void* user, void* dev_buf, size_t bytes, int root, void* stream);
typedef int (*pyipm_allreduce_fn)(void* user, double* dev_buf, size_t count, int op, void* stream);
int pyipm_newton_set_exchange(pyipm_newton_ctx* ctx, pyipm_bcast_fn bcast, pyipm_allreduce_fn allreduce, void* user);
/* The optional point-to-point half of variant 1 (round 5).  send / recv: `bytes` bytes at dev_buf to / from rank `peer`;
 * allgather: every rank contributes bytes_per_rank bytes at send_buf, recv_buf receives the W pieces in rank order (send_buf
 * may point into recv_buf at this rank's place: in place).  Same stream rule as above; `user` is set_exchange's.  With all
 * three installed the library runs over the callbacks everything it runs over its own communicator: the scatter + all-gather
 * form of the panel messages (after pyipm_newton_exchange_selftest has reproduced the plain broadcast with it) and the slice
 * messages of the two-message protocol as point-to-point sends (without them a slice travels as a broadcast).  serialize != 0:
 * the transport must not run two operations at once -- the library then issues every operation on its collective stream, one
 * after the other, hopping from the stream that asked with a pair of events (what it does for RCCL). */
typedef int (*pyipm_send_fn)(void* user, const void* dev_buf, size_t bytes, int peer, void* stream);
typedef int (*pyipm_recv_fn)(void* user, void* dev_buf, size_t bytes, int peer, void* stream);
typedef int (*pyipm_allgather_fn)(void* user, const void* send_buf, void* recv_buf, size_t bytes_per_rank, void* stream);
int pyipm_newton_set_exchange_p2p(pyipm_newton_ctx* ctx, pyipm_send_fn send, pyipm_recv_fn recv, pyipm_allgather_fn allgather,
                                  int serialize);
/* COLLECTIVE (every rank or none): the self-test comm_init runs on a fresh communicator, on whichever exchange is installed.
 * The ranks agree on whether every one of them wants and can run the scatter + all-gather form (three ranks or more, all
 * primitives present, PYIPM_DIST_SAG != 0), reproduce the plain broadcast with it (two roots, a count that does not divide by
 * the number of ranks), agree on the outcome, and switch the form on for all or for none (pyipm_newton_comm_bcast_mode). */
int pyipm_newton_exchange_selftest(pyipm_newton_ctx* ctx);
/* Wire accounting of the last factor_dist / step_dist on this rank: out[0] panel messages that travelled as a plain broadcast,
 * [1] their bytes, [2] panel messages in the scatter + all-gather form, [3] their bytes, [4] point-to-point pieces sent or
 * received, [5] all-gathers, [6] hops through the collective stream, [7] slice messages sent or received (two-message
 * protocol), [8] their bytes, [9] slices that travelled as a broadcast, [10] ms of the owner's rows work behind the chain path,
 * [11] tile chains of this rank's panels that took the second slice's rows along in their own launch (dist_slices = 2). */
int pyipm_newton_dist_wire(pyipm_newton_ctx* ctx, double out[12]);
/* Exchange, variant 2 -- the handle owns an RCCL communicator over the `world` ranks it was created for.  Rank 0
 * obtains an id (128 bytes), it reaches the other ranks out of band (torch.distributed, MPI, a file), every rank calls
 * comm_init.  RCCL is dlopen'ed at that point (pyipm_newton_rccl_library names the library to bind: a process that has
 * torch loaded should name torch's own librccl; default: one already loaded, then librccl.so.1). */
int pyipm_newton_rccl_library(const char* path);
int pyipm_newton_comm_unique_id(void* id128);
int pyipm_newton_comm_init(pyipm_newton_ctx* ctx, const void* id128);
/* Number of ranks of the handle's RCCL communicator as RCCL itself reports it (ncclCommCount); 0 = no communicator
 * (callback exchange or single rank), negative = error.  bench.py prints it so that a multi-GPU line proves RCCL saw N ranks. */
int pyipm_newton_comm_ranks(pyipm_newton_ctx* ctx);
/* How the panel messages of factor_dist travel over the handle's communicator: 1 = scatter + all-gather (the owner sends
 * piece r to rank r, then an in-place all-gather: W - 1 links of the xGMI mesh at once instead of the one link's bandwidth of
 * ncclBroadcast's ring), 0 = ncclBroadcast.  comm_init switches the first form on for three ranks or more after it has
 * reproduced ncclBroadcast on that communicator (two roots, a count that does not divide by the number of ranks; all ranks
 * agree -- first on whether every rank wants the form at all, so PYIPM_DIST_SAG=0 on ONE rank switches it off for all); the
 * environment variable PYIPM_DIST_SAG=0 or set_option("dist_sag", 0) keep ncclBroadcast.  set_option("dist_sag", 0) is
 * COLLECTIVE: call it on every rank or on none (a subset would mix the two forms inside one exchange).  Messages below
 * 4 MiB and the nb-long exchanges of the sweeps always use ncclBroadcast. */
int pyipm_newton_comm_bcast_mode(pyipm_newton_ctx* ctx);
/* Row-sharded staging: a rank assembles only the KKT columns it owns, and column j (j < n) of the lower triangle is
 * row j of triu(d2L) | Je | Ji -- so it needs only those rows.  owned_rows returns their number and (rows != NULL)
 * their global indices in the order the arrays must hold them (= the rank's local column order).  After
 * stage_blocks_owned, residual / assemble / kkt_matvec address the blocks by local row. */
int64_t pyipm_newton_owned_rows(pyipm_newton_ctx* ctx, int64_t* rows);
int pyipm_newton_stage_blocks_owned(pyipm_newton_ctx* ctx, const double* d2L_rows, int64_t ld_d2L, const double* Je_rows,
                                    int64_t ld_Je, const double* Ji_rows, int64_t ld_Ji, int memkind);
/* The phases of pyipm.py:1717-1725 over the ranks (vectors replicated: every rank passes the same s / lda / rhs and
 * receives the same g / dz; statistics reduced over the ranks).  refine as in pyipm_newton_solve (Hc applied from the
 * ranks' blocks, one N-long sum per product).  With world == 1 they run the same per-panel schedule on one GPU. */
int pyipm_newton_residual_dist(pyipm_newton_ctx* ctx, double* g_out, int memkind);
int pyipm_newton_factor_dist(pyipm_newton_ctx* ctx, pyipm_factor_stats* stats);
int pyipm_newton_solve_dist(pyipm_newton_ctx* ctx, const double* rhs, double* dz, int flip, int refine, int memkind);
int pyipm_newton_kkt_matvec_dist(pyipm_newton_ctx* ctx, const double* v, double* y, int memkind);
int pyipm_newton_step_dist(pyipm_newton_ctx* ctx, double delta, double delta_c, int refine, double* dz,
                           pyipm_factor_stats* stats, int memkind);
/* ms of the last factor_dist / solve_dist ("profile" = 1): out[0] factorisation wall, [1] this rank's panel
 * factorisations, [2] packing, [3] broadcasts as seen on the collective stream, [4] unpacking (rebuild of L),
 * [5] solve wall, [6] bytes this rank put on / took off the wire, [7] messages. */
int pyipm_newton_dist_timings(pyipm_newton_ctx* ctx, double out[8]);

/* ---- batched small systems (BASELINE.json configs[4]: 512 independent n=256 QPs) ------------------------
 * Independent problems of one shape, Npad = roundup(n+2mi+me,128) <= 1024.  One workgroup per problem: a
 * single launch factors the whole batch.  Blocks are caller-owned DEVICE arrays with a batch stride (in
 * doubles); pyipm_newton_stage_vectors on such a handle takes [batch][len] contiguous arrays.  Replaces a
 * loop of pyipm.py:1717-1725 over independent problems (multi-start); nothing in the reference batches. */
size_t pyipm_newton_workspace_bytes_batched(int64_t n, int64_t me, int64_t mi, int batch);
int pyipm_newton_create_batched(pyipm_newton_ctx** ctx, int64_t n, int64_t me, int64_t mi, int batch, int device,
                                void* workspace, size_t workspace_bytes, void* stream);
int pyipm_newton_stage_blocks_batched(pyipm_newton_ctx* ctx, const double* d2L, int64_t ld_d2L, int64_t stride_d2L,
                                      const double* Je, int64_t ld_Je, int64_t stride_Je,
                                      const double* Ji, int64_t ld_Ji, int64_t stride_Ji);
/* residual + assemble + factor + solve + flip for every problem; dz is [batch][N], stats (may be NULL) [batch].
 * set_option("condensed", 1) (round 5): per problem the (s_k, lambda_i_k) pairs with Sigma_k <= "condensed_sigma_max" are
 * eliminated analytically (pyipm.py:824-842's block structure), the rest stay as rows -1/Sigma_k: n + me + |A| columns are
 * factored instead of n + 2 mi + me (BASELINE config 5: 256 instead of 768).  Same inputs, same outputs: the full
 * [dx|ds|dle|dli] and the inertia of the FULL matrix.  pyipm_newton_last_timings on a batched handle: out[0] residual +
 * assembly, [6] factorisation, [3] substitutions, [1] the whole step (ms). */
int pyipm_newton_step_batched(pyipm_newton_ctx* ctx, double delta, double delta_c, double* dz,
                              pyipm_factor_stats* stats, int memkind);
/* The per-problem statistics of the last pyipm_newton_step_batched (B records), for a step that was called with
 * stats = NULL -- such a step returns once its launches are enqueued; this call synchronises the handle's stream.  Returns
 * PYIPM_E_NONFINITE when a problem met NaN/Inf (the records are filled in all the same).  (The reference reads the inertia of
 * every system eagerly, pyipm.py:1378-1381: one eigendecomposition per problem.) */
int pyipm_newton_stats_batched(pyipm_newton_ctx* ctx, pyipm_factor_stats* stats);

/* out[b] = |g - Hc raw_b| / |g| for every problem of the last step_batched: Hc applied from the staged blocks with that
 * step's shifts (never from the factor), raw = dz ([batch][N], DEVICE) with the multiplier flip undone.  The guard of the
 * condensed form (the host falls back to the full system when a problem misses its bar) and the batched counterpart of the
 * kkt_matvec check of a single system.  out: batch doubles, host or device per memkind. */
int pyipm_newton_backward_error_batched(pyipm_newton_ctx* ctx, const double* dz, double* out, int memkind);

/* ---- introspection for tests / bench ------------------------------------------------------ */

/* Device pointer + leading dimension of the local KKT storage (column-major lower).  The caller may write through the
 * pointer: handing it out makes EVERY following assemble() store every entry again (otherwise the zeros that no elimination
 * step can fill in are left in place from one assembly to the next, option "keep_zeros") -- the holder of the pointer may
 * write at any later time, so the shortcut stays off for this handle until set_option("keep_zeros", 1) is called again
 * (a promise that nothing writes through the pointer any more). */
int pyipm_newton_kkt_storage(pyipm_newton_ctx* ctx, double** ptr, int64_t* ld, int64_t* ncols);
/* Time (ms, HIP events on the handle's stream) of the phases of the last factor/solve call:
 * out[0]=assemble, [1]=panel work (factor time during which no update launch ran), [2]=trailing updates (sum of the launches'
 * durations -- a lookahead head on the side stream may overlap the bulk launch behind it, so [1]+[2] can exceed [6]),
 * [3]=solve, [4]=#trailing launches,
 * [5]=algorithmic flops of those launches, [6]=factor, [7]=Ji Sigma Ji' launch (condensed option) or, for the
 * full system, the number of matrix entries those launches update (their C-tile traffic is 16 B each). */
int pyipm_newton_last_timings(pyipm_newton_ctx* ctx, double out[8]);

/* The bulk trailing-update launches of the last factorisation by kernel instance (profile option on): out[0..3] = launches,
 * summed HIP-event duration (ms), flops, algorithmic bytes (C tiles read and written once + the two operand panels read
 * once) of the 128 x 128-tile launches (k_update<128,true,8>),
 * out[4..7] = the same for the 128 x 256-tile launches (k_update<256,true,8>) -- what a kernel trace lists under the two
 * names.  Measurement only; nothing in the reference. */
int pyipm_newton_trailing_instances(pyipm_newton_ctx* h, double out[8]);
/* The same launches' algorithmic bytes in both definitions: out[k] = C tiles only (16 B per updated entry), out[2 + k] = C tiles +
 * the two operand panels read once; k = 0: 128 x 128 tiles, k = 1: 128 x 256. */
int pyipm_newton_trailing_bytes(pyipm_newton_ctx* ctx, double out[4]);
/* Options.  Every one defaults to the measured-best setting; the public ones (28 names) choose numerics, forms of the
 * system and the few schedule parameters a deployment may have to adapt; the 12 EXPERT switches (test hooks, diagnostics
 * and the reference forms the bit-identity tests compare against) are refused unless the process has PYIPM_EXPERT=1 in
 * its environment or the handle was given set_option("expert", 1) -- so that nothing outside the tests and tools/ runs a
 * path nobody else runs.  Round 6 removed the 35 switches whose measurements HISTORY.md records as lost or neutral (with
 * the code paths behind the ones that lost): 40 names in all.  tests/test_gpu_symmetric.py checks both lists against the
 * library and runs the bitwise-neutrality sweep over every schedule option in them.
 *
 * PUBLIC OPTIONS: expert, condensed, condensed_sigma_max, condensed_refine, block_refine, refine_cond, refine_target,
 *   refine_max, pivtol_rel, tile_blocked, profile, skip_zeros, keep_zeros, fuse_forward, sweep_persist, lookahead, group,
 *   bulk_bn, reserve_cus, persist_rows, tile_chain, wide_sub, dist_sag, dist_sag_min_bytes, dist_slices, dist_comm2,
 *   dist_timeout_s, dist_selfmsg
 *
 *   "expert" 0|1        unlock the expert switches on this handle.
 *   "condensed" 0|1     handles with mi > 0: assemble / factor / solve work on the condensed system
 *                       [[d2L + delta I + Ji Sigma Ji', Je], [Je', -delta_c I]] of dimension n + me (+ active rows): s and
 *                       lambda_i eliminated analytically.  Same inputs, same outputs (full [dx|ds|dle|dli], inertia of the
 *                       full matrix); kkt_storage then exposes the condensed matrix.  Several ranks: with the FULL blocks
 *                       staged on every rank (stage_blocks, not stage_blocks_owned).  Batched handles: per problem.
 *   "condensed_sigma_max" (1e4): inequalities whose Sigma_k = lda_i/(s+eps) exceeds it are not folded into the x-x block but
 *                       kept as rows with -1/Sigma_k on the diagonal; "condensed_refine" (0): refinement steps against the
 *                       full blocks every condensed solve gets at least.
 *   "block_refine" 0..3 (2), "refine_cond" (1e3): refinement steps of the block solves L T = S / T z = y for diagonal tiles
 *                       whose pivot spread exceeds refine_cond (the tile inverses are explicit).
 *   "refine_target" (1e-14) / "refine_max" (8): adaptive refinement of solve(refine < 0): stop at this backward error or
 *                       after this many steps.
 *   "pivtol_rel" (1e-14): a pivot below this fraction of its tile column's magnitude has cancelled: static pivot.
 *   "tile_blocked" 0|1  (1): the 64 x 64
 *                       tile inversion 16 pivots at a time (in-register LDL' of the micro-block + fp64 MFMA block sweeps,
 *                       Bunch-Kaufman verified afterwards, fallback to the single sweeps) -- same pivots and inertia, another
 *                       order of rounding than the single sweeps (not bit-identical).
 *   "profile" 0|1       HIP-event timings (last_timings, trailing_instances, dist_timings, provider_stats).
 *   "skip_zeros" 0|1    (1) products with blocks that pyipm.py:824-842 makes identically zero are not formed; bitwise-neutral.
 *   "keep_zeros" 0|1    (1) K1 leaves in place the zeros nothing can fill in; single rank; bitwise-neutral.
 *   "fuse_forward" 0|1  (1) the forward substitution of a pending residual trails the factorisation; bitwise-neutral
 *                       against the per-panel sweeps.
 *   "sweep_persist" 0|1 (1; single rank, one right-hand side, nb <= 256): each substitution sweep as ONE device-driven
 *                       launch (k_fwd_sweep / k_bwd_sweep; flags and values cross workgroups through agent-scope atomics,
 *                       every poll has a 2 s timeout that poisons the result with NaN and is reported by the next
 *                       factorisation) -- equal to rounding, not to the bit (another summation order), deterministic.
 *                       Their workgroups wait for each other, so all of them must become resident: under a CU mask, or
 *                       beside a kernel that holds CUs for seconds, use 0.
 *   "lookahead" 0|1|2   (2) 1 = one-group lookahead of the single-rank schedule; 2 = that, and where the x block is one group
 *                       its update is applied panel by panel under its own chain; "group" 1..create-time
 *                       value: panels per bulk trailing update (K = group * nb).  Bitwise-neutral.
 *   "bulk_bn" 256|128   (256) column width of a bulk update tile; "reserve_cus" (16) / "persist_rows" (12288): in the
 *                       chain-bound phase (at most persist_rows rows left) the bulk update runs as a persistent launch
 *                       that leaves reserve_cus CUs to the panel chain; 0 = ordinary launches.  Bitwise-neutral.
 *   "tile_chain" 0|1|2  (1; round 6) the 64 x 64 tile steps of a diagonal block as ONE launch of persistent workgroups
 *                       (k_tile_chain: the chain workgroup + units that own row tiles; hand-overs through written-through
 *                       stores and progress words, every poll with a 2 s timeout that factor() reports as PYIPM_E_HIP)
 *                       instead of one launch per tile: 1 where the chain is what the step waits for (first group, last
 *                       12288 rows, the per-panel / multi-GPU schedule), 2 everywhere, 0 never.  Same condition on residency
 *                       as sweep_persist.  Bit-identical to the launches.
 *   "wide_sub" (256)    per-panel / multi-GPU schedule: a panel wider than this is factored by its owner as a block of
 *                       sub-panels this wide (the launches, and the bits, of the single-rank schedule at nb = 256).
 *   "dist_sag" 0|1      COLLECTIVE: 0 keeps the plain broadcast for the panel messages, 1 goes back to what the exchange
 *                       self-test decided (see pyipm_newton_comm_bcast_mode); "dist_sag_min_bytes" (4 MiB): smaller messages
 *                       always take the plain broadcast.
 *   "dist_slices" 0|1|2 (2) COLLECTIVE: the two-message protocol of the distributed factorisation -- the rows of panel p
 *                       that meet the diagonal block of panel p + 1 (and the rows of panel p + 2) go from owner(p) to
 *                       owner(p + 1) point to point AHEAD of the panel message, so the next owner's tile chain starts on an
 *                       nb x nb message; 2: the rows of the second slice take their stages inside the chain's own launch
 *                       (no launch of their own behind it); 0: one message per panel (rounds 1-4).  Bit-identical.
 *   "dist_comm2" 0|1    (0) COLLECTIVE, RCCL transport: the slice messages on a SECOND communicator over the same ranks, on
 *                       the owner's stream, so that they do not queue behind a panel broadcast in flight.  Setting it (before
 *                       or after comm_init) creates the communicator -- every rank must make the call; the ranks agree before
 *                       anyone enters ncclCommInitRank.  Off by default: two communicators in flight on one device have not
 *                       run on more than one GPU yet; bench.py's ladder tries it last.  Bit-identical.
 *   "dist_timeout_s"    (300) bound on the host's wait for the device at the end of a distributed factorisation; on expiry
 *                       PYIPM_E_COMM names the panel whose message / update / chain did not complete and the handle refuses
 *                       further distributed steps (destroy aborts its communicators).  <= 0: wait for ever.
 *   "dist_selfmsg" 0|1  world == 1 only: the distributed driver packs and "sends" every panel anyway (measures the
 *                       message path on one GPU).
 *
 * EXPERT OPTIONS: tail_group, group_chain, tile_step, tile_waves, bc_per_problem, chain_cpy, chain_whole, chain_lds_kb,
 *   sweep_max_blocks, debug_fault, debug_timeline_ptr, debug_chain_ptr
 *   (which stream runs what, in how many launches: tail_group = panels per group once at most 24576 columns remain (4; 8 for systems of at most 8192 rows);
 *   group_chain 0 = a group's panels one after the other instead of one tile chain; tile_step 0 = the two-launches-per-tile
 *   schedule of round 1; tile_waves 4|8|9 = the launch-per-tile kernel on four waves, on eight where a whole CU is to be had,
 *   on eight everywhere; bc_per_problem 0 = the batched condensed form's Gram part by one workgroup per tile; chain_cpy / chain_whole /
 *   chain_lds_kb = units per row tile, one launch per group or per sub-panel, and the shared-memory pad that gives a workgroup
 *   of k_tile_chain its compute unit to itself; sweep_max_blocks, debug_* = test hooks and diagnostics buffers.  All of them
 *   choose between implementations that accumulate the same products in the same order: bit-identical results,
 *   tests/test_gpu_symmetric.py, tests/test_gpu_tile_blocked.py.) */
int pyipm_newton_set_option(pyipm_newton_ctx* ctx, const char* name, double value);

/* Version of this interface: bumped whenever an entry point changes its argument list (round 5 added `dz` to
 * pyipm_newton_step_lengths in place: a caller built against the older header would have passed a host pointer where the
 * device direction goes).  A binding checks pyipm_newton_abi_version() == PYIPM_NEWTON_ABI_VERSION of the header it was
 * written against before any other call (pyipm_amd/newton.py does). */
#define PYIPM_NEWTON_ABI_VERSION 6
int pyipm_newton_abi_version(void);

/* fp64 MFMA peak micro-benchmark: register-resident v_mfma_f64_16x16x4_f64 only.
 * Returns achieved TFLOP/s in *tflops. */
int pyipm_mfma_f64_peak(int device, int iters, double* tflops);

#ifdef __cplusplus
}
#endif
#endif /* PYIPM_NEWTON_H */
