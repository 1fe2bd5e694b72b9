/* pyipm_lbfgs.h — C-ABI of the L-BFGS search direction on MI355X (gfx950), same shared object as
 * pyipm_newton.h (libpyipm_newton.so).  SURVEY.md section 8(f) rank 4.
 *
 * Replaces the compiled direction function of the reference's limited-memory mode,
 *     dz = self.lbfgs_dir_func(x, s, lda, g, zeta, S, Y, SS, L, D, B)      /root/reference/pyipm.py:872-875, 1201, 1228
 * whose expression graph is built by lbfgs_builder (:1007-1182).  The host keeps what the reference
 * keeps on the host (lbfgs_init :993, lbfgs_update :1282 — O(n m) vector work); the direction is:
 *
 *   constrained (me + mi > 0), the general branch :1099-1148.  With J = [Je | Ji] (n x p, p = me + mi),
 *   Sigma = lda_i / (s + eps), A = diag(zeta I_n, Sigma), B = [[Je, Ji], [0, -I]] (:582-607),
 *   W = [zeta S, Y] (n x 2m):
 *     G   = B' inv(A) B = J'J / zeta + diag(0, 1/Sigma)        p x p Gram matrix over the n variables   [MFMA]
 *     G  += reg I on the equality block when that block is numerically singular (:1106-1113)
 *     y   = inv(G) (B' inv(A) g1 - g2) ,  X00 = -inv(G) (J'W / zeta)      one factorisation, 2m+1 substitutions
 *     Zg  = [inv(A)(g1 - B y) ; y] ,      X01 = inv(A)([W;0] + B X00)
 *     v11 = inv(W'X01 - [[zeta SS, L], [L', -D]]) (W'Zg_x)                 2m x 2m, one workgroup
 *     dz  = Zg - [X01 ; -X00] v11                                          RAW: multiplier rows not yet negated
 *   evaluated so that J is read twice only: P = J'[g_x | W] (one pass, 2m+1 columns); W'X01_x = (W'W + P_w'X00)/zeta
 *   and W'Zg_x = (W'g_x - P_w'y)/zeta need no J; with u = y + X00 v11:  dz_x = (g_x - W v11 - J u)/zeta (one pass,
 *   one vector), dz_s = (g_s + u_i)/Sigma, dz_lambda = u.
 *   The factorisation of G is the block LDL' of pyipm_newton.h (an internal handle of order p); the
 *   reference's "rcond(G_ee) <= eps" test (an eigh, :1108-1109) is replaced by "the factorisation rejected
 *   a pivot or met a negative one" (G is positive semidefinite by construction), and additionally
 *   d_min/d_max <= eps when there are no inequalities.  The square-Jacobian shortcut (:1064-1097) is not
 *   reproduced: it returns the same direction through inv(B) and cannot be compiled in the reference as
 *   written (its input list names s_dev twice, :877-880).
 *
 *   unconstrained (me = mi = 0), :1149-1175 (inverse-Hessian form; SS holds Y'Y, L the upper-triangular S'Y):
 *     dz = zeta g - [S, zeta Y] c ,  [[0, L], [L', D + zeta SS]] c = [S'g ; zeta Y'g]
 *
 * Conventions are those of pyipm_newton.h: opaque handle, int return codes (PYIPM_OK / PYIPM_E_*),
 * no exceptions, caller-owned buffers flagged PYIPM_MEM_HOST / PYIPM_MEM_DEVICE, one handle per host
 * thread, work enqueued on the handle's stream and synchronised before a host result is returned.
 */
#ifndef PYIPM_LBFGS_H
#define PYIPM_LBFGS_H

#include "pyipm_newton.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pyipm_lbfgs_ctx pyipm_lbfgs_ctx;

typedef struct pyipm_lbfgs_stats {
    int64_t m;              /* displacement pairs used                                              */
    int64_t n_neg, n_zero;  /* negative / rejected pivots of the last factorisation of G (0, 0 = PD) */
    int64_t regularised;    /* 1: reg was added to the equality block (:1110-1113)                   */
    int64_t n_factor;       /* factorisations of G in this call (2 when regularised)                */
    double d_min, d_max;    /* pivot magnitudes of G                                                */
    double small_pivot_min; /* smallest |pivot| of the 2m x 2m LU (0: singular, direction is NaN)   */
} pyipm_lbfgs_stats;

/* max_pairs: largest number of stored pairs a call may pass (the reference lets the storage reach lbfgs+1,
 * :1300); 1 <= max_pairs <= 32.  nb: panel width of the internal factorisation (0 = 256).
 * stream: hipStream_t as void* (NULL = default stream). */
int pyipm_lbfgs_create(pyipm_lbfgs_ctx** out, int64_t n, int64_t me, int64_t mi, int max_pairs, int nb,
                       int device, void* stream);
int pyipm_lbfgs_destroy(pyipm_lbfgs_ctx* h);
const char* pyipm_lbfgs_last_error(pyipm_lbfgs_ctx* h);
int pyipm_lbfgs_set_stream(pyipm_lbfgs_ctx* h, void* stream);

/* Device bytes a handle of this shape allocates (queryable without a GPU). */
size_t pyipm_lbfgs_workspace_bytes(int64_t n, int64_t me, int64_t mi, int max_pairs, int nb);

/* Constraint Jacobians as the reference's dce / dci return them (:223-225, 486-487): Je n x me, Ji n x mi,
 * row-major with leading dimensions ld_*.  Copied into the library's padded operand buffer (strided copy,
 * no transposition); stage again whenever they change, once for linear constraints: J'J is computed on the
 * first direction after a staging and reused until the next one (it depends on neither zeta nor Sigma).
 * No-op when me = mi = 0. */
int pyipm_lbfgs_stage_jacobian(pyipm_lbfgs_ctx* h, const double* Je, int64_t ld_Je, const double* Ji,
                               int64_t ld_Ji, int memkind);

/* One search direction.  g: n + 2 mi + me (the NEGATED KKT residual, :1637, 1705-1710); s: mi; lda: me + mi;
 * S, Y: n x m row-major (ld_S, ld_Y >= m) — all flagged by memkind.  SS, L, D: m x m row-major, HOST memory
 * always (they are O(m^2) bookkeeping of lbfgs_update).  reg = reg_coef * eta * mu**beta (:1113).
 * dz: n + 2 mi + me, flagged by memkind; flip != 0 negates its multiplier rows (:1723-1725).
 * m = 0 is the empty storage (first iteration, or after a reset :1363-1368). */
int pyipm_lbfgs_direction(pyipm_lbfgs_ctx* h, const double* g, const double* s, const double* lda, double zeta,
                          int m, const double* S, int64_t ld_S, const double* Y, int64_t ld_Y,
                          const double* SS, const double* L, const double* D, double reg, double eps,
                          double* dz, int flip, int memkind, pyipm_lbfgs_stats* stats);

/* ms of the last direction call (HIP events on the handle's stream):
 * out[0] total, [1] Gram launch, [2] factorisation(s) of G, [3] the 2m+1 substitutions,
 * [4] the two passes over J, [5] small system + combination, [6] flop count of a Gram launch,
 * [7] Gram launches since create ([1] is ~0 for a direction that reused J'J). */
int pyipm_lbfgs_last_timings(pyipm_lbfgs_ctx* h, double out[8]);

/* Row-sharded use (one process per GPU; the direction shards by rows of J with three real exchanges).
 * Every rank creates its handle with n = ITS number of rows, stages its rows of Je / Ji and passes
 * g = [its rows of g_x | g_s | g_lambda], its rows of S and Y; zeta, SS, L, D, s, lda, reg are replicated.  With a
 * callback installed the library calls it — sum over ranks, in place, on device memory, enqueued on `stream` — on
 *   (1) J'J                 p_pad^2 doubles, once per staging of the Jacobians,
 *   (2) P = J'[g_x | W]     p_pad (2m+1) doubles per direction,
 *   (3) W'[g_x | W]         2m (2m+1) doubles per direction            (unconstrained problems: W'g, 2m doubles),
 * everything else is row-local (V, J u, dz_x) or replicated (G and its factor, the 2m x 2m system, dz_s, dz_lambda).
 * dz returns [its rows of dz_x | dz_s | dz_lambda].  fn returns 0 on success (anything else -> PYIPM_E_COMM).
 * pyipm_amd/lbfgs.py binds it to torch.distributed.all_reduce (backend "nccl" = RCCL). */
typedef int (*pyipm_lbfgs_allreduce_fn)(void* user, double* device_buf, int64_t count, void* stream);
int pyipm_lbfgs_set_allreduce(pyipm_lbfgs_ctx* h, pyipm_lbfgs_allreduce_fn fn, void* user);

/* Pass an option of pyipm_newton_set_option through to the internal factorisation handle. */
int pyipm_lbfgs_set_option(pyipm_lbfgs_ctx* h, const char* name, double value);

#ifdef __cplusplus
}
#endif
#endif
