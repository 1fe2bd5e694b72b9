// kernels_batched.hpp — many small independent Newton steps at once (BASELINE.json configs[4]: 512 QPs
// with n = 256, N = 768).  Independent systems shard with no exchange, and a system this small fits one
// workgroup: ONE launch factors the whole batch (grid = problems, 2 resident per CU, 512 in flight on an
// MI355X), one assembles, one forms the residuals, one substitutes.  Same algorithm as the big path
// (block LDL' with 64x64 Bunch-Kaufman block pivots, left-looking over the tile columns, refined block
// solves), same tile-inversion code (tile_invert_dev), MFMA for the tile products.
//
// Storage per problem: A (Npad x Npad, column-major, lower triangle = KKT / factor as in the big path);
// the strictly upper BLOCKS hold -S(t,u)' (the W operand of the updates), so no separate W buffer exists.
#pragma once
#include "ctx.hpp"
#include "kernels_assemble.hpp"
#include "kernels_factor.hpp"

namespace pyipm {

struct BatchPtrs {
    double* A; int64_t sA;                 // [B][Npad*Npad]
    double *Tinv, *Tsave; int64_t sT;      // [B][nt*64*64]
    double* Tflag; int64_t sF;             // [B][nt]
    DevStats* st;                          // [B]
    double *rhs, *sol; int64_t sV;         // [B][Npad]
    const double *d2L, *Je, *Ji; int64_t sH, sJe, sJi, ldh, ldje, ldji;      // caller blocks, batch strides in doubles
    const double *df, *ce, *ci, *s, *lda;  // staged vectors [B][n], [B][me], [B][mi], [B][mi], [B][me+mi]
    unsigned long long* anorm;             // [2 B]: per problem, bits of max |assembled entry| (scale of a static pivot) and the "assembly pending" word (always 0 here)
};

// The condensed form of a batched step (round 5; the big path's option, DESIGN section 7b, per problem): inequalities with
// Sigma_k = lda_k / (s_k + eps) <= sigma_max are eliminated analytically (their Ji Sigma Ji' joins the x-x block), the rest
// -- the constraints going active late in a run -- stay as rows with -1 / Sigma_k on the diagonal.  Per problem the system
// has n + me + |A| rows in the order [x | lambda_e | lambda_A]; |A| differs from problem to problem, so every kernel reads it
// from `cnt` and sizes its own loops (one workgroup per problem: nothing else has to agree).
struct BatchCond {
    int* pos;            // [B][mi]: index inside A, or -1
    int* idx;            // [B][mi]: members of A
    int* cnt;            // [B]: |A|
    double* sig;         // [B][mi]: Sigma_k of the eliminated pairs, 0 for the members of A (the scaling of the Gram operand)
    int64_t sP;          // stride of pos / idx / sig (>= mi)
    double sigma_max;
};

// K1 for the batch: grid (Npad/512, Npad/16, B)
__global__ __launch_bounds__(256) void k_b_assemble(BatchPtrs bp, Geo g, double eps, double delta, double delta_c)
{
    const int64_t b = blockIdx.z;
    double* A = bp.A + b * bp.sA;
    const double* d2L = bp.d2L + b * bp.sH;
    const double* Je = bp.Je ? bp.Je + b * bp.sJe : nullptr;
    const double* Ji = bp.Ji ? bp.Ji + b * bp.sJi : nullptr;
    const double* s = bp.s + b * g.mi;
    const double* lda = bp.lda + b * (g.me + g.mi);
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= g.Npad) return;                           // (wave-uniform: Npad is a multiple of 128)
    double amax = 0.0;
    #pragma unroll 4
    for (int c = 0; c < 16; ++c) {
        const int64_t j = (int64_t)blockIdx.y * 16 + c;
        if (j >= g.Npad) break;
        if (i + 1 < j) continue;
        const double v1 = kkt_entry(i + 1, j, j, g, d2L, bp.ldh, Je, bp.ldje, Ji, bp.ldji, s, lda, eps, delta, delta_c);
        const bool sig = j >= g.n && j < g.n + g.mi;   // (Sigma stays out of the static-pivot scale: see k_assemble)
        amax = fmax(amax, (sig && i + 1 == j) ? 0.0 : fabs(v1));
        if (i >= j) {
            dbl2_t v;
            v.x = kkt_entry(i, j, j, g, d2L, bp.ldh, Je, bp.ldje, Ji, bp.ldji, s, lda, eps, delta, delta_c);
            v.y = v1;
            amax = fmax(amax, (sig && i == j) ? 0.0 : fabs(v.x));
            *reinterpret_cast<dbl2_t*>(&A[i + j * g.Npad]) = v;
        } else {
            A[(i + 1) + j * g.Npad] = v1;
        }
    }
    anorm_publish(bp.anorm + 2 * b, amax);
}

// K2 for the batch: g = -grad (pyipm.py:655-668, 1717).  grid B, 256 threads (one wave per row of the x part).
__global__ __launch_bounds__(256) void k_b_residual(BatchPtrs bp, Geo g, double mu, double eps)
{
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = g.n, me = g.me, mi = g.mi;
    double* out = bp.rhs + b * bp.sV;
    const double* lda = bp.lda + b * (me + mi);
    const double* s = bp.s + b * mi;
    for (int64_t j = wave; j < n; j += 4) {
        double acc = 0.0;
        if (me) { const double* r = bp.Je + b * bp.sJe + j * bp.ldje; for (int64_t a = lane; a < me; a += 64) acc += r[a] * lda[a]; }
        if (mi) { const double* r = bp.Ji + b * bp.sJi + j * bp.ldji; for (int64_t a = lane; a < mi; a += 64) acc += r[a] * lda[me + a]; }
        acc = wave_sum(acc);
        if (lane == 0) out[j] = -(bp.df[b * n + j] - acc);
    }
    for (int64_t i = n + threadIdx.x; i < g.Npad; i += 256) {
        double v = 0.0;
        if (i < n + mi)            { const int64_t k = i - n;           v = -(lda[me + k] - mu / (s[k] + eps)); }
        else if (i < n + mi + me)  { const int64_t a = i - n - mi;      v = -bp.ce[b * me + a]; }
        else if (i < g.N)          { const int64_t k = i - n - mi - me; v = -(bp.ci[b * mi + k] - s[k]); }
        out[i] = v;
    }
}

// Condensed form, step 1 (grid B, 256 threads): the active set of problem b, its full right-hand side g = -grad (bp.rhs: the
// expansion and the backward-error check read it) and the condensed one (bp.sol, solved in place):
//   vc = [ g_x + Ji_I (Sigma_I g_i + g_s)_I ; g_e ; (g_i + g_s / Sigma)_A ; 0 ... ]
__global__ __launch_bounds__(256) void k_bc_prep(BatchPtrs bp, Geo g, double mu, double eps, BatchCond bc)
{
    __shared__ int part[256];
    __shared__ double tsh[512];                         // (mi <= 511: n + 2 mi + me <= 1024)
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = g.n, me = g.me, mi = g.mi;
    const double* lda = bp.lda + b * (me + mi);
    const double* s = bp.s + b * mi;
    int* pos = bc.pos + b * bc.sP;
    int* idx = bc.idx + b * bc.sP;
    double* out = bp.rhs + b * bp.sV;
    double* vc = bp.sol + b * bp.sV;
    // active set: an exclusive scan over 256 chunks
    const int64_t per = (mi + 255) / 256, k0 = tid * per;
    int64_t k1 = k0 + per; if (k1 > mi) k1 = mi;
    int c = 0;
    for (int64_t k = k0; k < k1; ++k) c += (lda[me + k] / (s[k] + eps) > bc.sigma_max) ? 1 : 0;
    part[tid] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int at = part[tid] - c;
    for (int64_t k = k0; k < k1; ++k) {
        if (lda[me + k] / (s[k] + eps) > bc.sigma_max) { pos[k] = at; idx[at] = (int)k; ++at; }
        else pos[k] = -1;
    }
    const int na = part[255];
    if (tid == 0) bc.cnt[b] = na;
    __syncthreads();
    for (int64_t k = tid; k < mi; k += 256) {
        const double sg = lda[me + k] / (s[k] + eps);
        const double gs = -(lda[me + k] - mu / (s[k] + eps)), gi = -(bp.ci[b * mi + k] - s[k]);
        out[n + k] = gs; out[n + mi + me + k] = gi;
        if (pos[k] < 0) { tsh[k] = sg * gi + gs; bc.sig[b * bc.sP + k] = sg; }
        else { tsh[k] = 0.0; bc.sig[b * bc.sP + k] = 0.0; vc[n + me + pos[k]] = gi + gs * (s[k] + eps) / lda[me + k]; }
    }
    for (int64_t a = tid; a < me; a += 256) { const double v = -bp.ce[b * me + a]; out[n + mi + a] = v; vc[n + a] = v; }
    for (int64_t i = g.N + tid; i < g.Npad; i += 256) out[i] = 0.0;
    for (int64_t i = n + me + na + tid; i < g.Npad; i += 256) vc[i] = 0.0;
    __syncthreads();
    // x rows: a wave takes FOUR rows per trip (their loads in flight together: one row at a time was one memory latency per row,
    // 64 of them in a row per wave -- most of this kernel's time)
    for (int64_t j0 = (int64_t)wave * 4; j0 < n; j0 += 16) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0}, acct[4] = {0.0, 0.0, 0.0, 0.0};
        if (me) {
            for (int64_t a = lane; a < me; a += 64) {
                const double la = lda[a];
                #pragma unroll
                for (int q = 0; q < 4; ++q) if (j0 + q < n) acc[q] += bp.Je[b * bp.sJe + (j0 + q) * bp.ldje + a] * la;
            }
        }
        if (mi) {
            for (int64_t a = lane; a < mi; a += 64) {
                const double la = lda[me + a], ta = tsh[a];
                #pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (j0 + q < n) { const double v = bp.Ji[b * bp.sJi + (j0 + q) * bp.ldji + a]; acc[q] += v * la; acct[q] += v * ta; }
            }
        }
        #pragma unroll
        for (int q = 0; q < 4; ++q) { acc[q] = wave_sum(acc[q]); acct[q] = wave_sum(acct[q]); }
        if (lane < 4 && j0 + lane < n) {
            const int64_t j = j0 + lane;
            const double a_ = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
            const double t_ = lane == 0 ? acct[0] : lane == 1 ? acct[1] : lane == 2 ? acct[2] : acct[3];
            const double gx = -(bp.df[b * n + j] - a_); out[j] = gx; vc[j] = gx + t_;
        }
    }
}

// Condensed form, step 2: the matrix  [[H + delta I + Ji_I Sigma_I Ji_I', Je, Ji_A], [Je', -delta_c I, 0], [Ji_A', 0, -1/Sigma_A]]
// (lower triangle, identity pad up to the next multiple of 64) of problem blockIdx.y, 64 x 64 tiles.  Workgroups 0 .. T0 - 1
// take one tile each of the first nt0 = ceil((n + me) / 64) tile rows (every problem has those: the x-x tiles with their Gram
// part and the lambda_e rows); workgroup T0 walks over the tile rows that exist only where a problem keeps active inequality
// rows (|A| > 0: late in a run) -- copies of Jacobian columns and a diagonal, no products -- and returns at once otherwise.
// (A grid over the tile pairs of the LARGEST possible system started 13000 workgroups per launch only to return: 0.1 ms.)
// The Gram part of an x-x tile is formed on the matrix pipe: both operand tiles of a 64-column chunk of Ji staged through
// shared memory (coalesced along the rows of the row-major Ji), the Sigma scaling applied to one of them on the way, the next
// chunk's loads in flight under the products of the current one.
// (sw: which 16-row strip of the 64 x 64 tile this wave stores; acc: its Gram part, columns t * 16 + l4 + 4 r)
__device__ __forceinline__ void bc_store_strip(const BatchPtrs& bp, const Geo& g, const BatchCond& bc, int64_t b, int rt, int ct, int sw,
                                               const double4_t (&acc)[4], double eps, double delta, double delta_c, int na);
__device__ __forceinline__ void bc_store_tile(const BatchPtrs& bp, const Geo& g, const BatchCond& bc, int64_t b, int rt, int ct,
                                              const double4_t (&acc)[4], double eps, double delta, double delta_c, int na)
{
    bc_store_strip(bp, g, bc, b, rt, ct, (int)(threadIdx.x >> 6), acc, eps, delta, delta_c, na);
}
__device__ __forceinline__ void bc_store_strip(const BatchPtrs& bp, const Geo& g, const BatchCond& bc, int64_t b, int rt, int ct, int wave,
                                               const double4_t (&acc)[4], double eps, double delta, double delta_c, int na)
{
    const int64_t n = g.n, me = g.me, mi = g.mi, ld = g.Npad, nc = n + me + na;
    const int tid = threadIdx.x, lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    double* A = bp.A + b * bp.sA;
    const double* d2L = bp.d2L + b * bp.sH;
    const double* Je = bp.Je ? bp.Je + b * bp.sJe : nullptr;
    const double* Ji = bp.Ji ? bp.Ji + b * bp.sJi : nullptr;
    const double* s = bp.s + b * mi;
    const double* lda = bp.lda + b * (me + mi);
    const int* idx = bc.idx + b * bc.sP;
    const int64_t i = (int64_t)rt * TB + wave * 16 + l15;
    double amax = 0.0;
    #pragma unroll
    for (int t = 0; t < 4; ++t)
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t c = (int64_t)ct * TB + t * 16 + l4 + 4 * r;
            if (i < c) continue;
            double v = 0.0;
            bool scale = true;
            if (c < n) {
                if (i < n)            v = d2L[c * bp.ldh + i] + (i == c ? delta : 0.0) + acc[t][r];
                else if (i < n + me)  v = Je[c * bp.ldje + (i - n)];
                else if (i < nc)      v = Ji[c * bp.ldji + idx[i - n - me]];
            } else if (i == c) {
                if (c < n + me)       v = -delta_c;
                else if (c < nc)      { const int k = idx[c - n - me]; v = -(s[k] + eps) / lda[me + k]; scale = false; }
                else                  v = 1.0;
            }
            A[i + c * ld] = v;
            if (scale) amax = fmax(amax, fabs(v));      // (1 / Sigma stays out of the static-pivot scale, as Sigma does in the full form)
        }
    anorm_publish(bp.anorm + 2 * b, amax);
}

__global__ __launch_bounds__(256) void k_bc_assemble(BatchPtrs bp, Geo g, double eps, double delta, double delta_c, BatchCond bc, int nt0)
{
    __shared__ double XA[TB][TB + 2];
    __shared__ double XB[TB][TB + 2];
    const int64_t b = blockIdx.y;
    const int64_t n = g.n, me = g.me, mi = g.mi;
    const int na = bc.cnt[b];
    const int64_t nc = n + me + na;
    const int T0 = nt0 * (nt0 + 1) / 2;
    double4_t acc[4];
    #pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    if ((int)blockIdx.x >= T0) {
        const int ntc = (int)((nc + TB - 1) / TB);
        for (int rt = nt0; rt < ntc; ++rt)
            for (int ct = 0; ct <= rt; ++ct) bc_store_tile(bp, g, bc, b, rt, ct, acc, eps, delta, delta_c, na);
        return;
    }
    int rt = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
    while (rt * (rt + 1) / 2 > (int)blockIdx.x) --rt;
    while ((rt + 1) * (rt + 2) / 2 <= (int)blockIdx.x) ++rt;
    const int ct = (int)blockIdx.x - rt * (rt + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t i0 = (int64_t)rt * TB, j0 = (int64_t)ct * TB;
    if (j0 < n && i0 < n && mi > 0) {
        const double* Ji = bp.Ji + b * bp.sJi;
        const double* sig = bc.sig + b * bc.sP;
        const int kk = tid & 63, r0q = tid >> 6;             // this thread stages column kk of rows r0q + 4 q of both operand tiles
        const double* pa = Ji + (j0 + r0q) * bp.ldji + kk;   // rows j0 + r0q + 4 q of the column-block operand
        const double* pb = Ji + (i0 + r0q) * bp.ldji + kk;   // rows i0 + r0q + 4 q of the row-block operand (scaled)
        const int64_t step = 4 * bp.ldji;
        const bool full = j0 + TB <= n && i0 + TB <= n && (mi % TB) == 0;     // no masks anywhere (block-uniform)
        double va[16], vb[16];
        // chunk 0
        if (full) {
            const double sg = sig[kk];
            #pragma unroll
            for (int q = 0; q < 16; ++q) { va[q] = pa[q * step]; vb[q] = sg * pb[q * step]; }
        } else {
            const bool kin = kk < mi;
            const double sg = kin ? sig[kk] : 0.0;
            #pragma unroll
            for (int q = 0; q < 16; ++q) {
                va[q] = (kin && j0 + r0q + 4 * q < n) ? pa[q * step] : 0.0;
                vb[q] = (kin && i0 + r0q + 4 * q < n) ? sg * pb[q * step] : 0.0;
            }
        }
        for (int64_t kc = 0; kc < mi; kc += TB) {
            if (kc > 0) __syncthreads();
            #pragma unroll
            for (int q = 0; q < 16; ++q) { XA[r0q + 4 * q][kk] = va[q]; XB[r0q + 4 * q][kk] = vb[q]; }
            __syncthreads();
            const int64_t kn = kc + TB;
            if (kn < mi) {                                    // the next chunk's loads fly under this chunk's products
                if (full) {
                    const double sg = sig[kn + kk];
                    #pragma unroll
                    for (int q = 0; q < 16; ++q) { va[q] = pa[kn + q * step]; vb[q] = sg * pb[kn + q * step]; }
                } else {
                    const bool kin = kn + kk < mi;
                    const double sg = kin ? sig[kn + kk] : 0.0;
                    #pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        va[q] = (kin && j0 + r0q + 4 * q < n) ? pa[kn + q * step] : 0.0;
                        vb[q] = (kin && i0 + r0q + 4 * q < n) ? sg * pb[kn + q * step] : 0.0;
                    }
                }
            }
            #pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const double bop = XB[wave * 16 + l15][ks * 4 + l4];
                #pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(XA[t * 16 + l15][ks * 4 + l4], bop, acc[t], 0, 0, 0);
            }
        }
    }
    if ((int64_t)rt * TB < nc) bc_store_tile(bp, g, bc, b, rt, ct, acc, eps, delta, delta_c, na);
}

// k_bc_assemble with ONE workgroup per problem (round 5) for n = 64 NX <= 256 and mi a multiple of 32 (BASELINE's batched
// configuration: n = mi = 256).  The tile-per-workgroup form above fetches a 64-row chunk of Ji twice per tile -- ten tiles per
// problem at n = 256: every entry of Ji crosses L2 -> CU ten times -- with two workgroups of four waves per CU: the matrix pipes
// ran at a third of their rate.  Here a chunk of 32 inequality rows of ALL n columns is staged once, double buffered (the next
// chunk's loads fly under this chunk's products), and every x-x tile is updated from it.  2 NX waves; wave w owns the 16-row
// strips w and 4 NX - 1 - w of the lower block triangle -- tile rows RLO = w / 4 and NX - 1 - RLO: 4 (NX + 1) accumulator tiles
// of 16 x 16 for every wave.  The Sigma scaling is applied to the row operand on its way out of shared memory (the same single
// product sigma_k Ji[k][i] as above), k ascends in MFMA groups of 4 as above: the same bits as k_bc_assemble.
template <int NX, int RLO>
__device__ __forceinline__ void bc_gram_problem(double (&V)[2][64 * NX][34], double (&SG)[2][32], const BatchPtrs& bp, const Geo& g,
                                                const BatchCond& bc, int64_t b, int na, double eps, double delta, double delta_c)
{
    constexpr int RHI = NX - 1 - RLO, NLO = 4 * (RLO + 1), NHI = 4 * (RHI + 1);
    constexpr int NT = 128 * NX, NVAR = 64 * NX, KC = 32, VPP = NT / 32, NPASS = NVAR / VPP;
    static_assert(RHI >= RLO, "the high strip's tile row is never above the low strip's");
    const int64_t mi = g.mi;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const double* Ji = bp.Ji + b * bp.sJi;
    const double* sig = bc.sig + b * bc.sP;
    const int s_lo = wave, s_hi = 4 * NX - 1 - wave;
    double4_t alo[NLO], ahi[NHI];
    #pragma unroll
    for (int t = 0; t < NLO; ++t) alo[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    #pragma unroll
    for (int t = 0; t < NHI; ++t) ahi[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int kk = tid & 31, v0 = tid >> 5;        // staging: 32 lanes along a column's rows, NT / 32 columns per pass
    constexpr int HP = NPASS / 2 > 0 ? NPASS / 2 : 1;      // the next chunk arrives in two halves (half the staging registers)
    double stg[HP], sgv = 0.0;
    #pragma unroll
    for (int h = 0; h < NPASS; h += HP) {
        #pragma unroll
        for (int q = 0; q < HP; ++q) stg[q] = Ji[(int64_t)(v0 + VPP * (h + q)) * bp.ldji + kk];
        #pragma unroll
        for (int q = 0; q < HP; ++q) V[0][v0 + VPP * (h + q)][kk] = stg[q];
    }
    if (tid < KC) SG[0][tid] = sig[tid];
    __syncthreads();
    int buf = 0;
    auto products = [&](int ks0, int ks1) {
        #pragma unroll 1
        for (int ks = ks0; ks < ks1; ++ks) {
            const double sg = SG[buf][ks * 4 + l4];
            const double blo = sg * V[buf][16 * s_lo + l15][ks * 4 + l4];
            const double bhi = sg * V[buf][16 * s_hi + l15][ks * 4 + l4];
            #pragma unroll
            for (int jj = 0; jj < NHI; ++jj) {
                const double a = V[buf][16 * jj + l15][ks * 4 + l4];
                ahi[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bhi, ahi[jj], 0, 0, 0);
                if (jj < NLO) alo[jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, blo, alo[jj], 0, 0, 0);
            }
        }
    };
    for (int64_t kc = 0; kc < mi; kc += KC) {
        const int64_t kn = kc + KC;
        const bool more = kn < mi;                         // (uniform)
        if (more) {
            #pragma unroll
            for (int q = 0; q < HP; ++q) stg[q] = Ji[(int64_t)(v0 + VPP * q) * bp.ldji + kn + kk];
            if (tid < KC) sgv = sig[kn + tid];
        }
        products(0, KC / 8);
        if (more) {                                        // the other buffer is nobody's operand during this chunk
            #pragma unroll
            for (int q = 0; q < HP; ++q) V[buf ^ 1][v0 + VPP * q][kk] = stg[q];
            if (tid < KC) SG[buf ^ 1][tid] = sgv;
            if (NPASS > HP) {
                #pragma unroll
                for (int q = 0; q < HP; ++q) stg[q] = Ji[(int64_t)(v0 + VPP * (HP + q)) * bp.ldji + kn + kk];
            }
        }
        products(KC / 8, KC / 4);
        if (more && NPASS > HP) {
            #pragma unroll
            for (int q = 0; q < HP; ++q) V[buf ^ 1][v0 + VPP * (HP + q)][kk] = stg[q];
        }
        __syncthreads();
        buf ^= 1;
    }
    #pragma unroll
    for (int J = 0; J <= RLO; ++J) {
        const double4_t a4[4] = {alo[4 * J], alo[4 * J + 1], alo[4 * J + 2], alo[4 * J + 3]};
        bc_store_strip(bp, g, bc, b, RLO, J, s_lo & 3, a4, eps, delta, delta_c, na);
    }
    #pragma unroll
    for (int J = 0; J <= RHI; ++J) {
        const double4_t a4[4] = {ahi[4 * J], ahi[4 * J + 1], ahi[4 * J + 2], ahi[4 * J + 3]};
        bc_store_strip(bp, g, bc, b, RHI, J, s_hi & 3, a4, eps, delta, delta_c, na);
    }
}

template <int NX>
__global__ __launch_bounds__(128 * NX) void k_bc_assemble_p(BatchPtrs bp, Geo g, double eps, double delta, double delta_c, BatchCond bc)
{
    __shared__ double V[2][64 * NX][34];
    __shared__ double SG[2][32];
    const int64_t b = blockIdx.x;
    const int na = bc.cnt[b];
    const int64_t nc = g.n + g.me + na;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (every wave runs the same number of barriers whatever its pair of strips)
    if (NX >= 3 && wave >= 4) bc_gram_problem<NX, (NX >= 3 ? 1 : 0)>(V, SG, bp, g, bc, b, na, eps, delta, delta_c);
    else                      bc_gram_problem<NX, 0>(V, SG, bp, g, bc, b, na, eps, delta, delta_c);
    // the tile rows below the x block (lambda_e rows, active inequality rows): copies and a diagonal, strip by strip
    const int ntc = (int)((nc + TB - 1) / TB);
    double4_t zero[4];
    #pragma unroll
    for (int t = 0; t < 4; ++t) zero[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    int idx = 0;
    for (int rt = NX; rt < ntc; ++rt)
        for (int ct = 0; ct <= rt; ++ct)
            for (int sw = 0; sw < 4; ++sw, ++idx)
                if (idx % (2 * NX) == wave) bc_store_strip(bp, g, bc, b, rt, ct, sw, zero, eps, delta, delta_c, na);
}

// Backward error of a batch of directions against the KKT blocks (never the factor): out[b] = |g - Hc raw| / |g| with raw the
// direction with the multiplier flip undone, Hc the full 4-block matrix with the shifts of the last step.  grid B, 256
// threads; g = -grad must be in bp.rhs (every batched step leaves it there).  The guard of the condensed form, and a check
// any caller can afford: O(N^2) per problem.
__global__ __launch_bounds__(256) void k_b_berr(BatchPtrs bp, Geo g, const double* __restrict__ dz, double eps, double delta,
                                                double delta_c, double* __restrict__ out)
{
    __shared__ double red[2][4];
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = g.n, me = g.me, mi = g.mi;
    const double* d = dz + b * g.N;
    const double* gr = bp.rhs + b * bp.sV;
    const double* d2L = bp.d2L + b * bp.sH;
    const double* s = bp.s + b * mi;
    const double* lda = bp.lda + b * (me + mi);
    const double* dx = d; const double* ds = d + n; const double* dle = d + n + mi; const double* dli = d + n + mi + me;   // (flipped)
    double rr = 0.0, gg = 0.0;
    for (int64_t j0 = (int64_t)wave * 4; j0 < n; j0 += 16) {    // x rows, four per trip: sym(triu(d2L)) dx + delta dx + Je dle + Ji dli
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int64_t i = lane; i < n; i += 64) {
            const double xi = dx[i];
            #pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t j = j0 + q;
                if (j < n) acc[q] += (i >= j ? d2L[j * bp.ldh + i] : d2L[i * bp.ldh + j]) * xi;
            }
        }
        for (int64_t a = lane; a < me; a += 64) {
            const double v = dle[a];
            #pragma unroll
            for (int q = 0; q < 4; ++q) if (j0 + q < n) acc[q] -= bp.Je[b * bp.sJe + (j0 + q) * bp.ldje + a] * v;
        }
        for (int64_t a = lane; a < mi; a += 64) {
            const double v = dli[a];
            #pragma unroll
            for (int q = 0; q < 4; ++q) if (j0 + q < n) acc[q] -= bp.Ji[b * bp.sJi + (j0 + q) * bp.ldji + a] * v;
        }
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double a_ = wave_sum(acc[q]);
            const int64_t j = j0 + q;
            if (lane == 0 && j < n) { const double r = gr[j] - (a_ + delta * dx[j]); rr += r * r; gg += gr[j] * gr[j]; }
        }
    }
    for (int64_t k = tid; k < mi; k += 256) {                // s rows and lambda_i rows
        double u = 0.0;
        const double* col = bp.Ji + b * bp.sJi + k;
        for (int64_t j = 0; j < n; ++j) u += col[j * bp.ldji] * dx[j];
        const double r1 = gr[n + k] - (lda[me + k] / (s[k] + eps) * ds[k] + dli[k]);
        const double r2 = gr[n + mi + me + k] - (u - ds[k]);
        rr += r1 * r1 + r2 * r2; gg += gr[n + k] * gr[n + k] + gr[n + mi + me + k] * gr[n + mi + me + k];
    }
    for (int64_t a = tid; a < me; a += 256) {                // lambda_e rows
        double u = 0.0;
        const double* col = bp.Je + b * bp.sJe + a;
        for (int64_t j = 0; j < n; ++j) u += col[j * bp.ldje] * dx[j];
        const double r = gr[n + mi + a] - (u + delta_c * dle[a]);
        rr += r * r; gg += gr[n + mi + a] * gr[n + mi + a];
    }
    rr = wave_sum(rr); gg = wave_sum(gg);
    if (lane == 0) { red[0][wave] = rr; red[1][wave] = gg; }
    __syncthreads();
    if (tid == 0) {
        const double R = red[0][0] + red[0][1] + red[0][2] + red[0][3], G = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        out[b] = G > 0.0 ? sqrt(R / G) : sqrt(R);
    }
}

// Can rows [lo, hi) hold a non-zero of the L / W operand whose source columns are [slo, shi)?  The block pattern
// of pyipm.py:824-842 (same argument as active_ranges in pyipm_newton.hip): x-block sources never reach the
// slack rows, slack-block sources only their own lambda_i rows.  Conservative at tile granularity.
__device__ __forceinline__ bool rows_active(const Geo& g, int64_t slo, int64_t shi, int64_t lo, int64_t hi) {
    if (g.mi == 0) return true;
    const int64_t s0 = g.n, s1 = g.n + g.mi, i0 = g.n + g.mi + g.me;
    if (shi <= s0) return !(lo >= s0 && hi <= s1);
    if (slo >= s0 && shi <= s1) return hi > i0 + (slo - s0) && lo < i0 + (shi - s0);
    return true;
}

// The tile inversion behind a real call: inlined into the loop over the tile columns, its ~100 lane-constant registers were
// hoisted over the whole kernel (256 + 256 registers and 102 of them spilled).  Round 4, 512 problems, whole step: 4.34 -> 3.97 ms
// with the call, 3.85 with two row tiles per pass instead of four.  Two workgroups per CU (launch bounds 256, 2: the inversion
// alone needs 242 registers) were slower in every variant tried (4.2 - 5.2 ms), and so was requesting the next step's operands
// before the products of the current one (3.98: the extra live registers spill inside the loops).  A problem is a chain of
// dependent round trips to memory (load L and W tiles, synchronise, 64 - 128 MFMAs, store): 1.3 ms alone on the GPU, 1.9 ms
// with 255 others.
__device__ __noinline__ void b_tile_invert(TileScratch& sm, const double* A, int64_t ld, int64_t j0, double* Tinv, double* Tsave,
                                           double* Tflag, double refine_cond, DevStats* st, int64_t Nreal, double pivtol_rel,
                                           const unsigned long long* anorm_bits, int64_t neg_from, bool blocked)
{
    tile_invert_dev(sm, A, ld, j0, j0, Tinv, Tsave, Tflag, refine_cond, st, Nreal, pivtol_rel, anorm_bits, neg_from, nullptr, false, blocked);
}

// Factor one problem per workgroup.  grid B, 256 threads.
constexpr int B_RT = 2;                   // row tiles per left-looking pass (their accumulators: 32 registers each per lane)
__global__ __launch_bounds__(256) void k_b_factor(BatchPtrs bp, Geo g, double refine_cond, int nref, double pivtol_rel, int blocked,
                                                  const int* __restrict__ cond_cnt)      // != NULL: the condensed system of each problem
{
    __shared__ TileScratch sm;
    __shared__ double X[TB][TB + 2];
    const int64_t bi = blockIdx.x, ld = g.Npad;
    const Geo gfull = g;
    int cond_na = 0;
    if (cond_cnt) {
        // condensed form: [x | lambda_e | lambda_A] of this problem, n + me + |A| rows padded to whole tiles inside the same
        // storage (leading dimension unchanged); no slack block, so no structural zeros to skip
        cond_na = cond_cnt[bi];
        g.me = gfull.me + cond_na; g.mi = 0; g.N = gfull.n + g.me; g.Npad = (g.N + TB - 1) / TB * TB;
    }
    double* A = bp.A + bi * bp.sA;
    double* Tinv = bp.Tinv + bi * bp.sT;
    double* Tsave = bp.Tsave + bi * bp.sT;
    double* Tflag = bp.Tflag + bi * bp.sF;
    DevStats* st = bp.st + bi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int nt = (int)(g.Npad / TB);
    if (tid == 0) {
        st->n_neg = st->n_zero = st->n_2x2 = st->n_pos = st->nonfinite = 0;
        st->d_min = 1.0e308; st->d_max = 0.0; st->growth_bits = 0ull;
    }
    double gmax = 0.0;
    for (int t = 0; t < nt; ++t) {
        const int64_t j0 = (int64_t)t * TB;
        // (1) left-looking: block column t, row tiles r >= t, gets the contributions of tile columns u < t.
        // B_RT row tiles per pass (a wave owns 16 rows of each): the staged W operand is reused B_RT times.
        if (t > 0) {
            for (int r0 = t; r0 < nt; r0 += B_RT) {
                const int nq = (nt - r0) < B_RT ? (nt - r0) : B_RT;
                const int64_t ib = (int64_t)r0 * TB + wave * 16 + l15;          // row of tile r0; tile r0+q: + 64 q
                double4_t acc[B_RT][4];
                #pragma unroll
                for (int q = 0; q < B_RT; ++q)
                    if (q < nq) {
                        #pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            #pragma unroll
                            for (int m = 0; m < 4; ++m) acc[q][cb][m] = A[(ib + q * TB) + (j0 + cb * 16 + l4 + 4 * m) * ld];
                    }
                for (int u = 0; u < t; ++u) {
                    // structural zeros (block-uniform): W(t,u) = 0, or every L(r,u) of this pass = 0
                    if (!rows_active(g, (int64_t)u * TB, (int64_t)(u + 1) * TB, j0, j0 + TB)) continue;
                    bool any = false;
                    for (int q = 0; q < nq; ++q)
                        any = any || rows_active(g, (int64_t)u * TB, (int64_t)(u + 1) * TB, (int64_t)(r0 + q) * TB, (int64_t)(r0 + q + 1) * TB);
                    if (!any) continue;
                    double b[16];
                    #pragma unroll
                    for (int ks = 0; ks < 16; ++ks) b[ks] = A[ib + ((int64_t)u * TB + ks * 4 + l4) * ld];   // L(r0,u)
                    __syncthreads();
                    {   // -S(t,u)[c][k] sits at A[64u + k][j0 + c]; the 16 loads of a thread go out together
                        double stg[TB * TB / 256];
                        #pragma unroll
                        for (int q = 0; q < TB * TB / 256; ++q) {
                            const int e = tid + 256 * q, k = e & 63, c = e >> 6;
                            stg[q] = A[((int64_t)u * TB + k) + (j0 + c) * ld];
                        }
                        #pragma unroll
                        for (int q = 0; q < TB * TB / 256; ++q) {
                            const int e = tid + 256 * q;
                            X[e >> 6][e & 63] = stg[q];
                        }
                    }
                    __syncthreads();
                    #pragma unroll
                    for (int q = 0; q < B_RT; ++q)
                        if (q < nq) {
                            double bn[16];
                            if (q + 1 < nq) {
                                #pragma unroll
                                for (int ks = 0; ks < 16; ++ks)
                                    bn[ks] = A[(ib + (q + 1) * TB) + ((int64_t)u * TB + ks * 4 + l4) * ld];
                            }
                            #pragma unroll
                            for (int ks = 0; ks < 16; ++ks)
                                #pragma unroll
                                for (int cb = 0; cb < 4; ++cb)
                                    acc[q][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[cb * 16 + l15][ks * 4 + l4], b[ks], acc[q][cb], 0, 0, 0);
                            if (q + 1 < nq) {
                                #pragma unroll
                                for (int ks = 0; ks < 16; ++ks) b[ks] = bn[ks];
                            }
                        }
                }
                #pragma unroll
                for (int q = 0; q < B_RT; ++q)
                    if (q < nq) {
                        #pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            #pragma unroll
                            for (int m = 0; m < 4; ++m) A[(ib + q * TB) + (j0 + cb * 16 + l4 + 4 * m) * ld] = acc[q][cb][m];
                    }
            }
        }
        __syncthreads();
        // (2) the block pivot
        b_tile_invert(sm, A, ld, j0, Tinv + (int64_t)t * TB * TB, Tsave + (int64_t)t * TB * TB, Tflag + t,
                      refine_cond, st, g.N, pivtol_rel, bp.anorm + 2 * bi, g.n + g.mi, blocked != 0);
        __syncthreads();
        // (3) rows below: keep -S' in the upper blocks, overwrite S with L = S X (refined when the tile is flagged)
        if (t + 1 < nt) {
            const double* Xi = Tinv + (int64_t)t * TB * TB;
            const double* Tt = Tsave + (int64_t)t * TB * TB;
            const int nr = (nref > 0 && Tflag[t] != 0.0) ? nref : 0;
            PYIPM_STAGE_TILE(X, 1.0, Xi)
            __syncthreads();
            for (int r = t + 1; r < nt; ++r) {
                if (!rows_active(g, j0, j0 + TB, (int64_t)r * TB, (int64_t)(r + 1) * TB)) continue;   // S(r,t) = 0 = L(r,t)
                const int64_t i = (int64_t)r * TB + wave * 16 + l15;
                double b[16];
                #pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    b[ks] = A[i + (j0 + ks * 4 + l4) * ld];
                    A[(j0 + ks * 4 + l4) + i * ld] = -b[ks];                   // -S(r,t)' into the upper block (t,r)
                }
                double4_t lac[4];
                #pragma unroll
                for (int cb = 0; cb < 4; ++cb) lac[cb] = (double4_t){0.0, 0.0, 0.0, 0.0};
                #pragma unroll
                for (int ks = 0; ks < 16; ++ks)
                    #pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        lac[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[cb * 16 + l15][ks * 4 + l4], b[ks], lac[cb], 0, 0, 0);
                for (int it = 0; it < nr; ++it) {                               // rare: ill-conditioned tile
                    __syncthreads();
                    PYIPM_STAGE_TILE(X, -1.0, Tt)
                    __syncthreads();
                    double4_t res[4];
                    #pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        #pragma unroll
                        for (int q = 0; q < 4; ++q) res[cb][q] = b[4 * cb + q];
                    #pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const double lop = lac[ks >> 2][ks & 3];
                        #pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            res[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[cb * 16 + l15][ks * 4 + l4], lop, res[cb], 0, 0, 0);
                    }
                    __syncthreads();
                    PYIPM_STAGE_TILE(X, 1.0, Xi)
                    __syncthreads();
                    #pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const double rop = res[ks >> 2][ks & 3];
                        #pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            lac[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[cb * 16 + l15][ks * 4 + l4], rop, lac[cb], 0, 0, 0);
                    }
                }
                #pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        A[i + (j0 + cb * 16 + l4 + 4 * q) * ld] = lac[cb][q];
                        gmax = fmax(gmax, fabs(lac[cb][q]));
                    }
            }
        }
        __syncthreads();
    }
    gmax = wave_max(gmax);
    if (lane == 0) atomicMax(&st->growth_bits, (unsigned long long)__double_as_longlong(gmax));
    // inertia of the FULL matrix: every eliminated (s_k, lambda_i_k) pair [[Sigma_k, -1], [-1, 0]] has one positive and one
    // negative eigenvalue; an active row kept in the system already counted its negative one, its eliminated Sigma_k is the
    // positive one (as the big path's factor_dispatch adds them)
    if (cond_cnt && tid == 0) { st->n_neg += gfull.mi - cond_na; st->n_pos += gfull.mi; }
}

// Substitutions + sign flip for one problem per workgroup.  grid B, block Npad (<= 1024), LDS 2*Npad doubles.
// cond != 0: the condensed system (k_bc_prep / k_bc_assemble): the right-hand side is the reduced one in bp.sol, the sweeps
// run over this problem's n + me + |A| rows, and the full direction is recovered from the condensed solution:
//   I: ds = Ji' dx - g_i , dli = Sigma ds - g_s ;   A: dli from the solve, ds = (dli + g_s) / Sigma.
__global__ void k_b_solve(BatchPtrs bp, Geo g, int nref, int flip, double* __restrict__ dz, BatchCond bc, int cond, double eps)
{
    extern __shared__ double lds[];
    double* y = lds;
    double* w = lds + g.Npad;
    const int64_t bi = blockIdx.x, ld = g.Npad;
    const double* A = bp.A + bi * bp.sA;
    const int tid = threadIdx.x, lane = tid & 63, t = tid >> 6;
    const int na = cond ? bc.cnt[bi] : 0;
    const int64_t nrows = cond ? (g.n + g.me + na + TB - 1) / TB * TB : g.Npad;      // rows that take part in the sweeps
    const int nt = (int)(nrows / TB);
    const bool act = tid < nrows;
    const double y0 = (cond ? bp.sol : bp.rhs)[bi * bp.sV + tid];
    y[tid] = y0;
    for (int u = 0; u + 1 < nt; ++u) {                       // forward (unit block lower triangular)
        __syncthreads();
        if (act && tid >= (u + 1) * TB) {
            const double* col = A + tid + ((int64_t)u * TB) * ld;
            double acc = 0.0;
            #pragma unroll 8
            for (int k = 0; k < TB; ++k) acc = fma(col[(int64_t)k * ld], y[u * TB + k], acc);
            y[tid] -= acc;
        }
    }
    __syncthreads();
    // block diagonal: z = inv(T) y, refined against T for flagged tiles (uniform trip count for the barriers)
    const double* Xi = bp.Tinv + bi * bp.sT + (int64_t)t * TB * TB;
    const double* Tt = bp.Tsave + bi * bp.sT + (int64_t)t * TB * TB;
    const bool flagged = act && bp.Tflag[bi * bp.sF + t] != 0.0;
    const double yt = y[tid];
    double z = 0.0;
    if (act) {
        #pragma unroll 8
        for (int j = 0; j < TB; ++j) z = fma(Xi[j * TB + lane], y[t * TB + j], z);
    }
    for (int it = 0; it < nref; ++it) {
        w[tid] = z;
        __syncthreads();
        double r = yt;
        if (flagged) {
            #pragma unroll 8
            for (int j = 0; j < TB; ++j) r = fma(-Tt[j * TB + lane], w[t * TB + j], r);
        }
        __syncthreads();
        w[tid] = r;
        __syncthreads();
        if (flagged) {
            #pragma unroll 8
            for (int j = 0; j < TB; ++j) z = fma(Xi[j * TB + lane], w[t * TB + j], z);
        }
        __syncthreads();
    }
    __syncthreads();
    y[tid] = z;
    for (int u = nt - 1; u >= 1; --u) {                      // backward
        __syncthreads();
        if (tid < u * TB) {
            const double* col = A + ((int64_t)u * TB) + (int64_t)tid * ld;
            double acc = 0.0;
            #pragma unroll 8
            for (int i = 0; i < TB; ++i) acc = fma(col[i], y[u * TB + i], acc);
            y[tid] -= acc;
        }
    }
    __syncthreads();
    double x = y[tid];
    if (cond) {
        // expand: w <- [dx ; ds ; dle ; dli] from the condensed solution y and the full right-hand side (bp.rhs)
        const int64_t n = g.n, me = g.me, mi = g.mi;
        const double* gr = bp.rhs + bi * bp.sV;
        const int* pos = bc.pos + bi * bc.sP;
        w[tid] = 0.0;
        __syncthreads();
        if (tid < n) w[tid] = y[tid];
        if (tid < me) w[n + mi + tid] = y[n + tid];
        if (tid < mi) {
            const int64_t k = tid;
            const double* col = bp.Ji + bi * bp.sJi + k;
            double u = 0.0;
            #pragma unroll 4
            for (int64_t j = 0; j < n; ++j) u = fma(col[j * bp.ldji], y[j], u);
            const double bs = gr[n + k], b_i = gr[n + mi + me + k];
            const double sg = bp.lda[bi * (me + mi) + me + k] / (bp.s[bi * mi + k] + eps);
            double ds, dl;
            if (pos[k] < 0) { ds = u - b_i; dl = sg * ds - bs; }
            else            { dl = y[n + me + pos[k]]; ds = (dl + bs) / sg; }
            w[n + k] = ds; w[n + mi + me + k] = dl;
        }
        __syncthreads();
        x = w[tid];
    }
    bp.sol[bi * bp.sV + tid] = x;
    if (tid < g.N) dz[bi * g.N + tid] = (flip && tid >= g.n + g.mi) ? -x : x;
}

}  // namespace pyipm
