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

namespace pyipm {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// K3: invert one 64x64 symmetric tile in LDS by symmetric sweeps with Bunch-Kaufman 1x1 / 2x2
// pivot selection restricted to the tile.  No row/column swaps: after sweeping a set S the
// unswept block is the Schur complement, so pivot choice is ordinary BK on what is left; after
// all 64 indices the working matrix is -inv(T).  Pivot search = wavefront shuffles (wave 0);
// the rank-1/2 update is spread over 4 waves.  Inertia from pivot signs.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_argmax(double v, int idx, double& vmax, int& imax) {
    // larger value wins; ties -> lower index (matches numpy.argmax in the CPU model)
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_xor(v, off, 64);
        int oi = __shfl_xor(idx, off, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    vmax = v; imax = idx;
}
__device__ __forceinline__ double wave_max(double v) {
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

#define PYIPM_BK_ALPHA 0.6403882032022076   /* (1+sqrt(17))/8 */

__global__ __launch_bounds__(256) void k_tile_invert(
    const double* __restrict__ A, int64_t ld, int64_t grow0, int64_t lcol0,
    double* __restrict__ Tinv, DevStats* __restrict__ st, int64_t Nreal, double pivtol_rel)
{
    __shared__ double B[TB][TB + 1];
    __shared__ double cp[TB], cq[TB], lp[TB], lq[TB];
    __shared__ double sh_piv[3];           // new values of B[p][p], B[p][q], B[q][q]
    __shared__ int sh_kind, sh_p, sh_q;
    __shared__ double sh_red[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // load the lower triangle and mirror it (coalesced along i)
    double amax = 0.0;
    for (int e = tid; e < TB * TB; e += 256) {
        const int i = e & 63, j = e >> 6;
        const double v = (i >= j) ? A[(grow0 + i) + (lcol0 + j) * ld]
                                  : A[(grow0 + j) + (lcol0 + i) * ld];
        B[i][j] = v;
        amax = fmax(amax, fabs(v));
    }
    amax = wave_max(amax);
    if (lane == 0) sh_red[wave] = amax;
    __syncthreads();
    const double scale = fmax(fmax(sh_red[0], sh_red[1]), fmax(sh_red[2], sh_red[3]));
    const double pivtol = pivtol_rel * scale;

    unsigned long long mask = ~0ull;       // unswept set, kept identically by every lane of wave 0
    long long neg = 0, zero = 0, n2 = 0, pos = 0, bad = 0;
    double dmin = 1.0e308, dmax = 0.0;

    int left = TB;                         // tracked identically by every thread
    for (int it = 0; it < TB; ++it) {
        if (left == 0) break;
        if (wave == 0) {
            const bool u = (mask >> lane) & 1ull;
            double app; int p;
            wave_argmax(u ? fabs(B[lane][lane]) : -1.0, lane, app, p);
            double lam; int r;
            wave_argmax((u && lane != p) ? fabs(B[lane][p]) : -1.0, lane, lam, r);
            int kind = 1, piv = p, q = p;
            if (lam > 0.0 && app < PYIPM_BK_ALPHA * lam) {
                const double sigma = wave_max((u && lane != r) ? fabs(B[r][lane]) : -1.0);
                const double arr = fabs(B[r][r]);
                if (app * sigma >= PYIPM_BK_ALPHA * lam * lam) { piv = p; }
                else if (arr >= PYIPM_BK_ALPHA * sigma)        { piv = r; }
                else                                            { kind = 2; piv = p; q = r; }
            }
            if (kind == 1) {
                double d = B[piv][piv];
                const double ad = fabs(d);
                const bool real = (grow0 + piv) < Nreal;
                if (!(ad <= 1.0e308)) bad = 1;                     // NaN or Inf
                if (ad <= pivtol) {
                    if (real) zero++;
                    double t = pivtol > 0.0 ? pivtol : 2.2250738585072014e-308;
                    d = (d >= 0.0) ? t : -t;
                } else if (real) {
                    if (d < 0.0) neg++; else pos++;
                    dmin = fmin(dmin, ad); dmax = fmax(dmax, ad);
                }
                const double c = B[lane][piv];
                cp[lane] = c; cq[lane] = 0.0;
                lp[lane] = c / d; lq[lane] = 0.0;
                if (lane == 0) {
                    sh_piv[0] = -1.0 / d; sh_piv[1] = 0.0; sh_piv[2] = 0.0;
                    sh_kind = 1; sh_p = piv; sh_q = piv;
                }
                mask &= ~(1ull << piv);
            } else {
                const double a = B[piv][piv], b = B[piv][q], c = B[q][q];
                double det = a * c - b * b;                        // < 0 by the BK test
                if (!(fabs(det) <= 1.0e308)) bad = 1;
                const double tr = a + c, disc = sqrt((a - c) * (a - c) + 4.0 * b * b);
                const double e1 = 0.5 * (tr + disc), e2 = 0.5 * (tr - disc);
                n2++;
                if (fabs(e1) <= pivtol) zero++; else { if (e1 < 0.0) neg++; else pos++;
                    dmin = fmin(dmin, fabs(e1)); dmax = fmax(dmax, fabs(e1)); }
                if (fabs(e2) <= pivtol) zero++; else { if (e2 < 0.0) neg++; else pos++;
                    dmin = fmin(dmin, fabs(e2)); dmax = fmax(dmax, fabs(e2)); }
                if (det == 0.0) det = -2.2250738585072014e-308;
                const double ia = c / det, ib = -b / det, ic = a / det;
                const double vp = B[lane][piv], vq = B[lane][q];
                cp[lane] = vp; cq[lane] = vq;
                lp[lane] = vp * ia + vq * ib;
                lq[lane] = vp * ib + vq * ic;
                if (lane == 0) {
                    sh_piv[0] = -ia; sh_piv[1] = -ib; sh_piv[2] = -ic;
                    sh_kind = 2; sh_p = piv; sh_q = q;
                }
                mask &= ~((1ull << piv) | (1ull << q));
            }
        }
        __syncthreads();
        {
            const int kind = sh_kind, p = sh_p, q = sh_q;
            left -= kind;
            const int i = lane, jb = wave * 16;
            const double lpi = lp[i], lqi = lq[i];
            #pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = jb + jj;
                double v = B[i][j] - lpi * cp[j] - lqi * cq[j];
                if (j == p) v = lpi;
                if (kind == 2 && j == q) v = lqi;
                if (i == p) v = lp[j];
                if (kind == 2 && i == q) v = lq[j];
                if (i == p && j == p) v = sh_piv[0];
                if (kind == 2) {
                    if ((i == p && j == q) || (i == q && j == p)) v = sh_piv[1];
                    if (i == q && j == q) v = sh_piv[2];
                }
                B[i][j] = v;
            }
        }
        __syncthreads();
    }

    for (int e = tid; e < TB * TB; e += 256) {
        const int i = e & 63, j = e >> 6;
        Tinv[e] = -B[i][j];
    }
    if (tid == 0) {
        st->n_neg += neg; st->n_zero += zero; st->n_2x2 += n2; st->n_pos += pos;
        st->nonfinite += bad;
        if (dmin < st->d_min) st->d_min = dmin;
        if (dmax > st->d_max) st->d_max = dmax;
    }
}

// ---------------------------------------------------------------------------------------------
// Panel scaling: for rows below a factored tile, keep the Schur-complemented block as W and
// overwrite it with the block factor L = W * inv(T).  One wave per 64 rows; inv(T) in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_panel_scale(
    double* __restrict__ Aout, int64_t ld_out, int64_t col_out,      // L written at Aout[i + (col_out+c)*ld_out]
    const double* __restrict__ Win, int64_t ld_in, int64_t col_in,   // S read from  Win[i + (col_in+k)*ld_in]
    double* __restrict__ Wcopy, int64_t ld_w, int64_t col_w,         // copy of S (may be NULL = no copy)
    const double* __restrict__ Tinv, int64_t row_begin, unsigned long long* __restrict__ growth_bits)
{
    __shared__ double T[TB][TB];
    const int lane = threadIdx.x;
    #pragma unroll 8
    for (int k = 0; k < TB; ++k) T[k][lane] = Tinv[k * TB + lane];   // symmetric: T[c][k] == T[k][c]
    __syncthreads();
    const int64_t i = row_begin + (int64_t)blockIdx.x * TB + lane;
    double w[TB];
    #pragma unroll
    for (int k = 0; k < TB; ++k) w[k] = Win[i + (col_in + k) * ld_in];
    if (Wcopy) {
        #pragma unroll
        for (int k = 0; k < TB; ++k) Wcopy[i + (col_w + k) * ld_w] = w[k];
    }
    double gmax = 0.0;
    for (int c = 0; c < TB; ++c) {
        double acc = 0.0;
        #pragma unroll
        for (int k = 0; k < TB; ++k) acc = fma(w[k], T[c][k], acc);
        Aout[i + (col_out + c) * ld_out] = acc;
        gmax = fmax(gmax, fabs(acc));
    }
    gmax = wave_max(gmax);
    if (lane == 0 && growth_bits) atomicMax(growth_bits, (unsigned long long)__double_as_longlong(gmax));
}

// ---------------------------------------------------------------------------------------------
// K4: C[i,j] -= sum_k L[i,k] * W[j,k]  on the lower block-triangle, fp64 MFMA
// (v_mfma_f64_16x16x4_f64).  Column-major everywhere, so i is the contiguous index of C, L and W.
// MFMA orientation: D[m][n] with m <- j (A operand = W), n <- i (B operand = L); the f64 C/D map
// is col = lane&15 (-> i, contiguous), row = (lane>>4) + 4*reg (-> j).
// Block tile 128(i) x BN(j), 4 waves, each wave 64 x (BN/2); K staged through LDS 16 at a time
// with register prefetch.  W is negated while staging so the accumulator starts at C.
// Column tiles are enumerated over LOCALLY owned panels (block-cyclic), rows are global.
// ---------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(256, 2) void k_update(
    double* __restrict__ C, int64_t ldc,
    const double* __restrict__ Lop, int64_t ldl,
    const double* __restrict__ Wop, int64_t ldw,
    int K, int64_t row_begin, int64_t Npad,
    int64_t first_lp, int64_t sub0, int nb, int world, int rank)
{
    constexpr int TJ = BN / 32;            // 16-wide MFMA tiles per wave along j
    constexpr int TI = 4;                  // along i (wave covers 64 rows)
    constexpr int LSTR = BM + 16;          // padded LDS row strides (doubles): rows k and k+1 hit
    constexpr int WSTR = BN + 16;          // disjoint halves of the 64 banks
    __shared__ double Ls[BKU][LSTR];
    __shared__ double Ws[BKU][WSTR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // column tile -> (global column, local column)
    const int tiles_per_panel = nb / BN;
    const int64_t ct = (int64_t)blockIdx.y + sub0;
    const int64_t lp = first_lp + ct / tiles_per_panel;
    const int sub = (int)(ct % tiles_per_panel);
    const int64_t jglob = (lp * world + rank) * (int64_t)nb + (int64_t)sub * BN;
    const int64_t jloc = lp * (int64_t)nb + (int64_t)sub * BN;
    if (jglob >= Npad) return;
    const int64_t i0 = row_begin + (int64_t)blockIdx.x * BM;
    if (i0 + BM <= jglob) return;          // tile strictly above the diagonal
    const int wi = (wave & 1) * 64, wj = (wave >> 1) * (BN / 2);
    const int l15 = lane & 15, l4 = lane >> 4;

    double4_t acc[TJ][TI];
    #pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
        #pragma unroll
        for (int ti = 0; ti < TI; ++ti)
            #pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[tj][ti][r] = C[(i0 + wi + ti * 16 + l15) + (jloc + wj + tj * 16 + l4 + 4 * r) * ldc];

    // staging registers: L tile 16 x 128 doubles = 1024 double2 -> 4 per thread;
    //                    W tile 16 x BN  doubles -> BN/32 per thread
    constexpr int WPASS = BN / 32;
    double2_t lreg[4], wreg[WPASS];
    const int lk = tid >> 6, li = (tid & 63) * 2;                     // L: 4 k-rows per pass
    constexpr int WTPR = BN / 2;                                      // threads per k-row of W
    const int wk = tid / WTPR, wjj = (tid % WTPR) * 2;                // W: 256/WTPR k-rows per pass
    constexpr int WKPP = 256 / WTPR;

    const double* lsrc = Lop + (i0 + li) + (int64_t)lk * ldl;
    const double* wsrc = Wop + (jglob + wjj) + (int64_t)wk * ldw;
#define PYIPM_LOAD_REGS(k0_)                                                                          \
    {                                                                                                 \
        _Pragma("unroll") for (int ps = 0; ps < 4; ++ps)                                              \
            lreg[ps] = *reinterpret_cast<const double2_t*>(lsrc + (int64_t)((k0_) + 4 * ps) * ldl);     \
        _Pragma("unroll") for (int ps = 0; ps < WPASS; ++ps)                                          \
            wreg[ps] = *reinterpret_cast<const double2_t*>(wsrc + (int64_t)((k0_) + WKPP * ps) * ldw);  \
    }

    PYIPM_LOAD_REGS(0)
    for (int k0 = 0; k0 < K; k0 += BKU) {
        __syncthreads();
        #pragma unroll
        for (int ps = 0; ps < 4; ++ps)
            *reinterpret_cast<double2_t*>(&Ls[lk + 4 * ps][li]) = lreg[ps];
        #pragma unroll
        for (int ps = 0; ps < WPASS; ++ps) {
            *reinterpret_cast<double2_t*>(&Ws[wk + WKPP * ps][wjj]) = -wreg[ps];
        }
        __syncthreads();
        if (k0 + BKU < K) PYIPM_LOAD_REGS(k0 + BKU)
        #pragma unroll
        for (int kk = 0; kk < BKU; kk += 4) {
            double a[TJ], b[TI];
            #pragma unroll
            for (int tj = 0; tj < TJ; ++tj) a[tj] = Ws[kk + l4][wj + tj * 16 + l15];
            #pragma unroll
            for (int ti = 0; ti < TI; ++ti) b[ti] = Ls[kk + l4][wi + ti * 16 + l15];
            #pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
                #pragma unroll
                for (int ti = 0; ti < TI; ++ti)
                    acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tj], b[ti], acc[tj][ti], 0, 0, 0);
        }
    }
#undef PYIPM_LOAD_REGS

    #pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
        #pragma unroll
        for (int ti = 0; ti < TI; ++ti)
            #pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(i0 + wi + ti * 16 + l15) + (jloc + wj + tj * 16 + l4 + 4 * r) * ldc] = acc[tj][ti][r];
}

// Register-resident MFMA-only loop for the fp64 matrix peak measurement.
__global__ __launch_bounds__(256) void k_mfma_peak(double* out, int iters)
{
    double4_t acc[8];
    #pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int t = 0; t < 8; ++t)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0.0;
    #pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 123.456) out[0] = s;
}

}  // namespace pyipm
