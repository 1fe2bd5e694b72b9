// kernels_assemble.hpp — K1 KKT assembly, K2 residual, KKT mat-vec from blocks (HBM-bound).
//
// K1 replaces self.hess (/root/reference/pyipm.py:816-844).  The device matrix is
// column-major with the LOWER triangle referenced, which read as row-major is exactly
// triu(H) of the reference: column j (j < n) is [ d2L[j, j:n] | 0 (mi) | Je[j,:] | Ji[j,:] ],
// i.e. row j of the reference's upper-triangular build, so the copy is fully coalesced.
#pragma once
#include "ctx.hpp"

namespace pyipm {

// Rows of the derivative blocks a rank works on.  Single rank: all n rows, identity.  Several ranks: the rows that
// belong to the x-columns the rank owns (block-cyclic by panels of nb, as the KKT columns: column j of the lower
// triangle is row j of triu(d2L) | Je | Ji), in local column order; `sharded` says the caller staged only those rows
// (pyipm_newton_stage_blocks_owned), so a block row is addressed by the LOCAL index.
// (struct RowMap, make_rowmap: ctx.hpp -- the distributed driver is a translation unit of its own)


// KKT entry (i, j), i >= j, in the reference's block order (pyipm.py:816-844 + reghess' shifts).
__device__ __forceinline__ double kkt_entry(
    int64_t i, int64_t j, int64_t jr, const Geo& g,      // jr: row of the derivative blocks that holds column j (j itself unless row-sharded)
    const double* __restrict__ d2L, int64_t ldh, const double* __restrict__ Je, int64_t ldje,
    const double* __restrict__ Ji, int64_t ldji, const double* __restrict__ s, const double* __restrict__ lda,
    double eps, double delta, double delta_c)
{
    const int64_t n = g.n, me = g.me, mi = g.mi, N = g.N;
    const int64_t o_s = n, o_e = n + mi, o_i = n + mi + me;
    if (j < n) {
        if (i < n)        return d2L[jr * ldh + i] + (i == j ? delta : 0.0);
        if (i < o_e)      return 0.0;
        if (i < o_i)      return Je[jr * ldje + (i - o_e)];
        if (i < N)        return Ji[jr * ldji + (i - o_i)];
        return 0.0;
    }
    if (j < o_e) {                                     // slack columns: Sigma and -I
        const int64_t b = j - o_s;
        if (i == j)       return lda[me + b] / (s[b] + eps);           // pyipm.py:498
        if (i == o_i + b) return -1.0;                                  // pyipm.py:838-842
        return 0.0;
    }
    if (j < o_i) return (i == j) ? -delta_c : 0.0;     // lambda_e columns: -delta_c I (reghess)
    if (j < N)   return 0.0;                           // lambda_i columns: zero block
    return (i == j) ? 1.0 : 0.0;                       // identity pad
}

// Largest |entry| a wave has assembled -> *anorm_bits (bit pattern of a non-negative double: monotone as an integer).
// One read of the running maximum per wave; the atomic only fires while the maximum still grows.
__device__ __forceinline__ void anorm_publish(unsigned long long* __restrict__ anorm_bits, double amax) {
    if (!anorm_bits) return;
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
    if ((threadIdx.x & 63) == 0 && amax <= 1.0e308) {          // (a NaN / Inf entry is the factorisation's to report)
        const unsigned long long b = (unsigned long long)__double_as_longlong(amax);
        if (b > *reinterpret_cast<volatile unsigned long long*>(anorm_bits)) atomicMax(anorm_bits, b);
    }
}

// One thread-block writes a 512(i) x 16(j) patch of the local storage; each thread owns two consecutive
// rows so every store is 16 bytes (and the dominant d2L reads too, when the leading dimension allows).
typedef double dbl2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_assemble(
    double* __restrict__ A, int64_t ld, Geo g,
    const double* __restrict__ d2L, int64_t ldh,
    const double* __restrict__ Je, int64_t ldje,
    const double* __restrict__ Ji, int64_t ldji,
    const double* __restrict__ s, const double* __restrict__ lda,
    double eps, double delta, double delta_c, unsigned long long* __restrict__ anorm_bits, int nt_store, int sharded,
    int zeros_in_place,      // the storage still holds the zeros of an earlier assembly wherever nothing can ever fill in
    int64_t lc_off,          // first local column of this launch (the assembly may come as two launches, see assemble_dev)
    int tri_nbx)             // > 0 (single rank, lc_off == 0): a 1-D grid over the patches on or below the diagonal only -- the 2-D
                             // grid launches as many patches above it, 65000 workgroups at N = 32768 that start only to return
{
    int64_t bx = blockIdx.x, by = blockIdx.y;
    if (tri_nbx > 0) {
        // 32 column patches (512 columns) share their first row patch G = by / 32; group G holds 32 (nbx - G) patches, the
        // groups before it P(G) = 32 (G nbx - G (G - 1) / 2).  Invert P with a square root and correct by one.
        const int64_t L = blockIdx.x, nbx = tri_nbx;
        const double b2 = (double)(2 * nbx + 1);
        int64_t G = (int64_t)((b2 - sqrt(b2 * b2 - 8.0 * ((double)L / 32.0))) * 0.5);
        if (G < 0) G = 0;
        while (G > 0 && 32 * (G * nbx - G * (G - 1) / 2) > L) --G;
        while (32 * ((G + 1) * nbx - (G + 1) * G / 2) <= L) ++G;
        const int64_t rem = L - 32 * (G * nbx - G * (G - 1) / 2), per = nbx - G;
        by = G * 32 + rem / per; bx = G + rem % per;
    }
    const int64_t i = (bx * 256 + threadIdx.x) * 2;          // even row; rows i, i+1
    const int64_t lc_base = lc_off + by * 16;
    if (lc_base >= g.ncols_local) return;
    if (i >= g.Npad) return;                          // (wave-uniform: Npad is a multiple of 128)
    const bool vec_h = ((ldh & 1) == 0) && ((reinterpret_cast<uintptr_t>(d2L) & 15) == 0);
    // the 16 columns of a block lie in ONE panel (nb is a multiple of 128): one division per block, not one per entry
    const int64_t lp = lc_base / g.nb;
    const int64_t j0 = (lp * g.world + g.rank) * (int64_t)g.nb + (lc_base - lp * g.nb);
    if (bx * 512 + 512 <= j0) return;     // the whole patch lies above the diagonal: nothing is stored there
    if (zeros_in_place) {
        // Zeros for ever (pyipm.py:824-842 and the elimination order x, s, lambda_e, lambda_i): the (s, x) block; below the
        // diagonal of the (s, s) block; the (lambda_e, s) block; the (lambda_i, s) block off its diagonal of -1.  Nothing
        // fills them in -- every update that reaches them adds an exact zero -- so once written they need no second
        // store (34 % of the lower triangle at the benchmark shape).  Decided per 512 x 16 patch, block-uniform.
        const int64_t r0 = bx * 512, r1 = r0 + 512;           // rows [r0, r1), columns [j0, j0 + 16)
        const int64_t o_s = g.n, o_e = g.n + g.mi, o_i = o_e + g.me;
        if (j0 + 16 <= o_s && r0 >= o_s && r1 <= o_e) return;                  // (s, x)
        if (j0 >= o_s && j0 + 16 <= o_e) {
            if (r0 > j0 + 15 && r1 <= o_i) return;                             // (s, s) below the patch's diagonal entries, (lambda_e, s)
            if (r0 >= o_i && r1 <= g.N) {
                const int64_t d0 = o_i + (j0 - o_s);                           // rows of the patch's -1 entries: [d0, d0 + 16)
                if (r1 <= d0 || r0 >= d0 + 16) return;
            }
        }
    }
    // (a lane whose rows lie above the whole patch has nothing to do, but stays for the wave reduction at the end)
    const int ncol = (i + 1 < j0) ? 0 : ((g.ncols_local - lc_base) < 16 ? (int)(g.ncols_local - lc_base) : 16);
    // values first (all loads of the thread in flight), stores afterwards
    dbl2_t val[16];
    #pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int64_t j = j0 + c;
        const int64_t jr = sharded ? lc_base + c : j;  // row-sharded blocks are laid out in local column order
        dbl2_t v; v.x = 0.0; v.y = 0.0;
        if (c < ncol && i + 1 >= j) {
            if (vec_h && j < g.n && i + 1 < g.n && i >= j) {
                v = *reinterpret_cast<const dbl2_t*>(&d2L[jr * ldh + i]);
                if (i == j) v.x += delta;
            } else {
                v.x = (i >= j) ? kkt_entry(i, j, jr, g, d2L, ldh, Je, ldje, Ji, ldji, s, lda, eps, delta, delta_c) : 0.0;
                v.y = kkt_entry(i + 1, j, jr, g, d2L, ldh, Je, ldje, Ji, ldji, s, lda, eps, delta, delta_c);
            }
        }
        val[c] = v;
    }
    double amax = 0.0;
    #pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int64_t j = j0 + c, lc = lc_base + c;
        if (c >= ncol || i + 1 < j) continue;         // both rows above the diagonal: not stored
        const dbl2_t v = val[c];
        {   // the scale of a static pivot leaves Sigma = lambda_i / (s + eps) out: late in an interior-point run those entries span
            // 1e-10 .. 1e+10 while the rows a static pivot lands in (zero Hessian rows: LPs, linear variables) keep the scale of
            // their Jacobian entries.  With Sigma in, the "perturbation" of iterate 8 of an LP was 0.005 on rows of norm 1 --
            // the reference (pyipm.py:18-20, LU over the whole matrix) solves that system unshifted; found by the LP trace
            // fixture of round 3 (tests/test_gpu_pivoting.py).
            const bool sig = j >= g.n && j < g.n + g.mi;
            amax = fmax(amax, fmax((sig && i == j) ? 0.0 : fabs(v.x), (sig && i + 1 == j) ? 0.0 : fabs(v.y)));
        }
        if (i >= j) {
            // (non-temporal stores measured no faster, r02: 1.99 vs 1.95 ms at N = 32768)
            if (nt_store) __builtin_nontemporal_store(v, reinterpret_cast<dbl2_t*>(&A[i + lc * ld]));
            else          *reinterpret_cast<dbl2_t*>(&A[i + lc * ld]) = v;
        } else {
            A[(i + 1) + lc * ld] = v.y;               // the pair straddles the diagonal: store the lower one only
        }
    }
    anorm_publish(anorm_bits, amax);
}

// ---- wave-level helpers -------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// y[j] (op)= sum_a M[j*ldm + a] * x[a]   for a in [0, ncol); one wave per row j.
// Two matrices are fused so that K2 needs one pass.  mode: 0 -> out = base[j] - acc (then negated
// if neg), used by the residual;  1 -> out[j] += acc;  2 -> out[j] = acc.
__global__ __launch_bounds__(256) void k_rowdot2(
    double* __restrict__ out, const double* __restrict__ base, int64_t nrow,
    const double* __restrict__ M1, int64_t ld1, const double* __restrict__ x1, int64_t nc1,
    const double* __restrict__ M2, int64_t ld2, const double* __restrict__ x2, int64_t nc2,
    int mode, int neg, RowMap rm)
{
    const int lane = threadIdx.x & 63;
    const int64_t jl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (jl >= nrow) return;
    const int64_t j = rm.glob(jl), jb = rm.brow(jl);     // out / base index; row of the blocks
    double acc = 0.0;
    if (nc1 > 0) {
        const double* r = M1 + jb * ld1;
        #pragma unroll 8
        for (int64_t a = lane; a < nc1; a += 64) acc += r[a] * x1[a];
    }
    if (nc2 > 0) {
        const double* r = M2 + jb * ld2;
        #pragma unroll 8
        for (int64_t a = lane; a < nc2; a += 64) acc += r[a] * x2[a];
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        if (mode == 0)      { double v = base[j] - acc; out[j] = neg ? -v : v; }
        else if (mode == 1) { out[j] += acc; }
        else                { out[j] = acc; }                 // mode 2: the plain product
    }
}

// K2 tail: the s, lambda_e, lambda_i parts of g = -grad (pyipm.py:655-668, negated at :1717),
// and zero of the pad.
__global__ __launch_bounds__(256) void k_residual_tail(
    double* __restrict__ g, Geo geo, const double* __restrict__ ce, const double* __restrict__ ci,
    const double* __restrict__ s, const double* __restrict__ lda, double mu, double eps)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;   // index into [n, Npad)
    const int64_t i = geo.n + t;
    if (i >= geo.Npad) return;
    const int64_t n = geo.n, me = geo.me, mi = geo.mi;
    double v = 0.0;
    if (i < n + mi)            { int64_t b = i - n;           v = -(lda[me + b] - mu / (s[b] + eps)); }
    else if (i < n + mi + me)  { int64_t a = i - n - mi;      v = -ce[a]; }
    else if (i < geo.N)        { int64_t b = i - n - mi - me; v = -(ci[b] - s[b]); }
    g[i] = v;
}

// Upper-triangle symmetric product, row part:  y[j] = sum_{k>=j} U[j,k] v[k] + delta v[j]
__global__ __launch_bounds__(256) void k_symv_row(
    double* __restrict__ y, const double* __restrict__ U, int64_t ldh, int64_t n,
    const double* __restrict__ v, double delta, RowMap rm)
{
    const int lane = threadIdx.x & 63;
    const int64_t jl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (jl >= rm.nloc) return;
    const int64_t j = rm.glob(jl);
    const double* r = U + rm.brow(jl) * ldh;
    double acc = 0.0;
    for (int64_t k = j + lane; k < n; k += 64) acc += r[k] * v[k];
    acc = wave_sum(acc);
    if (lane == 0) y[j] = acc + delta * v[j];
}

// sym(triu(U)) v in ONE pass over the upper triangle (round 4).  The two kernels above / below read the
// triangle once each -- row part and mirrored part -- i.e. 8 n^2 bytes for a product that needs 4 n^2.  Here a block takes the
// rows of chunk blockIdx.y times a segment of SYMV_SEG columns (blockIdx.x) and streams the rows 16 at a time, every thread
// one COLUMN of a 256-column window (a row is read in 2 KB pieces, 16 row streams per block -- a first version that staged
// 64 x 64 tiles through shared memory kept 64 streams of 512 B per block going and reached only 2.1 TB/s):
//   column sums  c[k] += U[j,k] v[j]  (j < k): the 16 values a thread has just loaded ARE its column's -- summed in registers,
//                one accumulator per window of the segment, kept over all rows of the chunk -> part_col[chunk][k];
//   row sums     r[j] += U[j,k] v[k]  (k >= j): accumulated per thread over the windows of the segment, folded over the 256
//                threads once per 16 rows (through shared memory) -> part_row[segment][local row].
// Every load is unconditional (clamped address, masked value).  Deterministic: every partial has one writer and a fixed
// order; k_symv_finish adds the partials in a fixed order.
constexpr int SYMV_SEG = 1024;           // columns per block: SYMV_SEG / 256 column accumulators per thread
constexpr int SYMV_CHUNKS = 128;         // row chunks (the column partials are [SYMV_CHUNKS][ncols])
// TRI: the upper triangle of a square matrix (row sums over k >= j, column sums over j < k), vcol = vrow = v.
// !TRI (round 4): a rectangular block M (rows = the x rows this rank works on, ncols columns) -- BOTH products of a Jacobian
// block from one pass:  part_row -> M vcol (Je v_e: into the x rows),  part_col -> M' vrow (Je' v_x: the multiplier rows).
template <bool TRI>
__device__ __forceinline__ void rowcol_tiles(
    double* __restrict__ part_row, double* __restrict__ part_col, const double* __restrict__ U, int64_t ldh, int64_t n,
    const double* __restrict__ vcol, const double* __restrict__ vrow, int64_t rows_per_chunk, const RowMap& rm)
{
    constexpr int NW = SYMV_SEG / 256;
    __shared__ double red[16][257];
    const int tid = threadIdx.x;
    const int64_t k_seg0 = (int64_t)blockIdx.x * SYMV_SEG;
    int64_t k_seg1 = k_seg0 + SYMV_SEG; if (k_seg1 > n) k_seg1 = n;
    const int64_t jl0 = (int64_t)blockIdx.y * rows_per_chunk;             // local rows of this chunk (a multiple of 64)
    int64_t jl1 = jl0 + rows_per_chunk; if (jl1 > rm.nloc) jl1 = rm.nloc;
    if (jl0 >= jl1 || k_seg0 >= n) return;
    double cacc[NW], vk[NW];
    #pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int64_t k = k_seg0 + 256 * w + tid;
        cacc[w] = 0.0; vk[w] = (k < n) ? vcol[k] : 0.0;
    }
    for (int64_t gl = jl0; gl < jl1; gl += 16) {                          // 16 local rows: contiguous global rows jg .. jg + 15
        const int64_t jg = rm.glob(gl);
        const int rows = (int)((jl1 - gl) < 16 ? (jl1 - gl) : 16);
        if (TRI && jg >= k_seg1) {                                        // (block-uniform) wholly left of / below the segment's columns
            if (tid < rows) part_row[(int64_t)blockIdx.x * rm.nloc + (gl + tid)] = 0.0;
            continue;
        }
        const double* rowp[16]; double vr[16];
        #pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rc = r < rows ? r : rows - 1;
            rowp[r] = U + rm.brow(gl + rc) * ldh;
            vr[r] = (r < rows) ? vrow[jg + rc] : 0.0;
        }
        double racc[16];
        #pragma unroll
        for (int r = 0; r < 16; ++r) racc[r] = 0.0;
        #pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int64_t kt = k_seg0 + 256 * w;
            if ((TRI && kt + 256 <= jg) || kt >= k_seg1) continue;        // (uniform) the window lies left of the diagonal / past the matrix
            const int64_t k = kt + tid, kc = k < n ? k : n - 1;
            double u[16];
            #pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = rowp[r][kc];
            double ca = 0.0;
            #pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t j = jg + r;
                const double uu = (r < rows && k < n && (!TRI || k >= j)) ? u[r] : 0.0;
                racc[r] = fma(uu, vk[w], racc[r]);
                ca = fma((!TRI || k > j) ? uu : 0.0, vr[r], ca);
            }
            cacc[w] += ca;
        }
        // fold the 16 row sums over the 256 threads: [row][thread] through shared memory, 16 threads per row
        __syncthreads();                                                  // (the group before has been read)
        #pragma unroll
        for (int r = 0; r < 16; ++r) red[r][tid] = racc[r];
        __syncthreads();
        {
            const int r = tid >> 4, q = tid & 15;
            double a = 0.0;
            #pragma unroll
            for (int c = 0; c < 16; ++c) a += red[r][c * 16 + q];
            a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 8, 64);
            if (q == 0 && r < rows) part_row[(int64_t)blockIdx.x * rm.nloc + (gl + r)] = a;
        }
    }
    #pragma unroll
    for (int w = 0; w < NW; ++w) {
        const int64_t k = k_seg0 + 256 * w + tid;
        if (k < k_seg1) part_col[(int64_t)blockIdx.y * n + k] = cacc[w];
    }
}
__global__ __launch_bounds__(256) void k_symv_tiles(
    double* __restrict__ part_row, double* __restrict__ part_col, const double* __restrict__ U, int64_t ldh, int64_t n,
    const double* __restrict__ v, int64_t rows_per_chunk, RowMap rm)
{
    rowcol_tiles<true>(part_row, part_col, U, ldh, n, v, v, rows_per_chunk, rm);
}
// One Jacobian block (rows = the x rows of this rank, ncols = me or mi): part_row[seg][local row] = partial of M vcol,
// part_col[chunk][k] = partial of M' vrow (vrow indexed by GLOBAL row).  k_jac_finish_rows adds the row partials into y.
__global__ __launch_bounds__(256) void k_jac_tiles(
    double* __restrict__ part_row, double* __restrict__ part_col, const double* __restrict__ M, int64_t ldm, int64_t ncols,
    const double* __restrict__ vcol, const double* __restrict__ vrow, int64_t rows_per_chunk, RowMap rm)
{
    rowcol_tiles<false>(part_row, part_col, M, ldm, ncols, vcol, vrow, rows_per_chunk, rm);
}
// y[glob(jl)] += sum_seg part_row[seg][jl]  (fixed order)
__global__ __launch_bounds__(256) void k_jac_finish_rows(double* __restrict__ y, const double* __restrict__ part_row, int nseg, RowMap rm)
{
    const int64_t jl = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (jl >= rm.nloc) return;
    double acc = 0.0;
    for (int sgi = 0; sgi < nseg; ++sgi) acc += part_row[(int64_t)sgi * rm.nloc + jl];
    y[rm.glob(jl)] += acc;
}

// y[a] = (a is a row this rank works on ? sum_seg part_row[seg][local row] + delta v[a] : 0) + sum_chunk part_col[chunk][a]
__global__ __launch_bounds__(256) void k_symv_finish(
    double* __restrict__ y, const double* __restrict__ part_row, const double* __restrict__ part_col, int64_t n,
    const double* __restrict__ v, double delta, int nseg, int nchunk, RowMap rm)
{
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= n) return;
    double acc = 0.0;
    int64_t jl = -1;
    if (rm.world == 1) jl = a;
    else { const int64_t p = a / rm.nb; if (p % rm.world == rm.rank) jl = (p / rm.world) * (int64_t)rm.nb + a % rm.nb; }
    if (jl >= 0 && jl < rm.nloc) {
        for (int sg = 0; sg < nseg; ++sg) acc += part_row[(int64_t)sg * rm.nloc + jl];
        acc += delta * v[a];
    }
    double cacc = 0.0;
    int c = 0;
    for (; c + 8 <= nchunk; c += 8) {                       // eight partials in flight, added in chunk order
        double t[8];
        #pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = part_col[(int64_t)(c + q) * n + a];
        #pragma unroll
        for (int q = 0; q < 8; ++q) cacc += t[q];
    }
    for (; c < nchunk; ++c) cacc += part_col[(int64_t)c * n + a];
    y[a] = acc + cacc;
}

// Column-walk partial products, deterministic two-pass:
//   part[chunk][a] = sum_{j in chunk, j < jmax(a)} M[j*ldm + a] * x[j]
// strict_upper != 0 restricts to j < a (the mirrored half of triu(d2L)).
__global__ __launch_bounds__(256) void k_coldot_partial(
    double* __restrict__ part, const double* __restrict__ M, int64_t ldm, int64_t nrow, int64_t ncol,
    const double* __restrict__ x, int64_t rows_per_chunk, int strict_upper, RowMap rm)
{
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= ncol) return;
    const int64_t j0 = (int64_t)blockIdx.y * rows_per_chunk;             // chunks of the rows this rank works on
    int64_t j1 = j0 + rows_per_chunk; if (j1 > nrow) j1 = nrow;
    double acc = 0.0;
    if (rm.world == 1) {
        if (strict_upper && j1 > a) j1 = a;
        // eight rows in flight per thread (one load per trip left the memory pipe waiting on the add); same order of additions
        int64_t j = j0;
        for (; j + 8 <= j1; j += 8) {
            double m[8];
            #pragma unroll
            for (int q = 0; q < 8; ++q) m[q] = M[(j + q) * ldm + a];
            #pragma unroll
            for (int q = 0; q < 8; ++q) acc += m[q] * x[j + q];
        }
        for (; j < j1; ++j) acc += M[j * ldm + a] * x[j];
    } else {
        for (int64_t jl = j0; jl < j1; ++jl) {
            const int64_t j = rm.glob(jl);                                // (ascending in jl)
            if (strict_upper && j >= a) break;
            acc += M[rm.brow(jl) * ldm + a] * x[j];
        }
    }
    part[(int64_t)blockIdx.y * ncol + a] = acc;
}

__global__ __launch_bounds__(256) void k_coldot_reduce(
    double* __restrict__ y, const double* __restrict__ part, int64_t ncol, int nchunk, int accumulate)
{
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= ncol) return;
    double acc = 0.0;
    int c = 0;
    for (; c + 8 <= nchunk; c += 8) {
        double t[8];
        #pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = part[(int64_t)(c + q) * ncol + a];
        #pragma unroll
        for (int q = 0; q < 8; ++q) acc += t[q];
    }
    for (; c < nchunk; ++c) acc += part[(int64_t)c * ncol + a];
    y[a] = accumulate ? y[a] + acc : acc;
}

// Element-wise remainder of y = Hc v:  Sigma v_s - v_i ;  -delta_c v_e ;  -v_s ; pad passthrough.
// (J' v_x terms are accumulated by the coldot kernels afterwards.)
__global__ __launch_bounds__(256) void k_matvec_tail(
    double* __restrict__ y, const double* __restrict__ v, Geo geo,
    const double* __restrict__ s, const double* __restrict__ lda, double eps, double delta_c)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = geo.n + t;
    if (i >= geo.Npad) return;
    const int64_t n = geo.n, me = geo.me, mi = geo.mi;
    double r;
    if (i < n + mi)           { int64_t b = i - n; r = lda[me + b] / (s[b] + eps) * v[i] - v[n + mi + me + b]; }
    else if (i < n + mi + me) { r = -delta_c * v[i]; }
    else if (i < geo.N)       { int64_t b = i - n - mi - me; r = -v[n + b]; }
    else                      { r = v[i]; }
    y[i] = r;
}

__global__ __launch_bounds__(256) void k_axpby(double* __restrict__ out, const double* __restrict__ a,
                                               const double* __restrict__ b, double alpha, double beta, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = alpha * a[i] + beta * b[i];
}

// dz[nvar+nineq:] = -dz[nvar+nineq:]  (pyipm.py:1723-1725) fused with the copy-out.
__global__ __launch_bounds__(256) void k_copy_flip(double* __restrict__ out, const double* __restrict__ in,
                                                   int64_t N, int64_t flip_from, int flip)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) out[i] = (flip && i >= flip_from) ? -in[i] : in[i];
}

// Fraction-to-the-boundary step lengths (pyipm.py:1408-1436 in closed form) for v = s and v = lda_i.
// One block; out[0] = alpha_s, out[1] = alpha_l.
__global__ __launch_bounds__(256) void k_step_lengths(
    double* __restrict__ out, const double* __restrict__ s, const double* __restrict__ lda_i,
    const double* __restrict__ ds, const double* __restrict__ dl, int64_t mi, double tau)
{
    __shared__ double red[2][4];
    double as = 1.0, al = 1.0;
    for (int64_t i = threadIdx.x; i < mi; i += 256) {
        const double a = ds[i], b = dl[i];
        if (a < 0.0) as = fmin(as, -tau * s[i] / a);
        if (b < 0.0) al = fmin(al, -tau * lda_i[i] / b);
    }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        as = fmin(as, __shfl_xor(as, off, 64));
        al = fmin(al, __shfl_xor(al, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = as; red[1][threadIdx.x >> 6] = al; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = fmin(fmin(red[0][0], red[0][1]), fmin(red[0][2], red[0][3]));
        out[1] = fmin(fmin(red[1][0], red[1][1]), fmin(red[1][2], red[1][3]));
    }
}

// ---- condensed KKT: eliminate s and lambda_i ---------------------------
// Inequalities are split by Sigma_k = lda_i[k] / (s[k] + eps) (pyipm.py:498):
//   I (Sigma_k <= theta): pair (s_k, lambda_i_k) eliminated ->  Ji_I Sigma_I Ji_I' added to the x-x block;
//   A (Sigma_k >  theta): only s_k eliminated; lambda_i_k stays as a row  Ji_k' dx - dli_k / Sigma_k = b_i + b_s / Sigma_k
// (the active constraints late in a run: folding a Sigma ~ 1e10 into the x-x block would make a dense
// matrix of that condition number and multiply the rounding of ds by Sigma; as a row it tends to an
// equality constraint).  System, dimension n + me + |A|, order [x | lambda_e | lambda_A]:
//   [[H + delta I + Ji_I Sigma_I Ji_I', Je, Ji_A], [Je', -delta_c I, 0], [Ji_A', 0, -Sigma_A^-1]]
//   rhs [ b_x + Ji_I (Sigma_I b_i + b_s)_I ; b_e ; (b_i + b_s / Sigma)_A ]
//   recovery  I: ds = Ji' dx - b_i , dli = Sigma ds - b_s ;   A: dli from the solve, ds = (dli + b_s) / Sigma

// pos[k] = index of k inside A (or -1), idx[j] = k of the j-th member, *count = |A|.  One block.
__global__ __launch_bounds__(1024) void k_active_scan(int* __restrict__ pos, int* __restrict__ idx, int* __restrict__ count,
                                                       const double* __restrict__ s, const double* __restrict__ lda_i,
                                                       double eps, double theta, int64_t mi)
{
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int64_t per = (mi + 1023) / 1024, k0 = t * per;
    int64_t k1 = k0 + per; if (k1 > mi) k1 = mi;
    int c = 0;
    for (int64_t k = k0; k < k1; ++k) c += (lda_i[k] / (s[k] + eps) > theta) ? 1 : 0;
    part[t] = c;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // inclusive Hillis-Steele scan
        const int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int at = part[t] - c;
    for (int64_t k = k0; k < k1; ++k) {
        if (lda_i[k] / (s[k] + eps) > theta) { pos[k] = at; idx[at] = (int)k; ++at; }
        else pos[k] = -1;
    }
    if (t == 1023) *count = part[1023];
}

// Jx = [ Je | Ji[:, A] ]  (n x (me + na), row-major): the "equality-like" Jacobian of the condensed system
__global__ __launch_bounds__(256) void k_cond_gather_J(double* __restrict__ Jx, int64_t ldx,
                                                       const double* __restrict__ Je, int64_t ldje, int64_t me,
                                                       const double* __restrict__ Ji, int64_t ldji,
                                                       const int* __restrict__ idx, int64_t na)
{
    const int64_t i = blockIdx.x;                         // row (grid.x: no 65535 limit)
    const int64_t a = (int64_t)blockIdx.y * 256 + threadIdx.x;
    if (a >= me + na) return;
    Jx[i * ldx + a] = (a < me) ? Je[i * ldje + a] : Ji[i * ldji + idx[a - me]];
}

// diagonal of the lambda_A rows: -1 / Sigma_k
__global__ __launch_bounds__(256) void k_cond_fix_diag(double* __restrict__ A, int64_t ld, int64_t r0,
                                                       const int* __restrict__ idx, int64_t na,
                                                       const double* __restrict__ s, const double* __restrict__ lda_i, double eps,
                                                       Geo gc)        // the condensed geometry: a rank stores the columns it owns
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= na) return;
    const int k = idx[j];
    const int64_t col = r0 + j, q = col / gc.nb;
    if (gc.owner(q) != gc.rank) return;
    const int64_t lcol = (q / gc.world) * (int64_t)gc.nb + (col - q * gc.nb);
    A[col + lcol * ld] = -(s[k] + eps) / lda_i[k];
}

// JT[i + k*ldt] = Ji[i][k]  and  WT[i + k*ldt] = Sigma_k Ji[i][k]  (i < n, k in I; zero elsewhere up to
// rows ldt / columns mi_pad): the operands of the rank-mi MFMA update  C += JT * WT' = C + Ji_I Sigma_I Ji_I'
// (k_update accumulates Lop * Wop'; the factorisation feeds it the pre-negated W).
// 32x32 tiles through LDS: reads coalesced along k, writes coalesced along i.
__global__ __launch_bounds__(256) void k_transpose_scale(
    double* __restrict__ JT, double* __restrict__ WT, int64_t ldt, const double* __restrict__ Ji, int64_t ldji,
    int64_t n, int64_t mi, int64_t mi_pad, const double* __restrict__ s, const double* __restrict__ lda_i, double eps,
    const int* __restrict__ pos)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const int64_t i0 = (int64_t)blockIdx.x * 32, k0 = (int64_t)blockIdx.y * 32;
    #pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + ty + 8 * r, k = k0 + tx;
        tile[ty + 8 * r][tx] = (i < n && k < mi) ? Ji[i * ldji + k] : 0.0;
    }
    __syncthreads();
    #pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t k = k0 + ty + 8 * r, i = i0 + tx;
        if (k < mi_pad && i < ldt) {
            const double v = tile[tx][ty + 8 * r];
            const double sg = (k < mi && pos[k] < 0) ? lda_i[k] / (s[k] + eps) : 0.0;
            JT[i + k * ldt] = v;
            WT[i + k * ldt] = sg * v;
        }
    }
}

// t[k] = -(Sigma_k b_i[k] + b_s[k]) for k in I, 0 for k in A   (negated: k_rowdot2's "base - acc" form yields b_x + Ji t)
__global__ __launch_bounds__(256) void k_cond_t(double* __restrict__ t, const double* __restrict__ b, Geo g,
                                                const double* __restrict__ s, const double* __restrict__ lda_i, double eps,
                                                const int* __restrict__ pos)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= g.mi) return;
    const double sg = lda_i[k] / (s[k] + eps);
    t[k] = (pos[k] < 0) ? -(sg * b[g.n + g.mi + g.me + k] + b[g.n + k]) : 0.0;
}

// condensed right-hand side tail: vc[n + a] = b_e[a] ; vc[n + me + j] = b_i[k] + b_s[k] / Sigma_k, k = idx[j] ; zero pad
__global__ __launch_bounds__(256) void k_cond_gather(double* __restrict__ vc, const double* __restrict__ b, Geo g, int64_t npad_c,
                                                     const int* __restrict__ idx, int64_t na,
                                                     const double* __restrict__ s, const double* __restrict__ lda_i, double eps)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = g.n + t;
    if (i >= npad_c) return;
    double v = 0.0;
    if (t < g.me) v = b[g.n + g.mi + t];
    else if (t < g.me + na) {
        const int k = idx[t - g.me];
        v = b[g.n + g.mi + g.me + k] + b[g.n + k] * (s[k] + eps) / lda_i[k];
    }
    vc[i] = v;
}

// full solution from the condensed one: v (holding b on entry) <- [dx ; ds ; dle ; dli],  u = Ji' dx
__global__ __launch_bounds__(256) void k_cond_expand(double* __restrict__ v, const double* __restrict__ vc,
                                                     const double* __restrict__ u, Geo g,
                                                     const double* __restrict__ s, const double* __restrict__ lda_i, double eps,
                                                     const int* __restrict__ pos)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= g.Npad) return;
    const int64_t n = g.n, me = g.me, mi = g.mi;
    if (i < n) { v[i] = vc[i]; return; }
    if (i < n + mi) {                                   // one thread does the (s, lambda_i) pair of index k
        const int64_t k = i - n;
        const double bs = v[n + k], bi = v[n + mi + me + k];
        const double sg = lda_i[k] / (s[k] + eps);
        double ds, dl;
        if (pos[k] < 0) { ds = u[k] - bi; dl = sg * ds - bs; }
        else            { dl = vc[n + me + pos[k]]; ds = (dl + bs) / sg; }
        v[n + k] = ds;
        v[n + mi + me + k] = dl;
        return;
    }
    if (i < n + mi + me) { v[i] = vc[n + (i - n - mi)]; return; }
    if (i >= g.N) v[i] = 0.0;
}

// out[0] = sum a_i^2, out[1] = sum b_i^2 (one block, fixed order: deterministic).  The adaptive refinement of
// pyipm_newton_solve(refine < 0) reads the two back to form |b - Hc x| / |b|.
__global__ __launch_bounds__(1024) void k_sumsq2(double* __restrict__ out, const double* __restrict__ a,
                                                 const double* __restrict__ b, int64_t n)
{
    __shared__ double red[2][16];
    double sa = 0.0, sb = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) { const double x = a[i], y = b[i]; sa = fma(x, x, sa); sb = fma(y, y, sb); }
    sa = wave_sum(sa); sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sa; red[1][threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int w = 0; w < 16; ++w) { ta += red[0][w]; tb += red[1][w]; }
        out[0] = ta; out[1] = tb;
    }
}

// Deterministic pseudo-random start vector in [-1, 1) for the eigenvalue estimates (zero in the pad).
__global__ __launch_bounds__(256) void k_hash_vector(double* __restrict__ out, int64_t N, int64_t Npad, unsigned long long seed)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Npad) return;
    unsigned long long z = (unsigned long long)i + seed * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;   // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    out[i] = (i < N) ? (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0 : 0.0;
}

// out = alpha * in
__global__ __launch_bounds__(256) void k_scale_copy(double* __restrict__ out, const double* __restrict__ in, double alpha, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = alpha * in[i];
}

__global__ __launch_bounds__(256) void k_fill(double* __restrict__ out, double v, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}

}  // namespace pyipm
