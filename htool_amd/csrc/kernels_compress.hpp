// kernels_compress.hpp -- compression of the admissible blocks (partialACA / sympartialACA, workgroup teams, host-generator form, fullACA, SVD, recompression).
// Part of the engine's device code: included by kernels_body.hpp inside namespace hmx::{f64,f32,z64,c32}, written against `scalar` / `real`.  No include guard on purpose.

// ---------------------------------------------------------------------------------------------
// Compression: partially pivoted ACA, one workgroup per admissible block.
// ---------------------------------------------------------------------------------------------
struct AcaArgs {
    KernelSpec ks;
    const double *tx, *ty, *tz; // target coordinates, cluster order (SoA)
    const double *sx, *sy, *sz; // source coordinates, cluster order (SoA)
    const int32_t *order;       // launch order -> block id (largest first)
    const int32_t *t_off, *t_size, *s_off, *s_size;
    int symmetric_pivoting;     // sympartialACA: pivot on the larger-offset cluster first
    double epsilon;
    int reqrank;
    scalar *pool;               // cross storage, bump allocated
    unsigned long long *pool_head;
    unsigned long long pool_cap;
    const int64_t *colptr;      // per block: first slot in cross_off
    const int32_t *colcap;      // per block: slots available
    int64_t *cross_off;         // per (block, k): pool offset of [uu_k (n1) | vv_k (n2)]
    unsigned char *visited;     // per block: n1 + n2 flags
    const int64_t *vis_ptr;
    int32_t *rank_out;          // > 0 rank; 0 = compressor failed (dense fallback); -2 = pool exhausted
    int32_t *swapped_out;       // 1 when index "1" is the source side (sympartialACA.hpp:48-63)
    int32_t *st_q, *st_I1, *st_I2; // per block: state of a suspended block (zero = fresh start): iterations completed, next row pivot, last column pivot
    real *st_frob, *st_aux;
    int team_min, team_q;       // blocks with n1 + n2 >= team_min hand over to the team kernels (rank -3) once team_q iterations are done; team_min = 0: never
};

template <int NT>
__device__ __forceinline__ void block_argmax(real &val, int &idx, real *sval, int *sidx) {
    // maximum of |.|, ties -> larger index (the reference scans upward and replaces on ">=")
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const real ov = hmx_shfl_xor(val, o);
        const int oi    = hmx_shfl_xor(idx, o);
        if (ov > val || (ov == val && oi > idx)) {
            val = ov;
            idx = oi;
        }
    }
    const int w = threadIdx.x / WAVE;
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        sval[w] = val;
        sidx[w] = idx;
    }
    __syncthreads();
    val = sval[0];
    idx = sidx[0];
#pragma unroll
    for (int k = 1; k < NT / WAVE; k++)
        if (sval[k] > val || (sval[k] == val && sidx[k] > idx)) {
            val = sval[k];
            idx = sidx[k];
        }
    __syncthreads();
}

template <int NT, int G, typename V>
__device__ __forceinline__ void block_sum_group(V (&acc)[G], V *sbuf) {
#pragma unroll
    for (int g = 0; g < G; g++)
        acc[g] = wave_sum_any(acc[g]);
    const int w = threadIdx.x / WAVE;
    if ((threadIdx.x & (WAVE - 1)) == 0)
#pragma unroll
        for (int g = 0; g < G; g++)
            sbuf[w * G + g] = acc[g];
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; g++) {
        V s = sbuf[g];
#pragma unroll
        for (int k = 1; k < NT / WAVE; k++)
            s += sbuf[k * G + g];
        acc[g] = s;
    }
    __syncthreads();
}

// One line of a cross (partialACA.hpp:93-99 the row, :112-118 the column): out[k] = A(line, k) - sum_j hist_j[coef_index] * hist_j[line_base + k],
// j in history order, (x gamma for the column), and the pivot search over the entries not visited yet.  Every iteration of the ACA walks
// the whole history of the block, so this is where a high-rank block spends its time: the coefficients and pool offsets of ACA_JT crosses
// are staged in LDS, a thread keeps KR entries of the line in registers and the loop over the history has KR independent, unconditional
// loads per cross (the index is clamped instead of predicated: no branch inside the loop, loads of several crosses stay in flight).
// Histories longer than ACA_JT are applied tile after tile with the partial line parked in `out` (same sums, same order).
constexpr int ACA_JT = 128;
template <int NT, int KR, typename F>
__device__ __forceinline__ void aca_cross_line(int k_lo, int n, int nq, const int64_t *cross, const scalar *pool, int64_t coef_index, int64_t line_base, scalar *out, F eval,
                                               bool scale, scalar gamma, const unsigned char *vis, int skip, real &best, int &besti, scalar *s_coef, int64_t *s_offs) {
    const int tid = threadIdx.x;
    best          = -1;
    besti         = -1;
    int j0        = 0;
    do {
        const int tile = (nq - j0) < ACA_JT ? (nq - j0) : ACA_JT;
        if (tid < tile) {
            const int64_t o = cross[j0 + tid];
            s_offs[tid]     = o + line_base;
            s_coef[tid]     = -pool[o + coef_index];
        }
        __syncthreads();
        const bool first = j0 == 0, last = j0 + tile >= nq;
        for (int k0 = k_lo; k0 < n; k0 += KR * NT) { // the entries [k_lo, n) of the line
            scalar v[KR];
            int kk[KR];
#pragma unroll
            for (int r = 0; r < KR; r++) {
                const int k = k0 + r * NT + tid;
                kk[r]       = k < n ? k : n - 1;
                v[r]        = first ? eval(kk[r]) : out[kk[r]];
            }
            auto apply = [&](int jj) {
                const scalar coef = s_coef[jj];
                const scalar *cj  = pool + s_offs[jj];
#pragma unroll
                for (int r = 0; r < KR; r++)
                    v[r] = coef * cj[kk[r]] + v[r];
            };
            if (KR == 1) { // 16 loads of the history in flight per thread on either path
#pragma unroll 16
                for (int jj = 0; jj < tile; jj++)
                    apply(jj);
            } else {
#pragma unroll 4
                for (int jj = 0; jj < tile; jj++)
                    apply(jj);
            }
#pragma unroll
            for (int r = 0; r < KR; r++) {
                const int k = k0 + r * NT + tid;
                if (k < n) {
                    if (last) {
                        if (scale)
                            v[r] = v[r] * gamma;
                        out[k] = v[r];
                        if (!vis[k] && k != skip) {
                            const real a = hmx_abs(v[r]);
                            if (a >= best) { // k increases per thread: ">=" keeps the last maximum
                                best  = a;
                                besti = k;
                            }
                        }
                    } else {
                        out[k] = v[r];
                    }
                }
            }
        }
        __syncthreads();
        j0 += tile;
    } while (j0 < nq);
}

// Error estimator, the sums of partialACA.hpp:141-147 for four crosses j0..j0+nj-1 of the history at once: acc[2g] = vv_j . r (over index 2),
// acc[2g+1] = uu_j . c (over index 1), per-thread partial sums (k increasing); the new cross is loaded once, four independent streams.
template <int NT>
__device__ __forceinline__ void aca_dots4(const scalar *pool, const int64_t *cross, int j0, int nj, int n1, int n2, const scalar *u1, const scalar *u2, scalar (&acc)[8]) {
    const int tid    = threadIdx.x;
    const scalar *c0 = pool + cross[j0], *c1 = pool + cross[j0 + (nj > 1 ? 1 : 0)], *c2 = pool + cross[j0 + (nj > 2 ? 2 : 0)], *c3 = pool + cross[j0 + (nj > 3 ? 3 : 0)];
    scalar a10 = 0, a11 = 0, a12 = 0, a13 = 0, a20 = 0, a21 = 0, a22 = 0, a23 = 0;
    if (nj == 4) {
#pragma unroll 2
        for (int k = tid; k < n2; k += NT) {
            const scalar x = u1[k];
            a10 += hmx_conj(c0[n1 + k]) * x;
            a11 += hmx_conj(c1[n1 + k]) * x;
            a12 += hmx_conj(c2[n1 + k]) * x;
            a13 += hmx_conj(c3[n1 + k]) * x;
        }
#pragma unroll 2
        for (int k = tid; k < n1; k += NT) {
            const scalar x = u2[k];
            a20 += hmx_conj(c0[k]) * x;
            a21 += hmx_conj(c1[k]) * x;
            a22 += hmx_conj(c2[k]) * x;
            a23 += hmx_conj(c3[k]) * x;
        }
    } else {
        for (int k = tid; k < n2; k += NT) {
            const scalar x = u1[k];
            a10 += hmx_conj(c0[n1 + k]) * x;
            if (nj > 1)
                a11 += hmx_conj(c1[n1 + k]) * x;
            if (nj > 2)
                a12 += hmx_conj(c2[n1 + k]) * x;
        }
        for (int k = tid; k < n1; k += NT) {
            const scalar x = u2[k];
            a20 += hmx_conj(c0[k]) * x;
            if (nj > 1)
                a21 += hmx_conj(c1[k]) * x;
            if (nj > 2)
                a22 += hmx_conj(c2[k]) * x;
        }
    }
    acc[0] = a10; acc[1] = a20; acc[2] = a11; acc[3] = a21; acc[4] = a12; acc[5] = a22; acc[6] = a13; acc[7] = a23;
}

// partialACA::copy_low_rank_approximation (hmatrix/lrmat/partialACA.hpp:42-184) and
// sympartialACA (hmatrix/lrmat/sympartialACA.hpp:41-216) share this kernel: index "1" is the
// row side unless symmetric pivoting asks for the larger-offset side.
// A block that finds the pool exhausted SUSPENDS: it records (q, I1, frob, aux) in A.st_* and reports rank -2; the host grows the pool and
// launches the suspended blocks again, which continue with their next iteration (crosses and visited flags are in global memory already).
#undef HMX_ACA_OCCUPANCY
#if HMX_COMPLEX
#define HMX_ACA_OCCUPANCY
#else
#ifndef HMX_ACA_WAVES_EU
#define HMX_ACA_WAVES_EU 4 // (the fp64 kernel wants 132 registers: 12 bytes of scratch at 4; with 3 and no scratch the N = 1e6 build is no faster, 87-110 against 79-100 ms)
#endif
#define HMX_ACA_OCCUPANCY __attribute__((amdgpu_waves_per_eu(HMX_ACA_WAVES_EU))) // <= 128 registers: the many small blocks want workgroups in flight, not loads
#endif
template <int NT>
__global__ __launch_bounds__(NT) HMX_ACA_OCCUPANCY void aca_kernel(AcaArgs A) {
    __shared__ real sval[NT / WAVE];
    __shared__ int sidx[NT / WAVE];
    __shared__ scalar sbuf[(NT / WAVE) * 8];
    __shared__ scalar s_coef[ACA_JT];
    __shared__ int64_t s_offs[ACA_JT];
    __shared__ unsigned long long s_off;

    const int b      = A.order[blockIdx.x];
    const int M      = A.t_size[b], N = A.s_size[b];
    const int roff   = A.t_off[b], coff = A.s_off[b];
    const bool swap  = A.symmetric_pivoting && !(roff >= coff);
    const int n1     = swap ? N : M, n2 = swap ? M : N;
    // coordinates of index-1 points (p1*) and index-2 points (p2*)
    const double *p1x = swap ? A.sx + coff : A.tx + roff, *p1y = swap ? A.sy + coff : A.ty + roff, *p1z = swap ? A.sz + coff : A.tz + roff;
    const double *p2x = swap ? A.tx + roff : A.sx + coff, *p2y = swap ? A.ty + roff : A.sy + coff, *p2z = swap ? A.tz + roff : A.sz + coff;
    unsigned char *vis1 = A.visited + A.vis_ptr[b];
    unsigned char *vis2 = vis1 + n1;
    int64_t *cross      = A.cross_off + A.colptr[b];
    const int cap       = A.colcap[b];
    const int tid       = threadIdx.x;

    int I1 = A.st_I1[b], I2 = A.st_I2[b], q = A.st_q[b];
    real frob = A.st_frob[b], aux = A.st_aux[b];
    const int reqrank = A.reqrank;
    const int minmn   = n1 < n2 ? n1 : n2;
    while (((reqrank > 0) && (q < (reqrank < minmn ? reqrank : minmn))) || ((reqrank < 0) && (q == 0 || sqrt(aux / frob) > (real)A.epsilon))) {
        auto suspend = [&](int completed) { // before the next iteration has changed anything
            if (tid == 0) {
                A.st_q[b]    = completed;
                A.st_I1[b]   = I1;
                A.st_I2[b]   = I2;
                A.st_frob[b] = frob;
                A.st_aux[b]  = aux;
            }
        };
        if (A.team_min > 0 && n1 + n2 >= A.team_min && q >= A.team_q) { // a large block whose rank keeps growing: several workgroups take over
            suspend(q);
            q = -3;
            break;
        }
        q += 1;
        if ((long long)q * ((long long)n1 + n2) > (long long)n1 * n2 || q > cap) { // not advantageous any more
            q = -1;
            break;
        }
        if (tid == 0)
            s_off = atomicAdd(A.pool_head, (unsigned long long)(n1 + n2));
        __syncthreads();
        const unsigned long long off = s_off;
        if (off + (unsigned long long)(n1 + n2) > A.pool_cap) {
            suspend(q - 1);
            q = -2;
            break;
        }
        scalar *u2 = A.pool + off;      // new uu (length n1)
        scalar *u1 = A.pool + off + n1; // new vv (length n2)
        // ---- cross row: entries (I1, k), k over index 2 ------------------------------------------
        const double ax = p1x[I1], ay = p1y[I1], az = p1z[I1];
        real best;
        int besti;
        auto row_entry = [&](int k) { return swap ? eval_scalar(A.ks, p2x[k], p2y[k], p2z[k], ax, ay, az) : eval_scalar(A.ks, ax, ay, az, p2x[k], p2y[k], p2z[k]); };
        if (n2 <= NT)
            aca_cross_line<NT, 1>(0, n2, q - 1, cross, A.pool, I1, n1, u1, row_entry, false, scalar(1), vis2, -1, best, besti, s_coef, s_offs);
        else
            aca_cross_line<NT, 4>(0, n2, q - 1, cross, A.pool, I1, n1, u1, row_entry, false, scalar(1), vis2, -1, best, besti, s_coef, s_offs);
        block_argmax<NT>(best, besti, sval, sidx); // also makes u1 visible to the whole workgroup
        if (besti >= 0)
            I2 = besti;
        if (tid == 0)
            vis1[I1] = 1;
        const scalar piv   = u1[I2];
        const scalar gamma = scalar(1) / piv;
        if (hmx_abs(piv) > 1e-15) {
            // ---- cross column: entries (k, I2), k over index 1 -----------------------------------
            const double bx = p2x[I2], by = p2y[I2], bz = p2z[I2];
            auto col_entry = [&](int k) { return swap ? eval_scalar(A.ks, bx, by, bz, p1x[k], p1y[k], p1z[k]) : eval_scalar(A.ks, p1x[k], p1y[k], p1z[k], bx, by, bz); };
            if (n1 <= NT)
                aca_cross_line<NT, 1>(0, n1, q - 1, cross, A.pool, (int64_t)n1 + I2, 0, u2, col_entry, true, gamma, vis1, I1, best, besti, s_coef, s_offs);
            else
                aca_cross_line<NT, 4>(0, n1, q - 1, cross, A.pool, (int64_t)n1 + I2, 0, u2, col_entry, true, gamma, vis1, I1, best, besti, s_coef, s_offs);
            block_argmax<NT>(best, besti, sval, sidx);
            const int nextI1 = besti >= 0 ? besti : I1;
            if (tid == 0) {
                vis2[I2]     = 1;
                cross[q - 1] = (int64_t)off;
            }
            if (reqrank < 0) {
                // error estimator (partialACA.hpp:136-148): |c.c||r.r| + 2 sum_j (vv_j.r)(uu_j.c)
                scalar acc2[2] = {scalar(0), scalar(0)};
                for (int k = tid; k < n1; k += NT)
                    acc2[0] += hmx_conj(u2[k]) * u2[k];
                for (int k = tid; k < n2; k += NT)
                    acc2[1] += hmx_conj(u1[k]) * u1[k];
                block_sum_group<NT, 2>(acc2, sbuf);
                aux             = hmx_abs(acc2[0]) * hmx_abs(acc2[1]);
                scalar frob_aux = 0;
                for (int j0 = 0; j0 < q - 1; j0 += 4) {
                    const int nj = (q - 1 - j0) < 4 ? (q - 1 - j0) : 4;
                    scalar acc[8];
                    aca_dots4<NT>(A.pool, cross, j0, nj, n1, n2, u1, u2, acc);
                    block_sum_group<NT, 8>(acc, sbuf);
                    frob_aux += acc[0] * acc[1];
                    if (nj > 1)
                        frob_aux += acc[2] * acc[3];
                    if (nj > 2)
                        frob_aux += acc[4] * acc[5];
                    if (nj > 3)
                        frob_aux += acc[6] * acc[7];
                }
                frob += aux + 2 * hmx_re(frob_aux);
            }
            __syncthreads();
            I1 = nextI1;
        } else {
            q -= 1;
            if (q == 0)
                q = -1;
            break;
        }
    }
    if (tid == 0) {
        A.rank_out[b]    = q > 0 ? q : (q <= -2 ? q : 0);
        A.swapped_out[b] = swap ? 1 : 0;
    }
}

// The same iteration for SMALL blocks (both sides <= 64 ACA_WAVE_KR points: most admissible blocks sit at the bottom levels of the block
// tree) by ONE WAVE per block, WAVES independent blocks per workgroup: no workgroup barrier, no LDS -- aca_kernel spends these blocks'
// iterations in some fifteen __syncthreads each, with four blocks per compute unit in flight where there is room for sixteen and more.
// Lane l keeps the entries l, l + 64, ... of the current cross (and their visited flags) in registers; the history's pool offsets and
// coefficients are fetched by the lanes (one cross each) and handed round by v_readlane.  Crosses, pivots, estimator and ranks are
// aca_kernel's bit for bit: every entry sums its history in the same order, and the estimator's sums are formed exactly as there --
// register r of the lanes is wave r of aca_kernel<256> (thread k holds entry k there too, blocks of <= 256 points per side), so
// wave_sum_any per register, then the registers in order, reproduces block_sum_group's order of additions.
// Pool space is taken ACA_WAVE_CHUNK crosses at a time: one returning atomic on the pool head per iteration of every small block would be
// more than the ~90 per microsecond a single address sustains (MI355X_MICROARCH.md, dequeue); what a block leaves unused of its last
// chunk stays a hole in the pool (cross_off holds every cross's own offset, nothing assumes they are adjacent).
constexpr int ACA_WAVE_KR    = 4; // largest instantiation: blocks of up to 256 points per side (KR = 1, 2, 4: one launch per size class)
constexpr int ACA_WAVE_CHUNK = 4;
template <int WAVES, int KR>
__global__ __launch_bounds__(WAVES *WAVE) void aca_wave_kernel(AcaArgs A, int nblocks) {
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pos  = blockIdx.x * WAVES + wv;
    if (pos >= nblocks)
        return; // (no workgroup barrier below: the waves are independent)
    const int b      = A.order[pos];
    const int M      = A.t_size[b], N = A.s_size[b];
    const int roff   = A.t_off[b], coff = A.s_off[b];
    const bool swap  = A.symmetric_pivoting && !(roff >= coff);
    const int n1     = swap ? N : M, n2 = swap ? M : N;
    const int nr1 = (n1 + WAVE - 1) / WAVE, nr2 = (n2 + WAVE - 1) / WAVE; // registers in use per lane (<= KR)
    const double *p1x = swap ? A.sx + coff : A.tx + roff, *p1y = swap ? A.sy + coff : A.ty + roff, *p1z = swap ? A.sz + coff : A.tz + roff;
    const double *p2x = swap ? A.tx + roff : A.sx + coff, *p2y = swap ? A.ty + roff : A.sy + coff, *p2z = swap ? A.tz + roff : A.sz + coff;
    unsigned char *vis1 = A.visited + A.vis_ptr[b];
    unsigned char *vis2 = vis1 + n1;
    int64_t *cross      = A.cross_off + A.colptr[b];
    const int cap       = A.colcap[b];
    bool seen1[KR], seen2[KR]; // visited flags of the lane's entries (kept in global memory too: the state of a suspended block)
#pragma unroll
    for (int r = 0; r < KR; r++) {
        const int k = lane + WAVE * r;
        seen1[r]    = k < n1 ? vis1[k] != 0 : true;
        seen2[r]    = k < n2 ? vis2[k] != 0 : true;
    }
    int I1 = A.st_I1[b], I2 = A.st_I2[b], q = A.st_q[b];
    real frob = A.st_frob[b], aux = A.st_aux[b];
    const int reqrank = A.reqrank;
    const int minmn   = n1 < n2 ? n1 : n2;
    unsigned long long chunk_off = 0;
    int granted = 0;
    auto bcast64 = [](int64_t v, int l) {
        const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
        return (int64_t)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    };
    // one line of a cross: v[r] = A(line, k) - sum_j hist_j[coef_index] * hist_j[line_base + k], j in history order (aca_cross_line)
    auto line = [&](int n, int nr, int nq, int64_t coef_index, int64_t line_base, auto eval, scalar(&v)[KR]) {
        int kk[KR];
#pragma unroll
        for (int r = 0; r < KR; r++) {
            const int k = lane + WAVE * r;
            kk[r]       = k < n ? k : n - 1;
            v[r]        = scalar(0);
            if (r < nr)
                v[r] = eval(kk[r]);
        }
        for (int j0 = 0; j0 < nq; j0 += WAVE) {
            const int tile = (nq - j0) < WAVE ? (nq - j0) : WAVE;
            int64_t o = 0;
            scalar cf = scalar(0);
            if (lane < tile) {
                o  = cross[j0 + lane];
                cf = -A.pool[o + coef_index];
                o += line_base;
            }
            // four crosses per step, their loads issued together (a clamped cross is loaded again and not applied); applied in history order
            for (int jj = 0; jj < tile; jj += 4) {
                scalar coef[4], h[4][KR];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int jc     = jj + u < tile ? jj + u : tile - 1;
                    coef[u]          = readlane_val(cf, jc);
                    const scalar *cj = A.pool + bcast64(o, jc);
#pragma unroll
                    for (int r = 0; r < KR; r++)
                        if (r < nr)
                            h[u][r] = cj[kk[r]];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (jj + u < tile)
#pragma unroll
                        for (int r = 0; r < KR; r++)
                            if (r < nr)
                                v[r] = coef[u] * h[u][r] + v[r];
            }
        }
    };
    // maximum of |.| over the entries not visited yet, ties -> larger index (block_argmax's rule: independent of the order of comparison)
    auto pivot = [&](const scalar(&v)[KR], const bool(&seen)[KR], int n, int skip) {
        real best = -1;
        int besti = -1;
#pragma unroll
        for (int r = 0; r < KR; r++) {
            const int k = lane + WAVE * r;
            if (k < n && !seen[r] && k != skip) {
                const real a = hmx_abs(v[r]);
                if (a >= best) {
                    best  = a;
                    besti = k;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const real ov = hmx_shfl_xor(best, o);
            const int oi  = hmx_shfl_xor(besti, o);
            if (ov > best || (ov == best && oi > besti)) {
                best  = ov;
                besti = oi;
            }
        }
        return besti;
    };
    auto entry_of = [&](const scalar(&v)[KR], int k) { // v's entry k, to every lane
        scalar e = v[0];
#pragma unroll
        for (int r = 1; r < KR; r++)
            e = (k >> 6) == r ? v[r] : e;
        return hmx_shfl(e, k & 63);
    };
    // sum over the entries k < n of term(r), as block_sum_group<256> adds them: per register (= wave of aca_kernel) the butterfly, then the registers in order
    auto sum_as_block = [&](auto term, int nr) {
        scalar s = wave_sum_any(term(0));
#pragma unroll
        for (int r = 1; r < KR; r++)
            if (r < nr)
                s += wave_sum_any(term(r));
        return s;
    };
    while (((reqrank > 0) && (q < (reqrank < minmn ? reqrank : minmn))) || ((reqrank < 0) && (q == 0 || sqrt(aux / frob) > (real)A.epsilon))) {
        auto suspend = [&](int completed) {
            if (lane == 0) {
                A.st_q[b]    = completed;
                A.st_I1[b]   = I1;
                A.st_I2[b]   = I2;
                A.st_frob[b] = frob;
                A.st_aux[b]  = aux;
            }
        };
        q += 1;
        if ((long long)q * ((long long)n1 + n2) > (long long)n1 * n2 || q > cap) { // not advantageous any more
            q = -1;
            break;
        }
        if (granted == 0) {
            unsigned long long got = 0;
            if (lane == 0)
                got = atomicAdd(A.pool_head, (unsigned long long)ACA_WAVE_CHUNK * (unsigned long long)(n1 + n2));
            chunk_off = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(got >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(got & 0xffffffffull));
            granted   = ACA_WAVE_CHUNK;
            if (chunk_off + (unsigned long long)ACA_WAVE_CHUNK * (unsigned long long)(n1 + n2) > A.pool_cap) {
                suspend(q - 1);
                q = -2;
                break;
            }
        }
        const unsigned long long off = chunk_off + (unsigned long long)(ACA_WAVE_CHUNK - granted) * (unsigned long long)(n1 + n2);
        granted--;
        scalar *u2 = A.pool + off;      // new uu (length n1)
        scalar *u1 = A.pool + off + n1; // new vv (length n2)
        // ---- cross row: entries (I1, k), k over index 2 ------------------------------------------
        const double ax = p1x[I1], ay = p1y[I1], az = p1z[I1];
        auto row_entry = [&](int k) { return swap ? eval_scalar(A.ks, p2x[k], p2y[k], p2z[k], ax, ay, az) : eval_scalar(A.ks, ax, ay, az, p2x[k], p2y[k], p2z[k]); };
        scalar r1[KR], r2[KR];
        line(n2, nr2, q - 1, I1, n1, row_entry, r1);
#pragma unroll
        for (int r = 0; r < KR; r++)
            if (lane + WAVE * r < n2)
                u1[lane + WAVE * r] = r1[r];
        const int besti2 = pivot(r1, seen2, n2, -1);
        if (besti2 >= 0)
            I2 = besti2;
#pragma unroll
        for (int r = 0; r < KR; r++)
            seen1[r] = seen1[r] || lane + WAVE * r == I1;
        if (lane == 0)
            vis1[I1] = 1;
        const scalar piv   = entry_of(r1, I2);
        const scalar gamma = scalar(1) / piv;
        if (hmx_abs(piv) > 1e-15) {
            // ---- cross column: entries (k, I2), k over index 1 -----------------------------------
            const double bx = p2x[I2], by = p2y[I2], bz = p2z[I2];
            auto col_entry = [&](int k) { return swap ? eval_scalar(A.ks, bx, by, bz, p1x[k], p1y[k], p1z[k]) : eval_scalar(A.ks, p1x[k], p1y[k], p1z[k], bx, by, bz); };
            line(n1, nr1, q - 1, (int64_t)n1 + I2, 0, col_entry, r2);
#pragma unroll
            for (int r = 0; r < KR; r++) {
                r2[r] = r2[r] * gamma;
                if (lane + WAVE * r < n1)
                    u2[lane + WAVE * r] = r2[r];
            }
            const int besti1 = pivot(r2, seen1, n1, I1);
            const int nextI1 = besti1 >= 0 ? besti1 : I1;
#pragma unroll
            for (int r = 0; r < KR; r++)
                seen2[r] = seen2[r] || lane + WAVE * r == I2;
            if (lane == 0) {
                vis2[I2]     = 1;
                cross[q - 1] = (int64_t)off;
            }
            if (reqrank < 0) {
                // error estimator (partialACA.hpp:136-148): |c.c||r.r| + 2 sum_j (vv_j.r)(uu_j.c)
                const scalar cc = sum_as_block([&](int r) { scalar a = scalar(0); if (lane + WAVE * r < n1) a += hmx_conj(r2[r]) * r2[r]; return a; }, nr1);
                const scalar rr = sum_as_block([&](int r) { scalar a = scalar(0); if (lane + WAVE * r < n2) a += hmx_conj(r1[r]) * r1[r]; return a; }, nr2);
                aux             = hmx_abs(cc) * hmx_abs(rr);
                scalar frob_aux = 0;
                for (int j0 = 0; j0 < q - 1; j0 += WAVE) {
                    const int tile  = (q - 1 - j0) < WAVE ? (q - 1 - j0) : WAVE;
                    const int64_t o = lane < tile ? cross[j0 + lane] : 0;
                    for (int jj = 0; jj < tile; jj++) {
                        const scalar *cj = A.pool + bcast64(o, jj);
                        scalar h1[KR], h2[KR]; // the loads first: 2 KR independent streams
#pragma unroll
                        for (int r = 0; r < KR; r++) {
                            const int k = lane + WAVE * r;
                            h2[r] = (r < nr2 && k < n2) ? cj[n1 + k] : scalar(0);
                            h1[r] = (r < nr1 && k < n1) ? cj[k] : scalar(0);
                        }
                        const scalar d2 = sum_as_block([&](int r) { scalar a = scalar(0); if (lane + WAVE * r < n2) a += hmx_conj(h2[r]) * r1[r]; return a; }, nr2);
                        const scalar d1 = sum_as_block([&](int r) { scalar a = scalar(0); if (lane + WAVE * r < n1) a += hmx_conj(h1[r]) * r2[r]; return a; }, nr1);
                        frob_aux += d2 * d1;
                    }
                }
                frob += aux + 2 * hmx_re(frob_aux);
            }
            // the crosses written above are read back (other lanes, later iterations) through this CU's own L1 / L2: stores complete first
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            I1 = nextI1;
        } else {
            q -= 1;
            if (q == 0)
                q = -1;
            break;
        }
    }
    if (lane == 0) {
        A.rank_out[b]    = q > 0 ? q : (q <= -2 ? q : 0);
        A.swapped_out[b] = swap ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// The same iteration for a TEAM of G workgroups per block.  One workgroup streams some tens of GB/s: a 15 625 x 15 625 block of rank 476
// walks 113 GB of its own history, seconds on one compute unit while the rest of the GPU has long finished.  So large blocks leave
// aca_kernel after team_q iterations (rank -3, state in st_*) and continue here, three launches per iteration over all such blocks:
//   aca_team_control_kernel  error estimator of the iteration just finished (the history crosses are dealt out to the workgroups, each sum is
//                            complete in one workgroup exactly as in aca_kernel), stopping test, pool grant for the next iteration
//   aca_team_row_kernel      the cross row, the workgroups share the entries; column pivot
//   aca_team_col_kernel      the cross column; next row pivot
// Row and column entries and the pivots are those of aca_kernel bit for bit (every entry sums its history in the same order, the pivot
// rule is order independent); only the final sum of the estimator's products runs over per-workgroup partial sums.  The last workgroup
// of a team to arrive (atomic counter, no spinning, so no co-residency is needed) does the team's scalar work.
// ---------------------------------------------------------------------------------------------
struct AcaTeamArgs {
    AcaArgs A;
    const int32_t *wg_team;    // workgroup -> team
    const int32_t *team_block; // team -> block id
    const int32_t *team_wg0;   // team -> its first workgroup
    const int32_t *team_G;     // team -> workgroups
    int32_t *status;           // block: 0 active, 1 finished, 2 suspended (pool exhausted)
    int32_t *need_dots;        // block: an iteration has completed whose estimator is due
    scalar *gamma;             // block: 1 / pivot of the iteration in progress
    unsigned long long *off;   // block: pool grant of the iteration in progress (of the last one while need_dots)
    unsigned int *counter;     // block: arrivals
    real *pval;                // workgroup: partial pivot search
    int32_t *pidx;
    scalar *pfrob;             // workgroup: partial sum of the estimator's products
    real *paux;                // block: |c.c||r.r|
};

struct AcaTeamBlock { // what every team kernel derives from its workgroup index
    int t, b, g, G, n1, n2;
    bool swap;
    const double *p1x, *p1y, *p1z, *p2x, *p2y, *p2z;
    unsigned char *vis1, *vis2;
    int64_t *cross;
};
__device__ __forceinline__ bool aca_team_setup(const AcaTeamArgs &T, AcaTeamBlock &B) {
    const AcaArgs &A = T.A;
    B.t              = T.wg_team[blockIdx.x];
    B.b              = T.team_block[B.t];
    if (T.status[B.b] != 0)
        return false;
    B.g            = (int)blockIdx.x - T.team_wg0[B.t];
    B.G            = T.team_G[B.t];
    const int b    = B.b;
    const int M    = A.t_size[b], N = A.s_size[b];
    const int roff = A.t_off[b], coff = A.s_off[b];
    B.swap         = A.symmetric_pivoting && !(roff >= coff);
    B.n1           = B.swap ? N : M;
    B.n2           = B.swap ? M : N;
    B.p1x = B.swap ? A.sx + coff : A.tx + roff; B.p1y = B.swap ? A.sy + coff : A.ty + roff; B.p1z = B.swap ? A.sz + coff : A.tz + roff;
    B.p2x = B.swap ? A.tx + roff : A.sx + coff; B.p2y = B.swap ? A.ty + roff : A.sy + coff; B.p2z = B.swap ? A.tz + roff : A.sz + coff;
    B.vis1  = A.visited + A.vis_ptr[b];
    B.vis2  = B.vis1 + B.n1;
    B.cross = A.cross_off + A.colptr[b];
    return true;
}
// true in the workgroup that arrives last: everything the other workgroups of the team wrote before arriving is visible to it
__device__ __forceinline__ bool aca_team_arrive(unsigned int *counter, int G, int *s_last) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
        *s_last = atomicAdd(counter, 1u) == (unsigned)(G - 1);
    __syncthreads();
    const bool last = *s_last != 0;
    if (last)
        __threadfence();
    return last;
}
// the share [lo, hi) of workgroup g in a line of n entries (whole wavefronts)
__device__ __forceinline__ void aca_team_share(int n, int g, int G, int &lo, int &hi) {
    const int per = (((n + G - 1) / G) + WAVE - 1) / WAVE * WAVE;
    lo            = g * per < n ? g * per : n;
    hi            = lo + per < n ? lo + per : n;
}

template <int NT>
__global__ __launch_bounds__(NT) void aca_team_control_kernel(AcaTeamArgs T) {
    __shared__ scalar sbuf[(NT / WAVE) * 8];
    __shared__ int s_last;
    AcaTeamBlock B;
    if (!aca_team_setup(T, B))
        return;
    const AcaArgs &A = T.A;
    const int tid = threadIdx.x, n1 = B.n1, n2 = B.n2, b = B.b;
    const int q   = A.st_q[b]; // iterations completed
    const bool dots = T.need_dots[B.b] != 0;
    if (dots) {
        const scalar *u2 = A.pool + T.off[B.b], *u1 = u2 + n1;
        if (B.g == 0) {
            scalar acc2[2] = {scalar(0), scalar(0)};
            for (int k = tid; k < n1; k += NT)
                acc2[0] += hmx_conj(u2[k]) * u2[k];
            for (int k = tid; k < n2; k += NT)
                acc2[1] += hmx_conj(u1[k]) * u1[k];
            block_sum_group<NT, 2>(acc2, sbuf);
            if (tid == 0)
                T.paux[B.b] = hmx_abs(acc2[0]) * hmx_abs(acc2[1]);
        }
        scalar part = 0;
        for (int j0 = 4 * B.g; j0 < q - 1; j0 += 4 * B.G) { // groups of four history crosses, dealt out round robin
            const int nj = (q - 1 - j0) < 4 ? (q - 1 - j0) : 4;
            scalar acc[8];
            aca_dots4<NT>(A.pool, B.cross, j0, nj, n1, n2, u1, u2, acc);
            block_sum_group<NT, 8>(acc, sbuf);
            part += acc[0] * acc[1];
            if (nj > 1)
                part += acc[2] * acc[3];
            if (nj > 2)
                part += acc[4] * acc[5];
            if (nj > 3)
                part += acc[6] * acc[7];
        }
        if (tid == 0)
            T.pfrob[blockIdx.x] = part;
    }
    if (!aca_team_arrive(T.counter + B.b, B.G, &s_last) || tid != 0)
        return;
    // ---- the team's scalar work: estimator, stopping test (partialACA.hpp:78-84), grant for the next iteration --------------------------
    real frob = A.st_frob[b], aux = A.st_aux[b];
    if (dots) {
        scalar frob_aux = 0;
        for (int g = 0; g < B.G; g++)
            frob_aux += T.pfrob[T.team_wg0[B.t] + g];
        aux = T.paux[B.b];
        frob += aux + 2 * hmx_re(frob_aux);
        A.st_frob[b] = frob;
        A.st_aux[b]  = aux;
    }
    T.need_dots[B.b] = 0;
    T.counter[B.b]   = 0;
    auto finish = [&](int rank) {
        A.rank_out[b]    = rank;
        A.swapped_out[b] = B.swap ? 1 : 0;
        T.status[B.b]    = rank == -2 ? 2 : 1;
    };
    if (!(q == 0 || sqrt(aux / frob) > (real)A.epsilon)) {
        finish(q);
        return;
    }
    const int qn = q + 1;
    if ((long long)qn * ((long long)n1 + n2) > (long long)n1 * n2 || qn > A.colcap[b]) { // not advantageous any more: the compressor fails
        finish(0);
        return;
    }
    const unsigned long long off = atomicAdd(A.pool_head, (unsigned long long)(n1 + n2));
    if (off + (unsigned long long)(n1 + n2) > A.pool_cap) {
        finish(-2);
        return;
    }
    T.off[B.b] = off;
}

template <int NT>
__global__ __launch_bounds__(NT) void aca_team_row_kernel(AcaTeamArgs T) {
    __shared__ real sval[NT / WAVE];
    __shared__ int sidx[NT / WAVE];
    __shared__ scalar s_coef[ACA_JT];
    __shared__ int64_t s_offs[ACA_JT];
    __shared__ int s_last;
    AcaTeamBlock B;
    if (!aca_team_setup(T, B))
        return;
    const AcaArgs &A = T.A;
    const int tid = threadIdx.x, n1 = B.n1, n2 = B.n2, b = B.b;
    const int q   = A.st_q[b] + 1; // the iteration in progress
    const int I1  = A.st_I1[b];
    scalar *u1    = A.pool + T.off[B.b] + n1;
    const double ax = B.p1x[I1], ay = B.p1y[I1], az = B.p1z[I1];
    auto row_entry = [&](int k) { return B.swap ? eval_scalar(A.ks, B.p2x[k], B.p2y[k], B.p2z[k], ax, ay, az) : eval_scalar(A.ks, ax, ay, az, B.p2x[k], B.p2y[k], B.p2z[k]); };
    int lo, hi;
    aca_team_share(n2, B.g, B.G, lo, hi);
    real best = -1;
    int besti = -1;
    if (lo < hi) {
        if (hi - lo <= NT)
            aca_cross_line<NT, 1>(lo, hi, q - 1, B.cross, A.pool, I1, n1, u1, row_entry, false, scalar(1), B.vis2, -1, best, besti, s_coef, s_offs);
        else
            aca_cross_line<NT, 4>(lo, hi, q - 1, B.cross, A.pool, I1, n1, u1, row_entry, false, scalar(1), B.vis2, -1, best, besti, s_coef, s_offs);
    }
    block_argmax<NT>(best, besti, sval, sidx);
    if (tid == 0) {
        T.pval[blockIdx.x] = best;
        T.pidx[blockIdx.x] = besti;
    }
    if (!aca_team_arrive(T.counter + B.b, B.G, &s_last))
        return;
    best  = tid < B.G ? T.pval[T.team_wg0[B.t] + tid] : (real)-1;
    besti = tid < B.G ? T.pidx[T.team_wg0[B.t] + tid] : -1;
    block_argmax<NT>(best, besti, sval, sidx);
    if (tid != 0)
        return;
    const int I2     = besti >= 0 ? besti : A.st_I2[b];
    const scalar piv = u1[I2];
    B.vis1[I1]       = 1;
    T.counter[B.b]   = 0;
    A.st_I2[b]       = I2;
    if (hmx_abs(piv) > 1e-15) {
        T.gamma[B.b] = scalar(1) / piv;
    } else { // zero row: the crosses found so far are the approximation (none: the compressor fails)
        A.rank_out[b]    = q - 1 > 0 ? q - 1 : 0;
        A.swapped_out[b] = B.swap ? 1 : 0;
        T.status[B.b]    = 1;
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void aca_team_col_kernel(AcaTeamArgs T) {
    __shared__ real sval[NT / WAVE];
    __shared__ int sidx[NT / WAVE];
    __shared__ scalar s_coef[ACA_JT];
    __shared__ int64_t s_offs[ACA_JT];
    __shared__ int s_last;
    AcaTeamBlock B;
    if (!aca_team_setup(T, B))
        return;
    const AcaArgs &A = T.A;
    const int tid = threadIdx.x, n1 = B.n1, b = B.b;
    const int q   = A.st_q[b] + 1;
    const int I1 = A.st_I1[b], I2 = A.st_I2[b];
    const unsigned long long off = T.off[B.b];
    scalar *u2         = A.pool + off;
    const scalar gamma = T.gamma[B.b];
    const double bx = B.p2x[I2], by = B.p2y[I2], bz = B.p2z[I2];
    auto col_entry = [&](int k) { return B.swap ? eval_scalar(A.ks, bx, by, bz, B.p1x[k], B.p1y[k], B.p1z[k]) : eval_scalar(A.ks, B.p1x[k], B.p1y[k], B.p1z[k], bx, by, bz); };
    int lo, hi;
    aca_team_share(n1, B.g, B.G, lo, hi);
    real best = -1;
    int besti = -1;
    if (lo < hi) {
        if (hi - lo <= NT)
            aca_cross_line<NT, 1>(lo, hi, q - 1, B.cross, A.pool, (int64_t)n1 + I2, 0, u2, col_entry, true, gamma, B.vis1, I1, best, besti, s_coef, s_offs);
        else
            aca_cross_line<NT, 4>(lo, hi, q - 1, B.cross, A.pool, (int64_t)n1 + I2, 0, u2, col_entry, true, gamma, B.vis1, I1, best, besti, s_coef, s_offs);
    }
    block_argmax<NT>(best, besti, sval, sidx);
    if (tid == 0) {
        T.pval[blockIdx.x] = best;
        T.pidx[blockIdx.x] = besti;
    }
    if (!aca_team_arrive(T.counter + B.b, B.G, &s_last))
        return;
    best  = tid < B.G ? T.pval[T.team_wg0[B.t] + tid] : (real)-1;
    besti = tid < B.G ? T.pidx[T.team_wg0[B.t] + tid] : -1;
    block_argmax<NT>(best, besti, sval, sidx);
    if (tid != 0)
        return;
    B.vis2[I2]       = 1;
    B.cross[q - 1]   = (int64_t)off;
    A.st_I1[b]       = besti >= 0 ? besti : I1;
    A.st_q[b]        = q;
    T.need_dots[B.b] = 1;
    T.counter[B.b]   = 0;
}

// ---------------------------------------------------------------------------------------------
// Partially pivoted ACA for a HOST generator (the user's VirtualGenerator::copy_submatrix, a C callback):
// the same algorithm as aca_kernel, run in lock step over a BATCH of admissible blocks (the host runs many batches concurrently, one
// or two per generator thread, each on its own stream: engine_body.hpp, "host generator on all cores").  Per iteration the host
// evaluates one cross row per active block (callback), aca_cb_row_kernel subtracts the previous crosses and
// picks the column pivot; the host evaluates those columns, aca_cb_col_kernel finishes the iteration (scaling,
// row pivot, error estimator, stopping test).  All arithmetic except the generator itself stays on the device.
// A launch covers the batch's active blocks; position p of the launch reads items[p] and leaves res[p] (one packed copy back).
// ---------------------------------------------------------------------------------------------
struct AcaCbArgs {
    const CbItem *items;   // per launch position: block id, first entry of its line in buf
    CbResult *res;         // per launch position: what the host needs for the next phase
    const int32_t *t_off, *t_size, *s_off, *s_size;
    int symmetric_pivoting;
    double epsilon;
    int reqrank;
    scalar *pool;
    unsigned long long *pool_head;
    unsigned long long pool_cap;
    const int64_t *colptr;
    const int32_t *colcap;
    int64_t *cross_off;
    unsigned char *visited;
    const int64_t *vis_ptr;
    // per-block state carried between launches
    int32_t *I1, *I2, *q;
    real *frob, *aux;
    scalar *gamma;
    unsigned long long *cur_off;
    const scalar *buf;        // host-evaluated entries of this phase, packed
    int32_t *rank_out, *swapped_out;
};

template <int NT>
__global__ __launch_bounds__(NT) void aca_cb_row_kernel(AcaCbArgs A) {
    __shared__ real sval[NT / WAVE];
    __shared__ int sidx[NT / WAVE];
    __shared__ unsigned long long s_off;
    const CbItem item = A.items[blockIdx.x];
    const int b = item.block;
    const int M = A.t_size[b], N = A.s_size[b];
    const bool swap = A.symmetric_pivoting && !(A.t_off[b] >= A.s_off[b]);
    const int n1 = swap ? N : M, n2 = swap ? M : N;
    unsigned char *vis1 = A.visited + A.vis_ptr[b], *vis2 = vis1 + n1;
    int64_t *cross = A.cross_off + A.colptr[b];
    const int tid  = threadIdx.x;
    int q          = A.q[b] + 1;
    const int I1   = A.I1[b];
    auto finish = [&](int rank) {
        if (tid == 0) {
            A.res[blockIdx.x] = CbResult{rank == -2 ? CB_SUSPENDED : CB_FINISHED, I1, 0, 0};
            A.rank_out[b]     = rank;
            A.swapped_out[b]  = swap ? 1 : 0;
        }
    };
    if ((long long)q * ((long long)n1 + n2) > (long long)n1 * n2 || q > A.colcap[b]) {
        finish(0); // not advantageous: the compressor reports failure, dense fallback
        return;
    }
    if (tid == 0)
        s_off = atomicAdd(A.pool_head, (unsigned long long)(n1 + n2));
    __syncthreads();
    const unsigned long long off = s_off;
    if (off + (unsigned long long)(n1 + n2) > A.pool_cap) {
        finish(-2); // nothing of the iteration has happened yet: the block continues from this row once the pool has grown
        return;
    }
    scalar *u1       = A.pool + off + n1;
    const scalar *in = A.buf + item.off;
    real best = -1;
    int besti = -1;
    for (int k = tid; k < n2; k += NT) {
        scalar v = in[k];
        for (int j = 0; j < q - 1; j++) {
            const scalar *cj  = A.pool + cross[j];
            const scalar coef = -cj[I1];
            v               = coef * cj[n1 + k] + v;
        }
        u1[k] = v;
        if (!vis2[k]) {
            const real a = hmx_abs(v);
            if (a >= best) {
                best  = a;
                besti = k;
            }
        }
    }
    block_argmax<NT>(best, besti, sval, sidx);
    const int I2   = besti >= 0 ? besti : A.I2[b];
    const scalar piv = u1[I2];
    if (tid == 0)
        vis1[I1] = 1;
    if (hmx_abs(piv) > 1e-15) {
        if (tid == 0) {
            A.I2[b]      = I2;
            A.gamma[b]   = scalar(1) / piv;
            A.cur_off[b] = off;
            A.q[b]       = q; // provisional: the column phase completes iteration q
            A.res[blockIdx.x] = CbResult{CB_ACTIVE, I1, I2, 0};
        }
    } else { // zero row: rank q-1, or failure when nothing was accepted yet
        finish(q - 1 > 0 ? q - 1 : 0);
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void aca_cb_col_kernel(AcaCbArgs A) {
    __shared__ real sval[NT / WAVE];
    __shared__ int sidx[NT / WAVE];
    __shared__ scalar sbuf[(NT / WAVE) * 8];
    const CbItem item = A.items[blockIdx.x];
    const int b = item.block;
    const int M = A.t_size[b], N = A.s_size[b];
    const bool swap = A.symmetric_pivoting && !(A.t_off[b] >= A.s_off[b]);
    const int n1 = swap ? N : M, n2 = swap ? M : N;
    unsigned char *vis1 = A.visited + A.vis_ptr[b], *vis2 = vis1 + n1;
    int64_t *cross = A.cross_off + A.colptr[b];
    const int tid  = threadIdx.x;
    const int q = A.q[b], I1 = A.I1[b], I2 = A.I2[b];
    const scalar gamma = A.gamma[b];
    const unsigned long long off = A.cur_off[b];
    scalar *u2 = A.pool + off, *u1 = A.pool + off + n1;
    const scalar *in = A.buf + item.off;
    real best = -1;
    int besti = -1;
    for (int k = tid; k < n1; k += NT) {
        scalar v = in[k];
        for (int j = 0; j < q - 1; j++) {
            const scalar *cj  = A.pool + cross[j];
            const scalar coef = -cj[n1 + I2];
            v               = coef * cj[k] + v;
        }
        v     = v * gamma;
        u2[k] = v;
        if (!vis1[k] && k != I1) {
            const real a = hmx_abs(v);
            if (a >= best) {
                best  = a;
                besti = k;
            }
        }
    }
    block_argmax<NT>(best, besti, sval, sidx);
    real frob = A.frob[b], aux = A.aux[b];
    if (A.reqrank < 0) {
        scalar acc2[2] = {scalar(0), scalar(0)};
        for (int k = tid; k < n1; k += NT)
            acc2[0] += hmx_conj(u2[k]) * u2[k];
        for (int k = tid; k < n2; k += NT)
            acc2[1] += hmx_conj(u1[k]) * u1[k];
        block_sum_group<NT, 2>(acc2, sbuf);
        aux           = hmx_abs(acc2[0]) * hmx_abs(acc2[1]);
        scalar frob_aux = 0;
        for (int j0 = 0; j0 < q - 1; j0 += 4) { // the sums of aca_kernel, four history crosses at a time (aca_dots4: no indexed private array)
            const int nj = (q - 1 - j0) < 4 ? (q - 1 - j0) : 4;
            scalar acc[8];
            aca_dots4<NT>(A.pool, cross, j0, nj, n1, n2, u1, u2, acc);
            block_sum_group<NT, 8>(acc, sbuf);
            frob_aux += acc[0] * acc[1];
            if (nj > 1)
                frob_aux += acc[2] * acc[3];
            if (nj > 2)
                frob_aux += acc[4] * acc[5];
            if (nj > 3)
                frob_aux += acc[6] * acc[7];
        }
        frob += aux + 2 * hmx_re(frob_aux);
    }
    const int minmn = n1 < n2 ? n1 : n2;
    const bool more = (A.reqrank > 0) ? (q < (A.reqrank < minmn ? A.reqrank : minmn)) : (sqrt(aux / frob) > (real)A.epsilon);
    if (tid == 0) {
        const int nextI1 = besti >= 0 ? besti : I1;
        vis2[I2]     = 1;
        cross[q - 1] = (int64_t)off;
        A.I1[b]      = nextI1;
        A.frob[b]    = frob;
        A.aux[b]     = aux;
        A.res[blockIdx.x] = CbResult{more ? CB_ACTIVE : CB_FINISHED, nextI1, I2, 0};
        if (!more) {
            A.rank_out[b]    = q;
            A.swapped_out[b] = swap ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Compression of an ASSEMBLED block: fully pivoted ACA and truncated SVD (small blocks; O(M N min(M,N))).
// One workgroup per block, the block lives in a scratch slab; results are written as crosses
// [U(:,k) | V(k,:)] into the same pool the partial ACA uses, so packing is shared.
// ---------------------------------------------------------------------------------------------
struct DenseCompressArgs {
    KernelSpec ks;
    const double *tx, *ty, *tz, *sx, *sy, *sz;
    const int32_t *order; // launch order -> block id
    const int32_t *t_off, *t_size, *s_off, *s_size;
    const int64_t *scratch_off; // per block: first scalar of its slab in `scratch`
    scalar *scratch;
    double epsilon;
    int reqrank;
    scalar *pool;
    unsigned long long *pool_head;
    unsigned long long pool_cap;
    const int64_t *colptr;
    const int32_t *colcap;
    int64_t *cross_off;
    int32_t *rank_out;
    const scalar *pre;        // != NULL: blocks assembled by the host generator (column-major M x N) at pre_off[b]
    const int64_t *pre_off;
};

// fullACA::copy_low_rank_approximation (hmatrix/lrmat/fullACA.hpp:38-88)
template <int NT>
__global__ __launch_bounds__(NT) void fullaca_kernel(DenseCompressArgs A) {
    __shared__ real sval[NT / WAVE];
    __shared__ int sidx[NT / WAVE];
    __shared__ real sbuf[(NT / WAVE) * 2];
    __shared__ unsigned long long s_off;
    const int b = A.order[blockIdx.x];
    const int M = A.t_size[b], N = A.s_size[b], roff = A.t_off[b], coff = A.s_off[b];
    const int64_t MN = (int64_t)M * N;
    scalar *mat      = A.scratch + A.scratch_off[b];
    int64_t *cross   = A.cross_off + A.colptr[b];
    const int cap    = A.colcap[b];
    const int tid    = threadIdx.x;
    real acc1[1]     = {0};
    for (int64_t e = tid; e < MN; e += NT) {
        const int i = (int)(e % M), j = (int)(e / M);
        const scalar v = A.pre ? A.pre[A.pre_off[b] + e] : eval_scalar(A.ks, A.tx[roff + i], A.ty[roff + i], A.tz[roff + i], A.sx[coff + j], A.sy[coff + j], A.sz[coff + j]);
        mat[e]         = v;
        acc1[0] += hmx_abs2(v);
    }
    block_sum_group<NT, 1>(acc1, sbuf);
    const real Norm = sqrt(acc1[0]);
    real cur        = Norm; // Frobenius norm of the current residual
    int q             = 0;
    const int reqrank = A.reqrank;
    const int minmn   = M < N ? M : N;
    while (((reqrank > 0) && (q < (reqrank < minmn ? reqrank : minmn))) || ((reqrank < 0) && (cur / Norm > (real)A.epsilon || q == 0))) {
        q += 1;
        if ((long long)q * ((long long)M + N) > MN || q > cap) {
            q = -1;
            break;
        }
        // std::max_element over the column-major array: first maximum of |.| (matrix/utils/math.hpp:18-23)
        real best = -1;
        int64_t bi  = -1;
        for (int64_t e = tid; e < MN; e += NT) {
            const real a = hmx_abs(mat[e]);
            if (a > best) {
                best = a;
                bi   = e;
            }
        }
        // block reduction with "smaller index wins ties": reuse block_argmax on (value, -index)
        int neg = bi >= 0 ? (int)(-bi) : -2147483647; // MN < 2^31 is guaranteed by the caller
        block_argmax<NT>(best, neg, sval, sidx);
        const int64_t pe = -(int64_t)neg;
        const int pi = (int)(pe % M), pj = (int)(pe / M);
        const scalar pivot = mat[pe];
        if (hmx_abs(pivot) < 1e-15) {
            q += -1;
            break;
        }
        if (tid == 0)
            s_off = atomicAdd(A.pool_head, (unsigned long long)(M + N));
        __syncthreads();
        const unsigned long long off = s_off;
        if (off + (unsigned long long)(M + N) > A.pool_cap) {
            q = -2;
            break;
        }
        scalar *u = A.pool + off, *v = A.pool + off + M;
        for (int i = tid; i < M; i += NT)
            u[i] = mat[i + (int64_t)M * pj];
        for (int j = tid; j < N; j += NT)
            v[j] = mat[pi + (int64_t)M * j] / pivot;
        __syncthreads();
        acc1[0] = 0;
        for (int64_t e = tid; e < MN; e += NT) {
            const int i = (int)(e % M), j = (int)(e / M);
            const scalar r = mat[e] - u[i] * v[j];
            mat[e]         = r;
            acc1[0] += hmx_abs2(r);
        }
        block_sum_group<NT, 1>(acc1, sbuf);
        cur = sqrt(acc1[0]);
        if (tid == 0)
            cross[q - 1] = (int64_t)off;
    }
    if (tid == 0)
        A.rank_out[b] = q > 0 ? q : (q == -2 ? -2 : 0);
}

// Cyclic one-sided Jacobi on the columns of W (m x n, column-major): on return the columns are mutually orthogonal
// (W_out = W_in * Vm, Vm accumulates the rotations, must hold the identity on entry).  The pairs of one round-robin
// round touch disjoint columns, so each wave rotates one pair; a workgroup barrier separates the rounds.
template <int NT>
__device__ void jacobi_orthogonalize(scalar *W, int m, int n, scalar *Vm, int *s_changed_ptr) {
    int &s_changed = *s_changed_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = NT / WAVE;
    const int np = (n + 1) & ~1; // players (one dummy when n is odd)
    for (int sweep = 0; sweep < 60; sweep++) {
        if (tid == 0)
            s_changed = 0;
        __syncthreads();
        for (int round = 0; round < np - 1; round++) {
            for (int k = wv; k < np / 2; k += NW) {
                // circle method: position 0 is fixed, the others rotate
                int p = k == 0 ? 0 : 1 + (k - 1 + round) % (np - 1);
                int qq = 1 + (np - 1 - k - 1 + round) % (np - 1);
                if (p > qq) {
                    const int t = p;
                    p           = qq;
                    qq          = t;
                }
                if (qq >= n || p == qq)
                    continue;
                scalar *wp = W + (int64_t)m * p, *wq = W + (int64_t)m * qq;
                scalar *vp = Vm + (int64_t)n * p, *vq = Vm + (int64_t)n * qq;
#if HMX_COMPLEX
                // complex columns: a^H c = |apq| e^{i phi}; column q is first turned by e^{-i phi}, which makes the inner
                // product real and positive, then the real rotation applies
                real app = 0, aqq = 0;
                scalar apq = scalar(0);
                for (int i = lane; i < m; i += WAVE) {
                    const scalar a = wp[i], c = wq[i];
                    app += a.re * a.re + a.im * a.im;
                    aqq += c.re * c.re + c.im * c.im;
                    apq += hmx_conj(a) * c;
                }
                app = wave_sum_any(app);
                aqq = wave_sum_any(aqq);
                apq = wave_sum(apq);
                const real absq = hmx_abs(apq);
                if (absq <= 1e-300 || absq <= 1e-17 * sqrt(app * aqq))
                    continue;
                if (absq / sqrt(app * aqq) >= 1e-15 && lane == 0)
                    s_changed = 1;
                const scalar ph = hmx_conj(apq) / absq; // e^{-i phi}
                const real zeta = (aqq - app) / (2.0 * absq);
                const real t    = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const real cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = lane; i < m; i += WAVE) {
                    const scalar a = wp[i], c = wq[i] * ph;
                    wp[i]          = cs * a - sn * c;
                    wq[i]          = sn * a + cs * c;
                }
                for (int i = lane; i < n; i += WAVE) {
                    const scalar a = vp[i], c = vq[i] * ph;
                    vp[i]          = cs * a - sn * c;
                    vq[i]          = sn * a + cs * c;
                }
#else
                real app = 0, aqq = 0, apq = 0;
                for (int i = lane; i < m; i += WAVE) {
                    const real a = wp[i], c = wq[i];
                    app += a * a;
                    aqq += c * c;
                    apq += a * c;
                }
                app = wave_sum(app);
                aqq = wave_sum(aqq);
                apq = wave_sum(apq);
                if (fabs(apq) <= 1e-300 || fabs(apq) <= 1e-17 * sqrt(app * aqq))
                    continue;
                if (fabs(apq) / sqrt(app * aqq) >= 1e-15 && lane == 0)
                    s_changed = 1;
                const real zeta = (aqq - app) / (2.0 * apq);
                const real t    = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const real cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = lane; i < m; i += WAVE) {
                    const real a = wp[i], c = wq[i];
                    wp[i]          = cs * a - sn * c;
                    wq[i]          = sn * a + cs * c;
                }
                for (int i = lane; i < n; i += WAVE) {
                    const real a = vp[i], c = vq[i];
                    vp[i]          = cs * a - sn * c;
                    vq[i]          = sn * a + cs * c;
                }
#endif
            }
            __syncthreads();
        }
        const int changed = s_changed;
        __syncthreads();
        if (!changed)
            break;
    }
}

// SVD::copy_low_rank_approximation (hmatrix/lrmat/SVD.hpp:27-92) with gesvd replaced by a one-sided Jacobi
// SVD (LAPACK is a third-party dependency of the reference; its contract -- A = u diag(s) vt, s descending --
// is what is reproduced) and the truncation rule of matrix/utils/SVD_truncation.hpp:37-52.
// Slab layout: W (m x n, m >= n, column-major; A or A^T) | Vm (n x n) | sv (n) | order (n, as doubles)
template <int NT>
__global__ __launch_bounds__(NT) void svd_kernel(DenseCompressArgs A) {
    __shared__ int s_changed;
    __shared__ int s_rank;
    __shared__ unsigned long long s_off;
    const int b = A.order[blockIdx.x];
    const int M = A.t_size[b], N = A.s_size[b], roff = A.t_off[b], coff = A.s_off[b];
    const bool tr = M < N;
    const int m = tr ? N : M, n = tr ? M : N;
    scalar *W  = A.scratch + A.scratch_off[b];
    scalar *Vm = W + (int64_t)m * n;
    real *sv   = reinterpret_cast<real *>(Vm + (int64_t)n * n); // n singular values and n order slots live in the 2n scalars behind Vm
    real *ord  = sv + n;
    int64_t *cross = A.cross_off + A.colptr[b];
    const int cap  = A.colcap[b];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = NT / WAVE;
    for (int64_t e = tid; e < (int64_t)M * N; e += NT) {
        const int i = (int)(e % M), j = (int)(e / M);
        const scalar v = A.pre ? A.pre[A.pre_off[b] + e] : eval_scalar(A.ks, A.tx[roff + i], A.ty[roff + i], A.tz[roff + i], A.sx[coff + j], A.sy[coff + j], A.sz[coff + j]);
        if (tr)
            W[j + (int64_t)m * i] = v;
        else
            W[i + (int64_t)m * j] = v;
    }
    for (int64_t e = tid; e < (int64_t)n * n; e += NT)
        Vm[e] = (e % n == e / n) ? scalar(1) : scalar(0);
    __syncthreads();
    jacobi_orthogonalize<NT>(W, m, n, Vm, &s_changed);
    // singular values = column norms, descending order by counting
    for (int j = wv; j < n; j += NW) {
        real nn = 0;
        for (int i = lane; i < m; i += WAVE)
            nn += hmx_re(hmx_conj(W[i + (int64_t)m * j]) * W[i + (int64_t)m * j]);
        nn = wave_sum_any(nn);
        if (lane == 0)
            sv[j] = sqrt(nn);
    }
    __syncthreads();
    for (int j = tid; j < n; j += NT) {
        int pos = 0;
        for (int k = 0; k < n; k++)
            pos += (sv[k] > sv[j] || (sv[k] == sv[j] && k < j)) ? 1 : 0;
        ord[pos] = (real)j;
    }
    __syncthreads();
    if (tid == 0) {
        int r;
        if (A.reqrank > 0) {
            r = A.reqrank < n ? A.reqrank : n;
        } else { // SVD_truncation.hpp:37-52: smallest k whose discarded tail stays below epsilon
            real norm2 = 0, err = 0;
            for (int k = 0; k < n; k++)
                norm2 += sv[(int)ord[k]] * sv[(int)ord[k]];
            const real nrm = sqrt(norm2);
            int j = n;
            do {
                j = j - 1;
                err += sv[(int)ord[j]] * sv[(int)ord[j]];
            } while (j > 0 && sqrt(err) / nrm < (real)A.epsilon);
            r = j + 1;
            if ((long long)r * ((long long)M + N) > (long long)M * N || r <= 0)
                r = 0;
        }
        if (r > cap)
            r = 0;
        s_rank = r;
        if (r > 0)
            s_off = atomicAdd(A.pool_head, (unsigned long long)r * (unsigned long long)(M + N));
    }
    __syncthreads();
    const int r = s_rank;
    if (r > 0) {
        const unsigned long long off = s_off;
        if (off + (unsigned long long)r * (unsigned long long)(M + N) > A.pool_cap) {
            if (tid == 0)
                A.rank_out[b] = -2;
            return;
        }
        for (int k = 0; k < r; k++) {
            const int j     = (int)ord[k];
            const real sj = sv[j], isj = sj > 0 ? 1.0 / sj : 0.0;
            scalar *u = A.pool + off + (unsigned long long)k * (M + N), *v = u + M;
            if (!tr) { // A = W Vm^H: U(:,k) = u_k s_k = W(:,j), V(k,:) = Vm(:,j)^H
                for (int i = tid; i < M; i += NT)
                    u[i] = W[i + (int64_t)m * j];
                for (int c = tid; c < N; c += NT)
                    v[c] = hmx_conj(Vm[c + (int64_t)n * j]);
            } else { // A^T = W Vm^H  =>  A = conj(Vm) W^T: U(:,k) = conj(Vm(:,j)) s_j, V(k,:) = W(:,j)^T / s_j
                for (int i = tid; i < M; i += NT)
                    u[i] = hmx_conj(Vm[i + (int64_t)n * j]) * sj;
                for (int c = tid; c < N; c += NT)
                    v[c] = W[c + (int64_t)m * j] * isj;
            }
            if (tid == 0)
                cross[k] = (int64_t)(off + (unsigned long long)k * (M + N));
        }
    }
    if (tid == 0)
        A.rank_out[b] = r;
}

// SVD_recompression (hmatrix/lrmat/utils/SVD_recompression.hpp:19-181) of an existing U (M x r) * V (r x N):
// the reference does QR(U), LQ(V), SVD(R L) with LAPACK; here both thin factors are orthogonalised by one-sided
// Jacobi (U G_u = Q_u S_u, V^T G_v = Q_v S_v), the r x r core C = S_u G_u^T G_v S_v gets a Jacobi SVD, the rank is
// truncated with SVD_truncation's rule and the factors are rebuilt as U' = Q_u u sqrt(s), V' = sqrt(s) vt Q_v^T.
// As in the reference the block is only rewritten when the rank drops.
// Slab: Uw (M x r) | Vw (N x r) | Gu, Gv, Cm, Gc (r x r each) | su, sv, sc, ord (r each)
struct RecompressArgs {
    const int32_t *order;
    const int32_t *t_size, *s_size;
    const int32_t *swapped;
    const int64_t *scratch_off;
    scalar *scratch;
    double epsilon;
    scalar *pool;
    const int64_t *colptr;
    const int64_t *cross_off;
    int32_t *rank; // in: current rank, out: new rank
};
template <int NT>
__global__ __launch_bounds__(NT) void recompress_kernel(RecompressArgs A) {
    __shared__ int s_changed;
    __shared__ int s_rank;
    const int b = A.order[blockIdx.x];
    const int M = A.t_size[b], N = A.s_size[b], r = A.rank[b];
    const bool sw = A.swapped[b] != 0;
    const int n1  = sw ? N : M; // length of the first vector of a cross
    const int64_t *cross = A.cross_off + A.colptr[b];
    scalar *Uw = A.scratch + A.scratch_off[b];
    scalar *Vw = Uw + (int64_t)M * r;
    scalar *Gu = Vw + (int64_t)N * r, *Gv = Gu + r * r, *Cm = Gv + r * r, *Gc = Cm + r * r;
    real *su = reinterpret_cast<real *>(Gc + r * r), *sv = su + r, *sc = sv + r, *ord = sc + r; // 4r reals in the 4r scalars behind Gc
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = NT / WAVE;
    for (int k = 0; k < r; k++) {
        const scalar *c  = A.pool + cross[k];
        const scalar *uk = sw ? c + n1 : c, *vk = sw ? c : c + n1; // U(:,k), V(k,:)
        for (int i = tid; i < M; i += NT)
            Uw[i + (int64_t)M * k] = uk[i];
        for (int j = tid; j < N; j += NT)
            Vw[j + (int64_t)N * k] = vk[j];
    }
    for (int e = tid; e < r * r; e += NT) {
        Gu[e] = Gv[e] = Gc[e] = (e % r == e / r) ? scalar(1) : scalar(0);
    }
    __syncthreads();
    jacobi_orthogonalize<NT>(Uw, M, r, Gu, &s_changed);
    jacobi_orthogonalize<NT>(Vw, N, r, Gv, &s_changed);
    for (int j = wv; j < 2 * r; j += NW) { // column norms of both factors
        const scalar *col = j < r ? Uw + (int64_t)M * j : Vw + (int64_t)N * (j - r);
        const int len     = j < r ? M : N;
        real nn = 0;
        for (int i = lane; i < len; i += WAVE)
            nn += hmx_re(hmx_conj(col[i]) * col[i]);
        nn = wave_sum_any(nn);
        if (lane == 0)
            (j < r ? su : sv)[j < r ? j : j - r] = sqrt(nn);
    }
    __syncthreads();
    // U V = Q_u [S_u G_u^H conj(G_v) S_v] Q_v^T  (V^T = Q_v S_v G_v^H): the core C
    for (int e = tid; e < r * r; e += NT) {
        const int i = e % r, j = e / r;
        scalar s = scalar(0);
        for (int l = 0; l < r; l++)
            s += hmx_conj(Gu[l + r * i]) * hmx_conj(Gv[l + r * j]);
        Cm[e] = su[i] * s * sv[j];
    }
    __syncthreads();
    jacobi_orthogonalize<NT>(Cm, r, r, Gc, &s_changed); // Cm <- C Gc = u_c diag(sc)
    for (int j = tid; j < r; j += NT) {
        real nn = 0;
        for (int i = 0; i < r; i++)
            nn += hmx_re(hmx_conj(Cm[i + r * j]) * Cm[i + r * j]);
        sc[j] = sqrt(nn);
    }
    __syncthreads();
    for (int j = tid; j < r; j += NT) {
        int pos = 0;
        for (int l = 0; l < r; l++)
            pos += (sc[l] > sc[j] || (sc[l] == sc[j] && l < j)) ? 1 : 0;
        ord[pos] = (real)j;
    }
    __syncthreads();
    if (tid == 0) { // SVD_truncation.hpp:37-52
        real norm2 = 0, err = 0;
        for (int l = 0; l < r; l++)
            norm2 += sc[l] * sc[l];
        const real nrm = sqrt(norm2);
        int j = r;
        do {
            j = j - 1;
            err += sc[(int)ord[j]] * sc[(int)ord[j]];
        } while (j > 0 && sqrt(err) / nrm < (real)A.epsilon);
        s_rank = j + 1;
    }
    __syncthreads();
    const int kr = s_rank;
    if (kr < r) {
        // U'(:,k) = sqrt(s_k) * sum_i Q_u(:,i) u_c(i,k) ,  Q_u(:,i) = Uw(:,i)/su_i ,  u_c(:,k) = Cm(:,jk)/sc_jk
        // V'(k,:) = sqrt(s_k) * sum_i conj(Gc(i,jk)) Q_v(:,i)^T ,  Q_v(:,i) = Vw(:,i)/sv_i      (C = u_c diag(sc) Gc^H)
        for (int k = 0; k < kr; k++) {
            const int jk   = (int)ord[k];
            const real sk  = sc[jk], rs = sqrt(sk), isk = sk > 0 ? real(1) / sk : real(0);
            scalar *c   = A.pool + cross[k];
            scalar *uk  = sw ? c + n1 : c, *vk = sw ? c : c + n1;
            for (int i = tid; i < M; i += NT) {
                scalar s = scalar(0);
                for (int l = 0; l < r; l++)
                    if (su[l] > 0)
                        s += Uw[i + (int64_t)M * l] / su[l] * (Cm[l + r * jk] * isk);
                uk[i] = rs * s;
            }
            for (int j = tid; j < N; j += NT) {
                scalar s = scalar(0);
                for (int l = 0; l < r; l++)
                    if (sv[l] > 0)
                        s += Vw[j + (int64_t)N * l] / sv[l] * hmx_conj(Gc[l + r * jk]);
                vk[j] = rs * s;
            }
        }
        if (tid == 0)
            A.rank[b] = kr;
    }
}
