// kernels_body.hpp -- the kernels, written against `scalar` / `scalar2` (coefficient type) and `real` (its underlying
// real type) and included four times by engine.hip: namespaces hmx::f64 / hmx::f32 (scalar = real = double / float;
// htool's HMatrix<float,double>: fp32 coefficients, fp64 coordinates) and hmx::z64 / hmx::c32 (scalar = cplx<real>,
// HMX_COMPLEX = 1: htool's HMatrix<std::complex<...>>).  No include guard on purpose.
// Layout and design notes: kernels_common.hpp.  Compiled with -ffp-contract=off, FMAs are explicit (hmx_fma).

__device__ __forceinline__ scalar wave_sum(scalar v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += hmx_shfl_xor(v, o);
    return v;
}

template <typename V>
__device__ __forceinline__ V wave_sum_any(V v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += hmx_shfl_xor(v, o);
    return v;
}

// generator entry in coefficient precision.  KS_INV_DIST: real 1/den; complex (cre + i cim sgn)/den, component-wise division as
// std::complex<double> / double does.  KS_HELMHOLTZ: exp(i k r) / (p0 + p1 r) (real types: its real part).  KS_LAPLACE_SL:
// (cre [+ i cim]) / (4 pi (p0 + r)).
__device__ __forceinline__ scalar eval_scalar(const KernelSpec &ks, double tx, double ty, double tz, double sx, double sy, double sz) {
    if (ks.kind == KS_INV_DIST) {
#if HMX_COMPLEX
        const double den = eval_kernel_den(ks, tx, ty, tz, sx, sy, sz);
        const double u   = tx - sx;
        const double sgn = ks.herm ? (u > 0 ? 1.0 : (u < 0 ? -1.0 : 0.0)) : 1.0;
        return scalar((real)(ks.cre / den), (real)((ks.cim * sgn) / den));
#else
        return (scalar)eval_kernel(ks, tx, ty, tz, sx, sy, sz);
#endif
    }
    const double r = sqrt(eval_dist2(ks, tx, ty, tz, sx, sy, sz));
    if (ks.kind == KS_HELMHOLTZ) {
        const double den = ks.p0 + ks.p1 * r;
        double sn, cs;
        hmx_sincos(ks.wavenumber * r, sn, cs);
#if HMX_COMPLEX
        return scalar((real)(cs / den), (real)(sn / den));
#else
        return (scalar)(cs / den);
#endif
    }
    const double den = HMX_FOUR_PI * (ks.p0 + r); // KS_LAPLACE_SL
#if HMX_COMPLEX
    return scalar((real)(ks.cre / den), (real)(ks.cim / den));
#else
    return (scalar)(ks.cre / den);
#endif
}



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


// ---------------------------------------------------------------------------------------------
// Pack: move compressed data into the matvec streams
// ---------------------------------------------------------------------------------------------
struct PackLrArgs {
    const scalar *pool;
    const int64_t *cross_off; // per (block,k)
    const int64_t *colptr;
    const int32_t *rank;
    const int32_t *swapped;
    const int32_t *t_off, *t_size, *s_off, *s_size;
    // pair lists
    const int32_t *pair_block, *pair_range, *pair_col; // column offset inside the range's stream
    const int32_t *range_off, *range_len;
    const int64_t *range_base;
    const int32_t *range_cols; // C of the range (R-stream only)
    const int32_t *range_cw;   // chunk width of the range (R-stream only)
    scalar *stream;
    int origin;                // global cluster position of local offset 0 (T0 for E-streams, S0 for R-streams)
    const int32_t *conjflag;   // 1: this entry of the layout holds the conjugate of the stored factors (Hermitian mirror)
};

// U slices -> E-stream (column-major len x C per target range)
__global__ void pack_lr_expand_kernel(PackLrArgs P, int64_t npairs) {
    const int64_t p = blockIdx.x;
    if (p >= npairs)
        return;
    const int b = P.pair_block[p], R = P.pair_range[p], col = P.pair_col[p];
    const int len = P.range_len[R], r = P.rank[b];
    const int n1  = P.swapped[b] ? P.s_size[b] : P.t_size[b]; // length of uu in a cross
    const int rel = P.range_off[R] + P.origin - P.t_off[b];
    const int64_t *cross = P.cross_off + P.colptr[b];
    scalar *dst          = P.stream + P.range_base[R] + (int64_t)col * len;
    for (int e = threadIdx.x; e < r * len; e += blockDim.x) {
        const int k = e / len, i = e - k * len;
        // U(:,k) = uu_k when index 1 is the row side, vv_k otherwise (sympartialACA.hpp:198-212)
        const scalar *src = P.pool + cross[k] + (P.swapped[b] ? n1 : 0);
        dst[e]            = P.conjflag[b] ? hmx_conj(src[rel + i]) : src[rel + i];
    }
}

// row pitch of an R-stream chunk of w columns: even (16-byte rows for 8-byte pairs); a multiple of 4 for 4-byte coefficients, whose
// reduce stage reads 16 bytes = 4 columns per lane
constexpr int HMX_WPAD = sizeof(scalar) == 4 ? 3 : 1;
__host__ __device__ __forceinline__ int hmx_wp(int w) { return (w + HMX_WPAD) & ~HMX_WPAD; }

__device__ __forceinline__ int64_t rstream_index(int64_t base, int len, int C, int cw, int i, int col) {
    // row-major, chunks of cw columns (cw even, <= 128, chosen per range so the chunks are balanced); the last
    // chunk may be narrower and is stored with its own row pitch rounded up to even
    const int ch = col / cw, within = col - ch * cw;
    int w        = C - ch * cw;
    w            = w > cw ? cw : w;
    w            = hmx_wp(w);
    return base + (int64_t)ch * len * cw + (int64_t)i * w + within;
}

// V slices -> R-stream
__global__ void pack_lr_reduce_kernel(PackLrArgs P, int64_t npairs) {
    const int64_t p = blockIdx.x;
    if (p >= npairs)
        return;
    const int b = P.pair_block[p], S = P.pair_range[p], col = P.pair_col[p];
    const int len = P.range_len[S], r = P.rank[b], C = P.range_cols[S];
    const int n1  = P.swapped[b] ? P.s_size[b] : P.t_size[b];
    const int rel = P.range_off[S] + P.origin - P.s_off[b];
    const int64_t *cross = P.cross_off + P.colptr[b];
    for (int e = threadIdx.x; e < r * len; e += blockDim.x) {
        const int k = e / len, i = e - k * len;
        const scalar *src = P.pool + cross[k] + (P.swapped[b] ? 0 : n1); // V(k,:) = vv_k, or uu_k when swapped
        P.stream[rstream_index(P.range_base[S], len, C, P.range_cw[S], i, col + k)] = P.conjflag[b] ? hmx_conj(src[rel + i]) : src[rel + i];
    }
}

struct PackDenseArgs {
    KernelSpec ks;
    const double *tx, *ty, *tz, *sx, *sy, *sz;
    const int32_t *pair_block, *pair_range, *pair_col;
    const int32_t *range_off, *range_len;
    const int64_t *range_base;
    const int32_t *t_off, *t_size, *s_off, *s_size;
    const int64_t *staged_off; // >= 0: uploaded dense block (column-major M x N) in `pool`; < 0: generate
    const int32_t *sym_uplo;   // 0 none, 1 'L', 2 'U' : uploaded symmetric leaf, only that triangle is valid
    const int32_t *transposed; // 1: this entry of the layout is the TRANSPOSE of a stored leaf (mirrored copy, or a transposed view)
    const int32_t *conjflag;   // 1: ... and conjugated (Hermitian mirror)
    const scalar *pool;
    scalar *stream;
    int origin; // T0
    int herm;   // Hermitian storage: mirrored entries are conjugated, the diagonal of a symmetric leaf is real (hemv)
};

// dense leaves -> E-stream: HMatrix::compute_dense_data (hmatrix/hmatrix.hpp:222-226) fused with the
// layout change; entries are generated straight into their final position.
__global__ void pack_dense_kernel(PackDenseArgs P, int64_t npairs) {
    const int64_t p = blockIdx.x;
    if (p >= npairs)
        return;
    const int b = P.pair_block[p], R = P.pair_range[p], col = P.pair_col[p];
    const int len = P.range_len[R], N = P.s_size[b], M = P.t_size[b];
    const int row0 = P.range_off[R] + P.origin; // global cluster position of the range's first row
    const int rel  = row0 - P.t_off[b];
    const int c0   = P.s_off[b];
    scalar *dst    = P.stream + P.range_base[R] + (int64_t)col * len;
    const int64_t st = P.staged_off[b];
    const int su     = P.sym_uplo[b];
    for (int e = threadIdx.x; e < N * len; e += blockDim.x) {
        const int j = e / len, i = e - j * len;
        scalar v;
        if (st >= 0) {
            int ii = rel + i, jj = j;
            bool cj = false;
            if ((su == 1 && ii < jj) || (su == 2 && ii > jj)) { // symv / hemv semantics: mirror the stored triangle
                const int t = ii;
                ii          = jj;
                jj          = t;
                cj          = P.herm != 0;
            }
            if (P.conjflag[b])
                cj = !cj;
            v = P.transposed[b] ? P.pool[st + jj + (int64_t)N * ii] : P.pool[st + ii + (int64_t)M * jj];
            if (cj)
                v = hmx_conj(v);
            if (P.herm && su && ii == jj)
                v = scalar(hmx_re(v));
        } else if (P.transposed[b]) {
            // entry (i, j) of the transpose of a stored leaf: the generator is evaluated at (target = the column's point,
            // source = the row's point), i.e. at the stored leaf's own (row, column), then conjugated for a Hermitian mirror
            v = eval_scalar(P.ks, P.sx[c0 + j], P.sy[c0 + j], P.sz[c0 + j], P.tx[row0 + i], P.ty[row0 + i], P.tz[row0 + i]);
            if (P.conjflag[b])
                v = hmx_conj(v);
        } else {
            v = eval_scalar(P.ks, P.tx[row0 + i], P.ty[row0 + i], P.tz[row0 + i], P.sx[c0 + j], P.sy[c0 + j], P.sz[c0 + j]);
            if (P.conjflag[b])
                v = hmx_conj(v);
        }
        dst[e] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// H-matvec, trans = 'N'
// ---------------------------------------------------------------------------------------------
// Stage 1 (add_lrmat_vector_product.hpp:16, a = V x): one wave per (source range, column chunk).
// lane owns two adjacent columns and walks the rows; the x slice is loaded 64 rows at a time (one
// coalesced load) and broadcast with v_readlane, so the row loop contains only the 16-B stream loads.
// The two R-stream columns a lane owns.  Real and complex-float coefficients: two ADJACENT columns, one 16-byte load per
// row.  Complex double (16-byte coefficients): columns lane and lane + 64, two loads that are each one contiguous KiB per wave.
#if HMX_SPLIT_COLS
#define HMX_COL0(lane) (lane)
#define HMX_COL1(lane) ((lane) + 64)
__device__ __forceinline__ scalar2 load_pair(const scalar *row, int col0, int col1, int wp) {
    scalar2 v;
    v.x = col0 < wp ? stream_load(row + col0) : scalar(0);
    v.y = col1 < wp ? stream_load(row + col1) : scalar(0);
    return v;
}
#else
#define HMX_COL0(lane) (2 * (lane))
#define HMX_COL1(lane) (2 * (lane) + 1)
__device__ __forceinline__ scalar2 load_pair(const scalar *row, int col0, int col1, int wp) {
    return stream_load(reinterpret_cast<const scalar2 *>(row + (col0 < wp ? col0 : 0))); // wp is even: both columns or none
}
#endif

struct ReduceArgs {
    const scalar *stream;
    const int32_t *task_range, *task_chunk;
    const int32_t *range_off, *range_len, *range_cols, *range_cw;
    const int64_t *range_base;
    const int64_t *range_colbase; // first entry of the range in out_idx
    const int32_t *out_idx;       // per column: destination in Z (an `a` slot or a partial slot)
    const scalar *x;              // input vector, local to the source root
    scalar *Z;
    int ntasks;
};

#ifndef HMX_REDUCE_ROWS
#define HMX_REDUCE_ROWS 1
#endif
// 4-byte real coefficients: a chunk is at most 128 columns = 512 bytes per row, so with the 8-byte pair loads of the generic path a
// wave-wide load moves at most 512 bytes (5.2 TB/s).  Here every lane loads 16 bytes = 4 adjacent columns and a wave-wide load covers
// R = 256 / wp whole rows of the contiguous row-major chunk (lane l: row group 4 l / wp); the R partial sums of a column are folded
// in a fixed tree at the end.  (Compiled for every coefficient type, called for float only.)
__device__ __forceinline__ void reduce_rows_x4(const ReduceArgs &A, int lane, int S, int ch, int len, int w, int wp, int cw, const scalar *src, const scalar *xs) {
    const int R = 256 / wp, hw = wp / 4; // rows per load (>= 2), lanes per row
    const int g = lane / hw;
    const bool lane_ok = g < R;
    const scalar *p  = src + 4 * lane;
    const scalar *xg = xs + g;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 8 * R <= len; j += 8 * R) {
        hmx_f4v v[8];
        float xi[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            v[u]  = lane_ok ? __builtin_nontemporal_load(reinterpret_cast<const hmx_f4v *>(p + (int64_t)(j + u * R) * wp)) : hmx_f4v{0.f, 0.f, 0.f, 0.f};
            xi[u] = lane_ok ? (float)hmx_re(xg[j + u * R]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            a[0] = __builtin_fmaf(v[u].x, xi[u], a[0]);
            a[1] = __builtin_fmaf(v[u].y, xi[u], a[1]);
            a[2] = __builtin_fmaf(v[u].z, xi[u], a[2]);
            a[3] = __builtin_fmaf(v[u].w, xi[u], a[3]);
        }
    }
    for (; j < len; j += R) {
        const bool ok  = lane_ok && j + g < len;
        const float xi = ok ? (float)hmx_re(xg[j]) : 0.f;
        const hmx_f4v v = ok ? __builtin_nontemporal_load(reinterpret_cast<const hmx_f4v *>(p + (int64_t)j * wp)) : hmx_f4v{0.f, 0.f, 0.f, 0.f};
        a[0] = __builtin_fmaf(v.x, xi, a[0]);
        a[1] = __builtin_fmaf(v.y, xi, a[1]);
        a[2] = __builtin_fmaf(v.z, xi, a[2]);
        a[3] = __builtin_fmaf(v.w, xi, a[3]);
    }
    for (int n = R; n > 1;) { // row groups 0..n-1 hold partial sums; fold the upper half onto the lower one
        const int h = (n + 1) >> 1;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float t = __shfl(a[k], lane + h * hw, WAVE);
            if (g + h < n)
                a[k] += t;
        }
        n = h;
    }
    if (lane < hw) {
        const int64_t cb = A.range_colbase[S] + ch * cw;
        const int c0     = 4 * lane;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (c0 + k < w)
                A.Z[A.out_idx[cb + c0 + k]] = scalar(a[k]);
    }
}
#ifndef HMX_REDUCE_UNROLL_NARROW
#define HMX_REDUCE_UNROLL_NARROW 16
#endif
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) void reduce_kernel(ReduceArgs A) {
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6)); // wave-uniform
    if (task >= A.ntasks)
        return;
    const int lane = threadIdx.x & 63;
    const int S = A.task_range[task], ch = A.task_chunk[task];
    const int len = A.range_len[S], C = A.range_cols[S], cw = A.range_cw[S];
    int w = C - ch * cw;
    w     = w > cw ? cw : w;
    const int wp      = hmx_wp(w);
    const int col0 = HMX_COL0(lane), col1 = HMX_COL1(lane); // the two columns of this lane
    const bool active = col0 < wp;
    const scalar *src = A.stream + A.range_base[S] + (int64_t)ch * len * cw;
    const scalar *xs  = A.x + A.range_off[S];
    scalar a0 = scalar(0), a1 = scalar(0);
    // Narrow chunks (at most half a wave wide: the per-rank share of a multi-GPU run, small problems): the chunk is one
    // contiguous row-major array, so a wave-wide load covers R = floor(wave elements / wp) whole rows; lane l holds the
    // columns of row group g = EPL*l / wp.  R times fewer loads for the same bytes; the R partial sums of a column are
    // added in a fixed tree at the end.  HMX_REDUCE_ROWS=0 (compile time) keeps one row per load.
    if (HMX_REDUCE_ROWS && sizeof(scalar) == 4) { // fp32: 16-byte loads for every chunk (wp <= 128 is a multiple of 4)
        reduce_rows_x4(A, lane, S, ch, len, w, wp, cw, src, xs);
        return;
    }
    constexpr int EPL = HMX_SPLIT_COLS ? 1 : 2; // stream elements per lane and load
    if (HMX_REDUCE_ROWS && wp <= 32 * EPL) {
        const int R  = (64 * EPL) / wp;         // rows per load, >= 2
        const int hw = wp / EPL;                // lanes per row
        const int g = lane / hw, e0 = EPL * lane; // row group of this lane, its offset in the R-row window
        const bool lane_ok = g < R;
        // the x value of a lane's row comes straight from memory: R distinct addresses per load, always cache hits
        const scalar *p  = src + e0;
        const scalar *xg = xs + g;
        int j = 0;
        for (; j + 8 * R <= len; j += 8 * R) {
#if HMX_SPLIT_COLS
            scalar v[8];
#else
            scalar2 v[8];
#endif
            scalar xi[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
#if HMX_SPLIT_COLS
                v[u] = lane_ok ? stream_load(p + (int64_t)(j + u * R) * wp) : scalar(0);
#else
                v[u] = lane_ok ? stream_load(reinterpret_cast<const scalar2 *>(p + (int64_t)(j + u * R) * wp)) : scalar2{scalar(0), scalar(0)};
#endif
                xi[u] = lane_ok ? xg[j + u * R] : scalar(0);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
#if HMX_SPLIT_COLS
                a0 = hmx_fma(v[u], xi[u], a0);
#else
                a0 = hmx_fma(v[u].x, xi[u], a0);
                a1 = hmx_fma(v[u].y, xi[u], a1);
#endif
            }
        }
        for (; j < len; j += R) {
            const bool ok   = lane_ok && j + g < len;
            const scalar xi = ok ? xg[j] : scalar(0);
#if HMX_SPLIT_COLS
            const scalar v = ok ? stream_load(p + (int64_t)j * wp) : scalar(0);
            a0             = hmx_fma(v, xi, a0);
#else
            const scalar2 v = ok ? stream_load(reinterpret_cast<const scalar2 *>(p + (int64_t)j * wp)) : scalar2{scalar(0), scalar(0)};
            a0              = hmx_fma(v.x, xi, a0);
            a1              = hmx_fma(v.y, xi, a1);
#endif
        }
        for (int n = R; n > 1;) { // row groups 0..n-1 hold partial sums; fold the upper half onto the lower one
            const int h     = (n + 1) >> 1;
            const scalar t0 = hmx_shfl(a0, lane + h * hw), t1 = hmx_shfl(a1, lane + h * hw);
            if (g + h < n) {
                a0 += t0;
                a1 += t1;
            }
            n = h;
        }
        if (lane < hw) {
            const int64_t cb = A.range_colbase[S] + ch * cw;
            const int c0     = EPL * lane;
            if (c0 < w)
                A.Z[A.out_idx[cb + c0]] = a0;
            if (EPL == 2 && c0 + 1 < w)
                A.Z[A.out_idx[cb + c0 + 1]] = a1;
        }
        return;
    }
    // rows in flight per wave: 8 for 16-byte loads (1 KiB per row and wave), 16 when a lane's pair is only 8 bytes (fp32: 512 B per row)
    constexpr int RU = sizeof(scalar2) <= 8 ? HMX_REDUCE_UNROLL_NARROW : 8;
    for (int i0 = 0; i0 < len; i0 += 64) {
        const int nr    = (len - i0) < 64 ? (len - i0) : 64;
        const scalar xv = lane < nr ? xs[i0 + lane] : scalar(0);
        const scalar *p = src + (int64_t)i0 * wp;
        int j = 0;
        for (; j + RU <= nr; j += RU) {
            scalar2 v[RU];
#pragma unroll
            for (int u = 0; u < RU; u++)
                v[u] = load_pair(p + (int64_t)(j + u) * wp, col0, col1, wp);
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const scalar xi = readlane_val(xv, j + u);
                a0              = hmx_fma(v[u].x, xi, a0);
                a1              = hmx_fma(v[u].y, xi, a1);
            }
        }
        for (; j < nr; j++) {
            const scalar2 v = load_pair(p + (int64_t)j * wp, col0, col1, wp);
            const scalar xi = readlane_val(xv, j);
            a0              = hmx_fma(v.x, xi, a0);
            a1              = hmx_fma(v.y, xi, a1);
        }
    }
    if (active) {
        const int64_t cb = A.range_colbase[S] + ch * cw;
        if (col0 < w)
            A.Z[A.out_idx[cb + col0]] = a0;
        if (col1 < w)
            A.Z[A.out_idx[cb + col1]] = a1;
    }
}

// Stage 1b: blocks whose source cluster spans several ranges: a_b[k] = sum_s partial[b][s][k]
struct CombineArgs {
    const int32_t *dst, *src, *stride, *count;
    scalar *Z;
    int n;
};
__global__ void combine_kernel(CombineArgs A) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.n)
        return;
    const scalar *p = A.Z + A.src[e];
    const int st = A.stride[e], cnt = A.count[e];
    // four independent partial sums keep four loads in flight; the order is fixed, so results stay reproducible
    scalar s0 = scalar(0), s1 = scalar(0), s2 = scalar(0), s3 = scalar(0);
    int k = 0;
    for (; k + 4 <= cnt; k += 4) {
        s0 += p[(int64_t)k * st];
        s1 += p[(int64_t)(k + 1) * st];
        s2 += p[(int64_t)(k + 2) * st];
        s3 += p[(int64_t)(k + 3) * st];
    }
    for (; k < cnt; k++)
        s0 += p[(int64_t)k * st];
    A.Z[A.dst[e]] = (s0 + s1) + (s2 + s3);
}

// fixed-order all-reduce over the 64 lanes without LDS (v_permlane32_swap, v_permlane16_swap, DPP row operations): every lane returns
// the same sum, the order of the additions does not depend on anything but the lane layout -- used where one wave folds many partial
// sums (combine_list_wave_kernel)
__device__ __forceinline__ scalar wave_sum_dpp(scalar s) {
    scalar a = s, b = s;
    lane_swap32(a, b);
    s = a + b;
    a = s, b = s;
    lane_swap16(a, b);
    s = a + b;
    s += dpp_move<0x128>(s);
    s += dpp_move<0x141>(s);
    s += dpp_move<0xB1>(s);
    s += dpp_move<0x4E>(s);
    return s;
}
// Stage 2 (dense leaves: add_matrix_vector_product.hpp:18; low rank: add_lrmat_vector_product.hpp:17,
// y += U a; final alpha/beta as openmp_internal_add_hmatrix_vector_product :134-136,168):
// one workgroup per target row range, lane = row, the waves split the columns in 64-column chunks.
struct ExpandArgs {
    const scalar *stream;
    const int32_t *order; // launch position -> range (heaviest ranges first)
    const int32_t *range_off, *range_len, *range_cols;
    const int64_t *range_base;
    const int64_t *range_colbase;
    const int32_t *z_idx; // per column: index into Z = [x | a | ...]
    const scalar *Z;
    scalar *y;            // output, local to the target root
    scalar alpha, beta;
    int nranges;
    // Z = [x | a | partials]: indices below nx are read straight from the caller's input vector instead of a copy in Z
    const scalar *x;
    int nx;
};
__device__ __forceinline__ const scalar *expand_operand(const ExpandArgs &A, int zi, int mu) {
    return (zi < A.nx ? A.x : A.Z) + (int64_t)zi * mu;
}

// loads in flight per wave in the expand stage: 8 for 8- and 16-byte coefficients, 16 for 4-byte ones (a wave's load is then
// only 256 bytes; N=1e6 fp32: 0.916 -> 0.899 ms); -DHMX_EXPAND_UNROLL=8 restores 8 for A/B comparison
#ifndef HMX_EXPAND_UNROLL
#define HMX_EXPAND_UNROLL 16
#endif
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) void expand_kernel(ExpandArgs A) {
    constexpr int EU = sizeof(scalar) == 4 ? HMX_EXPAND_UNROLL : 8;
    __shared__ scalar part[WAVES][WAVE];
    const int R = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = A.range_len[R], C = A.range_cols[R];
    const scalar *E     = A.stream + A.range_base[R];
    const int32_t *zidx = A.z_idx + A.range_colbase[R];
    const bool active   = lane < len;
    const int row       = active ? lane : 0;
    scalar acc = scalar(0);
    for (int c0 = wv * 64; c0 < C; c0 += WAVES * 64) {
        const int nc   = (C - c0) < 64 ? (C - c0) : 64;
        const scalar z = lane < nc ? *expand_operand(A, zidx[c0 + lane], 1) : scalar(0);
        const scalar *col = E + (int64_t)c0 * len + row;
        int j = 0;
        for (; j + EU <= nc; j += EU) {
            scalar v[EU];
#pragma unroll
            for (int u = 0; u < EU; u++)
                v[u] = stream_load(col + (int64_t)(j + u) * len);
#pragma unroll
            for (int u = 0; u < EU; u++)
                acc = hmx_fma(v[u], readlane_val(z, j + u), acc);
        }
        for (; j < nc; j++)
            acc = hmx_fma(col[(int64_t)j * len], readlane_val(z, j), acc);
    }
    part[wv][lane] = active ? acc : scalar(0);
    __syncthreads();
    if (wv == 0 && active) {
        scalar s = part[0][lane];
#pragma unroll
        for (int k = 1; k < WAVES; k++)
            s += part[k][lane];
        scalar *yo = A.y + A.range_off[R] + lane;
        *yo        = hmx_is_zero(A.beta) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused multi-RHS (row-major, mu fastest) H-matvec, trans = 'N':
// openmp_internal_add_hmatrix_matrix_product_row_major (hmatrix/linalg/add_hmatrix_matrix_product_row_major.hpp:112-178),
// leaf products add_matrix_matrix_product_row_major / add_lrmat_matrix_product_row_major (K7-K9 of SURVEY.md 2.2).
// The streams are read ONCE for MU right-hand sides; Z, x and y are [index][mu] with a row pitch of `mu`
// doubles and this launch handles the MU columns starting at `cbase`.  The wave-uniform operand (x rows
// in the reduce stage, gathered coefficients in the expand stage) is staged in a wave-private LDS tile and read
// back as broadcast ds_read_b128, so the inner loops are one stream load + MU FMAs per lane.
// ---------------------------------------------------------------------------------------------
// Narrow chunk (at most half a wave wide) of the multi-RHS reduce stage, same idea as in reduce_kernel: a wave-wide load
// covers R = floor(wave elements / wp) whole rows of the contiguous row-major chunk, lane l works on row group EPL*l / wp and
// reads ITS row's MU operands from the wave-private LDS tile (R distinct rows per ds_read instead of one broadcast row:
// the same LDS time for R rows of stream).  The R partial sums per column are folded in a fixed tree at the end.
template <int MU>
__device__ __forceinline__ void reduce_mu_narrow(const ReduceArgs &A, scalar (*xt)[MU], int lane, int S, int ch, int len, int w, int wp, int cw,
                                                 const scalar *src, const scalar *xs, int mu, int cbase) {
    constexpr int EPL = HMX_SPLIT_COLS ? 1 : 2;
    const int R = (64 * EPL) / wp, hw = wp / EPL;
    const int g = lane / hw, e0 = EPL * lane;
    const bool lane_ok = g < R;
    scalar a0[MU], a1[MU];
#pragma unroll
    for (int c = 0; c < MU; c++)
        a0[c] = a1[c] = scalar(0);
    for (int i0 = 0; i0 < len; i0 += 64) {
        const int nr = (len - i0) < 64 ? (len - i0) : 64;
        {
            __builtin_amdgcn_wave_barrier();
            if (lane < nr) {
#pragma unroll
                for (int c = 0; c < MU; c++)
                    xt[lane][c] = xs[(int64_t)(i0 + lane) * mu + c];
            }
            __builtin_amdgcn_wave_barrier();
        }
        scalar(*const xrow)[MU] = xt;
        const scalar *p = src + (int64_t)i0 * wp + e0;
        for (int j = 0; j < nr; j += 4 * R) {
#if HMX_SPLIT_COLS
            scalar v[4];
#else
            scalar2 v[4];
#endif
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ok[u] = lane_ok && j + u * R + g < nr;
#if HMX_SPLIT_COLS
                v[u] = ok[u] ? stream_load(p + (int64_t)(j + u * R) * wp) : scalar(0);
#else
                v[u] = ok[u] ? stream_load(reinterpret_cast<const scalar2 *>(p + (int64_t)(j + u * R) * wp)) : scalar2{scalar(0), scalar(0)};
#endif
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (ok[u]) { // masked lanes do not touch their sums (no 0 * inf)
                    const scalar *xr = xrow[j + u * R + g];
#pragma unroll
                    for (int c = 0; c < MU; c++) {
#if HMX_SPLIT_COLS
                        a0[c] = hmx_fma(v[u], xr[c], a0[c]);
#else
                        a0[c] = hmx_fma(v[u].x, xr[c], a0[c]);
                        a1[c] = hmx_fma(v[u].y, xr[c], a1[c]);
#endif
                    }
                }
        }
    }
    for (int n = R; n > 1;) {
        const int h = (n + 1) >> 1;
#pragma unroll
        for (int c = 0; c < MU; c++) {
            const scalar t0 = hmx_shfl(a0[c], lane + h * hw), t1 = hmx_shfl(a1[c], lane + h * hw);
            if (g + h < n) {
                a0[c] += t0;
                a1[c] += t1;
            }
        }
        n = h;
    }
    if (lane < hw) {
        const int64_t cb = A.range_colbase[S] + ch * cw;
        const int c0     = EPL * lane;
        if (c0 < w) {
            scalar *dst = A.Z + (int64_t)A.out_idx[cb + c0] * mu + cbase;
#pragma unroll
            for (int c = 0; c < MU; c++)
                dst[c] = a0[c];
        }
        if (EPL == 2 && c0 + 1 < w) {
            scalar *dst = A.Z + (int64_t)A.out_idx[cb + c0 + 1] * mu + cbase;
#pragma unroll
            for (int c = 0; c < MU; c++)
                dst[c] = a1[c];
        }
    }
}

template <int WAVES, int MU>
__global__ __launch_bounds__(WAVES *WAVE) void reduce_mu_kernel(ReduceArgs A, int mu, int cbase) {
    __shared__ __attribute__((aligned(16))) scalar xt[WAVES][WAVE][MU];
    const int wv   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * WAVES + wv;
    if (task >= A.ntasks)
        return;
    const int lane = threadIdx.x & 63;
    const int S = A.task_range[task], ch = A.task_chunk[task];
    const int len = A.range_len[S], C = A.range_cols[S], cw = A.range_cw[S];
    int w = C - ch * cw;
    w     = w > cw ? cw : w;
    const int wp      = hmx_wp(w);
    const int col0 = HMX_COL0(lane), col1 = HMX_COL1(lane);
    const bool active = col0 < wp;
    const scalar *src = A.stream + A.range_base[S] + (int64_t)ch * len * cw;
    const scalar *xs  = A.x + (int64_t)A.range_off[S] * mu + cbase;
    if (HMX_REDUCE_ROWS && wp <= (HMX_SPLIT_COLS ? 32 : 64)) {
        reduce_mu_narrow<MU>(A, xt[wv], lane, S, ch, len, w, wp, cw, src, xs, mu, cbase);
        return;
    }
    scalar a0[MU], a1[MU];
#pragma unroll
    for (int c = 0; c < MU; c++)
        a0[c] = a1[c] = scalar(0);
    for (int i0 = 0; i0 < len; i0 += 64) {
        const int nr = (len - i0) < 64 ? (len - i0) : 64;
        __builtin_amdgcn_wave_barrier();
        if (lane < nr) {
#pragma unroll
            for (int c = 0; c < MU; c++)
                xt[wv][lane][c] = xs[(int64_t)(i0 + lane) * mu + c];
        }
        __builtin_amdgcn_wave_barrier();
        const scalar *p = src + (int64_t)i0 * wp;
        int j = 0;
        for (; j + 4 <= nr; j += 4) {
            scalar2 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                v[u] = load_pair(p + (int64_t)(j + u) * wp, col0, col1, wp);
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int c = 0; c < MU; c++) {
                    const scalar xi = xt[wv][j + u][c];
                    a0[c]           = hmx_fma(v[u].x, xi, a0[c]);
                    a1[c]           = hmx_fma(v[u].y, xi, a1[c]);
                }
            }
        }
        for (; j < nr; j++) {
            const scalar2 v = load_pair(p + (int64_t)j * wp, col0, col1, wp);
#pragma unroll
            for (int c = 0; c < MU; c++) {
                const scalar xi = xt[wv][j][c];
                a0[c]           = hmx_fma(v.x, xi, a0[c]);
                a1[c]           = hmx_fma(v.y, xi, a1[c]);
            }
        }
    }
    if (active) {
        const int64_t cb = A.range_colbase[S] + ch * cw;
        if (col0 < w) {
            scalar *dst = A.Z + (int64_t)A.out_idx[cb + col0] * mu + cbase;
#pragma unroll
            for (int c = 0; c < MU; c++)
                dst[c] = a0[c];
        }
        if (col1 < w) {
            scalar *dst = A.Z + (int64_t)A.out_idx[cb + col1] * mu + cbase;
#pragma unroll
            for (int c = 0; c < MU; c++)
                dst[c] = a1[c];
        }
    }
}

__global__ void combine_mu_kernel(CombineArgs A, int mu) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)A.n * mu)
        return;
    const int e = (int)(t / mu), c = (int)(t - (int64_t)e * mu);
    const scalar *p = A.Z + (int64_t)A.src[e] * mu + c;
    const int st = A.stride[e], cnt = A.count[e];
    scalar s = scalar(0);
    for (int k = 0; k < cnt; k++)
        s += p[(int64_t)k * st * mu];
    A.Z[(int64_t)A.dst[e] * mu + c] = s;
}

template <int WAVES, int MU>
__global__ __launch_bounds__(WAVES *WAVE) void expand_mu_kernel(ExpandArgs A, int mu, int cbase) {
    __shared__ __attribute__((aligned(16))) scalar zt[WAVES][WAVE][MU]; // coefficient tiles, reused for the final reduction
    const int R = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = A.range_len[R], C = A.range_cols[R];
    const scalar *E     = A.stream + A.range_base[R];
    const int32_t *zidx = A.z_idx + A.range_colbase[R];
    const bool active   = lane < len;
    const int row       = active ? lane : 0;
    scalar acc[MU];
#pragma unroll
    for (int c = 0; c < MU; c++)
        acc[c] = scalar(0);
    for (int c0 = wv * 64; c0 < C; c0 += WAVES * 64) {
        const int nc = (C - c0) < 64 ? (C - c0) : 64;
        __builtin_amdgcn_wave_barrier();
        if (lane < nc) {
            const scalar *zr = expand_operand(A, zidx[c0 + lane], mu) + cbase;
#pragma unroll
            for (int c = 0; c < MU; c++)
                zt[wv][lane][c] = zr[c];
        }
        __builtin_amdgcn_wave_barrier();
        const scalar *col = E + (int64_t)c0 * len + row;
        int j = 0;
        // columns in flight per wave: 16 for 4-byte coefficients (a wave's load is only 256 bytes then), 8 otherwise -- as in expand_kernel
        constexpr int EU = sizeof(scalar) == 4 ? 16 : 8;
        for (; j + EU <= nc; j += EU) {
            scalar v[EU];
#pragma unroll
            for (int u = 0; u < EU; u++)
                v[u] = stream_load(col + (int64_t)(j + u) * len);
#pragma unroll
            for (int u = 0; u < EU; u++)
#pragma unroll
                for (int c = 0; c < MU; c++)
                    acc[c] = hmx_fma(v[u], zt[wv][j + u][c], acc[c]);
        }
        for (; j < nc; j++) {
            const scalar v = col[(int64_t)j * len];
#pragma unroll
            for (int c = 0; c < MU; c++)
                acc[c] = hmx_fma(v, zt[wv][j][c], acc[c]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < MU; c++)
        zt[wv][lane][c] = active ? acc[c] : scalar(0);
    __syncthreads();
    // rows x MU outputs, summed over the waves; consecutive threads write consecutive right-hand sides
    for (int e = threadIdx.x; e < len * MU; e += WAVES * WAVE) {
        const int i = e / MU, c = e - i * MU;
        scalar s = zt[0][i][c];
#pragma unroll
        for (int k = 1; k < WAVES; k++)
            s += zt[k][i][c];
        scalar *yo = A.y + (int64_t)(A.range_off[R] + i) * mu + cbase + c;
        *yo        = hmx_is_zero(A.beta) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
    }
}

#if !HMX_COMPLEX
// ---------------------------------------------------------------------------------------------
// Multi-RHS kernels with the wave-uniform operand in SCALAR registers.  The MU coefficients a streamed row (reduce) or
// column (expand) is multiplied with are the same for all 64 lanes: instead of staging them in LDS and reading them
// back as broadcast ds_read_b128 (4 per row for 16 floats -- the LDS pipe, not HBM, then bounds the fp32 kernels), they
// are fetched through the scalar cache (s_load_dwordx16 from a constant-address-space view of X / Z, which no wave of these
// kernels writes) and enter the packed FMAs as SGPR operands.  Same arithmetic, same order as the *_mu kernels.
// ---------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) scalar *uniform_ptr;
typedef const __attribute__((address_space(4))) int32_t *uniform_iptr;

template <int WAVES, int MU>
__global__ __launch_bounds__(WAVES *WAVE) void reduce_mus_kernel(ReduceArgs A, int mu, int cbase) {
    const int wv   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * WAVES + wv;
    if (task >= A.ntasks)
        return;
    const int lane = threadIdx.x & 63;
    const int S = A.task_range[task], ch = A.task_chunk[task];
    const int len = A.range_len[S], C = A.range_cols[S], cw = A.range_cw[S];
    int w = C - ch * cw;
    w     = w > cw ? cw : w;
    const int wp      = hmx_wp(w);
    const int col0 = HMX_COL0(lane), col1 = HMX_COL1(lane);
    const bool active = col0 < wp;
    const scalar *src = A.stream + A.range_base[S] + (int64_t)ch * len * cw;
    if (HMX_REDUCE_ROWS && wp <= 64) { // narrow chunk: rows differ between lanes, so the operand cannot be wave-uniform
        __shared__ __attribute__((aligned(16))) scalar xt[WAVES][WAVE][MU];
        reduce_mu_narrow<MU>(A, xt[wv], lane, S, ch, len, w, wp, cw, src, A.x + (int64_t)A.range_off[S] * mu + cbase, mu, cbase);
        return;
    }
    uniform_ptr xs    = (uniform_ptr)(A.x + (int64_t)A.range_off[S] * mu + cbase);
    scalar a0[MU], a1[MU];
#pragma unroll
    for (int c = 0; c < MU; c++)
        a0[c] = a1[c] = scalar(0);
    int i = 0;
    constexpr int RU = sizeof(scalar2) <= 8 ? 8 : 4; // rows in flight: a wave's load of 4-byte pairs is at most 512 bytes
    for (; i + RU <= len; i += RU) {
        scalar2 v[RU];
#pragma unroll
        for (int u = 0; u < RU; u++)
            v[u] = load_pair(src + (int64_t)(i + u) * wp, col0, col1, wp);
#pragma unroll
        for (int u = 0; u < RU; u++) {
            uniform_ptr xr = xs + (int64_t)(i + u) * mu;
#pragma unroll
            for (int c = 0; c < MU; c++) {
                const scalar xi = xr[c];
                a0[c]           = hmx_fma(v[u].x, xi, a0[c]);
                a1[c]           = hmx_fma(v[u].y, xi, a1[c]);
            }
        }
    }
    for (; i < len; i++) {
        const scalar2 v = load_pair(src + (int64_t)i * wp, col0, col1, wp);
        uniform_ptr xr  = xs + (int64_t)i * mu;
#pragma unroll
        for (int c = 0; c < MU; c++) {
            const scalar xi = xr[c];
            a0[c]           = hmx_fma(v.x, xi, a0[c]);
            a1[c]           = hmx_fma(v.y, xi, a1[c]);
        }
    }
    if (active) {
        const int64_t cb = A.range_colbase[S] + ch * cw;
        if (col0 < w) {
            scalar *dst = A.Z + (int64_t)A.out_idx[cb + col0] * mu + cbase;
#pragma unroll
            for (int c = 0; c < MU; c++)
                dst[c] = a0[c];
        }
        if (col1 < w) {
            scalar *dst = A.Z + (int64_t)A.out_idx[cb + col1] * mu + cbase;
#pragma unroll
            for (int c = 0; c < MU; c++)
                dst[c] = a1[c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 16 right-hand sides on the matrix cores.  With mu = 16 the leaf products are real GEMMs
// (K7-K9 of SURVEY.md 2.2: [rows x cols] x [cols x 16]); the VALU kernels above then spend their time re-reading the
// 16 wave-uniform operands from LDS (8 broadcast ds_read_b128 per streamed column).  v_mfma_*_16x16x4 takes that
// operand as ONE register per lane: stream tile = A (16 x 4), operand tile = B (4 x 16), 16 x 16 accumulators.
// fp64/fp32 MFMA peak equals the vector peak on gfx950, so this is not about FLOP/s: it takes the LDS and VALU-issue
// pressure off a kernel that should be HBM-bound.  Results differ from the VALU kernels only by summation order.
// ---------------------------------------------------------------------------------------------
typedef Acc4<real>::type acc4;

// The stream tile is STAGED THROUGH LDS.  In the first version of this kernel (round 2) every lane fetched its own MFMA operand element:
// one load instruction of a wave is four 128-byte pieces of four different columns (16 rows x 8 bytes each, and a column of a 61-row
// range starts at an odd multiple of 8 bytes, so most pieces straddle two lines) -- the kernel moves its bytes at 5 TB/s where the
// single-vector expand_kernel, whose loads are whole columns (lane = row: 488 contiguous bytes), reaches 6.5 TB/s.  Here the loads
// are those of expand_kernel -- 16 whole columns per step, the next step's 16 in flight under the current step's MFMAs -- and the
// 64 x 16 tile goes through a wave-private LDS buffer (80-element column pitch: the operand reads 16 rows x 4 columns are free of bank
// conflicts) to reach the lanes in operand layout.  LDS traffic is 16 bytes per streamed 8, a quarter of the pipe.
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) void expand_mfma16s_kernel(ExpandArgs A, int mu, int cbase, int nrhs) {
    constexpr int PITCH = 80; // 64 rows + 16: consecutive tile columns are 32 banks apart, so the 64-bit operand reads (16 rows x 4 columns) do not conflict
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 16 * PITCH > WAVES * WAVE * 16 ? WAVES * 16 * PITCH : WAVES * WAVE * 16]; // (HMX_EXPAND_PERMLANE: only the final fold of the waves uses it)
    const int R = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = A.range_len[R], C = A.range_cols[R];
    const real *E       = A.stream + A.range_base[R];
    const int32_t *zidx = A.z_idx + A.range_colbase[R];
    const int m = lane & 15, kk = lane >> 4;
    real(*tile)[PITCH] = reinterpret_cast<real(*)[PITCH]>(lds + wv * 16 * PITCH);
    const int row      = lane < len ? lane : len - 1; // idle lanes re-read the last row: their tile rows only reach accumulator rows that are never stored
    // nrhs < 16, a ragged last group: operand column m of the MFMA only reaches result column m, and the columns >= nrhs are never stored, so
    // their lanes just read a valid element (the group's first right-hand side).  NOT a select on the loaded value: the compiler then moves the
    // load under an exec-mask branch with a vmcnt(0) behind it (fp32 config 5: 8.4 -> 12.2 ms for this kernel)
    const int mo = cbase + (m < nrhs ? m : 0);
    acc4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
        acc[t] = acc4{0, 0, 0, 0};
    // The wave's work as ONE sequence of steps of 16 columns: wave w owns the 64-column tiles w, w + WAVES, ... of the range, step s covers the
    // columns col_of(s) ... + 16; every tile is full except the range's last.
    const int ntile_all = (C + 63) >> 6;
    int n = 0; // steps of this wave
    if (wv < ntile_all) {
        n = 4 * ((ntile_all - 1 - wv) / WAVES + 1);
        if ((ntile_all - 1 - wv) % WAVES == 0) // the range's last tile is this wave's
            n -= 4 - ((C - 64 * (ntile_all - 1) + 15) >> 4);
    }
    auto col_of = [&](int s) { return (((s >> 2) * WAVES + wv) << 6) + ((s & 3) << 4); };
    // THREE stages in flight, every load unconditional: the indices of step s + 2, the operand gathers and the 16 stream columns of step
    // s + 1, the arithmetic of step s.  Steps beyond the wave's last re-load its last step (nobody uses the result).  Unconditional because
    // s_waitcnt vmcnt counts loads in issue order and the compiler derives the count at a use from what is CERTAINLY outstanding there: behind
    // an `if (more columns) prefetch;` that is the path without the prefetch, and every use then waits for the prefetch itself -- an
    // s_waitcnt vmcnt(0) per step, the pipeline drained once per 16 columns (round 5, read off the ISA of the round-3 kernel; the same rule as
    // in expand_sym_kernel).  The zero operand of a column beyond the range is selected when the step is applied, not behind the load (a
    // select waits on the spot), and HMX_SCHED_FENCE() keeps the load groups in program order (the scheduler sinks independent loads towards
    // their use otherwise: the gathers ended up LAST in the queue).  tests/test_isa_shape.py: no vmcnt(0) in the loop.
    auto load_idx = [&](int s) { // lane l: the Z index of column col_of(s) + (l & 15)
        const int c = col_of(s < n ? s : n - 1) + m;
        return zidx[c < C ? c : C - 1];
    };
    auto gathers = [&](real(&b)[4], int zi) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int zc = __shfl(zi, 4 * g + kk, WAVE);
            b[g]         = expand_operand(A, zc, mu)[mo];
        }
    };
    auto load_cols = [&](real(&v)[16], int s) { // 16 whole columns, clamped to the range's last one (zero operand there)
        const int c = col_of(s < n ? s : n - 1);
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int col = c + u < C ? c + u : C - 1;
            v[u]          = stream_load(E + (int64_t)col * len + row);
        }
    };
    // Round 4: the operand layout WITHOUT LDS for 4-byte coefficients.  The loads fill register u of lane r with E[row r][column c + u]; the
    // MFMA wants, for column group g and row tile t, lane (m, kk) to hold E[row 16 t + m][column c + 4 g + kk] -- register 4 g + kk of lane
    // quarter t.  That is a 4 x 4 transposition between register index and lane quarter per column group: v_permlane32_swap on (0, 2),
    // (1, 3), then v_permlane16_swap on (0, 1), (2, 3).  Sixteen swaps instead of sixteen LDS stores + sixteen LDS loads + a fence per step;
    // the same MFMAs on the same operands in the same order, so the results are bitwise those of the staged form.  8-byte coefficients (two
    // swaps per register) keep the LDS tile.  -DHMX_EXPAND_PERMLANE=0 / 1 forces one form for both.
#ifdef HMX_EXPAND_PERMLANE
    constexpr bool PERM = HMX_EXPAND_PERMLANE != 0;
#else
    constexpr bool PERM = sizeof(real) == 4;
#endif
    auto apply = [&](real(&v)[16], const real(&braw)[4], int s) {
        const int c = col_of(s);
        real b[4];
#pragma unroll
        for (int g = 0; g < 4; g++)
            b[g] = (c + 4 * g + kk < C) ? braw[g] : real(0);
        if constexpr (PERM) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                lane_swap32(v[4 * g + 0], v[4 * g + 2]);
                lane_swap32(v[4 * g + 1], v[4 * g + 3]);
                lane_swap16(v[4 * g + 0], v[4 * g + 1]);
                lane_swap16(v[4 * g + 2], v[4 * g + 3]);
            }
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[t] = mfma16(v[4 * g + t], b[g], acc[t]);
        } else {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 16; u++)
                tile[u][lane] = v[u];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            real a[4][4];
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    a[g][t] = tile[4 * g + kk][16 * t + m];
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[t] = mfma16(a[g][t], b[g], acc[t]);
        }
    };
    if (n > 0) {
        real v0[16], v1[16], b0[4], b1[4];
        int i0 = load_idx(0), i1 = load_idx(1);
        gathers(b0, i0);
        load_cols(v0, 0);
        HMX_SCHED_FENCE();
        for (int s = 0; s < n; s += 2) {
            i0 = load_idx(s + 2);
            gathers(b1, i1);
            load_cols(v1, s + 1);
            HMX_SCHED_FENCE();
            apply(v0, b0, s);
            HMX_SCHED_FENCE();
            i1 = load_idx(s + 3);
            gathers(b0, i0);
            load_cols(v0, s + 2);
            HMX_SCHED_FENCE();
            if (s + 1 < n)
                apply(v1, b1, s + 1);
            HMX_SCHED_FENCE();
        }
    }
    // accumulator tile t, register j of lane l = (row 16t + mfma16_row, rhs l & 15): stage as [row][rhs] (the tile buffers are done with)
    real(*red)[WAVE][16] = reinterpret_cast<real(*)[WAVE][16]>(lds);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            red[wv][16 * t + mfma16_row(real(0), lane, j)][m] = acc[t][j];
    __syncthreads();
    for (int e = threadIdx.x; e < len * 16; e += WAVES * WAVE) {
        const int i = e >> 4, c = e & 15;
        if (c >= nrhs)
            continue;
        real s = red[0][i][c];
#pragma unroll
        for (int w = 1; w < WAVES; w++)
            s += red[w][i][c];
        real *yo = A.y + (int64_t)(A.range_off[R] + i) * mu + cbase + c;
        *yo      = A.beta == real(0) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
    }
}

// expand_mfma16s_kernel for groups of up to 32 right-hand sides: every tile element read from LDS feeds two MFMAs (operand sets m and
// 16 + m), so a sweep over the E-stream serves twice the columns -- at 32 right-hand sides the product needs 8 flops per streamed byte
// and the matrix cores, not HBM, set the pace (fp64: 82 % of their peak at full HBM speed).
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) __attribute__((amdgpu_waves_per_eu(2))) void expand_mfma32s_kernel(ExpandArgs A, int mu, int cbase, int nrhs) {
    constexpr int PITCH = 80; // 64 rows + 16: consecutive tile columns are 32 banks apart, so the 64-bit operand reads (16 rows x 4 columns) do not conflict
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 16 * PITCH > WAVES * WAVE * 16 ? WAVES * 16 * PITCH : WAVES * WAVE * 16];
    const int R = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = A.range_len[R], C = A.range_cols[R];
    const real *E       = A.stream + A.range_base[R];
    const int32_t *zidx = A.z_idx + A.range_colbase[R];
    const int m = lane & 15, kk = lane >> 4;
    real(*tile)[PITCH] = reinterpret_cast<real(*)[PITCH]>(lds + wv * 16 * PITCH);
    const int row      = lane < len ? lane : len - 1; // idle lanes re-read the last row: their tile rows only reach accumulator rows that are never stored
    // nrhs < 16, a ragged last group: operand column m of the MFMA only reaches result column m, and the columns >= nrhs are never stored, so
    // their lanes just read a valid element (the group's first right-hand side).  NOT a select on the loaded value: the compiler then moves the
    // load under an exec-mask branch with a vmcnt(0) behind it (fp32 config 5: 8.4 -> 12.2 ms for this kernel)
    const int mo = cbase + (m < nrhs ? m : 0), mo2 = cbase + (16 + m < nrhs ? 16 + m : 0);
    acc4 acc[4], acc2[4]; // right-hand sides 0..15 and 16..31 of the group
#pragma unroll
    for (int t = 0; t < 4; t++)
        acc[t] = acc2[t] = acc4{0, 0, 0, 0};
    auto load_cols = [&](real(&v)[16], int c) { // 16 whole columns, clamped to the last one (zero operand there)
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int col = c + u < C ? c + u : C - 1;
            v[u]          = stream_load(E + (int64_t)col * len + row);
        }
    };
    auto operands = [&](real(&b)[8], int c, int zi, int base) { // b[g]: right-hand side m, b[4 + g]: right-hand side 16 + m of operand row g
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int zc   = __shfl(zi, base + 4 * g + kk, WAVE);
            const real *zr = expand_operand(A, zc, mu);
            const real bv = zr[mo], bw = zr[mo2];
            b[g]     = (c + 4 * g + kk < C) ? bv : real(0);
            b[4 + g] = (c + 4 * g + kk < C) ? bw : real(0);
        }
    };
    auto apply = [&](const real(&v)[16], const real(&b)[8]) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 16; u++)
            tile[u][lane] = v[u];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 4; g++) {
            real a[4];
#pragma unroll
            for (int t = 0; t < 4; t++)
                a[t] = tile[4 * g + kk][16 * t + m];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                acc[t]  = mfma16(a[t], b[g], acc[t]); // the tile element is read from LDS once for both halves of the group
                acc2[t] = mfma16(a[t], b[4 + g], acc2[t]);
            }
        }
    };
    // wave w takes the 64-column tiles w, w + WAVES, ...; inside a tile four steps of 16 columns, loads one step ahead
    for (int t0 = wv * 64; t0 < C; t0 += WAVES * 64) {
        const int tend = (t0 + 64) < C ? (t0 + 64) : C;
        const int zi   = (t0 + lane < C) ? zidx[t0 + lane] : zidx[C - 1];
        real v0[16], v1[16], b0[8], b1[8];
        load_cols(v0, t0);
        operands(b0, t0, zi, 0);
        if (t0 + 16 < tend) {
            load_cols(v1, t0 + 16);
            operands(b1, t0 + 16, zi, 16);
        }
        apply(v0, b0);
        if (t0 + 32 < tend) {
            load_cols(v0, t0 + 32);
            operands(b0, t0 + 32, zi, 32);
        }
        if (t0 + 16 < tend)
            apply(v1, b1);
        if (t0 + 48 < tend) {
            load_cols(v1, t0 + 48);
            operands(b1, t0 + 48, zi, 48);
        }
        if (t0 + 32 < tend)
            apply(v0, b0);
        if (t0 + 48 < tend)
            apply(v1, b1);
    }
    // accumulator tile t, register j of lane l = (row 16t + mfma16_row, rhs l & 15): stage as [row][rhs] (the tile buffers are done with),
    // first the right-hand sides 0..15, then 16..31 through the same buffer
    real(*red)[WAVE][16] = reinterpret_cast<real(*)[WAVE][16]>(lds);
#pragma unroll
    for (int half = 0; half < 2; half++) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                red[wv][16 * t + mfma16_row(real(0), lane, j)][m] = half ? acc2[t][j] : acc[t][j];
        __syncthreads();
        for (int e = threadIdx.x; e < len * 16; e += WAVES * WAVE) {
            const int i = e >> 4, c = 16 * half + (e & 15);
            if (c >= nrhs)
                continue;
            real s = red[0][i][e & 15];
#pragma unroll
            for (int w = 1; w < WAVES; w++)
                s += red[w][i][e & 15];
            real *yo = A.y + (int64_t)(A.range_off[R] + i) * mu + cbase + c;
            *yo      = A.beta == real(0) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
        }
    }
}

// The reduce stage on the matrix cores, stream tile staged through LDS as in expand_mfma16s_kernel: the loads are those of the single-vector
// reduce_kernel (a lane fetches two adjacent columns, a wave one whole row of the chunk: up to 1 KiB contiguous), eight rows per step
// with the next eight in flight, and the 8 x 128 tile reaches the lanes in operand layout through a wave-private LDS buffer
// (144-element row pitch: rows 32 banks apart).
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) void reduce_mfma16s_kernel(ReduceArgs A, int mu, int cbase, int nrhs) {
    // Rows per wave-wide load: a chunk of <= 64 (<= 32) columns puts 2 (4) consecutive rows into one load instruction -- the chunk is a
    // contiguous row-major block, lane l reads the column pair 2 (l mod LPR) of row l / LPR -- instead of leaving half (three quarters) of
    // the lanes idle; a step is then 16 (32) rows and 4 (8) k-steps over 4 (2) column tiles: the same 16 MFMAs per 8 loads.  On one
    // rank's share of a row-partitioned operator up to half of the R-stream sits in such chunks (few leaves share a source cluster).
    // Tile pitch per variant: 144 / 80 / 48 elements (= 16 mod 32: the operand reads of 4 rows x 16 columns do not conflict).
    // fp32 only: measured on one box (profiles/r3_ab_rpl.log), fp32 reduce stage -8 % on the whole N = 1e6 operator and -4 % on one rank's
    // share of config 5; the fp64 stage does not gain from the 2-row form (a row of 64 fp64 columns already is a 512-byte load) and loses 3 %.
#ifndef HMX_REDUCE_RPL64
#define HMX_REDUCE_RPL64 4
#endif
    constexpr int RPL_MAX = sizeof(real) == 8 ? HMX_REDUCE_RPL64 : 4;
    constexpr int TILE    = RPL_MAX == 4 ? 32 * 48 : 8 * 144;
    static_assert(TILE >= 8 * 144, "tile buffer");
    __shared__ __attribute__((aligned(16))) real lds[WAVES * TILE];
    const int wv   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * WAVES + wv;
    if (task >= A.ntasks)
        return;
    const int lane = threadIdx.x & 63;
    const int S = A.task_range[task], ch = A.task_chunk[task];
    const int len = A.range_len[S], C = A.range_cols[S], cw = A.range_cw[S];
    int w = C - ch * cw;
    w     = w > cw ? cw : w;
    const int wp    = hmx_wp(w);
    const int ntile = (w + 15) >> 4; // <= 8 column tiles of 16
    const real *src = A.stream + A.range_base[S] + (int64_t)ch * len * cw;
    const real *xs  = A.x + (int64_t)A.range_off[S] * mu + cbase;
    const int m = lane & 15, kk = lane >> 4; // A: column m of the tile, row kk of the step; B: row kk, rhs m
    real *tile   = lds + wv * TILE;
    const int mo = m < nrhs ? m : 0; // ragged group: see expand_mfma16s_kernel (xs already points at the group's first right-hand side)
    acc4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++)
        acc[t] = acc4{0, 0, 0, 0};
    auto sweep = [&](auto rpl_c) {
        constexpr int RPL = decltype(rpl_c)::value, LPR = 64 / RPL, RS = 8 * RPL, KS = 2 * RPL, NT = 8 / RPL;
        constexpr int PITCH = RPL == 1 ? 144 : (RPL == 2 ? 80 : 48);
        const int rl = lane / LPR, lr = lane % LPR;
        const int c2 = 2 * lr < wp ? 2 * lr : 0; // lanes beyond the chunk re-read its first pair (their tile columns are never used)
        auto load_rows = [&](scalar2(&v)[8], int i0) { // 8 loads of RPL whole rows each, clamped to the last row (its operand is zero there)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int row = i0 + u * RPL + rl < len ? i0 + u * RPL + rl : len - 1;
                v[u]          = stream_load(reinterpret_cast<const scalar2 *>(src + (int64_t)row * wp + c2));
            }
        };
        // operand loads only ISSUE here (rows clamped into the piece); the zero of a row beyond it is selected when the step is applied: a
        // select right behind the load makes the wave wait for every load issued before it -- the step in flight included (see expand_mfma16s_kernel)
        auto operands = [&](real(&b)[KS], int i0) {
#pragma unroll
            for (int h = 0; h < KS; h++) {
                const int row = i0 + 4 * h + kk;
                b[h]          = xs[(int64_t)(row < len ? row : len - 1) * mu + mo];
            }
        };
        // All NT column tiles of the variant, unconditionally: a tile beyond the chunk's last column multiplies what the idle lanes re-read
        // (finite stream data) into accumulators nobody stores.  With one `if (t < ntile)` per MFMA the compiler emitted ds_read -> s_waitcnt
        // lgkmcnt(0) -> v_mfma -> branch sixteen times in a row (round 5, read off the ISA): an LDS round trip exposed per MFMA.
        auto apply = [&](const scalar2(&v)[8], const real(&braw)[KS], int i0) {
            real b[KS];
#pragma unroll
            for (int h = 0; h < KS; h++)
                b[h] = (i0 + 4 * h + kk < len) ? braw[h] : real(0);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 8; u++)
                *reinterpret_cast<scalar2 *>(&tile[(u * RPL + rl) * PITCH + 2 * lr]) = v[u];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < KS; h++) {
                real a[NT];
#pragma unroll
                for (int t = 0; t < NT; t++)
                    a[t] = tile[(4 * h + kk) * PITCH + 16 * t + m];
#pragma unroll
                for (int t = 0; t < NT; t++)
                    acc[t] = mfma16(a[t], b[h], acc[t]);
            }
        };
        scalar2 v0[8], v1[8];
        real b0[KS], b1[KS];
        operands(b0, 0);
        load_rows(v0, 0);
        HMX_SCHED_FENCE();
        // every prefetch unconditional (rows beyond the piece are clamped into it, their operand is zeroed at use): the compiler can then
        // count the loads outstanding at each use -- behind an `if (more rows)` it assumes the path without the prefetch and waits for
        // everything (see expand_mfma16s_kernel)
        for (int i0 = 0; i0 < len; i0 += 2 * RS) {
            operands(b1, i0 + RS);
            load_rows(v1, i0 + RS);
            HMX_SCHED_FENCE();
            apply(v0, b0, i0);
            HMX_SCHED_FENCE();
            operands(b0, i0 + 2 * RS);
            load_rows(v0, i0 + 2 * RS);
            HMX_SCHED_FENCE();
            if (i0 + RS < len)
                apply(v1, b1, i0 + RS);
            HMX_SCHED_FENCE();
        }
    };
    if (RPL_MAX >= 4 && wp <= 32)
        sweep(std::integral_constant<int, (RPL_MAX >= 4 ? 4 : 1)>{});
    else if (RPL_MAX >= 2 && wp <= 64)
        sweep(std::integral_constant<int, (RPL_MAX >= 2 ? 2 : 1)>{});
    else
        sweep(std::integral_constant<int, 1>{});
    const int64_t cb = A.range_colbase[S] + ch * cw;
    // destinations first: the chunk's (<= 128) indices in two coalesced loads, handed to the lanes by shuffles.  With the index fetched
    // under each store's own predicate the compiler emits load -> vmcnt(0) -> store thirty-two times in a row (and it moves plain
    // unpredicated index loads back under the predicates; a shuffle cannot be moved into divergent code)
    const int32_t ilo = A.out_idx[cb + (lane < w ? lane : 0)], ihi = A.out_idx[cb + (64 + lane < w ? 64 + lane : 0)];
    int32_t dst[8][4];
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            dst[t][j] = __shfl(t < 4 ? ilo : ihi, 16 * (t & 3) + mfma16_row(real(0), lane, j), WAVE);
#pragma unroll
    for (int t = 0; t < 8; t++)
        if (t < ntile)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int col = 16 * t + mfma16_row(real(0), lane, j);
                if (col < w && m < nrhs)
                    A.Z[(int64_t)dst[t][j] * mu + cbase + m] = acc[t][j];
            }
}

// reduce_mfma16s_kernel for groups of up to 32 right-hand sides.  Sixteen accumulator tiles (8 column tiles x 2 operand sets) do not fit
// the registers, so a task walks its rows once per HALF of its (<= 128) columns -- the halves are different coefficients, nothing is read
// twice -- with 2 rows per wave-wide load (64 columns x 2 rows: every lane busy), 4 column tiles and both operand sets per k-step.
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) __attribute__((amdgpu_waves_per_eu(2))) void reduce_mfma32s_kernel(ReduceArgs A, int mu, int cbase, int nrhs) {
    constexpr int PITCH = 80, RS = 16, KS = 4; // 16 rows per step = 4 k-steps, tile pitch 80 (= 16 mod 32: conflict-free operand reads)
    __shared__ __attribute__((aligned(16))) real lds[WAVES * RS * PITCH];
    const int wv   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * WAVES + wv;
    if (task >= A.ntasks)
        return;
    const int lane = threadIdx.x & 63;
    const int S = A.task_range[task], ch = A.task_chunk[task];
    const int len = A.range_len[S], C = A.range_cols[S], cw = A.range_cw[S];
    int w = C - ch * cw;
    w     = w > cw ? cw : w;
    const int wp    = hmx_wp(w);
    const real *src = A.stream + A.range_base[S] + (int64_t)ch * len * cw;
    const real *xs  = A.x + (int64_t)A.range_off[S] * mu + cbase;
    const int m = lane & 15, kk = lane >> 4;
    real *tile   = lds + wv * RS * PITCH;
    const int mo = m < nrhs ? m : 0, mo2 = 16 + m < nrhs ? 16 + m : 0; // ragged group: see expand_mfma16s_kernel
    const int rl = lane >> 5, lr = lane & 31;                          // row of the load, column pair in the row
    const int64_t cb = A.range_colbase[S] + ch * cw;
    for (int c0 = 0; c0 < w; c0 += 64) { // columns [c0, c0 + 64) of the chunk
        const int wh    = w - c0 < 64 ? w - c0 : 64;
        const int ntile = (wh + 15) >> 4;
        const int c2    = c0 + 2 * lr < wp ? c0 + 2 * lr : c0; // lanes beyond the chunk re-read the half's first pair (their tile columns are never used)
        acc4 acc[4], acc2[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
            acc[t] = acc2[t] = acc4{0, 0, 0, 0};
        auto load_rows = [&](scalar2(&v)[8], int i0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int row = i0 + 2 * u + rl < len ? i0 + 2 * u + rl : len - 1;
                v[u]          = stream_load(reinterpret_cast<const scalar2 *>(src + (int64_t)row * wp + c2));
            }
        };
        auto operands = [&](real(&b)[2 * KS], int i0) {
#pragma unroll
            for (int h = 0; h < KS; h++) {
                const int row  = i0 + 4 * h + kk;
                const real *xr = xs + (int64_t)(row < len ? row : len - 1) * mu;
                const real bv = xr[mo], bw = xr[mo2];
                b[h]      = row < len ? bv : real(0);
                b[KS + h] = row < len ? bw : real(0);
            }
        };
        auto apply = [&](const scalar2(&v)[8], const real(&b)[2 * KS]) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 8; u++)
                *reinterpret_cast<scalar2 *>(&tile[(2 * u + rl) * PITCH + 2 * lr]) = v[u];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < KS; h++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (t < ntile) {
                        const real a = tile[(4 * h + kk) * PITCH + 16 * t + m];
                        acc[t]       = mfma16(a, b[h], acc[t]);
                        acc2[t]      = mfma16(a, b[KS + h], acc2[t]);
                    }
        };
        scalar2 v0[8], v1[8];
        real b0[2 * KS], b1[2 * KS];
        load_rows(v0, 0);
        operands(b0, 0);
        for (int i0 = 0; i0 < len; i0 += 2 * RS) {
            if (i0 + RS < len) {
                load_rows(v1, i0 + RS);
                operands(b1, i0 + RS);
            }
            apply(v0, b0);
            if (i0 + 2 * RS < len) {
                load_rows(v0, i0 + 2 * RS);
                operands(b0, i0 + 2 * RS);
            }
            if (i0 + RS < len)
                apply(v1, b1);
        }
        // destinations of the half's columns: one coalesced load, handed out by shuffles (see reduce_mfma16s_kernel)
        const int32_t ih = A.out_idx[cb + (c0 + lane < w ? c0 + lane : 0)];
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (t < ntile)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int cl      = 16 * t + mfma16_row(real(0), lane, j);
                    const int32_t dst = __shfl(ih, cl, WAVE);
                    if (c0 + cl < w) {
                        if (m < nrhs)
                            A.Z[(int64_t)dst * mu + cbase + m] = acc[t][j];
                        if (16 + m < nrhs)
                            A.Z[(int64_t)dst * mu + cbase + 16 + m] = acc2[t][j];
                    }
                }
    }
}

#endif // !HMX_COMPLEX

#if HMX_COMPLEX
// ---------------------------------------------------------------------------------------------
// Groups of 8 COMPLEX right-hand sides on the matrix cores (matrix/linalg/add_matrix_matrix_product_row_major.hpp:49-84,113-139: the
// complex gemm of the leaf products).  A row of 8 complex operands is 16 reals (re0, im0, re1, im1, ...): with n = 2 rhs + part as the
// MFMA's free index,
//     [Y_re | Y_im interleaved] += E_re * Z  +  E_im * Z',        Z'[n] = n even ? -Z[n + 1] : Z[n - 1]
// i.e. TWO real 16x16x4 MFMAs per complex tile; Z' is the operand register of the neighbouring lane (DPP quad_perm [1,0,3,2]) with the
// sign of the even lanes flipped, and the accumulator rows come out as interleaved complex numbers.  The stream tiles are staged through
// LDS as in expand_mfma16s_kernel / reduce_mfma16s_kernel (whole-column / whole-row loads, the next step in flight), real and imaginary
// parts in two planes.
// ---------------------------------------------------------------------------------------------
typedef Acc4<real>::type zacc4;
__device__ __forceinline__ real zmfma_swapped(real b, int lane) {
    const real o = hmx_shfl_xor(b, 1);
    return (lane & 1) ? o : -o;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) void expand_zmfma8s_kernel(ExpandArgs A, int mu, int cbase, int nrhs) {
    constexpr int PITCH = 80, STEP = 8; // 8 columns per step: two planes of 8 x 80 reals per wave
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 2 * STEP * PITCH > WAVES * WAVE * 16 ? WAVES * 2 * STEP * PITCH : WAVES * WAVE * 16];
    const int R = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = A.range_len[R], C = A.range_cols[R];
    const scalar *E     = A.stream + A.range_base[R];
    const int32_t *zidx = A.z_idx + A.range_colbase[R];
    const int m = lane & 15, kk = lane >> 4;
    real(*tre)[PITCH] = reinterpret_cast<real(*)[PITCH]>(lds + wv * 2 * STEP * PITCH);
    real(*tim)[PITCH] = tre + STEP;
    const int row     = lane < len ? lane : len - 1;
    const int mo      = m < 2 * nrhs ? m : 0; // ragged group: see expand_mfma16s_kernel
    zacc4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
        acc[t] = zacc4{0, 0, 0, 0};
    auto load_cols = [&](scalar(&v)[STEP], int c) {
#pragma unroll
        for (int u = 0; u < STEP; u++) {
            const int col = c + u < C ? c + u : C - 1;
            v[u]          = stream_load(E + (int64_t)col * len + row);
        }
    };
    auto operands = [&](real(&b)[2], real(&bs)[2], int c, int zi, int base) {
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const int zc   = __shfl(zi, base + 4 * g + kk, WAVE);
            const real *zr = reinterpret_cast<const real *>(expand_operand(A, zc, mu) + cbase);
            const real zv  = zr[mo]; // nrhs < 8, a ragged last group: see expand_mfma16s_kernel (lanes of the missing right-hand sides read a valid pair)
            const real bv  = (c + 4 * g + kk < C) ? zv : real(0);
            b[g]           = bv;
            bs[g]          = zmfma_swapped(bv, lane);
        }
    };
    auto apply = [&](const scalar(&v)[STEP], const real(&b)[2], const real(&bs)[2]) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < STEP; u++) {
            tre[u][lane] = v[u].re;
            tim[u][lane] = v[u].im;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                acc[t] = mfma16(tre[4 * g + kk][16 * t + m], b[g], acc[t]);
                acc[t] = mfma16(tim[4 * g + kk][16 * t + m], bs[g], acc[t]);
            }
    };
    // wave w takes the 64-column tiles w, w + WAVES, ...; inside a tile eight steps of 8 columns, loads one step ahead
    for (int t0 = wv * 64; t0 < C; t0 += WAVES * 64) {
        const int tend = (t0 + 64) < C ? (t0 + 64) : C;
        const int zi   = (t0 + lane < C) ? zidx[t0 + lane] : zidx[C - 1];
        scalar v0[STEP], v1[STEP];
        real b0[2], s0[2], b1[2], s1[2];
        load_cols(v0, t0);
        operands(b0, s0, t0, zi, 0);
        for (int c = t0; c < tend; c += 2 * STEP) {
            if (c + STEP < tend) {
                load_cols(v1, c + STEP);
                operands(b1, s1, c + STEP, zi, c + STEP - t0);
            }
            apply(v0, b0, s0);
            if (c + 2 * STEP < tend) {
                load_cols(v0, c + 2 * STEP);
                operands(b0, s0, c + 2 * STEP, zi, c + 2 * STEP - t0);
            }
            if (c + STEP < tend)
                apply(v1, b1, s1);
        }
    }
    // accumulator tile t, register j of lane l = (row 16t + mfma16_row, real column l & 15 = 2 rhs + part): stage as [row][16 reals]
    real(*red)[WAVE][16] = reinterpret_cast<real(*)[WAVE][16]>(lds);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            red[wv][16 * t + mfma16_row(real(0), lane, j)][m] = acc[t][j];
    __syncthreads();
    for (int e = threadIdx.x; e < len * 8; e += WAVES * WAVE) {
        const int i = e >> 3, c = e & 7;
        if (c >= nrhs)
            continue;
        scalar s(red[0][i][2 * c], red[0][i][2 * c + 1]);
#pragma unroll
        for (int w = 1; w < WAVES; w++)
            s += scalar(red[w][i][2 * c], red[w][i][2 * c + 1]);
        scalar *yo = A.y + (int64_t)(A.range_off[R] + i) * mu + cbase + c;
        *yo        = hmx_is_zero(A.beta) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
    }
}

// expand_zmfma8s_kernel for groups of up to 16 complex right-hand sides: the (re, im) planes of a tile element are read from LDS once and
// feed four MFMAs (two operand sets), as expand_mfma32s_kernel does for real coefficients.  The reduce stage keeps its sweeps of 8 (the two
// stages need not cut the right-hand sides into the same groups: stage 2 starts when all of stage 1 is done).
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) __attribute__((amdgpu_waves_per_eu(2))) void expand_zmfma16s_kernel(ExpandArgs A, int mu, int cbase, int nrhs) {
    constexpr int PITCH = 80, STEP = 8; // 8 columns per step: two planes of 8 x 80 reals per wave
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 2 * STEP * PITCH > WAVES * WAVE * 16 ? WAVES * 2 * STEP * PITCH : WAVES * WAVE * 16];
    const int R = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = A.range_len[R], C = A.range_cols[R];
    const scalar *E     = A.stream + A.range_base[R];
    const int32_t *zidx = A.z_idx + A.range_colbase[R];
    const int m = lane & 15, kk = lane >> 4;
    real(*tre)[PITCH] = reinterpret_cast<real(*)[PITCH]>(lds + wv * 2 * STEP * PITCH);
    real(*tim)[PITCH] = tre + STEP;
    const int row     = lane < len ? lane : len - 1;
    const int mo      = m < 2 * nrhs ? m : 0;                  // ragged group: see expand_mfma16s_kernel
    const int mo2     = 16 + (m < 2 * (nrhs - 8) ? m : 0);     // right-hand sides 8..15 of the group (nrhs > 8 here)
    zacc4 acc[4], acc2[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
        acc[t] = acc2[t] = zacc4{0, 0, 0, 0};
    auto load_cols = [&](scalar(&v)[STEP], int c) {
#pragma unroll
        for (int u = 0; u < STEP; u++) {
            const int col = c + u < C ? c + u : C - 1;
            v[u]          = stream_load(E + (int64_t)col * len + row);
        }
    };
    auto operands = [&](real(&b)[4], real(&bs)[4], int c, int zi, int base) { // [g]: right-hand sides 0..7, [2 + g]: 8..15
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const int zc   = __shfl(zi, base + 4 * g + kk, WAVE);
            const real *zr = reinterpret_cast<const real *>(expand_operand(A, zc, mu) + cbase);
            const real zv = zr[mo], zw = zr[mo2];
            const real bv = (c + 4 * g + kk < C) ? zv : real(0), bw = (c + 4 * g + kk < C) ? zw : real(0);
            b[g]      = bv;
            bs[g]     = zmfma_swapped(bv, lane);
            b[2 + g]  = bw;
            bs[2 + g] = zmfma_swapped(bw, lane);
        }
    };
    auto apply = [&](const scalar(&v)[STEP], const real(&b)[4], const real(&bs)[4]) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < STEP; u++) {
            tre[u][lane] = v[u].re;
            tim[u][lane] = v[u].im;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const real are = tre[4 * g + kk][16 * t + m], aim = tim[4 * g + kk][16 * t + m]; // read once for both halves of the group
                acc[t]  = mfma16(are, b[g], acc[t]);
                acc[t]  = mfma16(aim, bs[g], acc[t]);
                acc2[t] = mfma16(are, b[2 + g], acc2[t]);
                acc2[t] = mfma16(aim, bs[2 + g], acc2[t]);
            }
    };
    // wave w takes the 64-column tiles w, w + WAVES, ...; inside a tile eight steps of 8 columns, loads one step ahead
    for (int t0 = wv * 64; t0 < C; t0 += WAVES * 64) {
        const int tend = (t0 + 64) < C ? (t0 + 64) : C;
        const int zi   = (t0 + lane < C) ? zidx[t0 + lane] : zidx[C - 1];
        scalar v0[STEP], v1[STEP];
        real b0[4], s0[4], b1[4], s1[4];
        load_cols(v0, t0);
        operands(b0, s0, t0, zi, 0);
        for (int c = t0; c < tend; c += 2 * STEP) {
            if (c + STEP < tend) {
                load_cols(v1, c + STEP);
                operands(b1, s1, c + STEP, zi, c + STEP - t0);
            }
            apply(v0, b0, s0);
            if (c + 2 * STEP < tend) {
                load_cols(v0, c + 2 * STEP);
                operands(b0, s0, c + 2 * STEP, zi, c + 2 * STEP - t0);
            }
            if (c + STEP < tend)
                apply(v1, b1, s1);
        }
    }
    // accumulator tile t, register j of lane l = (row 16t + mfma16_row, real column l & 15 = 2 rhs + part): stage as [row][16 reals],
    // first the right-hand sides 0..7, then 8..15 through the same buffer
    real(*red)[WAVE][16] = reinterpret_cast<real(*)[WAVE][16]>(lds);
#pragma unroll
    for (int half = 0; half < 2; half++) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                red[wv][16 * t + mfma16_row(real(0), lane, j)][m] = half ? acc2[t][j] : acc[t][j];
        __syncthreads();
        for (int e = threadIdx.x; e < len * 8; e += WAVES * WAVE) {
            const int i = e >> 3, cl = e & 7, c = 8 * half + cl;
            if (c >= nrhs)
                continue;
            scalar s(red[0][i][2 * cl], red[0][i][2 * cl + 1]);
#pragma unroll
            for (int w = 1; w < WAVES; w++)
                s += scalar(red[w][i][2 * cl], red[w][i][2 * cl + 1]);
            scalar *yo = A.y + (int64_t)(A.range_off[R] + i) * mu + cbase + c;
            *yo        = hmx_is_zero(A.beta) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
        }
    }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) void reduce_zmfma8s_kernel(ReduceArgs A, int mu, int cbase, int nrhs) {
    constexpr int PITCH = 144, STEP = 4; // 4 rows per step (one MFMA k-step): two planes of 4 x 144 reals per wave
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 2 * STEP * PITCH];
    const int wv   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * WAVES + wv;
    if (task >= A.ntasks)
        return;
    const int lane = threadIdx.x & 63;
    const int S = A.task_range[task], ch = A.task_chunk[task];
    const int len = A.range_len[S], C = A.range_cols[S], cw = A.range_cw[S];
    int w = C - ch * cw;
    w     = w > cw ? cw : w;
    const int wp      = hmx_wp(w);
    const int ntile   = (w + 15) >> 4; // <= 8 column tiles of 16
    const scalar *src = A.stream + A.range_base[S] + (int64_t)ch * len * cw;
    const real *xs    = reinterpret_cast<const real *>(A.x + (int64_t)A.range_off[S] * mu + cbase);
    const int m = lane & 15, kk = lane >> 4; // A: column m of the tile, row kk of the step; B: row kk, real column m = 2 rhs + part
    real(*tre)[PITCH] = reinterpret_cast<real(*)[PITCH]>(lds + wv * 2 * STEP * PITCH);
    real(*tim)[PITCH] = tre + STEP;
    const int col0 = HMX_COL0(lane), col1 = HMX_COL1(lane);
    const int mo   = m < 2 * nrhs ? m : 0; // ragged group: see expand_mfma16s_kernel
    zacc4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++)
        acc[t] = zacc4{0, 0, 0, 0};
    auto load_rows = [&](scalar2(&v)[STEP], int i0) {
#pragma unroll
        for (int u = 0; u < STEP; u++) {
            const int row = i0 + u < len ? i0 + u : len - 1;
            v[u]          = load_pair(src + (int64_t)row * wp, col0, col1, wp);
        }
    };
    auto operands = [&](real &b, real &bs, int i0) {
        const int row = i0 + kk;
        const real bv = xs[(int64_t)(row < len ? row : len - 1) * 2 * mu + mo];
        b             = row < len ? bv : real(0);
        bs            = zmfma_swapped(b, lane);
    };
    auto apply = [&](const scalar2(&v)[STEP], real b, real bs) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < STEP; u++) {
            if (col0 < wp) {
                tre[u][col0] = v[u].x.re;
                tim[u][col0] = v[u].x.im;
            }
            if (col1 < wp) {
                tre[u][col1] = v[u].y.re;
                tim[u][col1] = v[u].y.im;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 8; t++)
            if (t < ntile) {
                acc[t] = mfma16(tre[kk][16 * t + m], b, acc[t]);
                acc[t] = mfma16(tim[kk][16 * t + m], bs, acc[t]);
            }
    };
    scalar2 v0[STEP], v1[STEP];
    real b0, s0, b1, s1;
    load_rows(v0, 0);
    operands(b0, s0, 0);
    for (int i0 = 0; i0 < len; i0 += 2 * STEP) {
        if (i0 + STEP < len) {
            load_rows(v1, i0 + STEP);
            operands(b1, s1, i0 + STEP);
        }
        apply(v0, b0, s0);
        if (i0 + 2 * STEP < len) {
            load_rows(v0, i0 + 2 * STEP);
            operands(b0, s0, i0 + 2 * STEP);
        }
        if (i0 + STEP < len)
            apply(v1, b1, s1);
    }
    const int64_t cb = A.range_colbase[S] + ch * cw;
    real *Zr         = reinterpret_cast<real *>(A.Z);
    // destinations first, two coalesced loads + shuffles: see reduce_mfma16s_kernel
    const int32_t ilo = A.out_idx[cb + (lane < w ? lane : 0)], ihi = A.out_idx[cb + (64 + lane < w ? 64 + lane : 0)];
    int32_t dst[8][4];
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            dst[t][j] = __shfl(t < 4 ? ilo : ihi, 16 * (t & 3) + mfma16_row(real(0), lane, j), WAVE);
#pragma unroll
    for (int t = 0; t < 8; t++)
        if (t < ntile)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int col = 16 * t + mfma16_row(real(0), lane, j);
                if (col < w && m < 2 * nrhs)
                    Zr[((int64_t)dst[t][j] * mu + cbase) * 2 + m] = acc[t][j];
            }
}

// reduce_zmfma8s_kernel for groups of up to 16 complex right-hand sides (as reduce_mfma32s_kernel for real coefficients): sixteen
// accumulator tiles do not fit the registers, so a task walks its rows once per HALF of its (<= 128) columns with both operand sets per
// k-step.  Complex double (columns lane / lane + 64 of a row are separate loads anyway): a pass issues the one load of its half, 4 rows
// = one k-step per step.  Complex float (two adjacent columns per 16-byte load): 2 rows per wave-wide load, 8 rows = two k-steps per step.
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) __attribute__((amdgpu_waves_per_eu(2))) void reduce_zmfma16s_kernel(ReduceArgs A, int mu, int cbase, int nrhs) {
    constexpr int PITCH = 80, RPL = HMX_SPLIT_COLS ? 1 : 2, RS = 4 * RPL, KS = RPL; // rows per load / per step, k-steps per step
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 2 * RS * PITCH];
    const int wv   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * WAVES + wv;
    if (task >= A.ntasks)
        return;
    const int lane = threadIdx.x & 63;
    const int S = A.task_range[task], ch = A.task_chunk[task];
    const int len = A.range_len[S], C = A.range_cols[S], cw = A.range_cw[S];
    int w = C - ch * cw;
    w     = w > cw ? cw : w;
    const int wp      = hmx_wp(w);
    const scalar *src = A.stream + A.range_base[S] + (int64_t)ch * len * cw;
    const real *xs    = reinterpret_cast<const real *>(A.x + (int64_t)A.range_off[S] * mu + cbase);
    const int m = lane & 15, kk = lane >> 4;
    real *tre = lds + wv * 2 * RS * PITCH, *tim = tre + RS * PITCH;
    const int mo  = m < 2 * nrhs ? m : 0;              // ragged group: see expand_mfma16s_kernel
    const int mo2 = 16 + (m < 2 * (nrhs - 8) ? m : 0); // right-hand sides 8..15 (nrhs > 8 here)
    const int64_t cb = A.range_colbase[S] + ch * cw;
    real *Zr         = reinterpret_cast<real *>(A.Z);
    for (int c0 = 0; c0 < w; c0 += 64) { // columns [c0, c0 + 64) of the chunk
        const int wh    = w - c0 < 64 ? w - c0 : 64;
        const int ntile = (wh + 15) >> 4;
#if HMX_SPLIT_COLS
        typedef scalar loaded; // one column per lane
        const int rl = 0, lc = lane;
        const int cl = c0 + lane < wp ? c0 + lane : c0;
#else
        typedef scalar2 loaded; // two adjacent columns per lane, 32 lanes per row
        const int rl = lane >> 5, lc = 2 * (lane & 31);
        const int cl = c0 + lc < wp ? c0 + lc : c0;
#endif
        zacc4 acc[4], acc2[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
            acc[t] = acc2[t] = zacc4{0, 0, 0, 0};
        auto load_rows = [&](loaded(&v)[4], int i0) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int row = i0 + u * RPL + rl < len ? i0 + u * RPL + rl : len - 1;
                v[u]          = stream_load(reinterpret_cast<const loaded *>(src + (int64_t)row * wp + cl));
            }
        };
        auto operands = [&](real(&b)[2 * KS], real(&bs)[2 * KS], int i0) { // [h]: right-hand sides 0..7, [KS + h]: 8..15 of k-step h
#pragma unroll
            for (int h = 0; h < KS; h++) {
                const int row  = i0 + 4 * h + kk;
                const real *xr = xs + (int64_t)(row < len ? row : len - 1) * 2 * mu;
                const real bv = xr[mo], bw = xr[mo2];
                b[h]       = row < len ? bv : real(0);
                b[KS + h]  = row < len ? bw : real(0);
                bs[h]      = zmfma_swapped(b[h], lane);
                bs[KS + h] = zmfma_swapped(b[KS + h], lane);
            }
        };
        auto apply = [&](const loaded(&v)[4], const real(&b)[2 * KS], const real(&bs)[2 * KS]) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = u * RPL + rl;
#if HMX_SPLIT_COLS
                tre[r * PITCH + lc] = v[u].re;
                tim[r * PITCH + lc] = v[u].im;
#else
                tre[r * PITCH + lc]     = v[u].x.re;
                tim[r * PITCH + lc]     = v[u].x.im;
                tre[r * PITCH + lc + 1] = v[u].y.re;
                tim[r * PITCH + lc + 1] = v[u].y.im;
#endif
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < KS; h++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (t < ntile) {
                        const real are = tre[(4 * h + kk) * PITCH + 16 * t + m], aim = tim[(4 * h + kk) * PITCH + 16 * t + m];
                        acc[t]  = mfma16(are, b[h], acc[t]);
                        acc[t]  = mfma16(aim, bs[h], acc[t]);
                        acc2[t] = mfma16(are, b[KS + h], acc2[t]);
                        acc2[t] = mfma16(aim, bs[KS + h], acc2[t]);
                    }
        };
        loaded v0[4], v1[4];
        real b0[2 * KS], s0[2 * KS], b1[2 * KS], s1[2 * KS];
        load_rows(v0, 0);
        operands(b0, s0, 0);
        for (int i0 = 0; i0 < len; i0 += 2 * RS) {
            if (i0 + RS < len) {
                load_rows(v1, i0 + RS);
                operands(b1, s1, i0 + RS);
            }
            apply(v0, b0, s0);
            if (i0 + 2 * RS < len) {
                load_rows(v0, i0 + 2 * RS);
                operands(b0, s0, i0 + 2 * RS);
            }
            if (i0 + RS < len)
                apply(v1, b1, s1);
        }
        // destinations of the half's columns: one coalesced load, handed out by shuffles (see reduce_mfma16s_kernel)
        const int32_t ih = A.out_idx[cb + (c0 + lane < w ? c0 + lane : 0)];
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (t < ntile)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int col     = 16 * t + mfma16_row(real(0), lane, j);
                    const int32_t dst = __shfl(ih, col, WAVE);
                    if (c0 + col < w) {
                        if (m < 2 * nrhs)
                            Zr[((int64_t)dst * mu + cbase) * 2 + m] = acc[t][j];
                        if (m < 2 * (nrhs - 8))
                            Zr[((int64_t)dst * mu + cbase) * 2 + 16 + m] = acc2[t][j];
                    }
                }
    }
}
#endif // HMX_COMPLEX

// ---------------------------------------------------------------------------------------------
// Wave reductions of the mirrored / transposed sweeps (columns of the E-streams, rows of the R-streams read "the other way")
// ---------------------------------------------------------------------------------------------
// Eight wave-wide sums for the price of ~1.25: each butterfly step halves the number of live values while
// halving the lane group that owns them.  On return lane l with (l & 7) == 0 holds the complete sum of input
// value number 4*bit5(l) + 2*bit4(l) + bit3(l).
// The lane exchanges are v_permlane32_swap / v_permlane16_swap / DPP row operations: no LDS traffic and none of ds_bpermute's
// latency in the dependent chain (-DHMX_REDUCE8_DPP=0 restores the ds_bpermute butterfly for A/B runs).  After the three halving
// steps the eight lanes of a group all-reduce with row_half_mirror (l <-> 7 - l) and the two quad permutations.
#ifndef HMX_REDUCE8_DPP
#define HMX_REDUCE8_DPP 1
#endif
__device__ __forceinline__ scalar reduce8(const scalar (&v)[8], int lane) {
#if HMX_REDUCE8_DPP
    scalar t[4], u[2];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        scalar a = v[k], b = v[k + 4];
        lane_swap32(a, b);
        t[k] = a + b;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        scalar a = t[k], b = t[k + 2];
        lane_swap16(a, b);
        u[k] = a + b;
    }
    const bool b3 = lane & 8;
    // (component-wise select: a lane-dependent choice between two complex values otherwise becomes a dynamically indexed private array --
    // 48 bytes of scratch and four scratch instructions per group of eight columns in the complex-double kernels until round 4)
    scalar r = hmx_select(b3, u[1], u[0]) + dpp_move<0x128>(hmx_select(b3, u[0], u[1])); // row_ror:8 = lane ^ 8 inside a row of 16
    r += dpp_move<0x141>(r);                                           // row_half_mirror
    r += dpp_move<0xB1>(r);                                            // quad_perm [1,0,3,2]
    r += dpp_move<0x4E>(r);                                            // quad_perm [2,3,0,1]
    return r;
#else
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    scalar t[4], u[2];
#pragma unroll
    for (int k = 0; k < 4; k++)
        t[k] = (b5 ? v[k + 4] : v[k]) + hmx_shfl_xor(b5 ? v[k] : v[k + 4], 32);
#pragma unroll
    for (int k = 0; k < 2; k++)
        u[k] = (b4 ? t[k + 2] : t[k]) + hmx_shfl_xor(b4 ? t[k] : t[k + 2], 16);
    scalar r = (b3 ? u[1] : u[0]) + hmx_shfl_xor(b3 ? u[0] : u[1], 8);
    r += hmx_shfl_xor(r, 4);
    r += hmx_shfl_xor(r, 2);
    r += hmx_shfl_xor(r, 1);
    return r;
#endif
}
__device__ __forceinline__ int reduce8_slot(int lane) { return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1); }

// ---------------------------------------------------------------------------------------------
// Fused symmetric / Hermitian product on the COMPACT layout (only the stored triangle is in HBM):
// add_hmatrix_vector_product.hpp:97-103,158-161 -- every leaf of leaves_for_symmetry is applied twice, out[t] += B in[s] and
// out[s] += B^T in[t] (B^H for 'H').  Here that is ONE sweep over the E-streams: while the tile of a row range sits in
// registers as lane = row for the forward product, the same registers give, per mirrored column, the column sum
// sum_i E[i,c] x_t[i] (eight wave reductions at a time, reduce8).  For a dense leaf that is the leaf's contribution to an output
// row; for a low-rank leaf B = U V it is a slice of a' = U^T x_t, and y_s += V^T a' needs a second sweep over the R-streams once
// a' is complete (the one factor a streaming product must read twice: U-expand needs V x_s and V-expand needs U^T x_t, so with
// one read of U the two V passes lie before and after it).  Nothing is accumulated with atomics: the column sums have their own
// slots in W = [a' | EW] assigned at layout time (E-column order: every row range writes one contiguous run), a' of a leaf spanning
// several ranges is folded in a fixed order (combine_list_kernel), and the second sweep (rowsym_kernel) owns the output rows it
// updates -- results are bit-reproducible.
// ---------------------------------------------------------------------------------------------
struct ExpandSymArgs {
    ExpandArgs X;
    const int32_t *mdst; // per E column: slot in W of its column sum, -1: not a mirrored column
    scalar *W;
    const scalar *xrow;  // the input at the TARGET positions of this operator: xrow[range_off + i]
    int herm;            // 'H' storage: the mirrored leaf is the conjugate transpose
};
// The wave's columns are walked in groups of eight (sixteen for 4-byte coefficients), flattened over its 64-column tiles and
// software-pipelined: the loads of group g + 1 are issued before group g is reduced, so the dependent chain of the eight-way
// reduction never leaves the wave without loads in flight.
// FWD = false: the mirrored column sums only -- the first sweep of the TRANSPOSED product of an ordinary operator on its stored data (every
// column is then a mirrored one, the forward operands and y are not touched; run_transposed_fused).
template <int WAVES, bool FWD = true>
__global__ __launch_bounds__(WAVES *WAVE) void expand_sym_kernel(ExpandSymArgs S) {
    const ExpandArgs &A = S.X;
    __shared__ scalar part[FWD ? WAVES : 1][WAVE];
    const int R = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = A.range_len[R], C = A.range_cols[R];
    const scalar *E     = A.stream + A.range_base[R];
    const int32_t *zidx = A.z_idx + A.range_colbase[R];
    const int32_t *mdst = S.mdst + A.range_colbase[R];
    const bool active   = lane < len;
    const int row       = active ? lane : 0;
    const scalar xr     = active ? S.xrow[A.range_off[R] + lane] : scalar(0); // idle lanes contribute exact zeros to the column sums
    const bool herm     = S.herm != 0;
    scalar acc = scalar(0);
    // always eight loads, no branches (the compiler can then count them: s_waitcnt vmcnt(8) keeps the next group in flight while
    // this one is used): beyond the last column of the range the last column is read again -- its products meet the zero
    // coefficients of the lanes >= nc and its column sums are never stored
    // (4-byte coefficients: groups of sixteen, reduced as two eights -- a wave's load is then only 256 bytes, sixteen are needed in flight)
    constexpr int GS = sizeof(scalar) == 4 ? 16 : 8;
    auto load_group = [&](scalar(&v)[GS], int c0, int j) {
        const int last    = C - c0 - j - 1; // >= 0
        const scalar *col = E + (int64_t)(c0 + j) * len + row;
#pragma unroll
        for (int u = 0; u < GS; u++)
            v[u] = stream_load(col + (int64_t)(u < last ? u : last) * len);
    };
    auto advance = [&](int &c0, int &j) {
        j += GS;
        if (j >= 64 || c0 + j >= C) {
            c0 += WAVES * 64;
            j = 0;
        }
    };
    scalar z = scalar(0), mine = scalar(0);
    int md = -1, nc = 0;
    bool mir = false;
    auto tile_setup = [&](int c0) { // gathered coefficients and mirror slots of the (up to) 64 columns of a tile
        nc  = (C - c0) < 64 ? (C - c0) : 64;
        if constexpr (FWD)
            z = lane < nc ? *expand_operand(A, zidx[c0 + lane], 1) : scalar(0);
        md  = lane < nc ? mdst[c0 + lane] : -1;
        mir = __any(md >= 0); // wave-uniform: tiles without mirrored columns (diagonal leaves, off-diagonal stripes) skip the reductions
    };
    auto process = [&](const scalar(&v)[GS], int jg) {
        if constexpr (FWD)
#pragma unroll
            for (int u = 0; u < GS; u++)
                acc = hmx_fma(v[u], readlane_val(z, (jg + u) & 63), acc);
        if (mir)
#pragma unroll
            for (int h = 0; h < GS; h += 8) {
            const int j = jg + h;
            if (j >= nc)
                break;
            scalar p[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                p[u] = (herm ? hmx_conj(v[h + u]) : v[h + u]) * xr;
            // every lane of lane group s = lane >> 3 now holds the sum of column j + s; lane 8 s + g keeps the one of group g = j / 8,
            // so that after the tile's last group an 8 x 8 transposition of the lane index (one ds_bpermute) puts the sum of
            // column c into lane c: ONE coalesced store per tile instead of eight 8-lane stores.  (What the stores cost is the write
            // stream itself: on MI355X 1.6 % of written bytes takes 15-30 % off a streaming read, tools/read_write_mix.hip; staging
            // the sums in LDS until the end of the workgroup, 128-byte aligned runs or non-temporal stores change nothing.)
            const scalar r = reduce8(p, lane);
            mine           = hmx_select((lane & 7) == (j >> 3), r, mine);
            if (j + 8 >= nc) {
                const scalar t = hmx_shfl(mine, 8 * (lane & 7) + (lane >> 3));
                if (md >= 0)
                    S.W[md] = t;
            }
        }
    };
    // order inside a step: (tile setup, its own dependent loads) -> prefetch of the next group -> arithmetic on the current
    // one; the prefetch is unconditional (past the end it re-reads the current group) so that exactly eight newer loads are
    // outstanding whenever a group is consumed
    scalar va[GS], vb[GS];
    int c0 = wv * 64, j = 0;
    if (c0 < C)
        load_group(va, c0, 0);
    while (c0 < C) {
        int n0 = c0, nj = j;
        advance(n0, nj);
        bool more = n0 < C;
        if (j == 0)
            tile_setup(c0);
        load_group(vb, more ? n0 : c0, more ? nj : j);
        process(va, j);
        if (!more)
            break;
        c0 = n0, j = nj;
        advance(n0, nj);
        more = n0 < C;
        if (j == 0)
            tile_setup(c0);
        load_group(va, more ? n0 : c0, more ? nj : j);
        process(vb, j);
        c0 = n0, j = nj;
    }
    if constexpr (FWD) {
        part[wv][lane] = active ? acc : scalar(0);
        __syncthreads();
        if (wv == 0 && active) {
            scalar s = part[0][lane];
#pragma unroll
            for (int k = 1; k < WAVES; k++)
                s += part[k][lane];
            scalar *yo = A.y + A.range_off[R] + lane;
            *yo        = hmx_is_zero(A.beta) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
        }
    }
}

// Second sweep over the R-streams, owner-computes: y_s += V^T a' (conjugated for 'H').  The target rows are cut into intervals of
// SYM_IR rows; one workgroup per interval walks the (parts of) (source piece, column chunk) tasks whose rows lie in it -- an R-stream
// chunk is row-major, so any row sub-range of a piece is one contiguous block -- wave w taking the sub-tasks w, w + WAVES, ... of the
// interval's list.  Per sub-task: lane = column pair, eight (fp32: sixteen) rows per group with the next group's loads in flight,
// reduce8 over the rows, and after 64 rows one transposing ds_bpermute that puts the sum of row i into lane i, which adds it to the
// wave's slice of an LDS accumulator.  At the end the waves' slices are added in order, the interval's dense mirrored contributions
// (column sums expand_sym_kernel left in EW, found through the level-major index) are added, and y is updated ONCE per row: no
// partial row sums go through HBM, no folding kernel.  Fixed order everywhere: bit-reproducible.
#ifndef HMX_SYM_IR
#define HMX_SYM_IR 256
#endif
#ifndef HMX_SYM_WAVES
#define HMX_SYM_WAVES 4
#endif
constexpr int SYM_IR    = HMX_SYM_IR; // rows per interval
constexpr int SYM_WAVES = HMX_SYM_WAVES;
struct RowSymArgs {
    const scalar *stream;
    const int32_t *task_range, *task_chunk;
    const int32_t *range_len, *range_cols, *range_cw;
    const int64_t *range_base, *range_colbase;
    const int32_t *coef;      // per R column: slot of a'[col] in W, -1: not a mirrored column
    const int32_t *order;     // launch position -> interval (heaviest first)
    const int64_t *sub_ptr;   // per interval: its sub-tasks [sub_ptr[I], sub_ptr[I + 1])
    const int32_t *sub_task, *sub_row0, *sub_nrows, *sub_dst; // task, first row inside the piece, rows, first row inside the interval
    const scalar *W;          // [a' | EW]
    const int32_t *fidx;      // dense mirrored contributions of output row j: W[fidx[k * n + j]], k < count[j]
    const int32_t *count;
    scalar *y;
    scalar alpha;
    int n;                    // rows of the operator (stride of fidx)
    int herm;
    scalar beta;              // accumulate = 0 (transposed product on the stored data: this sweep owns y): y = alpha * sums + beta * y
    int accumulate;           // 1: y += alpha * sums (the forward sweep of the symmetric product has written y already)
};
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) void rowsym_kernel(RowSymArgs A) {
    __shared__ scalar acc[WAVES][SYM_IR];
    const int I    = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int r = lane; r < SYM_IR; r += WAVE)
        acc[wv][r] = scalar(0);
    const bool herm = A.herm != 0;
    constexpr int GS = sizeof(scalar2) <= 8 ? 16 : 8;
    for (int64_t q = A.sub_ptr[I] + wv; q < A.sub_ptr[I + 1]; q += WAVES) {
        const int task = A.sub_task[q], row0 = A.sub_row0[q], len = A.sub_nrows[q];
        scalar *dst = &acc[wv][A.sub_dst[q]];
        const int S = A.task_range[task], ch = A.task_chunk[task];
        const int plen = A.range_len[S], C = A.range_cols[S], cw = A.range_cw[S];
        int w = C - ch * cw;
        w     = w > cw ? cw : w;
        const int wp      = hmx_wp(w);
        const int col0 = HMX_COL0(lane), col1 = HMX_COL1(lane);
        const scalar *src = A.stream + A.range_base[S] + (int64_t)ch * plen * cw + (int64_t)row0 * wp;
        const int64_t cb  = A.range_colbase[S] + ch * cw;
        scalar c0 = scalar(0), c1 = scalar(0);
        if (col0 < w) {
            const int d = A.coef[cb + col0];
            c0          = d >= 0 ? A.W[d] : scalar(0);
        }
        if (col1 < w) {
            const int d = A.coef[cb + col1];
            c1          = d >= 0 ? A.W[d] : scalar(0);
        }
        scalar mine = scalar(0);
        // always GS loads, no branches: rows beyond the sub-task re-read its last row (their sums are dropped), lanes beyond the
        // chunk read column 0 and multiply it with their zero coefficients
        auto load_rows = [&](scalar2(&e)[GS], int i0) {
#pragma unroll
            for (int u = 0; u < GS; u++) {
                const int i = i0 + u < len ? i0 + u : len - 1;
                e[u]        = load_pair(src + (int64_t)i * wp, col0, col1, wp);
            }
        };
        auto process = [&](const scalar2(&e)[GS], int ig) {
#pragma unroll
            for (int h = 0; h < GS; h += 8) {
                const int i0 = ig + h;
                if (i0 >= len)
                    break;
                scalar v[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    v[u] = herm ? hmx_fma(hmx_conj(e[h + u].x), c0, hmx_conj(e[h + u].y) * c1) : hmx_fma(e[h + u].x, c0, e[h + u].y * c1);
                // as in expand_sym_kernel: lane 8 s + g keeps the sum of row 64 b + 8 g + s; after 64 rows one transposing ds_bpermute
                const scalar r = reduce8(v, lane);
                const int g    = (i0 >> 3) & 7;
                mine           = hmx_select((lane & 7) == g, r, mine);
                if (g == 7 || i0 + 8 >= len) {
                    const scalar t = hmx_shfl(mine, 8 * (lane & 7) + (lane >> 3));
                    const int i    = (i0 & ~63) + lane;
                    if (i < len)
                        dst[i] += t; // this wave's slice: no other wave touches it, the sub-tasks of a wave run one after the other
                }
            }
        };
        scalar2 ea[GS], eb[GS];
        load_rows(ea, 0);
        for (int i0 = 0; i0 < len; i0 += 2 * GS) { // unconditional prefetches (clamped to the last row): exactly GS newer loads outstanding at every use
            load_rows(eb, i0 + GS);
            process(ea, i0);
            load_rows(ea, i0 + 2 * GS);
            if (i0 + GS < len)
                process(eb, i0 + GS);
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < SYM_IR; r += WAVES * WAVE) {
        const int j = I * SYM_IR + r;
        if (j >= A.n)
            break;
        scalar sum = acc[0][r];
#pragma unroll
        for (int k = 1; k < WAVES; k++)
            sum += acc[k][r];
        const int cnt = A.count[j];
        for (int k = 0; k < cnt; k++)
            sum += A.W[A.fidx[(int64_t)k * A.n + j]];
        if (A.accumulate)
            A.y[j] += A.alpha * sum;
        else
            A.y[j] = hmx_is_zero(A.beta) ? A.alpha * sum : A.alpha * sum + A.beta * A.y[j];
    }
}

// a'[dst] = sum_i W[list[lp + i] + k]: the partial column sums of a mirrored low-rank leaf that spans several row ranges, one list
// entry (position of the leaf's column group in EW) per range
struct CombineListArgs {
    const int32_t *dst, *lp, *count, *k, *list;
    scalar *W;
    int n;
};
__global__ void combine_list_kernel(CombineListArgs A) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.n)
        return;
    const int32_t *l = A.list + A.lp[e];
    const int cnt = A.count[e], k = A.k[e];
    scalar s = scalar(0);
    for (int i = 0; i < cnt; i++)
        s += A.W[l[i] + k];
    A.W[A.dst[e]] = s;
}
__global__ __launch_bounds__(256) void combine_list_wave_kernel(CombineListArgs A) {
    const int e = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (e >= A.n)
        return;
    const int lane   = threadIdx.x & 63;
    const int32_t *l = A.list + A.lp[e];
    const int cnt = A.count[e], k = A.k[e];
    scalar s = scalar(0);
    for (int i = lane; i < cnt; i += 64)
        s += A.W[l[i] + k];
    s = wave_sum_dpp(s);
    if (lane == 0)
        A.W[A.dst[e]] = s;
}


// ---------------------------------------------------------------------------------------------
// Several right-hand sides on the stored data: partial sums live in SW16 = [slot][SWW] (the slots of the single-vector product, SWW
// coefficients each: 16 real or 8 complex right-hand sides per sweep -- 128 resp. 64 bytes per slot in single, twice that in double precision).
// ---------------------------------------------------------------------------------------------
constexpr int SWW = HMX_COMPLEX ? 8 : 16;
// a'[dst][0..SWW) = sum_i SW16[list[lp + i] + k][0..SWW): the partial column sums of a mirrored low-rank leaf that spans several row ranges.
// One wave per entry for the entries with many partial sums (the first `A.n` entries handed to this kernel): SWW lanes take the right-hand
// sides, the 64 / SWW lane groups every (64 / SWW)-th partial sum; fixed order
__global__ __launch_bounds__(256) void combine_list_mu_wave_kernel(CombineListArgs A) {
    const int e = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (e >= A.n)
        return;
    constexpr int NG = 64 / SWW;
    const int lane = threadIdx.x & 63, m = lane % SWW, g = lane / SWW;
    const int32_t *l = A.list + A.lp[e];
    const int cnt = A.count[e], k = A.k[e];
    scalar s0 = scalar(0), s1 = scalar(0);
    int i = g;
    for (; i + NG < cnt; i += 2 * NG) { // two loads in flight per lane
        s0 += A.W[(int64_t)(l[i] + k) * SWW + m];
        s1 += A.W[(int64_t)(l[i + NG] + k) * SWW + m];
    }
    if (i < cnt)
        s0 += A.W[(int64_t)(l[i] + k) * SWW + m];
    scalar s = s0 + s1;
#pragma unroll
    for (int o = SWW; o < 64; o <<= 1)
        s += hmx_shfl_xor(s, o);
    if (g == 0)
        A.W[(int64_t)A.dst[e] * SWW + m] = s;
}
// ... one thread per (entry, right-hand side) for the rest
__global__ void combine_list_mu_kernel(CombineListArgs A) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = (int)(id / SWW), m = (int)(id % SWW);
    if (e >= A.n)
        return;
    const int32_t *l = A.list + A.lp[e];
    const int cnt = A.count[e], k = A.k[e];
    // four independent sums keep four (index, value) load pairs in flight; fixed order
    scalar s0 = scalar(0), s1 = scalar(0), s2 = scalar(0), s3 = scalar(0);
    int i = 0;
    for (; i + 4 <= cnt; i += 4) {
        const int32_t l0 = l[i], l1 = l[i + 1], l2 = l[i + 2], l3 = l[i + 3];
        s0 += A.W[(int64_t)(l0 + k) * SWW + m];
        s1 += A.W[(int64_t)(l1 + k) * SWW + m];
        s2 += A.W[(int64_t)(l2 + k) * SWW + m];
        s3 += A.W[(int64_t)(l3 + k) * SWW + m];
    }
    for (; i < cnt; i++)
        s0 += A.W[(int64_t)(l[i] + k) * SWW + m];
    A.W[(int64_t)A.dst[e] * SWW + m] = (s0 + s1) + (s2 + s3);
}

// The fused symmetric / Hermitian product (expand_sym_kernel, rowsym_kernel) for MU right-hand sides at a time on the VALU: what complex
// coefficients run on the stored triangle and, with every leaf mirrored and nothing applied forward (FWD = false), in the transposed product on
// the stored data (the reference: the mirror pass of hmatrix/linalg/add_hmatrix_matrix_product_row_major.hpp:100-106,160-170 with the complex
// symm / hemm leaf products of matrix/linalg/add_matrix_matrix_product_row_major.hpp:113-139).  Same sweeps, same slots, same fixed order as for
// one vector; per column of the E-streams MU forward FMAs and MU column sums (reduce8 per right-hand side), per row of the R-streams MU row
// sums.  The streams are read once for the whole group where the fallback before round 5 ran one single-vector product per right-hand side.
template <int WAVES, int MU, bool FWD = true>
__global__ __launch_bounds__(WAVES *WAVE) void expand_sym_mu_kernel(ExpandSymArgs S, int mu, int cbase, int nrhs) {
    const ExpandArgs &A = S.X;
    __shared__ scalar part[FWD ? WAVES : 1][FWD ? WAVE : 1][FWD ? MU : 1];
    const int R = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = A.range_len[R], C = A.range_cols[R];
    const scalar *E     = A.stream + A.range_base[R];
    const int32_t *zidx = A.z_idx + A.range_colbase[R];
    const int32_t *mdst = S.mdst + A.range_colbase[R];
    const bool active   = lane < len;
    const int row       = active ? lane : 0;
    const bool herm     = S.herm != 0;
    scalar xr[MU], acc[MU]; // the input at this lane's row (idle lanes and missing right-hand sides: exact zeros), the forward sums
#pragma unroll
    for (int j = 0; j < MU; j++) {
        xr[j]  = (active && j < nrhs) ? S.xrow[(int64_t)(A.range_off[R] + lane) * mu + cbase + j] : scalar(0);
        acc[j] = scalar(0);
    }
    constexpr int GS = 8;
    auto load_group = [&](scalar(&v)[GS], int c0, int j) { // always eight loads, no branches: see expand_sym_kernel
        const int last    = C - c0 - j - 1; // >= 0
        const scalar *col = E + (int64_t)(c0 + j) * len + row;
#pragma unroll
        for (int u = 0; u < GS; u++)
            v[u] = stream_load(col + (int64_t)(u < last ? u : last) * len);
    };
    auto advance = [&](int &c0, int &j) {
        j += GS;
        if (j >= 64 || c0 + j >= C) {
            c0 += WAVES * 64;
            j = 0;
        }
    };
    scalar z[MU], mine[MU];
#pragma unroll
    for (int j = 0; j < MU; j++)
        z[j] = mine[j] = scalar(0);
    int md = -1, nc = 0;
    bool mir = false;
    auto tile_setup = [&](int c0) {
        nc = (C - c0) < 64 ? (C - c0) : 64;
        if constexpr (FWD) {
            const scalar *zr = expand_operand(A, zidx[c0 + (lane < nc ? lane : 0)], mu) + cbase;
#pragma unroll
            for (int j = 0; j < MU; j++)
                z[j] = (lane < nc && j < nrhs) ? zr[j < nrhs ? j : 0] : scalar(0);
        }
        md  = lane < nc ? mdst[c0 + lane] : -1;
        mir = __any(md >= 0);
    };
    auto process = [&](const scalar(&v)[GS], int jg) {
        if constexpr (FWD)
#pragma unroll
            for (int u = 0; u < GS; u++)
#pragma unroll
                for (int j = 0; j < MU; j++)
                    acc[j] = hmx_fma(v[u], readlane_val(z[j], (jg + u) & 63), acc[j]);
        if (mir && jg < nc) {
#pragma unroll
            for (int j = 0; j < MU; j++) {
                scalar p[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    p[u] = (herm ? hmx_conj(v[u]) : v[u]) * xr[j];
                const scalar r = reduce8(p, lane); // lane 8 s + g of lane group s = lane >> 3 keeps the sum of column jg + s: see expand_sym_kernel
                mine[j]        = hmx_select((lane & 7) == (jg >> 3), r, mine[j]);
            }
            if (jg + 8 >= nc) {
#pragma unroll
                for (int j = 0; j < MU; j++) {
                    const scalar t = hmx_shfl(mine[j], 8 * (lane & 7) + (lane >> 3)); // the sum of column c in lane c
                    if (md >= 0 && j < nrhs)
                        S.W[(int64_t)md * SWW + j] = t;
                }
            }
        }
    };
    scalar va[GS], vb[GS];
    int c0 = wv * 64, j = 0;
    if (c0 < C)
        load_group(va, c0, 0);
    while (c0 < C) {
        int n0 = c0, nj = j;
        advance(n0, nj);
        bool more = n0 < C;
        if (j == 0)
            tile_setup(c0);
        load_group(vb, more ? n0 : c0, more ? nj : j);
        process(va, j);
        if (!more)
            break;
        c0 = n0, j = nj;
        advance(n0, nj);
        more = n0 < C;
        if (j == 0)
            tile_setup(c0);
        load_group(va, more ? n0 : c0, more ? nj : j);
        process(vb, j);
        c0 = n0, j = nj;
    }
    if constexpr (FWD) {
#pragma unroll
        for (int j = 0; j < MU; j++)
            part[wv][lane][j] = active ? acc[j] : scalar(0);
        __syncthreads();
        for (int e = threadIdx.x; e < len * MU; e += WAVES * WAVE) {
            const int i = e / MU, jj = e - i * MU;
            if (jj >= nrhs)
                continue;
            scalar s = part[0][i][jj];
#pragma unroll
            for (int k = 1; k < WAVES; k++)
                s += part[k][i][jj];
            scalar *yo = A.y + (int64_t)(A.range_off[R] + i) * mu + cbase + jj;
            *yo        = hmx_is_zero(A.beta) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
        }
    }
}

// second sweep over the R-streams for MU right-hand sides (rowsym_kernel's scheme on intervals of SYM_IR_MU rows: one workgroup per
// interval, wave w its sub-tasks w, w + WAVES, ..., row sums folded in LDS, the dense mirrored contributions and the y update at the end)
constexpr int SYM_IR_MU = 64;
template <int WAVES, int MU>
__global__ __launch_bounds__(WAVES *WAVE) void rowsym_mu_kernel(RowSymArgs A, const scalar *W16, int mu, int cbase, int nrhs) {
    __shared__ scalar acc[WAVES][SYM_IR_MU][MU];
    const int I    = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int r = lane; r < SYM_IR_MU * MU; r += WAVE)
        (&acc[wv][0][0])[r] = scalar(0);
    const bool herm = A.herm != 0;
    constexpr int GS = 8;
    for (int64_t q = A.sub_ptr[I] + wv; q < A.sub_ptr[I + 1]; q += WAVES) {
        const int task = A.sub_task[q], row0 = A.sub_row0[q], len = A.sub_nrows[q];
        scalar(*dst)[MU] = &acc[wv][A.sub_dst[q]];
        const int S = A.task_range[task], ch = A.task_chunk[task];
        const int plen = A.range_len[S], C = A.range_cols[S], cw = A.range_cw[S];
        int w = C - ch * cw;
        w     = w > cw ? cw : w;
        const int wp      = hmx_wp(w);
        const int col0 = HMX_COL0(lane), col1 = HMX_COL1(lane);
        const scalar *src = A.stream + A.range_base[S] + (int64_t)ch * plen * cw + (int64_t)row0 * wp;
        const int64_t cb  = A.range_colbase[S] + ch * cw;
        const int d0 = col0 < w ? A.coef[cb + col0] : -1, d1 = col1 < w ? A.coef[cb + col1] : -1;
        scalar c0[MU], c1[MU], mine[MU];
#pragma unroll
        for (int j = 0; j < MU; j++) {
            c0[j]   = (d0 >= 0 && j < nrhs) ? W16[(int64_t)d0 * SWW + j] : scalar(0);
            c1[j]   = (d1 >= 0 && j < nrhs) ? W16[(int64_t)d1 * SWW + j] : scalar(0);
            mine[j] = scalar(0);
        }
        auto load_rows = [&](scalar2(&e)[GS], int i0) {
#pragma unroll
            for (int u = 0; u < GS; u++) {
                const int i = i0 + u < len ? i0 + u : len - 1;
                e[u]        = load_pair(src + (int64_t)i * wp, col0, col1, wp);
            }
        };
        auto process = [&](const scalar2(&e)[GS], int i0) {
            if (i0 >= len)
                return;
            const int g = (i0 >> 3) & 7;
#pragma unroll
            for (int j = 0; j < MU; j++) {
                scalar v[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    v[u] = herm ? hmx_fma(hmx_conj(e[u].x), c0[j], hmx_conj(e[u].y) * c1[j]) : hmx_fma(e[u].x, c0[j], e[u].y * c1[j]);
                const scalar r = reduce8(v, lane); // lane 8 s + g keeps the sum of row 64 b + 8 g + s: see rowsym_kernel
                mine[j]        = hmx_select((lane & 7) == g, r, mine[j]);
            }
            if (g == 7 || i0 + 8 >= len) {
                const int i = (i0 & ~63) + lane;
#pragma unroll
                for (int j = 0; j < MU; j++) {
                    const scalar t = hmx_shfl(mine[j], 8 * (lane & 7) + (lane >> 3));
                    if (i < len)
                        dst[i][j] += t; // this wave's slice: no other wave touches it
                }
            }
        };
        scalar2 ea[GS], eb[GS];
        load_rows(ea, 0);
        for (int i0 = 0; i0 < len; i0 += 2 * GS) { // unconditional prefetches (clamped to the last row)
            load_rows(eb, i0 + GS);
            process(ea, i0);
            load_rows(ea, i0 + 2 * GS);
            process(eb, i0 + GS);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < SYM_IR_MU * MU; e += WAVES * WAVE) {
        const int r = e / MU, jj = e - r * MU;
        const int j = I * SYM_IR_MU + r;
        if (j >= A.n || jj >= nrhs)
            continue;
        scalar sum = acc[0][r][jj];
#pragma unroll
        for (int k = 1; k < WAVES; k++)
            sum += acc[k][r][jj];
        const int cnt = A.count[j];
        for (int k = 0; k < cnt; k++)
            sum += W16[(int64_t)A.fidx[(int64_t)k * A.n + j] * SWW + jj];
        scalar *yo = A.y + (int64_t)j * mu + cbase + jj;
        if (A.accumulate)
            *yo += A.alpha * sum;
        else
            *yo = hmx_is_zero(A.beta) ? A.alpha * sum : A.alpha * sum + A.beta * (*yo);
    }
}

#if !HMX_COMPLEX
// ---------------------------------------------------------------------------------------------
// Several right-hand sides on the STORED TRIANGLE (symmetric storage, real coefficients): the fused product above for groups of up to 16
// right-hand sides on the matrix cores.  The reference runs the mirror pass on the same leaves for any number of right-hand sides
// (hmatrix/linalg/add_hmatrix_matrix_product_row_major.hpp:100-106,160-170; symm for the diagonal leaves,
// matrix/linalg/add_matrix_matrix_product_row_major.hpp:87-106); until round 4 such products ran here on an expanded copy of the
// operator (twice the footprint, twice the traffic).  Three sweeps, as for one vector:
//   reduce_mfma16s_kernel           a = V X_s over the R-streams (the ordinary multi-RHS reduce stage)
//   expand_sym_mfma16_kernel        ONE pass over the E-streams: per 64 x 16 stream tile the forward product Y_t += E Z (tile = A operand,
//                                   rows on the M index) AND the mirrored column sums EW = E^T X_t (the same tile as A operand with its
//                                   columns on the M index and the rows contracted) -- 16 + 16 MFMAs per tile.  An MFMA contracts over
//                                   the lane bits 4-5 of both operands, so the two products need the tile in two lane layouts: the
//                                   forward operands come straight from the registers the loads filled (lane = row) by a 4 x 4
//                                   transposition between register index and lane quarter (v_permlane32_swap + v_permlane16_swap: no LDS),
//                                   the mirrored ones from a wave-private LDS copy [row][column] written with 16-byte stores.
//   combine_list_mu_kernel          a' of the leaves that span several row ranges
//   rowsym_mfma16_kernel            second pass over the R-streams, Y_s += V^T a': one WAVE owns 64 output rows (accumulators in
//                                   registers, nothing to fold between waves), stream tiles 16 rows x 64 columns staged through LDS
//                                   transposed and swizzled so that stores and operand reads both run at two lanes per bank.
// Partial sums live in SW16 = [slot][16] (the slots of the single-vector product, 16 values each).  Fixed summation order: bit-reproducible.
// ---------------------------------------------------------------------------------------------
// FWD = false: the mirrored column sums only (transposed product of an ordinary operator on its stored data, several right-hand sides)
template <int WAVES, bool FWD = true>
__global__ __launch_bounds__(WAVES *WAVE) void expand_sym_mfma16_kernel(ExpandSymArgs S, int mu, int cbase, int nrhs) {
    const ExpandArgs &A = S.X;
#ifndef HMX_SYMMU_PT
#define HMX_SYMMU_PT 24
#endif
    constexpr int PT = HMX_SYMMU_PT; // row pitch of the mirrored tile [64 rows][16 columns]: operand reads (16 columns x 4 rows) at two lanes per bank
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 64 * PT > WAVES * WAVE * 16 ? WAVES * 64 * PT : WAVES * WAVE * 16];
    const int R = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int len = A.range_len[R], C = A.range_cols[R];
    const real *E       = A.stream + A.range_base[R];
    const int32_t *zidx = A.z_idx + A.range_colbase[R];
    const int32_t *mdst = S.mdst + A.range_colbase[R];
    const int m = lane & 15, kk = lane >> 4;
    real *tile    = lds + wv * 64 * PT;
    const int row = lane < len ? lane : len - 1; // idle lanes re-read the last row: forward, they only reach accumulator rows that are never stored; mirrored, their X_t operand is zero
    const int mo  = cbase + (m < nrhs ? m : 0); // ragged group: see expand_mfma16s_kernel
    // B operand of the mirrored product, constant over the range: X_t[row 4h + kk][rhs m] for the 16 k-steps h (zero beyond the range)
    real xt[16];
#pragma unroll
    for (int h = 0; h < 16; h++) {
        const int r   = 4 * h + kk;
        const real xv = S.xrow[(int64_t)(A.range_off[R] + (r < len ? r : len - 1)) * mu + mo];
        xt[h]         = r < len ? xv : real(0);
    }
    acc4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
        acc[t] = acc4{0, 0, 0, 0};
    // the wave's columns as one sequence of 16-column steps, three stages in flight, every load unconditional: see expand_mfma16s_kernel
    const int ntile_all = (C + 63) >> 6;
    int n = 0;
    if (wv < ntile_all) {
        n = 4 * ((ntile_all - 1 - wv) / WAVES + 1);
        if ((ntile_all - 1 - wv) % WAVES == 0)
            n -= 4 - ((C - 64 * (ntile_all - 1) + 15) >> 4);
    }
    auto col_of = [&](int s) { return (((s >> 2) * WAVES + wv) << 6) + ((s & 3) << 4); };
    struct Idx {
        int z, md; // lane l: Z index and mirror slot of column col_of(s) + (l & 15)
    };
    auto load_idx = [&](int s) {
        const int c  = col_of(s < n ? s : n - 1) + m;
        const int cc = c < C ? c : C - 1;
        Idx ix;
        ix.z  = FWD ? zidx[cc] : 0;
        ix.md = mdst[cc];
        return ix;
    };
    auto gathers = [&](real(&b)[4], const Idx &ix) {
        if constexpr (FWD)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int zc = __shfl(ix.z, 4 * g + kk, WAVE);
                b[g]         = expand_operand(A, zc, mu)[mo];
            }
    };
    auto load_cols = [&](real(&v)[16], int s) { // 16 whole columns, clamped to the range's last one (zero operand there, sums never stored)
        const int c = col_of(s < n ? s : n - 1);
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int col = c + u < C ? c + u : C - 1;
            v[u]          = stream_load(E + (int64_t)col * len + row);
        }
    };
    // one step = 16 columns: mirrored column sums (if any of the 16 is a mirrored column), then the forward product
    auto apply = [&](real(&v)[16], const real(&braw)[4], int mdi, int s) {
        const int c  = col_of(s);
        const int md = (c + m < C) ? mdi : -1;
        if (__any(md >= 0)) { // wave-uniform: steps without mirrored columns (diagonal leaves, the other ranks' columns of a row-partitioned operator) skip all of it
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 16; u++)
                tile[lane * PT + u] = v[u]; // 16 consecutive elements per lane: 16-byte stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            real ta[16];
#pragma unroll
            for (int h = 0; h < 16; h++)
                ta[h] = tile[(4 * h + kk) * PT + m];
            acc4 am = acc4{0, 0, 0, 0};
#pragma unroll
            for (int h = 0; h < 16; h++)
                am = mfma16(ta[h], xt[h], am); // A[m = column][k = row 4h + kk], B[k][n = rhs]
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int d = __shfl(md, mfma16_row(real(0), lane, j), WAVE);
                if (d >= 0)
                    S.W[(int64_t)d * 16 + m] = am[j];
            }
        }
        // forward: a[g][t] (row 16 t + m, column 4 g + kk) = register 4 g + kk of lane quarter t -- a 4 x 4 transposition per column group
        if constexpr (FWD) {
            real b[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                b[g] = (c + 4 * g + kk < C) ? braw[g] : real(0);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                lane_swap32(v[4 * g + 0], v[4 * g + 2]);
                lane_swap32(v[4 * g + 1], v[4 * g + 3]);
                lane_swap16(v[4 * g + 0], v[4 * g + 1]);
                lane_swap16(v[4 * g + 2], v[4 * g + 3]);
            }
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[t] = mfma16(v[4 * g + t], b[g], acc[t]);
        }
    };
    if (n > 0) {
        real v0[16], v1[16], b0[4], b1[4];
        Idx i0 = load_idx(0), i1 = load_idx(1);
        gathers(b0, i0);
        load_cols(v0, 0);
        HMX_SCHED_FENCE();
        for (int s = 0; s < n; s += 2) {
            const int md0 = i0.md;
            i0 = load_idx(s + 2);
            gathers(b1, i1);
            load_cols(v1, s + 1);
            HMX_SCHED_FENCE();
            apply(v0, b0, md0, s);
            HMX_SCHED_FENCE();
            const int md1 = i1.md;
            i1 = load_idx(s + 3);
            gathers(b0, i0);
            load_cols(v0, s + 2);
            HMX_SCHED_FENCE();
            if (s + 1 < n)
                apply(v1, b1, md1, s + 1);
            HMX_SCHED_FENCE();
        }
    }
    if constexpr (!FWD)
        return;
    // forward result: the waves' accumulators folded through LDS as in expand_mfma16s_kernel
    real(*red)[WAVE][16] = reinterpret_cast<real(*)[WAVE][16]>(lds);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            red[wv][16 * t + mfma16_row(real(0), lane, j)][m] = acc[t][j];
    __syncthreads();
    for (int e = threadIdx.x; e < len * 16; e += WAVES * WAVE) {
        const int i = e >> 4, c = e & 15;
        if (c >= nrhs)
            continue;
        real s = red[0][i][c];
#pragma unroll
        for (int w = 1; w < WAVES; w++)
            s += red[w][i][c];
        real *yo = A.y + (int64_t)(A.range_off[R] + i) * mu + cbase + c;
        *yo      = A.beta == real(0) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
    }
}

// Second pass over the R-streams for up to 16 right-hand sides.  Interval = 64 output rows = one wave; its sub-tasks are the parts of the
// (source piece, column chunk) tasks whose rows lie in the interval.  Per sub-task and half of the chunk's (<= 128) columns: the B operands
// a'[column][rhs] of the 16 k-steps are gathered once, then every 16-row tile of the interval the sub-task touches is loaded (whole rows: two
// rows of 64 columns per wave-wide load), staged transposed in LDS and multiplied -- 16 MFMAs per 16 x 64 tile; rows of the tile that are
// not the sub-task's are dropped when the tile's result is added to the interval's accumulators.
struct RowSymMuArgs {
    RowSymArgs A;         // (sub_* / order refer to the 64-row intervals)
    const scalar *W16;    // [slot][16]
    int zero_slot;        // a slot whose 16 values are zero
    int nint;
};
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) void rowsym_mfma16_kernel(RowSymMuArgs P, int mu, int cbase, int nrhs) {
    const RowSymArgs &A = P.A;
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 64 * 16];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pos = blockIdx.x * WAVES + wv;
    if (pos >= P.nint)
        return; // (no workgroup barrier below: the waves are independent)
    const int I  = A.order[pos];
    const int m = lane & 15, kk = lane >> 4;
    // [64 columns][16 rows], element (row i, column c) at 16 (c ^ ((c >> 1) & 1)) + (i ^ ((c >> 1) & 15)): the stores of a load's two
    // rows x 64 columns and the operand reads of 16 rows x 4 columns both touch every bank exactly twice
    real *tile = lds + wv * 64 * 16;
    acc4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
        acc[t] = acc4{0, 0, 0, 0};
    const int lrow = lane >> 5, lc = 2 * (lane & 31); // loads: lane = (row parity, column pair) of a 2-row x 64-column slab
    // what a sub-task needs, fetched one sub-task ahead: the chain sub-task -> task -> range -> geometry is a dozen dependent loads
    struct Sub {
        const real *src;
        int w, wp, n, dst;
        int32_t dlo, dhi; // slots of a' for the chunk's columns lane and 64 + lane (-1: not a mirrored leaf's column / beyond the chunk)
    };
    auto fetch = [&](int64_t q) {
        Sub s;
        const int task = A.sub_task[q], row0 = A.sub_row0[q];
        s.n   = A.sub_nrows[q];
        s.dst = A.sub_dst[q];
        const int S = A.task_range[task], ch = A.task_chunk[task];
        const int plen = A.range_len[S], C = A.range_cols[S], cw = A.range_cw[S];
        int w = C - ch * cw;
        w     = w > cw ? cw : w;
        s.w   = w;
        s.wp  = hmx_wp(w);
        s.src = A.stream + A.range_base[S] + (int64_t)ch * plen * cw + (int64_t)row0 * s.wp;
        const int64_t cb = A.range_colbase[S] + ch * cw;
        const int32_t a = A.coef[cb + (lane < w ? lane : 0)], b = A.coef[cb + (64 + lane < w ? 64 + lane : 0)];
        s.dlo = lane < w ? a : -1;
        s.dhi = 64 + lane < w ? b : -1;
        return s;
    };
    // a tile = 16 interval rows x 64 columns of one sub-task: rows clamped into the sub-task's (the others are dropped when the result is added)
    auto load_tile = [&](scalar2(&v)[8], const Sub &s, int c0, int t) {
        const int cl = c0 + lc < s.wp ? c0 + lc : 0; // lanes beyond the chunk re-read its first pair (their operand is zero)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int r = 16 * t + 2 * u + lrow - s.dst;
            r     = r < 0 ? 0 : (r >= s.n ? s.n - 1 : r);
            v[u]  = stream_load(reinterpret_cast<const scalar2 *>(s.src + (int64_t)r * s.wp + cl));
        }
    };
    // B operands of a segment (sub-task, 64-column half): a'[column c0 + 4 h + kk][rhs m]; columns beyond the chunk and columns that are
    // no mirrored leaf's read a zero slot
    auto gather_b = [&](real(&b)[16], const Sub &s, int c0) {
#pragma unroll
        for (int h = 0; h < 16; h++) {
            const int d = __shfl(c0 ? s.dhi : s.dlo, 4 * h + kk, WAVE);
            b[h]        = P.W16[(int64_t)(d >= 0 ? d : P.zero_slot) * 16 + m];
        }
    };
    const int64_t q0 = A.sub_ptr[I], q1 = A.sub_ptr[I + 1];
    Sub cur{};
    if (q0 < q1)
        cur = fetch(q0);
    // one tile: staged transposed in LDS, 16 MFMAs, rows that are not the sub-task's dropped when the result joins the accumulators
    auto tile_product = [&](const scalar2(&v)[8], const real(&b)[16], const Sub &s, int t) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = 2 * u + lrow, sw = lane & 15, fl = lane & 1; // (c >> 1) & 15 and (c >> 1) & 1 of both columns lc, lc + 1
            tile[16 * (lc ^ fl) + (i ^ sw)]       = v[u].x;
            tile[16 * ((lc + 1) ^ fl) + (i ^ sw)] = v[u].y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        real ta[16];
#pragma unroll
        for (int h = 0; h < 16; h++) {
            const int c = 4 * h + kk;
            ta[h]       = tile[16 * (c ^ ((c >> 1) & 1)) + (m ^ ((c >> 1) & 15))];
        }
        acc4 tm = acc4{0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < 16; h++)
            tm = mfma16(ta[h], b[h], tm); // A[m = row][k = column c], B[k][n = rhs]
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i    = 16 * t + mfma16_row(real(0), lane, j);
            const real add = (i >= s.dst && i < s.dst + s.n) ? tm[j] : real(0);
#pragma unroll
            for (int tt = 0; tt < 4; tt++)
                if (tt == t)
                    acc[tt][j] += add;
        }
    };
    for (int64_t q = q0; q < q1; q++) {
        const Sub nxt = fetch(q + 1 < q1 ? q + 1 : q);
        const int t_lo = cur.dst >> 4, t_hi = (cur.dst + cur.n - 1) >> 4;
        for (int c0 = 0; c0 < cur.w; c0 += 64) {
            real b[16];
            gather_b(b, cur, c0);
            // two tile buffers used in turn (no register copies: a copy waits for the load it copies), the next tile's loads always issued --
            // clamped to the segment's last tile -- before the current tile is worked on
            scalar2 va[8], vb[8];
            load_tile(va, cur, c0, t_lo);
            for (int t = t_lo; t <= t_hi; t += 2) {
                load_tile(vb, cur, c0, t + 1 <= t_hi ? t + 1 : t_hi);
                HMX_SCHED_FENCE();
                tile_product(va, b, cur, t);
                HMX_SCHED_FENCE();
                load_tile(va, cur, c0, t + 2 <= t_hi ? t + 2 : t_hi);
                HMX_SCHED_FENCE();
                if (t + 1 <= t_hi)
                    tile_product(vb, b, cur, t + 1);
                HMX_SCHED_FENCE();
            }
        }
        cur = nxt;
    }
    // dense mirrored contributions of the interval's rows (column sums the first pass left in SW16, found through the level index), y update.
    // A lane holds 16 rows (one right-hand side each): level k of all sixteen is fetched together -- sixteen independent chains of two
    // loads per level instead of one (the levels of a row are few, but every one is two dependent trips to memory)
    int jr[16], cn[16], kmax = 0;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int jrow = I * SYM_IR_MU + 16 * t + mfma16_row(real(0), lane, j);
            jr[4 * t + j]  = jrow < A.n ? jrow : A.n - 1;
            cn[4 * t + j]  = jrow < A.n ? A.count[jr[4 * t + j]] : 0;
            kmax           = cn[4 * t + j] > kmax ? cn[4 * t + j] : kmax;
        }
    // the sixteen y values of the lane are fetched NOW, together, unconditionally (rows beyond the operator read its last row, right-hand sides
    // beyond the group the group's first): with the load inside each row's own `if (row exists) y = ...` the compiler emitted load -> wait ->
    // store sixteen times in a row, sixteen trips to memory one after the other at the end of every interval (round 5, read off the ISA)
    const bool need_y = A.accumulate || !(A.beta == real(0));
    const int mcol    = cbase + (m < nrhs ? m : 0);
    real yv[16];
#pragma unroll
    for (int e = 0; e < 16; e++)
        yv[e] = need_y ? A.y[(int64_t)jr[e] * mu + mcol] : real(0);
    for (int k = 0; k < kmax; k++) {
        int32_t d[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            d[e] = A.fidx[(int64_t)(k < cn[e] ? k : 0) * A.n + jr[e]]; // (level 0 of the row when it has fewer: a valid entry, dropped below)
        real w[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            w[e] = P.W16[(int64_t)(k < cn[e] ? d[e] : P.zero_slot) * 16 + m];
#pragma unroll
        for (int e = 0; e < 16; e++)
            acc[e >> 2][e & 3] += w[e];
    }
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int jrow = I * SYM_IR_MU + 16 * t + mfma16_row(real(0), lane, j);
            const real y0  = yv[4 * t + j];
            const real out = A.accumulate ? y0 + A.alpha * acc[t][j] : (A.beta == real(0) ? A.alpha * acc[t][j] : A.alpha * acc[t][j] + A.beta * y0);
            if (jrow < A.n && m < nrhs)
                A.y[(int64_t)jrow * mu + cbase + m] = out;
        }
}

#endif // !HMX_COMPLEX

// small helpers -----------------------------------------------------------------------------------
__global__ void axpby_kernel(int n, scalar alpha, const scalar *w, scalar beta, scalar *y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        y[i] = hmx_is_zero(beta) ? alpha * w[i] : alpha * w[i] + beta * y[i];
}
// user_to_cluster: out[i] = in[perm[i] - base]; cluster_to_user: out[perm[i] - base] = in[i]
// (clustering/cluster_node.hpp:150-175)
__global__ void gather_kernel(int n, const int32_t *perm, int base, const scalar *in, scalar *out, int mu) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)n * mu) {
        const int i = e / mu, c = e - (int64_t)i * mu;
        out[e]      = in[(int64_t)(perm[i] - base) * mu + c];
    }
}
__global__ void scatter_kernel(int n, const int32_t *perm, int base, const scalar *in, scalar *out, int mu) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)n * mu) {
        const int i = e / mu, c = e - (int64_t)i * mu;
        out[(int64_t)(perm[i] - base) * mu + c] = in[e];
    }
}
// column-major user numbering <-> row-major cluster numbering (add_hmatrix_matrix_product.hpp:44-60: user_to_cluster per column +
// transpose): rm[i][c] = cm[(perm[i] - base) + n * c]
__global__ void gather_cm_kernel(int n, int mu, const int32_t *perm, int base, const scalar *cm, scalar *rm) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)n * mu) {
        const int i = (int)(e / mu), c = (int)(e - (int64_t)i * mu);
        rm[e]       = cm[(int64_t)(perm[i] - base) + (int64_t)n * c];
    }
}
__global__ void scatter_cm_kernel(int n, int mu, const int32_t *perm, int base, const scalar *rm, scalar *cm) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)n * mu) {
        const int i = (int)(e / mu), c = (int)(e - (int64_t)i * mu);
        cm[(int64_t)(perm[i] - base) + (int64_t)n * c] = rm[e];
    }
}
// strided column extract / insert for row-major multi-RHS (X[n][mu])
__global__ void col_extract_kernel(int n, int mu, int c, const scalar *X, scalar *x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        x[i] = X[(int64_t)i * mu + c];
}
__global__ void col_insert_kernel(int n, int mu, int c, const scalar *y, scalar *Y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        Y[(int64_t)i * mu + c] = y[i];
}
// conjugation in place (trans = 'C' is conj o 'T' o conj)
// ---- bulk download (api_get_blocks): blocks gathered on the device into one staging array in htool's layouts -------------------------
// item = one requested leaf; dst = its first entry in `stage`: low rank U (M x r, column-major) then V (r x N, column-major) --
// LowRankMatrix, hmatrix/lrmat/lrmat.hpp:15-45 --, dense M x N column-major
struct GetItem {
    int64_t dst, colptr;
    int32_t leaf, rank, M, N, swapped, t_rel; // rank -1: dense; t_rel: first row of the leaf, local to the operator's rows
};
// grid (items, KS): workgroup (it, ks) copies the crosses ks, ks + KS, ... of item it
__global__ void get_lr_blocks_kernel(const GetItem *items, const scalar *pool, const int64_t *cross_off, scalar *stage) {
    const GetItem it = items[blockIdx.x];
    if (it.rank <= 0)
        return;
    const int M = it.M, N = it.N, r = it.rank;
    const int n1 = it.swapped ? N : M;
    scalar *U = stage + it.dst, *V = U + (int64_t)M * r;
    for (int k = blockIdx.y; k < r; k += gridDim.y) {
        const scalar *c    = pool + cross_off[it.colptr + k];
        const scalar *ucol = it.swapped ? c + n1 : c, *vrow = it.swapped ? c : c + n1;
        for (int i = threadIdx.x; i < M; i += blockDim.x)
            U[i + (int64_t)k * M] = ucol[i];
        for (int j = threadIdx.x; j < N; j += blockDim.x)
            V[k + (int64_t)r * j] = vrow[j];
    }
}
// one workgroup per (dense item, row range) slice: the leaf's columns of that range are one contiguous len x N block of the E-stream
__global__ void get_dense_blocks_kernel(const GetItem *items, const int32_t *p_item, const int32_t *p_range, const int32_t *p_col, const scalar *stream, const int64_t *base,
                                        const int32_t *range_off, const int32_t *range_len, scalar *stage) {
    const GetItem it = items[p_item[blockIdx.x]];
    const int r = p_range[blockIdx.x], len = range_len[r], rel = range_off[r] - it.t_rel;
    const scalar *src = stream + base[r] + (int64_t)p_col[blockIdx.x] * len;
    scalar *dst       = stage + it.dst;
    const int64_t tot = (int64_t)len * it.N;
    for (int64_t idx = threadIdx.x; idx < tot; idx += blockDim.x) {
        const int i = (int)(idx % len);
        const int64_t j = idx / len;
        dst[(rel + i) + (int64_t)it.M * j] = src[idx];
    }
}

__global__ void conj_kernel(int64_t n, const scalar *in, scalar *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = hmx_conj(in[i]);
}
