// kernels_symmetric.hpp -- mirrored sweeps: the fused symmetric / Hermitian product and the transposed product on the stored data, single vector and several right-hand sides.
// Part of the engine's device code: included by kernels_body.hpp inside namespace hmx::{f64,f32,z64,c32}, written against `scalar` / `real`.  No include guard on purpose.

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
    // groups of row ranges (build_mirror_tables): X.order is the launch order of the GROUPS, group g = ranges [g G, (g + 1) G); mdst <= -2 is
    // accumulator -2 - mdst of the group (dynamic LDS: grp_na[g] accumulators of one value -- several right-hand sides: SWW values -- each),
    // written to W[grp_flush[g] + ...] when the group's last range is done
    const int32_t *grp_flush, *grp_na;
    int G;
};
// the group's accumulators: zeroed before its first range, written out after its last one (all threads of the workgroup; vals = values per accumulator)
extern __shared__ __attribute__((aligned(16))) unsigned char hmx_group_lds[];
__device__ __forceinline__ void group_acc_zero(scalar *gacc, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        gacc[i] = scalar(0);
}
// a range's geometry inside the loop over a group's ranges: loaded after barriers and stores of the same kernel, the compiler no longer reads it
// through the scalar cache and treats it as one value per lane -- every bound and base address of the sweep then lives in vector registers
// (the 16-RHS sweep over the E-streams: 1.93 instead of 1.50 ms with identical instructions otherwise, round 6).  It IS wave-uniform: say so.
__device__ __forceinline__ int uniform_value(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uniform_value(int64_t v) {
    const int lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ void group_acc_flush(const scalar *gacc, scalar *W, int64_t first, int n) {
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        W[first + i] = gacc[i];
}
// The wave's columns are walked in groups of eight (sixteen for 4-byte coefficients), flattened over its 64-column tiles and
// software-pipelined: the loads of group g + 1 are issued before group g is reduced, so the dependent chain of the eight-way
// reduction never leaves the wave without loads in flight.
// FWD = false: the mirrored column sums only -- the first sweep of the TRANSPOSED product of an ordinary operator on its stored data (every
// column is then a mirrored one, the forward operands and y are not touched; run_transposed_fused).
template <int WAVES, bool FWD = true>
__global__ __launch_bounds__(WAVES *WAVE) HMX_WPE_EXPAND_SYM_KERNEL void expand_sym_kernel(ExpandSymArgs S) {
    const ExpandArgs &A = S.X;
    __shared__ scalar part[FWD ? WAVES : 1][WAVE];
    const int grp  = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    scalar *gacc   = reinterpret_cast<scalar *>(hmx_group_lds);
    const int gna  = S.grp_na[grp];
    group_acc_zero(gacc, gna);
    const int R_end = (grp + 1) * S.G < A.nranges ? (grp + 1) * S.G : A.nranges;
    for (int R = grp * S.G; R < R_end; R++) {
    __syncthreads(); // the accumulators are zero / the previous range is done with `part`
    const int len = uniform_value(A.range_len[R]), C = uniform_value(A.range_cols[R]), roff = uniform_value(A.range_off[R]);
    const int64_t rbase = uniform_value(A.range_base[R]), rcolbase = uniform_value(A.range_colbase[R]);
    const scalar *E     = A.stream + rbase;
    const int32_t *zidx = A.z_idx + rcolbase;
    const int32_t *mdst = S.mdst + rcolbase;
    const bool active   = lane < len;
    const int row       = active ? lane : 0;
    const scalar xr     = active ? S.xrow[roff + lane] : scalar(0); // idle lanes contribute exact zeros to the column sums
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
        mir = __any(md != -1); // wave-uniform: tiles without mirrored columns (diagonal leaves, off-diagonal stripes) skip the reductions
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
                else if (md <= -2)
                    gacc[-2 - md] += t; // (one lane of the workgroup per accumulator and range: see build_mirror_tables)
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
            scalar *yo = A.y + roff + lane;
            *yo        = hmx_is_zero(A.beta) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
        }
    }
    } // ranges of the group
    group_acc_flush(gacc, S.W, S.grp_flush[grp], gna);
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
    const int32_t *coef;      // per R column: slot of a'[col] in W, -1: not a mirrored column
    const int32_t *order;     // launch position -> interval (heaviest first)
    const int64_t *sub_ptr;   // per interval: its sub-tasks [sub_ptr[I], sub_ptr[I + 1])
    // per sub-task a ready record (round 6; before: task -> piece -> geometry, a chain of a dozen dependent index loads in front of every
    // sub-task's first stream load): first stream element (the sub-task's first row, the chunk's first column), first entry of the chunk in
    // `coef`, columns of the chunk, rows, first row inside the interval
    const int64_t *sub_src, *sub_cb;
    const int32_t *sub_w, *sub_nrows, *sub_dst;
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
__global__ __launch_bounds__(WAVES *WAVE) HMX_WPE_ROWSYM_KERNEL void rowsym_kernel(RowSymArgs A) {
    __shared__ scalar acc[WAVES][SYM_IR];
    const int I    = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int r = lane; r < SYM_IR; r += WAVE)
        acc[wv][r] = scalar(0);
    const bool herm = A.herm != 0;
    constexpr int GS = sizeof(scalar2) <= 8 ? 16 : 8;
    const int col0 = HMX_COL0(lane), col1 = HMX_COL1(lane);
    const int64_t q_end = A.sub_ptr[I + 1];
    // the coefficients a' of a sub-task's two columns per lane: gathered one sub-task ahead (record -> slot -> value: two trips to memory that
    // used to stand in front of every sub-task's first stream load)
    auto coefficients = [&](int64_t q, scalar &c0, scalar &c1) {
        const int w      = A.sub_w[q];
        const int64_t cb = A.sub_cb[q];
        const int d0 = col0 < w ? A.coef[cb + col0] : -1, d1 = col1 < w ? A.coef[cb + col1] : -1;
        c0 = d0 >= 0 ? A.W[d0] : scalar(0);
        c1 = d1 >= 0 ? A.W[d1] : scalar(0);
    };
    int64_t q = A.sub_ptr[I] + wv;
    scalar c0 = scalar(0), c1 = scalar(0);
    if (q < q_end)
        coefficients(q, c0, c1);
    for (; q < q_end; q += WAVES) {
        const int len = A.sub_nrows[q], w = A.sub_w[q];
        scalar *dst = &acc[wv][A.sub_dst[q]];
        const int wp      = hmx_wp(w);
        const scalar *src = A.stream + A.sub_src[q];
        scalar c0n = scalar(0), c1n = scalar(0);
        coefficients(q + WAVES < q_end ? q + WAVES : q, c0n, c1n); // (past the wave's last sub-task: its own again, unused)
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
        c0 = c0n, c1 = c1n;
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
// a'[dst][0..SWW) = sum_i SW16[list[lp + i] + k][0..SWW): the partial column sums of a mirrored low-rank leaf that spans several row ranges / groups.
// One wave per entry for the entries with many partial sums: SWW lanes take the right-hand sides, the 64 / SWW lane groups every (64 / SWW)-th
// partial sum; one thread per (entry, right-hand side) for the rest; fixed order.  Both kinds in ONE launch (combine_list_mu_both_kernel: the wave
// entries in the first `wave_blocks` workgroups): the two are independent and each too small to fill the chip for long (45 + 90 us one after the
// other at N = 1e6, round 6)
__device__ __forceinline__ void combine_list_mu_wave_body(const CombineListArgs &A, int block) {
    const int e = __builtin_amdgcn_readfirstlane(block * 4 + (threadIdx.x >> 6));
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
__device__ __forceinline__ void combine_list_mu_thread_body(const CombineListArgs &A, int64_t id) {
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
__global__ __launch_bounds__(256) void combine_list_mu_both_kernel(CombineListArgs Wv, CombineListArgs Th, int wave_blocks) {
    if ((int)blockIdx.x < wave_blocks)
        combine_list_mu_wave_body(Wv, blockIdx.x);
    else
        combine_list_mu_thread_body(Th, (int64_t)(blockIdx.x - wave_blocks) * 256 + threadIdx.x);
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
    const int grp  = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    scalar *gacc   = reinterpret_cast<scalar *>(hmx_group_lds); // [accumulator][SWW]
    const int gna  = S.grp_na[grp];
    group_acc_zero(gacc, gna * SWW);
    const int R_end = (grp + 1) * S.G < A.nranges ? (grp + 1) * S.G : A.nranges;
    for (int R = grp * S.G; R < R_end; R++) {
    __syncthreads(); // the accumulators are zero / the previous range is done with `part`
    const int len = uniform_value(A.range_len[R]), C = uniform_value(A.range_cols[R]), roff = uniform_value(A.range_off[R]);
    const int64_t rbase = uniform_value(A.range_base[R]), rcolbase = uniform_value(A.range_colbase[R]);
    const scalar *E     = A.stream + rbase;
    const int32_t *zidx = A.z_idx + rcolbase;
    const int32_t *mdst = S.mdst + rcolbase;
    const bool active   = lane < len;
    const int row       = active ? lane : 0;
    const bool herm     = S.herm != 0;
    scalar xr[MU], acc[MU]; // the input at this lane's row (idle lanes and missing right-hand sides: exact zeros), the forward sums
#pragma unroll
    for (int j = 0; j < MU; j++) {
        xr[j]  = (active && j < nrhs) ? S.xrow[(int64_t)(roff + lane) * mu + cbase + j] : scalar(0);
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
        mir = __any(md != -1);
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
                    else if (md <= -2 && j < nrhs)
                        gacc[(-2 - md) * SWW + j] += t;
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
            scalar *yo = A.y + (int64_t)(roff + i) * mu + cbase + jj;
            *yo        = hmx_is_zero(A.beta) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
        }
    }
    } // ranges of the group
    group_acc_flush(gacc, S.W, (int64_t)S.grp_flush[grp] * SWW, gna * SWW);
}

// ---- second sweep over the R-streams for several right-hand sides: intervals and segments ------------------------------------------------
// The output rows are cut into INTERVALS of at most SYM_IR_MU rows at the boundaries of the mirrored pieces (build_mirror_tables), so every
// (piece, chunk) task covers whole intervals.  Per interval the kernels walk SEGMENTS -- one per (task over the interval, 64-column half of
// its chunk), a ready record each: first stream element (row 0 of the interval, first column of the half), row pitch, columns, first entry
// of the coefficient-slot table -- in the launch order of the tasks (fixed: bit-reproducible).
constexpr int SYM_IR_MU = 64;
struct RowSegArgs {
    const scalar *stream;
    const int32_t *order;   // launch position -> interval (heaviest first)
    const int32_t *int_off; // interval I = output rows [int_off[I], int_off[I + 1])
    const int64_t *seg_ptr; // per interval: its segments [seg_ptr[I], seg_ptr[I + 1])
    const int64_t *seg_src, *seg_cb;
    const int32_t *seg_wp, *seg_w;
    const int32_t *coef;    // per R column: slot of a'[col] in W16, -1: not a mirrored column
    const scalar *W16;      // [slot][SWW]
    int zero_slot;          // a slot whose values are zero
    int nint;
    const int32_t *fidx;    // dense mirrored contributions of output row j: W16[fidx[k * n + j]], k < count[j]
    const int32_t *count;
    scalar *y;
    scalar alpha, beta;     // accumulate = 0: y = alpha * sums + beta * y (transposed product on the stored data: this sweep owns y)
    int n;                  // rows of the operator (stride of fidx)
    int herm;
    int accumulate;         // 1: y += alpha * sums (the forward sweep of the symmetric product has written y already)
};
// VALU form (what complex coefficients run when HMX_OPT_MATRIX_CORES is 0): one workgroup per interval, wave w its segments w, w + WAVES, ...,
// lane = column pair, row sums by reduce8 and folded in LDS, the dense mirrored contributions and the y update at the end
template <int WAVES, int MU>
__global__ __launch_bounds__(WAVES *WAVE) void rowsym_mu_kernel(RowSegArgs A, int mu, int cbase, int nrhs) {
    __shared__ scalar acc[WAVES][SYM_IR_MU][MU];
    const int I    = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = A.int_off[I], len = A.int_off[I + 1] - r0;
    for (int r = lane; r < SYM_IR_MU * MU; r += WAVE)
        (&acc[wv][0][0])[r] = scalar(0);
    const bool herm = A.herm != 0;
    constexpr int GS = 8;
    for (int64_t q = A.seg_ptr[I] + wv; q < A.seg_ptr[I + 1]; q += WAVES) {
        scalar(*dst)[MU] = &acc[wv][0];
        const int w = A.seg_w[q], wp = A.seg_wp[q];
        const int col0 = HMX_COL0(lane), col1 = HMX_COL1(lane);
        const scalar *src = A.stream + A.seg_src[q];
        const int64_t cb  = A.seg_cb[q];
        const int d0 = col0 < w ? A.coef[cb + col0] : -1, d1 = col1 < w ? A.coef[cb + col1] : -1;
        scalar c0[MU], c1[MU], mine[MU];
#pragma unroll
        for (int j = 0; j < MU; j++) {
            c0[j]   = (d0 >= 0 && j < nrhs) ? A.W16[(int64_t)d0 * SWW + j] : scalar(0);
            c1[j]   = (d1 >= 0 && j < nrhs) ? A.W16[(int64_t)d1 * SWW + j] : scalar(0);
            mine[j] = scalar(0);
        }
        const int wl = (w + 1) & ~1; // columns of the segment that exist in the stream (an odd chunk keeps a zero column)
        auto load_rows = [&](scalar2(&e)[GS], int i0) {
#pragma unroll
            for (int u = 0; u < GS; u++) {
                const int i = i0 + u < len ? i0 + u : len - 1;
                e[u]        = load_pair(src + (int64_t)i * wp, col0, col1, HMX_SPLIT_COLS ? w : wl);
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
        const int j = r0 + r;
        if (r >= len || jj >= nrhs)
            continue;
        scalar sum = acc[0][r][jj];
#pragma unroll
        for (int k = 1; k < WAVES; k++)
            sum += acc[k][r][jj];
        const int cnt = A.count[j];
        for (int k = 0; k < cnt; k++)
            sum += A.W16[(int64_t)A.fidx[(int64_t)k * A.n + j] * SWW + jj];
        scalar *yo = A.y + (int64_t)j * mu + cbase + jj;
        if (A.accumulate)
            *yo += A.alpha * sum;
        else
            *yo = hmx_is_zero(A.beta) ? A.alpha * sum : A.alpha * sum + A.beta * (*yo);
    }
}

#if HMX_COMPLEX
// ---------------------------------------------------------------------------------------------
// Groups of 8 COMPLEX right-hand sides on the stored data on the MATRIX CORES (round 5; the VALU kernels above are what every group width
// falls back to when HMX_OPT_MATRIX_CORES is 0).  Planes as in expand_zmfma8s_kernel: a row of 8 complex operands is 16 reals
// (re0, im0, re1, im1, ...), n = 2 rhs + part is the MFMA's free index,
//     forward   Y[row][n]  = sum_col  E_re[row][col] Z[col][n]  + E_im[row][col] Z'[col][n],       Z'[n] = n even ? -Z[n + 1] : Z[n - 1]
//     mirrored  EW[col][n] = sum_row  E_re[row][col] X[row][n] +- E_im[row][col] X'[row][n]        (-: Hermitian storage, conj(E))
// The mirrored product packs BOTH planes of 8 columns into the M index of ONE 16 x 16 x 4 MFMA (M < 8: re of column M, M >= 8: im of column
// M - 8) against the operand X alone: D[M < 8] = E_re^T X, D[M >= 8] = E_im^T X, and E_im^T X' is D[M >= 8] with neighbouring n exchanged and
// the even ones negated -- a lane exchange AFTER the 16 k-steps instead of a second MFMA per k-step.  One wave-private LDS tile
// [64 rows][8 re | 8 im] (pitch 20 reals) serves both products: read down the rows for the mirrored operand, across the columns for the forward one.
// ---------------------------------------------------------------------------------------------
typedef Acc4<real>::type zsacc4;
// the packed result of the mirrored / row product: tm = D of the M-packed MFMA.  Returns nval values val[k] with their M index idx[k] (< 8)
// -- P1[idx][n] + s P2[idx][n'] -- valid in the lanes `ok` says (all lanes for 8-byte reals: register j and j + 2 of one lane are M and M + 8;
// the lower two lane quarters for 4-byte reals: M + 8 sits 32 lanes up)
__device__ __forceinline__ int zpack_combine(const zsacc4 &tm, int lane, bool conj, real (&val)[4], int (&idx)[4], bool &ok) {
    if constexpr (sizeof(real) == 8) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const real o  = hmx_shfl_xor(tm[j + 2], 1);
            const real sw = (lane & 1) ? o : -o;
            val[j]        = conj ? tm[j] - sw : tm[j] + sw;
            idx[j]        = (lane >> 4) + 4 * j;
        }
        ok = true;
        return 2;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const real o  = hmx_shfl(tm[j], ((lane ^ 1) + 32) & 63);
            const real sw = (lane & 1) ? o : -o;
            val[j]        = conj ? tm[j] - sw : tm[j] + sw;
            idx[j]        = 4 * ((lane >> 4) & 1) + j;
        }
        ok = lane < 32;
        return 4;
    }
}

template <int WAVES, bool FWD = true>
__global__ __launch_bounds__(WAVES *WAVE) void expand_sym_zmfma8_kernel(ExpandSymArgs S, int mu, int cbase, int nrhs) {
    const ExpandArgs &A = S.X;
    constexpr int P = 20, STEP = 8; // tile pitch in reals: conflict-free (8-byte reals) / two lanes per bank (4-byte) for both read directions
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 64 * P];
    static_assert(64 * P >= WAVE * 16, "the final fold of the waves reuses the tiles");
    const int grp  = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    real *gacc     = reinterpret_cast<real *>(hmx_group_lds); // [accumulator][16 reals = 8 complex right-hand sides]
    const int gna  = S.grp_na[grp];
    group_acc_zero(reinterpret_cast<scalar *>(gacc), gna * SWW);
    const int R_end = (grp + 1) * S.G < A.nranges ? (grp + 1) * S.G : A.nranges;
    auto range_pass = [&](const int R) {
    const int len = uniform_value(A.range_len[R]), C = uniform_value(A.range_cols[R]), roff = uniform_value(A.range_off[R]);
    const int64_t rbase = uniform_value(A.range_base[R]), rcolbase = uniform_value(A.range_colbase[R]);
    const scalar *E     = A.stream + rbase;
    const int32_t *zidx = A.z_idx + rcolbase;
    const int32_t *mdst = S.mdst + rcolbase;
    const int m = lane & 15, kk = lane >> 4;
    real *tile     = lds + wv * 64 * P;
    real *W16r     = reinterpret_cast<real *>(S.W); // [slot][16 reals = 8 complex right-hand sides]
    const int row  = lane < len ? lane : len - 1;
    const int mo   = m < 2 * nrhs ? m : 0; // ragged group: see expand_mfma16s_kernel
    const bool herm = S.herm != 0;
    // B operand of the mirrored product, constant over the range: X_t[row 4h + kk] as 16 reals, element n = m (zero beyond the range)
    real xt[16];
#pragma unroll
    for (int h = 0; h < 16; h++) {
        const int r   = 4 * h + kk;
        const real xv = reinterpret_cast<const real *>(S.xrow + (int64_t)(roff + (r < len ? r : len - 1)) * mu + cbase)[mo];
        xt[h]         = r < len ? xv : real(0);
    }
    zsacc4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
        acc[t] = zsacc4{0, 0, 0, 0};
    // the wave's columns as one sequence of 8-column steps, three stages in flight, every load unconditional: see expand_mfma16s_kernel
    const int ntile_all = (C + 63) >> 6;
    int n = 0;
    if (wv < ntile_all) {
        n = 8 * ((ntile_all - 1 - wv) / WAVES + 1);
        if ((ntile_all - 1 - wv) % WAVES == 0)
            n -= 8 - ((C - 64 * (ntile_all - 1) + 7) >> 3);
    }
    auto col_of = [&](int s) { return (((s >> 3) * WAVES + wv) << 6) + ((s & 7) << 3); };
    struct Idx {
        int z, md; // lane l: Z index and mirror slot of column col_of(s) + (l & 7)
    };
    auto load_idx = [&](int s) {
        const int c  = col_of(s < n ? s : n - 1) + (lane & 7);
        const int cc = c < C ? c : C - 1;
        Idx ix;
        ix.z  = FWD ? zidx[cc] : 0;
        ix.md = mdst[cc];
        return ix;
    };
    auto gathers = [&](real(&b)[2], const Idx &ix) {
        if constexpr (FWD)
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const int zc = __shfl(ix.z, 4 * g + kk, WAVE);
                b[g]         = reinterpret_cast<const real *>(expand_operand(A, zc, mu) + cbase)[mo];
            }
    };
    auto load_cols = [&](scalar(&v)[STEP], int s) {
        const int c = col_of(s < n ? s : n - 1);
#pragma unroll
        for (int u = 0; u < STEP; u++) {
            const int col = c + u < C ? c + u : C - 1;
            v[u]          = stream_load(E + (int64_t)col * len + row);
        }
    };
    auto apply = [&](const scalar(&v)[STEP], const real(&braw)[2], int mdi, int s) {
        const int c = col_of(s);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < STEP; u++) {
            tile[lane * P + u]     = v[u].re;
            tile[lane * P + 8 + u] = v[u].im;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int md = (c + (lane & 7) < C) ? mdi : -1;
        if (__any(md != -1)) { // wave-uniform: steps without mirrored columns skip all of it
            real ta[16];
#pragma unroll
            for (int h = 0; h < 16; h++)
                ta[h] = tile[(4 * h + kk) * P + m]; // A[M = m][k = row 4h + kk]: m < 8 re of column m, m >= 8 im of column m - 8
            zsacc4 tm = zsacc4{0, 0, 0, 0};
#pragma unroll
            for (int h = 0; h < 16; h++)
                tm = mfma16(ta[h], xt[h], tm);
            real val[4];
            int idx[4];
            bool ok;
            const int nv = zpack_combine(tm, lane, herm, val, idx, ok);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < nv) {
                    const int d = __shfl(md, idx[k], WAVE); // slot of column c + idx[k]
                    if (ok && d >= 0)
                        W16r[(int64_t)d * 16 + m] = val[k];
                    else if (ok && d <= -2)
                        gacc[(-2 - d) * 16 + m] += val[k];
                }
        }
        if constexpr (FWD) {
            real b[2], bs[2];
#pragma unroll
            for (int g = 0; g < 2; g++) {
                b[g]  = (c + 4 * g + kk < C) ? braw[g] : real(0);
                bs[g] = zmfma_swapped(b[g], lane);
            }
#pragma unroll
            for (int g = 0; g < 2; g++) {
                real are[4], aim[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    are[t] = tile[(16 * t + m) * P + 4 * g + kk];
                    aim[t] = tile[(16 * t + m) * P + 8 + 4 * g + kk];
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    acc[t] = mfma16(are[t], b[g], acc[t]);
                    acc[t] = mfma16(aim[t], bs[g], acc[t]);
                }
            }
        }
    };
    if (n > 0) {
        scalar v0[STEP], v1[STEP];
        real b0[2], b1[2];
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
    // forward result: accumulator tile t, register j of lane l = (row 16 t + mfma16_row, real column l & 15 = 2 rhs + part), folded over the waves
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
        scalar sum(red[0][i][2 * c], red[0][i][2 * c + 1]);
#pragma unroll
        for (int w = 1; w < WAVES; w++)
            sum += scalar(red[w][i][2 * c], red[w][i][2 * c + 1]);
        scalar *yo = A.y + (int64_t)(roff + i) * mu + cbase + c;
        *yo        = hmx_is_zero(A.beta) ? A.alpha * sum : A.alpha * sum + A.beta * (*yo);
    }
    }; // range_pass
    for (int R = grp * S.G; R < R_end; R++) {
        __syncthreads(); // the accumulators are zero / the previous range is done with the tiles
        range_pass(R);
    }
    group_acc_flush(reinterpret_cast<const scalar *>(gacc), S.W, (int64_t)S.grp_flush[grp] * SWW, gna * SWW);
}

// Second pass over the R-streams for 8 complex right-hand sides: Y_s[row][n] += sum_col op(V[row][col]) a'[col][n], one WAVE per interval
// as in rowsym_mfma16_kernel.  Tiles of 8 rows x 64 columns, both planes packed into the M index of one MFMA per k-step (M < 8: re of row M,
// M >= 8: im of row M - 8; zpack_combine turns D into P1 +- P2'), staged transposed and swizzled in LDS exactly as the real kernel stages its
// 16-row tiles.  Accumulators hold the combined values of the interval's eight tiles.
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) void rowsym_zmfma8_kernel(RowSegArgs A, int mu, int cbase, int nrhs) {
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 64 * 16];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pos = blockIdx.x * WAVES + wv;
    if (pos >= A.nint)
        return; // (no workgroup barrier below: the waves are independent)
    const int I  = A.order[pos];
    const int r0 = A.int_off[I], ilen = A.int_off[I + 1] - r0;
    const int m = lane & 15, kk = lane >> 4;
    const int mo = m < 2 * nrhs ? m : 0;
    const bool herm = A.herm != 0;
    const real *W16r = reinterpret_cast<const real *>(A.W16); // [slot][16 reals]
    real *tile = lds + wv * 64 * 16;
    auto taddr = [](int c, int i) { return 16 * (c ^ ((c >> 1) & 1)) + (i ^ ((c >> 1) & 15)); }; // element (slot i, column c): see rowsym_mfma16_kernel
    constexpr int NV = sizeof(real) == 8 ? 2 : 4; // combined values per lane and tile (zpack_combine)
    real acc[8][NV];
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int k = 0; k < NV; k++)
            acc[t][k] = real(0);
    struct Seg {
        const scalar *src;
        int w, wp;
        int32_t slot; // lane l: slot of a' of the segment's column l (the zero slot: no mirrored leaf's column / beyond the segment)
    };
    const int64_t q0 = A.seg_ptr[I], q1 = A.seg_ptr[I + 1];
    auto fetch = [&](int64_t q) {
        q = q < q1 ? q : q1 - 1;
        Seg s;
        s.src = A.stream + A.seg_src[q];
        s.w   = A.seg_w[q];
        s.wp  = A.seg_wp[q];
        const int32_t c = A.coef[A.seg_cb[q] + (lane < s.w ? lane : 0)];
        s.slot          = (lane < s.w && c >= 0) ? c : A.zero_slot;
        return s;
    };
#if HMX_SPLIT_COLS
    constexpr int NL = 8; // loads per tile: one row x 64 columns (16-byte coefficients) each
    typedef scalar tile_vec;
#else
    constexpr int NL = 4; // two rows x 64 columns (8-byte coefficients, a column pair per lane) each
    typedef scalar2 tile_vec;
    const int lrow = lane >> 5, lc = 2 * (lane & 31);
#endif
    // tile t8 = interval rows 8 t8 ... 8 t8 + 7, the segment's columns; rows clamped into the interval
    auto load_tile = [&](tile_vec(&v)[NL], const Seg &s, int t8) {
#pragma unroll
        for (int u = 0; u < NL; u++) {
#if HMX_SPLIT_COLS
            int r = 8 * t8 + u;
            r     = r >= ilen ? ilen - 1 : r;
            v[u]  = stream_load(s.src + (int64_t)r * s.wp + (lane < s.w ? lane : 0));
#else
            int r = 8 * t8 + 2 * u + lrow;
            r     = r >= ilen ? ilen - 1 : r;
            v[u]  = stream_load(reinterpret_cast<const scalar2 *>(s.src + (int64_t)r * s.wp + (lc < s.w ? lc : 0)));
#endif
        }
    };
    auto gather_b = [&](real(&b)[16], const Seg &s) {
#pragma unroll
        for (int h = 0; h < 16; h++) {
            const int d = __shfl(s.slot, 4 * h + kk, WAVE);
            b[h]        = W16r[(int64_t)d * 16 + mo];
        }
    };
    auto tile_product = [&](const tile_vec(&v)[NL], const real(&b)[16], int t8) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < NL; u++) {
#if HMX_SPLIT_COLS
            tile[taddr(lane, u)]     = v[u].re;
            tile[taddr(lane, 8 + u)] = v[u].im;
#else
            const int i = 2 * u + lrow;
            tile[taddr(lc, i)]         = v[u].x.re;
            tile[taddr(lc, 8 + i)]     = v[u].x.im;
            tile[taddr(lc + 1, i)]     = v[u].y.re;
            tile[taddr(lc + 1, 8 + i)] = v[u].y.im;
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        real ta[16];
#pragma unroll
        for (int h = 0; h < 16; h++)
            ta[h] = tile[taddr(4 * h + kk, m)]; // A[M = m][k = column 4h + kk]: m < 8 re of row m, m >= 8 im of row m - 8
        zsacc4 tm = zsacc4{0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < 16; h++)
            tm = mfma16(ta[h], b[h], tm);
        real val[4];
        int idx[4];
        bool ok;
        const int nv = zpack_combine(tm, lane, herm, val, idx, ok);
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int i    = 8 * t8 + idx[k < nv ? k : 0];
            const real add = (ok && i < ilen) ? val[k] : real(0);
#pragma unroll
            for (int tt = 0; tt < 8; tt++)
                if (tt == t8)
                    acc[tt][k] += add;
        }
    };
    const int t_hi = (ilen - 1) >> 3;
    Seg cur{};
    if (q0 < q1)
        cur = fetch(q0);
    for (int64_t q = q0; q < q1; q++) {
        const Seg nxt = fetch(q + 1);
        real b[16];
        gather_b(b, cur);
        tile_vec va[NL], vb[NL];
        load_tile(va, cur, 0);
        for (int t = 0; t <= t_hi; t += 2) {
            load_tile(vb, cur, t + 1 <= t_hi ? t + 1 : t_hi);
            HMX_SCHED_FENCE();
            tile_product(va, b, t);
            HMX_SCHED_FENCE();
            load_tile(va, cur, t + 2 <= t_hi ? t + 2 : t_hi);
            HMX_SCHED_FENCE();
            if (t + 1 <= t_hi)
                tile_product(vb, b, t + 1);
            HMX_SCHED_FENCE();
        }
        cur = nxt;
    }
    // dense mirrored contributions + y update.  A lane holds, per tile, NV values: real column n = m (= 2 rhs + part) of the rows 8 t8 + idx.
    // The complex factors alpha / beta need both parts of a value: the partner is the neighbouring lane (m ^ 1).
    const bool lane_ok = sizeof(real) == 8 || lane < 32;
    const real a_re = A.alpha.re, a_im = (lane & 1) ? A.alpha.im : -A.alpha.im;
    const real b_re = A.beta.re, b_im = (lane & 1) ? A.beta.im : -A.beta.im;
    real *yr = reinterpret_cast<real *>(A.y);
#pragma unroll
    for (int t8 = 0; t8 < 8; t8++) {
        int jr[NV], cn[NV], kmax = 0;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int idxk = sizeof(real) == 8 ? (lane >> 4) + 4 * k : 4 * ((lane >> 4) & 1) + k;
            const int ir   = 8 * t8 + idxk;
            jr[k]          = r0 + (ir < ilen ? ir : ilen - 1);
            cn[k]          = (lane_ok && ir < ilen) ? A.count[jr[k]] : 0;
            kmax           = cn[k] > kmax ? cn[k] : kmax;
        }
        real yv[NV];
#pragma unroll
        for (int k = 0; k < NV; k++)
            yv[k] = yr[((int64_t)jr[k] * mu + cbase) * 2 + mo];
        for (int lev = 0; lev < kmax; lev++) {
            int32_t d[NV];
#pragma unroll
            for (int k = 0; k < NV; k++)
                d[k] = A.fidx[(int64_t)(lev < cn[k] ? lev : 0) * A.n + jr[k]];
#pragma unroll
            for (int k = 0; k < NV; k++)
                acc[t8][k] += W16r[(int64_t)(lev < cn[k] ? d[k] : A.zero_slot) * 16 + mo];
        }
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int idxk = sizeof(real) == 8 ? (lane >> 4) + 4 * k : 4 * ((lane >> 4) & 1) + k;
            const int ir   = 8 * t8 + idxk;
            const real own = acc[t8][k], oth = hmx_shfl_xor(own, 1);
            const real av  = a_re * own + a_im * oth; // this lane's part of alpha * value
            const real yo  = yv[k], yp = hmx_shfl_xor(yo, 1);
            const real out = A.accumulate ? yo + av : (hmx_is_zero(A.beta) ? av : av + (b_re * yo + b_im * yp));
            if (lane_ok && ir < ilen && m < 2 * nrhs)
                yr[((int64_t)(r0 + ir) * mu + cbase) * 2 + m] = out;
        }
    }
}
#endif // HMX_COMPLEX

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
//   combine_list_mu_both_kernel     a' of the leaves that span several row ranges / groups
//   rowsym_mfma16_kernel            second pass over the R-streams, Y_s += V^T a': one WAVE owns 64 output rows (accumulators in
//                                   registers, nothing to fold between waves), stream tiles 16 rows x 64 columns staged through LDS
//                                   transposed and swizzled so that stores and operand reads both run at two lanes per bank.
// Partial sums live in SW16 = [slot][16] (the slots of the single-vector product, 16 values each).  Fixed summation order: bit-reproducible.
// ---------------------------------------------------------------------------------------------
// FWD = false: the mirrored column sums only (transposed product of an ordinary operator on its stored data, several right-hand sides)
template <int WAVES, bool FWD = true>
__global__ __launch_bounds__(WAVES *WAVE) HMX_WPE_EXPAND_SYM_MFMA16_KERNEL void expand_sym_mfma16_kernel(ExpandSymArgs S, int mu, int cbase, int nrhs) {
    const ExpandArgs &A = S.X;
#ifndef HMX_SYMMU_PT
#define HMX_SYMMU_PT (sizeof(real) == 8 ? 18 : 20)
#endif
    // row pitch of the mirrored tile [64 rows][16 columns]: 16-byte stores stay aligned and conflict-free (a lane's row starts 36 / 20 dwords after
    // its neighbour's), operand reads (16 columns x 4 rows) at two lanes per bank at most.  Round 6: 18 / 20 instead of 24 -- LDS, not registers,
    // decides how many of these workgroups a CU holds once a group's accumulators sit next to the tiles
    constexpr int PT = HMX_SYMMU_PT;
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 64 * PT > WAVES * WAVE * 16 ? WAVES * 64 * PT : WAVES * WAVE * 16];
    const int grp  = A.order[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    real *gacc     = reinterpret_cast<real *>(hmx_group_lds); // [accumulator][16 right-hand sides]
    const int gna  = S.grp_na[grp];
    group_acc_zero(gacc, gna * 16);
    const int R_end = (grp + 1) * S.G < A.nranges ? (grp + 1) * S.G : A.nranges;
    auto range_pass = [&](const int R) {
    const int len = uniform_value(A.range_len[R]), C = uniform_value(A.range_cols[R]), roff = uniform_value(A.range_off[R]);
    const int64_t rbase = uniform_value(A.range_base[R]), rcolbase = uniform_value(A.range_colbase[R]);
    const real *E       = A.stream + rbase;
    const int32_t *zidx = A.z_idx + rcolbase;
    const int32_t *mdst = S.mdst + rcolbase;
    const int m = lane & 15, kk = lane >> 4;
    real *tile    = lds + wv * 64 * PT;
    const int row = lane < len ? lane : len - 1; // idle lanes re-read the last row: forward, they only reach accumulator rows that are never stored; mirrored, their X_t operand is zero
    const int mo  = cbase + (m < nrhs ? m : 0); // ragged group: see expand_mfma16s_kernel
    // B operand of the mirrored product, constant over the range: X_t[row 4h + kk][rhs m] for the 16 k-steps h (zero beyond the range)
    real xt[16];
#pragma unroll
    for (int h = 0; h < 16; h++) {
        const int r   = 4 * h + kk;
        const real xv = S.xrow[(int64_t)(roff + (r < len ? r : len - 1)) * mu + mo];
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
    // one step = 16 columns: mirrored column sums (if any of the 16 is a mirrored column) and the forward product.  Order inside a step: the
    // tile goes to LDS, then the registers are transposed for the forward operands (vector ALU work under the LDS round trip), then the
    // mirrored MFMAs, the stores of their sums, the forward MFMAs
    auto apply = [&](real(&v)[16], const real(&braw)[4], int mdi, int s) {
        const int c  = col_of(s);
        const int md = (c + m < C) ? mdi : -1;
        const bool mirrored = __any(md != -1); // wave-uniform: steps without mirrored columns (diagonal leaves, the other ranks' columns of a row-partitioned operator) skip all of it
        if (mirrored) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < 16; u++)
                tile[lane * PT + u] = v[u]; // 16 consecutive elements per lane: 16-byte stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // forward: a[g][t] (row 16 t + m, column 4 g + kk) = register 4 g + kk of lane quarter t -- a 4 x 4 transposition per column group
        real b[4];
        if constexpr (FWD) {
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
        }
        if (mirrored) {
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
                else if (d <= -2)
                    gacc[(-2 - d) * 16 + m] += am[j]; // (one lane of the workgroup per accumulator value and range: see build_mirror_tables)
            }
        }
        if constexpr (FWD) {
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[t] = mfma16(v[4 * g + t], b[g], acc[t]);
        }
    };
    if (n > 0) {
#ifndef HMX_SYMMU_BUFS
#define HMX_SYMMU_BUFS 2
#endif
#if HMX_SYMMU_BUFS == 3
        // THREE column buffers used in turn: while step s is computed the columns of steps s + 1 and s + 2 are in flight (and the indices of
        // step s + 3).  What a wave has in flight is all it has against the stream's latency: these kernels hold one (8-byte coefficients) or
        // two waves per SIMD, and with one step ahead -- 8 KB / 4 KB per wave, 32 KB per CU -- the sweep stood at 4 TB/s whatever else was done
        // to it (round 6: bytes in flight per CU / latency under load, not matrix-core time, is what bounded it)
        real v0[16], v1[16], v2[16], b0[4], b1[4], b2[4];
        Idx i0 = load_idx(0), i1 = load_idx(1), i2 = load_idx(2);
        gathers(b0, i0);
        load_cols(v0, 0);
        gathers(b1, i1);
        load_cols(v1, 1);
        HMX_SCHED_FENCE();
        for (int s = 0; s < n; s += 3) {
            const int md0 = i0.md;
            i0 = load_idx(s + 3);
            gathers(b2, i2);
            load_cols(v2, s + 2);
            HMX_SCHED_FENCE();
            apply(v0, b0, md0, s);
            HMX_SCHED_FENCE();
            const int md1 = i1.md;
            i1 = load_idx(s + 4);
            gathers(b0, i0);
            load_cols(v0, s + 3);
            HMX_SCHED_FENCE();
            if (s + 1 < n)
                apply(v1, b1, md1, s + 1);
            HMX_SCHED_FENCE();
            const int md2 = i2.md;
            i2 = load_idx(s + 5);
            gathers(b1, i1);
            load_cols(v1, s + 4);
            HMX_SCHED_FENCE();
            if (s + 2 < n)
                apply(v2, b2, md2, s + 2);
            HMX_SCHED_FENCE();
        }
#else
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
#endif
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
        real *yo = A.y + (int64_t)(roff + i) * mu + cbase + c;
        *yo      = A.beta == real(0) ? A.alpha * s : A.alpha * s + A.beta * (*yo);
    }
    }; // range_pass
    for (int R = grp * S.G; R < R_end; R++) {
        __syncthreads(); // the accumulators are zero / the previous range is done with the tiles
        range_pass(R);
    }
    group_acc_flush(gacc, S.W, (int64_t)S.grp_flush[grp] * 16, gna * 16);
}

// Second pass over the R-streams for up to 16 right-hand sides: Y_s[row][rhs] += sum_col V[row][col] a'[col][rhs].  One WAVE per interval
// (accumulators in registers, nothing to fold between waves); per segment the B operands a'[column][rhs] of the 16 k-steps are gathered once
// and every 16-row tile of the interval is loaded (whole rows: two rows of 64 columns per wave-wide load), staged transposed in LDS and
// multiplied: 16 MFMAs per 16 x 64 tile.
// Round 6: ONE software pipeline over all tiles of all segments of the interval.  Before, every (sub-task, half) started its own two-tile
// pipeline: four or five tiles, then a drain -- and the first tile's loads, the segment's sixteen operand gathers and the next sub-task's chain
// of index loads all waited for in the open (3.2 TB/s where the sweep moves 1.5 x its stream).  Now tile (segment q, t) is computed while the
// loads of the NEXT tile -- of this segment or the first of segment q + 1 -- are in flight, the operands of segment q + 1 are gathered a share per
// step during segment q, and the record of segment q + 2 is fetched at the start of segment q.  NT (tiles per segment = ceil(rows / 16)) is a
// template parameter: every load is unconditional and in program order (kernels_common.hpp, HMX_SCHED_FENCE), buffers and accumulators are
// indexed by constants.  Reads past the interval's last segment re-read that segment (nobody uses them).
template <int NT>
__device__ __forceinline__ void rowsym_mfma16_run(const RowSegArgs &A, real *tile, int I, int r0, int ilen, int lane, int mu, int cbase, int nrhs) {
    const int m = lane & 15, kk = lane >> 4;
    const int lrow = lane >> 5, lc = 2 * (lane & 31); // loads: lane = (row parity, column pair) of a 2-row x 64-column slab
    acc4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
        acc[t] = acc4{0, 0, 0, 0};
    struct Seg {
        const real *src;
        int w, wp;
        int32_t slot; // lane l: slot of a' of the segment's column l (the zero slot: no mirrored leaf's column / beyond the segment)
    };
    const int64_t q0 = A.seg_ptr[I], q1 = A.seg_ptr[I + 1];
    auto fetch = [&](int64_t q) {
        q = q < q1 ? q : q1 - 1;
        Seg s;
        s.src = A.stream + A.seg_src[q];
        s.w   = A.seg_w[q];
        s.wp  = A.seg_wp[q];
        const int32_t c = A.coef[A.seg_cb[q] + (lane < s.w ? lane : 0)];
        s.slot          = (lane < s.w && c >= 0) ? c : A.zero_slot;
        return s;
    };
    // a tile = 16 interval rows x 64 columns of one segment; rows beyond the interval re-read its last row (their results are never stored),
    // lanes beyond the segment its first pair (their operand is the zero slot)
    auto load_tile = [&](scalar2(&v)[8], const Seg &s, int t) {
        const int cl = lc < s.w ? lc : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int r = 16 * t + 2 * u + lrow;
            r     = r < ilen ? r : ilen - 1;
            v[u]  = stream_load(reinterpret_cast<const scalar2 *>(s.src + (int64_t)r * s.wp + cl));
        }
    };
    auto gather_b = [&](real(&b)[16], const Seg &s, int h0, int h1) { // a'[column 4 h + kk][rhs m] for the k-steps h0 <= h < h1
#pragma unroll
        for (int h = 0; h < 16; h++)
            if (h >= h0 && h < h1) {
                const int d = __shfl(s.slot, 4 * h + kk, WAVE);
                b[h]        = A.W16[(int64_t)d * 16 + m];
            }
    };
    // [64 columns][16 rows], element (row i, column c) at 16 (c ^ ((c >> 1) & 1)) + (i ^ ((c >> 1) & 15)): the stores of a load's two
    // rows x 64 columns and the operand reads of 16 rows x 4 columns both touch every bank exactly twice
    // refill: the segment's last tile -- every operand register is loaded with the NEXT segment's value right behind the MFMA that read it last
    // (one set of sixteen operand registers instead of two: the second set had cost the kernel its second wave per SIMD, or spills)
    auto tile_product = [&](const scalar2(&v)[8], real(&b)[16], acc4 &dst, const Seg &next, bool refill) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = 2 * u + lrow, sw = lane & 15, fl = lane & 1; // (c >> 1) & 15 and (c >> 1) & 1 of both columns lc, lc + 1
            tile[16 * (lc ^ fl) + (i ^ sw)]       = v[u].x;
            tile[16 * ((lc + 1) ^ fl) + (i ^ sw)] = v[u].y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // two halves of eight k-steps: eight operand registers live at a time (the sixteen cost two waves per SIMD their place in the file).
        // Every segment covers whole intervals, so the tile's result joins the interval's accumulator as the MFMAs' own C operand
#pragma unroll
        for (int half = 0; half < 2; half++) {
            real ta[8];
#pragma unroll
            for (int h = 0; h < 8; h++) {
                const int c = 4 * (8 * half + h) + kk;
                ta[h]       = tile[16 * (c ^ ((c >> 1) & 1)) + (m ^ ((c >> 1) & 15))];
            }
#pragma unroll
            for (int h = 0; h < 8; h++)
                dst = mfma16(ta[h], b[8 * half + h], dst); // A[m = row][k = column c], B[k][n = rhs]
            if (refill)
                gather_b(b, next, 8 * half, 8 * half + 8);
        }
    };
    if (q0 < q1) {
        // DIST tiles are in flight while one is computed (D = DIST + 1 buffers, used in turn)
        constexpr int DMAX  = 2;
        constexpr int D     = NT + 1 < DMAX ? NT + 1 : DMAX;
        constexpr int DIST  = D - 1;                   // <= NT: a prefetch never reaches beyond the next segment
        constexpr int SPI   = (NT % D) ? D : 1;        // segments per trip of the loop: the buffer of step n is n mod D, the pattern repeats after SPI segments
        static_assert((SPI * NT) % D == 0 && DIST <= NT, "buffer rotation");
        Seg cur = fetch(q0), nxt = fetch(q0 + 1);
        real bc[16];
        gather_b(bc, cur, 0, 16);
        scalar2 v[D][8];
#pragma unroll
        for (int t = 0; t < DIST; t++)
            load_tile(v[t], cur, t);
        HMX_SCHED_FENCE();
        for (int64_t q = q0; q < q1; q += SPI) {
#pragma unroll
            for (int p = 0; p < SPI; p++) {
                if (p > 0 && q + p >= q1)
                    break;
                const Seg nx2 = fetch(q + p + 2);
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const int n = p * NT + t; // (a constant after unrolling)
                    if (t + DIST < NT)
                        load_tile(v[(n + DIST) % D], cur, t + DIST);
                    else
                        load_tile(v[(n + DIST) % D], nxt, t + DIST - NT);
                    HMX_SCHED_FENCE();
                    tile_product(v[n % D], bc, acc[t], nxt, t == NT - 1);
                    HMX_SCHED_FENCE();
                }
                cur = nxt;
                nxt = nx2;
            }
        }
    }
    // dense mirrored contributions of the interval's rows (column sums the first pass left in SW16, found through the level index), y update.
    // A lane holds 4 NT rows (one right-hand side each): level k of all of them is fetched together -- independent chains of two loads per
    // level instead of one (the levels of a row are few, but every one is two dependent trips to memory)
    constexpr int NR = 4 * NT;
    int jr[NR], cn[NR], kmax = 0;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ir  = 16 * t + mfma16_row(real(0), lane, j);
            jr[4 * t + j] = r0 + (ir < ilen ? ir : ilen - 1);
            cn[4 * t + j] = ir < ilen ? A.count[jr[4 * t + j]] : 0;
            kmax          = cn[4 * t + j] > kmax ? cn[4 * t + j] : kmax;
        }
    // the y values of the lane are fetched NOW, together, unconditionally (rows beyond the interval read its last row, right-hand sides beyond
    // the group the group's first): with the load inside each row's own `if (row exists) y = ...` the compiler emitted load -> wait -> store
    // once per row, one trip to memory after the other at the end of every interval (round 5, read off the ISA)
    const bool need_y = A.accumulate || !(A.beta == real(0));
    const int mcol    = cbase + (m < nrhs ? m : 0);
    real yv[NR];
#pragma unroll
    for (int e = 0; e < NR; e++)
        yv[e] = need_y ? A.y[(int64_t)jr[e] * mu + mcol] : real(0);
    for (int k = 0; k < kmax; k++) {
        int32_t d[NR];
#pragma unroll
        for (int e = 0; e < NR; e++)
            d[e] = A.fidx[(int64_t)(k < cn[e] ? k : 0) * A.n + jr[e]]; // (level 0 of the row when it has fewer: a valid entry, dropped below)
        real w[NR];
#pragma unroll
        for (int e = 0; e < NR; e++)
            w[e] = A.W16[(int64_t)(k < cn[e] ? d[e] : A.zero_slot) * 16 + m];
#pragma unroll
        for (int e = 0; e < NR; e++)
            acc[e >> 2][e & 3] += w[e];
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ir   = 16 * t + mfma16_row(real(0), lane, j);
            const real y0  = yv[4 * t + j];
            const real out = A.accumulate ? y0 + A.alpha * acc[t][j] : (A.beta == real(0) ? A.alpha * acc[t][j] : A.alpha * acc[t][j] + A.beta * y0);
            if (ir < ilen && m < nrhs)
                A.y[(int64_t)(r0 + ir) * mu + cbase + m] = out;
        }
}
template <int WAVES>
__global__ __launch_bounds__(WAVES *WAVE) HMX_WPE_ROWSYM_MFMA16_KERNEL void rowsym_mfma16_kernel(RowSegArgs A, int mu, int cbase, int nrhs) {
    __shared__ __attribute__((aligned(16))) real lds[WAVES * 64 * 16];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pos = blockIdx.x * WAVES + wv;
    if (pos >= A.nint)
        return; // (no workgroup barrier below: the waves are independent)
    const int I  = A.order[pos];
    const int r0 = A.int_off[I], ilen = A.int_off[I + 1] - r0;
    real *tile = lds + wv * 64 * 16;
    switch ((ilen + 15) >> 4) { // wave-uniform
    case 4: rowsym_mfma16_run<4>(A, tile, I, r0, ilen, lane, mu, cbase, nrhs); break;
    case 3: rowsym_mfma16_run<3>(A, tile, I, r0, ilen, lane, mu, cbase, nrhs); break;
    case 2: rowsym_mfma16_run<2>(A, tile, I, r0, ilen, lane, mu, cbase, nrhs); break;
    case 1: rowsym_mfma16_run<1>(A, tile, I, r0, ilen, lane, mu, cbase, nrhs); break;
    default: break; // an empty interval
    }
}

#endif // !HMX_COMPLEX
