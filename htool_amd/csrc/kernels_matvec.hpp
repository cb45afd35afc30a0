// kernels_matvec.hpp -- the single-vector H-matvec, trans = 'N': reduce, combine, expand.
// Part of the engine's device code: included by kernels_body.hpp inside namespace hmx::{f64,f32,z64,c32}, written against `scalar` / `real`.  No include guard on purpose.

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
