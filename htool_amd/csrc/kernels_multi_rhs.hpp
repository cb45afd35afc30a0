// kernels_multi_rhs.hpp -- several right-hand sides in one sweep: VALU kernels, matrix-core kernels (real: 16 / 32, complex: 8 / 16 right-hand sides).
// Part of the engine's device code: included by kernels_body.hpp inside namespace hmx::{f64,f32,z64,c32}, written against `scalar` / `real`.  No include guard on purpose.

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
    // the partial sums four at a time: four loads in flight, the additions in the order they always had (bitwise the same result)
    const int64_t step = (int64_t)st * mu;
    scalar s = scalar(0);
    int k = 0;
    for (; k + 4 <= cnt; k += 4) {
        const scalar v0 = p[k * step], v1 = p[(k + 1) * step], v2 = p[(k + 2) * step], v3 = p[(k + 3) * step];
        s += v0;
        s += v1;
        s += v2;
        s += v3;
    }
    for (; k < cnt; k++)
        s += p[k * step];
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
__global__ __launch_bounds__(WAVES *WAVE) HMX_WPE_EXPAND_MFMA16S_KERNEL void expand_mfma16s_kernel(ExpandArgs A, int mu, int cbase, int nrhs) {
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
__global__ __launch_bounds__(WAVES *WAVE) HMX_WPE_REDUCE_MFMA16S_KERNEL void reduce_mfma16s_kernel(ReduceArgs A, int mu, int cbase, int nrhs) {
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
