// kernels_util.hpp -- permutations, conjugation, bulk download and other small kernels.
// Part of the engine's device code: included by kernels_body.hpp inside namespace hmx::{f64,f32,z64,c32}, written against `scalar` / `real`.  No include guard on purpose.

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
