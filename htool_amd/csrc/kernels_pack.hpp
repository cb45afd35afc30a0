// kernels_pack.hpp -- packing of the compressed blocks into the E- / R-streams.
// Part of the engine's device code: included by kernels_body.hpp inside namespace hmx::{f64,f32,z64,c32}, written against `scalar` / `real`.  No include guard on purpose.

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
    // consecutive threads take consecutive COLUMNS of one row: the leaf's r coefficients of a row are adjacent in the (row-major) chunk, one
    // run of r elements per row instead of 64 scattered 8-byte stores per wave; the reads (r crosses, neighbouring elements by neighbouring
    // rows) come out of L1 / L2.  (With the rows on consecutive threads this kernel moved its 15.7 GB at 1.1 TB/s.)
    for (int e = threadIdx.x; e < r * len; e += blockDim.x) {
        const int i = e / r, k = e - i * r;
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

// The index arrays of the product kernels, filled on the device from the pair lists the pack kernels already have (round 6: the host filled
// 23.6 M entries and uploaded 94 MB at N = 1e6 -- 24 ms of a 150 ms build):
//   E: column c of (leaf b, range r) gathers Z[z0[b] + c], z0 = nS + a-offset of a low-rank leaf, the first source position of a dense one
//   R: column k of (leaf b, piece s) writes Z[o0 + k], o0 = the leaf's a slots (one piece) or its partial slots of that piece
struct FillIndexArgs {
    const int32_t *pair_block, *pair_range, *pair_col;
    const int64_t *range_colbase;
    const int32_t *ncols; // per leaf: rank (low rank) or source size (dense)
    const int32_t *z0;    // E: per leaf; R: per PAIR (o0)
    int32_t *out;
    int per_pair;         // z0 indexed by pair (R) instead of by leaf (E)
};
__global__ __launch_bounds__(256) void fill_index_kernel(FillIndexArgs A, int64_t npairs) {
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npairs)
        return;
    const int lane = threadIdx.x & 63;
    const int b = A.pair_block[p], n = A.ncols[b], z = A.per_pair ? A.z0[p] : A.z0[b];
    int32_t *dst = A.out + A.range_colbase[A.pair_range[p]] + A.pair_col[p];
    for (int j = lane; j < n; j += 64)
        dst[j] = z + j;
}
