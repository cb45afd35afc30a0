// engine_state.hpp -- the HBM-resident operator of one coefficient type: launch-order helper, stream sets, struct HMat.
// Part of the engine's host code: included by engine_body.hpp inside namespace hmx::{f64,f32,z64,c32}.  No include guard on purpose.

#ifndef HMX_ROWSYM_WAVES
#define HMX_ROWSYM_WAVES 4 // intervals (= waves) per workgroup of rowsym_mfma16_kernel
#endif
// Launch order that keeps the tasks of one UNIT (tasks that gather the same operand rows: the row ranges of a few hundred consecutive
// rows, the chunks of the pieces over the same rows of x) on one XCD, one after the other, so that a unit's operand rows are fetched from
// HBM once and then found in that XCD's L2.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one: observed, used for
// speed only -- MI355X_MICROARCH.md, Workgroup dispatch); `per_group` consecutive launch positions belong to one workgroup.  Units are dealt
// heaviest first, round-robin over the eight lists, each list exactly as long as the number of positions of its label (a unit that does not
// fit is continued on the next list with room).  With several right-hand sides an operand row is 16 values: without this the multi-RHS
// kernels fetched 19-45 % more than their streams (profiles/r5_*_pmc_summary.json).
static std::vector<int32_t> xcd_deal(const std::vector<int64_t> &unit, const std::vector<int64_t> &weight, int per_group) {
    const int64_t n = (int64_t)unit.size();
    std::vector<int32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return unit[a] < unit[b]; }); // tasks of a unit adjacent, in their given order
    struct U {
        int64_t first, count, w;
    };
    std::vector<U> units;
    for (int64_t i = 0; i < n;) {
        int64_t j = i, w = 0;
        while (j < n && unit[idx[j]] == unit[idx[i]])
            w += weight[idx[j++]];
        units.push_back({i, j - i, w});
        i = j;
    }
    std::stable_sort(units.begin(), units.end(), [](const U &a, const U &b) { return a.w > b.w; });
    constexpr int X = 8;
    const int64_t ngroups = (n + per_group - 1) / per_group;
    int64_t cap[X] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t g = 0; g < ngroups; g++)
        cap[g % X] += std::min<int64_t>(per_group, n - g * per_group);
    std::vector<int32_t> list[X];
    int cursor = 0;
    for (const U &u : units) {
        int64_t done = 0;
        while (done < u.count) {
            while ((int64_t)list[cursor].size() >= cap[cursor])
                cursor = (cursor + 1) % X;
            const int64_t take = std::min<int64_t>(u.count - done, cap[cursor] - (int64_t)list[cursor].size());
            for (int64_t k = 0; k < take; k++)
                list[cursor].push_back(idx[u.first + done + k]);
            done += take;
            if (done < u.count)
                cursor = (cursor + 1) % X;
        }
        cursor = (cursor + 1) % X;
    }
    std::vector<int32_t> order(n);
    int64_t used[X] = {0, 0, 0, 0, 0, 0, 0, 0}, p = 0;
    for (int64_t g = 0; g < ngroups; g++) {
        const int x = (int)(g % X);
        for (int64_t k = 0; k < per_group && p < n; k++)
            order[p++] = list[x][used[x]++];
    }
    return order;
}

struct StreamSet {
    std::vector<int32_t> off, len, cols, cw; // per range: local offset, rows, columns, chunk width (R only)
    std::vector<int64_t> base, colbase;  // per range: first element in `stream`, first entry in index arrays
    int64_t elems = 0, total_cols = 0;
    DArr<int32_t> d_off, d_len, d_cols, d_cw;
    DArr<int64_t> d_base, d_colbase;
    DArr<scalar> stream;
    // R: one task per (range, column chunk), heaviest first.  E: task_range = launch order of the ranges.
    std::vector<int32_t> task_range, task_chunk;
    DArr<int32_t> d_task_range, d_task_chunk;
    int nranges() const { return (int)off.size(); }
    hipError_t upload_meta() {
        hipError_t e;
        if ((e = d_off.upload(off)) != hipSuccess) return e;
        if ((e = d_len.upload(len)) != hipSuccess) return e;
        if ((e = d_cols.upload(cols)) != hipSuccess) return e;
        if ((e = d_cw.upload(cw)) != hipSuccess) return e;
        if ((e = d_base.upload(base)) != hipSuccess) return e;
        if ((e = d_colbase.upload(colbase)) != hipSuccess) return e;
        if ((e = d_task_range.upload(task_range)) != hipSuccess) return e;
        return d_task_chunk.upload(task_chunk);
    }
};


struct HMat {
    int device = 0;
    Options opt = Options::from_environment(); // hmx_hmatrix_set_option; the environment gives the initial values only (read here, once)
    // structure (copied from the block tree)
    std::vector<hmx_leaf> leaves;
    std::vector<int> kind; // LeafKind per leaf
    int T0 = 0, nT = 0, S0 = 0, nS = 0;
    int nT_total = 0, nS_total = 0;
    char symmetry_for_leaves = 'N', uplo_for_leaves = 'N';
    double build_epsilon = 0;  // accuracy the low-rank leaves were built with (LowRankMatrix::get_epsilon)
    bool has_mirror = false;   // the block tree has leaves_for_symmetry
    bool sym_expanded = false; // ... and they were laid out explicitly (no mirror pass needed)
    // compact symmetric storage, fused product (expand_sym_kernel / rowsym_kernel): slots in SW = [a' | EW (column sums, E-column order)]
    bool sym_fused = false;
    bool trans_tables_failed = false;
    bool trans_fused = false; // tables of the transposed product on the stored data present (build_trans_tables): s_* below, output rows = source positions
    DArr<int32_t> s_mdst, s_coef, s_count, s_list, s_fidx;
    DArr<int64_t> s_sub_ptr, s_sub_src, s_sub_cb;
    DArr<int32_t> s_sub_w, s_sub_nrows, s_sub_dst, s_int_order; // second R sweep: per row interval the (parts of) tasks inside it, a ready record each (RowSymArgs)
    int s_nint = 0;
    // ... and for the multi-RHS form (rowsym_mfma16_kernel: intervals of 64 rows, one wave each); SW16 = [slot][16] partial sums of one sweep
    DArr<int32_t> s64_int_off, s64_int_order; // interval I = output rows [int_off[I], int_off[I + 1]) (at most SYM_IR_MU, cut at the mirrored pieces' boundaries)
    DArr<int64_t> s64_seg_ptr, s64_seg_src, s64_seg_cb; // per interval its segments; per segment: first stream element, first entry of the coefficient-slot table
    DArr<int32_t> s64_seg_wp, s64_seg_w;                // ... row pitch of the chunk, columns of the segment (<= 64)
    int s64_nint = 0;
    int64_t s_slots = 0; // slots of SW (a' | column sums)
    // groups of row ranges of the mirrored sweeps (build_mirror_tables): launch order, first flush slot and number of LDS accumulators per group
    DArr<int32_t> s_grp_order, s_grp_flush, s_grp_na;
    int s_ngroups = 0, s_group = 1, s_gcap = 0; // groups, ranges per group, the largest number of accumulators of a group
    DArr<scalar> SW16;
    DArr<int32_t> sc_dst, sc_lp, sc_count, sc_k;
    int n_sym_combine = 0, n_sym_combine_wave = 0; // the first n_sym_combine_wave entries have >= 32 partial sums: one wave each
    int s_kmax        = 0;
    DArr<scalar> SW;
    std::vector<int64_t> staged_off;
    std::vector<int32_t> perm_t, perm_s; // full permutations (cluster -> user)
    bool t_root_is_tree_root = false, perm_local = false;
    // the cluster trees' nodes as (offset, size, first child, number of children), GLOBAL cluster positions: the R-stream pieces of a
    // source cluster larger than SR_MAX follow the tree (its descendants of at most SR_MAX rows), so the pieces of all cluster levels nest
    struct TreeNode {
        int32_t off, size, first_child, n_children;
    };
    std::vector<TreeNode> tree_t, tree_s;

    // generator
    // host generator: VirtualGenerator::copy_submatrix semantics (user numbering, column-major output)
    void (*callback)(void *, int, int, const int32_t *, const int32_t *, scalar *) = nullptr;
    void *callback_user = nullptr;
    int callback_threads = 0; // host threads that may call the generator concurrently (hmx_hmatrix_set_callback_threads): 0 = option HMX_OPT_CALLBACK_THREADS (whose 0 = all cores), 1 = the calling thread only
    DArr<scalar> dense_stage; // dense leaves evaluated by the host generator (pack_dense reads them from here)
    bool has_kernel = false;
    KernelSpec ks{};
    DArr<double> tx, ty, tz, sx, sy, sz; // cluster-order coordinates (SoA)

    // per-leaf metadata on device
    DArr<int32_t> d_t_off, d_t_size, d_s_off, d_s_size, d_rank, d_swapped, d_sym_uplo, d_transposed, d_conj;
    DArr<int64_t> d_colptr, d_cross_off, d_staged_off;
    std::vector<int64_t> colptr;
    std::vector<int32_t> swapped;
    // compressed data before packing ("crosses": [uu_k | vv_k]) and staged dense uploads
    DArr<scalar> pool;
    unsigned long long pool_used = 0;
    // host staging for the upload path
    std::vector<std::vector<scalar>> staged_U, staged_V, staged_D;

    // streams
    StreamSet E, R;
    std::vector<int32_t> dp_leaf, dp_range, dp_col; // (dense leaf, row range, first column in the range) of every slice of a dense leaf, leaf-major
    DArr<int32_t> e_zidx;
    DArr<int32_t> r_outidx;
    DArr<int32_t> c_dst, c_src, c_stride, c_count;
    int n_combine       = 0;
    int64_t A_total     = 0, P_total = 0;
    int64_t zero_slot   = 0;
    DArr<scalar> Z, Zmu;
    PlacementReport placed_z, placed_zmu, placed_sw, placed_sw16; // what place_written measured for the arrays the sweeps write
    // Where the arrays written next to the E- / R-streams go (fraction of the slab's extent, -1: first fit; *_known: measured or found in the
    // PlacementCache).  Measured only while may_probe is set: inside hmx_hmatrix_compress / finalize / recompress / prepare, never in a product
    // call -- arrays a product allocates later go to the known place without measuring anything (place_array).
    double place_e = -1, place_r = -1;
    bool place_e_known = false, place_r_known = false, may_probe = false;
    DArr<scalar> tmp_in, tmp_out, tmp_in2, tmp_out2; // staging for host vectors / permutations / multi-RHS
    DArr<scalar> conj_in;                             // conjugated input of a trans = 'C' product
    DArr<scalar> mm_in, mm_out;                       // row-major cluster-numbered operands of the column-major front end
    // trans = 'T': the transposed operator laid out in its own streams (built on first use from the same crosses /
    // generator, see ensure_transposed_operator); `view_of` is set in that object and points back to the owner
    std::unique_ptr<HMat> T_op;
    const HMat *view_of = nullptr;
    bool T_op_failed    = false;
    // compact symmetric storage, several right-hand sides: the fused multi-RHS kernels run on an EXPANDED view of the operator
    // (same orientation, mirrored leaves laid out explicitly), built on the first such product (ensure_expanded_view)
    std::unique_ptr<HMat> X_op;
    bool X_op_failed      = false;
    bool view_transposed  = true; // of a view: rows and columns exchanged with respect to the owner
    bool factors_released = false; // hmx_hmatrix_release_factors: the cross pool was given back, only the streams remain
    DArr<int32_t> d_perm_t, d_perm_s;
    bool finalized = false;
    // expand stage in row chunks (hmx_dist overlap: the exchange of chunk c runs under the kernel of chunk c + 1): contiguous groups of
    // row ranges with about equal work, each group launched heaviest-first
    int chunk_plan_n = 0;
    std::vector<int32_t> chunk_first, chunk_count, chunk_row_lo, chunk_row_hi;
    DArr<int32_t> d_chunk_order;

    std::map<void *, std::unique_ptr<DArr<scalar>>> user_vectors; // hmx_hmatrix_alloc_vector

    hmx_stats stats{};
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    std::vector<const char *> ev_names;
    std::vector<float> last_ms;
    std::vector<const char *> last_names;

    ~HMat() {
        for (auto e : ev)
            (void)hipEventDestroy(e);
    }
};

// An array a sweep writes while it reads the E-stream (pair = 0) or the R-stream (pair = 1): where that pair runs fastest.  The place is
// measured at most once per operator and stream, and only while H.may_probe is set (builds and hmx_hmatrix_prepare); everywhere else -- every
// product call -- the array goes to the place already known, or to wherever first fit puts it, and nothing is launched but its zero-fill on `st`.
static hipError_t place_array(HMat &H, DArr<scalar> &arr, size_t count, int pair, hipStream_t st, PlacementReport *rep = nullptr, hipEvent_t after = nullptr) {
    const StreamSet &S = pair == 0 ? H.E : H.R;
    double &frac       = pair == 0 ? H.place_e : H.place_r;
    bool &known        = pair == 0 ? H.place_e_known : H.place_r_known;
    if (H.may_probe && !known && H.opt.i(HMX_OPT_PLACE_WRITTEN) != 0) {
        PlacementReport r;
        const hipError_t e = place_written(arr, count, S.stream.d, (size_t)S.elems * sizeof(scalar), true, &r, after);
        if (e == hipSuccess && r.known)
            frac = r.frac, known = true;
        if (rep)
            *rep = r;
        return e;
    }
    return place_like(arr, count, known ? frac : -1.0, st);
}
struct ProbeScope { // placement probes are allowed inside this scope only (builds, hmx_hmatrix_prepare)
    HMat &H;
    const bool before;
    explicit ProbeScope(HMat &h) : H(h), before(h.may_probe) { H.may_probe = true; }
    ~ProbeScope() { H.may_probe = before; }
};
