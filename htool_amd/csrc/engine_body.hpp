// engine_body.hpp -- host orchestration of the device engine, written against `scalar` (coefficient type; `real` is its
// underlying real type) and included four times by engine.hip: namespaces hmx::f64, hmx::f32 (scalar = double / float) and
// hmx::z64, hmx::c32 (scalar = cplx<double> / cplx<float>, HMX_COMPLEX = 1).  No include guard on purpose.

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
    DArr<int64_t> s_sub_ptr;
    DArr<int32_t> s_sub_task, s_sub_row0, s_sub_nrows, s_sub_dst, s_int_order; // second R sweep: per row interval the (parts of) tasks inside it
    int s_nint = 0;
    // ... and for the multi-RHS form (rowsym_mfma16_kernel: intervals of 64 rows, one wave each); SW16 = [slot][16] partial sums of one sweep
    DArr<int64_t> s64_sub_ptr;
    DArr<int32_t> s64_sub_task, s64_sub_row0, s64_sub_nrows, s64_sub_dst, s64_int_order;
    int s64_nint = 0;
    int64_t s_slots = 0; // slots of SW (a' | column sums)
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
    hvec32 h_e_zidx;
    DArr<int32_t> c_dst, c_src, c_stride, c_count;
    int n_combine       = 0;
    int64_t A_total     = 0, P_total = 0;
    int64_t zero_slot   = 0;
    DArr<scalar> Z, Zmu;
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


// ---- mirrored products: slots of the partial results ------------------------------------------------------------------------------
// The tables behind the fused symmetric product (every leaf of the stored triangle is also applied transposed) and -- `tmode`, round 4 --
// behind the TRANSPOSED product of an ordinary operator on its stored data (every leaf is applied transposed ONLY; the reference swaps the
// cluster roles on the same leaves, hmatrix/linalg/add_hmatrix_vector_product.hpp:74-81): the output rows are then the SOURCE positions.
struct MirrorCtx {
    const std::vector<hmx_leaf> &XL;
    const std::vector<int> &XK;
    int64_t nb;
    const std::vector<int32_t> &elr_b, &elr_r, &elr_c, &ed_b, &ed_r, &ed_c, &rlr_b, &rlr_r, &rlr_c; // (leaf, range, first column) pairs of the E- / R-streams
    const std::vector<int64_t> &aoff;
    int64_t A_total;
    bool tmode;
    std::function<void(const char *)> phase;
};
static int build_mirror_tables(HMat &H, const MirrorCtx &M) {
    StreamSet &E = H.E, &R = H.R;
    const std::vector<hmx_leaf> &XL = M.XL;
    const std::vector<int> &XK      = M.XK;
    const int64_t nb = M.nb, A_total = M.A_total;
    const std::vector<int32_t> &elr_b = M.elr_b, &elr_r = M.elr_r, &elr_c = M.elr_c, &ed_b = M.ed_b, &ed_r = M.ed_r, &ed_c = M.ed_c, &rlr_b = M.rlr_b, &rlr_r = M.rlr_r, &rlr_c = M.rlr_c;
    const std::vector<int64_t> &aoff = M.aoff;
    const bool tmode = M.tmode;
    auto phase_nosync = [&](const char *name) {
        if (M.phase)
            M.phase(name);
    };
    const int nOut    = tmode ? H.nS : H.nT;     // output rows of the mirrored products
    const int r_shift = tmode ? 0 : H.S0 - H.T0; // R piece offset (source-local) -> output row
    const int d_base  = tmode ? H.S0 : H.T0;     // global column of a dense leaf -> output row
    auto is_mir       = [&](int64_t b) { return tmode || XL[b].mirror != 0; };
    // W = [a' | EW].  expand_sym_kernel stores the column sums of a row range at EW[epad(range) + column] (E-column order: one
    // contiguous, 128-byte aligned run per range).  combine_list_kernel folds the partial a' of a leaf that spans several ranges
    // through a list of its column-group positions.  The second R sweep (rowsym_kernel) is owner-computes: one workgroup per interval
    // of SYM_IR target rows applies every (part of a) task inside it, folds the row sums in LDS, adds the interval's dense mirrored
    // column sums (EW, through a level-major index) and updates y once.  All in a fixed order: results are bit-reproducible.
    hvec32 s_mdst, s_coef;
    std::vector<int32_t> s_cnt, s_cd, s_clp, s_cc, s_ck, s_list;
    std::unique_ptr<int32_t[]> s_fidx; // level-major, s_kmax x nT: left uninitialised (only the entries below count[j] are ever read)
    size_t s_fidx_n = 0;
    std::vector<int64_t> s_sub_ptr;
    std::vector<int32_t> s_sub_task, s_sub_row0, s_sub_nrows, s_sub_dst, s_int_order;
    std::vector<int64_t> p64; // the same tables for intervals of 64 rows (multi-RHS form of the second sweep)
    std::vector<int32_t> t64, r64, n64, d64, o64;
    H.n_sym_combine = 0;
    H.s_kmax        = 0;
    int64_t s_total = 0;
    s_mdst.resize(E.total_cols); // sized without initialisation, filled by several threads
    s_coef.resize(R.total_cols);
    parallel_for(s_mdst.size(), [&](size_t lo, size_t hi) { std::fill(s_mdst.begin() + lo, s_mdst.begin() + hi, -1); });
    parallel_for(s_coef.size(), [&](size_t lo, size_t hi) { std::fill(s_coef.begin() + lo, s_coef.begin() + hi, -1); });
    s_cnt.assign(nOut, 0);
    bool bad = false;
    std::vector<int64_t> epad(E.nranges());
    int64_t EWN = 0;
    for (int r = 0; r < E.nranges(); r++) {
        epad[r] = EWN;
        EWN += (E.cols[r] + 15) & ~15;
    }
    const int64_t EWBASE = (A_total + 15) & ~int64_t(15), RWBASE = EWBASE + EWN;
    // low-rank mirrored leaves: column sums land in EW; a leaf inside ONE range is complete there (a' is read from EW),
    // otherwise a list of its column-group positions feeds combine_list_kernel, which writes a'[aoff + k]
    std::vector<int32_t> nrange(nb, 0);
    for (size_t p = 0; p < elr_b.size(); p++)
        nrange[elr_b[p]]++;
    std::vector<int64_t> lptr(nb, -1);
    int64_t LN = 0;
    for (int64_t b = 0; b < nb; b++)
        if (is_mir(b) && XK[b] == LK_LOWRANK && XL[b].rank > 0 && nrange[b] > 1) {
            lptr[b] = LN;
            LN += nrange[b];
        }
    phase_nosync("  sym: setup");
    s_list.assign(LN, 0);
    std::vector<int64_t> single_slot(nb, -1);
    {
        // the pairs of a leaf are consecutive in elr_* (leaf-major) and cover consecutive ranges: position in the leaf's list = r - first range
        std::vector<int32_t> first_range(nb, -1);
        for (size_t p = 0; p < elr_b.size(); p++)
            if (first_range[elr_b[p]] < 0)
                first_range[elr_b[p]] = elr_r[p];
        parallel_for(elr_b.size(), [&](size_t lo, size_t hi) {
            for (size_t p = lo; p < hi; p++) {
                const int b = elr_b[p], r = elr_r[p];
                if (!is_mir(b))
                    continue;
                const int64_t base = EWBASE + epad[r] + elr_c[p];
                if (nrange[b] == 1)
                    single_slot[b] = base;
                else
                    s_list[lptr[b] + (r - first_range[b])] = (int32_t)base;
                int32_t *dst = s_mdst.data() + E.colbase[r] + elr_c[p];
                for (int k = 0; k < XL[b].rank; k++)
                    dst[k] = (int32_t)(base + k);
            }
        });
    }
    phase_nosync("  sym: lr columns");
    H.n_sym_combine_wave = 0;
    {
        size_t entries = 0;
        for (int64_t b = 0; b < nb; b++)
            if (lptr[b] >= 0)
                entries += (size_t)XL[b].rank;
        for (auto *v : {&s_cd, &s_clp, &s_cc, &s_ck})
            v->reserve(entries);
    }
    for (int pass = 0; pass < 2; pass++) // entries with many partial sums first (one wave each), then the rest (one thread each)
        for (int64_t b = 0; b < nb; b++)
            if (lptr[b] >= 0 && (nrange[b] >= 32) == (pass == 0)) {
                for (int k = 0; k < XL[b].rank; k++) {
                    s_cd.push_back((int32_t)(aoff[b] + k));
                    s_clp.push_back((int32_t)lptr[b]);
                    s_cc.push_back(nrange[b]);
                    s_ck.push_back(k);
                }
                if (pass == 0)
                    H.n_sym_combine_wave += XL[b].rank;
            }
    phase_nosync("  sym: combine entries");
    parallel_for(rlr_b.size(), [&](size_t lo, size_t hi) {
        for (size_t p = lo; p < hi; p++) {
            const int b = rlr_b[p];
            if (!is_mir(b))
                continue;
            const int64_t base = single_slot[b] >= 0 ? single_slot[b] : aoff[b];
            int32_t *dst       = s_coef.data() + R.colbase[rlr_r[p]] + rlr_c[p];
            for (int k = 0; k < XL[b].rank; k++)
                dst[k] = (int32_t)(base + k);
        }
    });
    phase_nosync("  sym: coef");
    // Second R sweep, owner-computes: the target rows are cut into intervals of SYM_IR rows and ONE workgroup per interval applies
    // every (piece, chunk) task -- or the part of it -- whose rows lie in the interval, folds the row sums of its waves in LDS,
    // adds the interval's dense mirrored contributions (EW, through the level index) and updates y once.  No partial row sums
    // leave the chip (they were 76 MB per product at N=1e6, written and read again), no separate folding kernel.
    const size_t ntask = R.task_range.size();
    std::vector<char> task_mirror(ntask, 0);
    for (size_t t = 0; t < ntask; t++) {
        const int r = R.task_range[t], ch = R.task_chunk[t], cw = R.cw[r];
        const int w = std::min(cw, R.cols[r] - ch * cw);
        const int32_t *cf = s_coef.data() + R.colbase[r] + (int64_t)ch * cw;
        bool any = false;
        for (int c = 0; c < w && !any; c++)
            any = cf[c] >= 0;
        if (!any)
            continue;
        const int j0 = R.off[r] + r_shift;
        if (j0 < 0 || j0 + R.len[r] > nOut) {
            bad = true;
            break;
        }
        task_mirror[t] = 1;
    }
    // the sub-task lists of the intervals of IR rows (launch order of the tasks = order inside every interval's list), heaviest interval first
    auto build_intervals = [&](int IR, std::vector<int64_t> &sub_ptr, std::vector<int32_t> &sub_task, std::vector<int32_t> &sub_row0, std::vector<int32_t> &sub_nrows,
                               std::vector<int32_t> &sub_dst, std::vector<int32_t> &int_order, int per_group) -> int {
        const int nint = (nOut + IR - 1) / IR;
        std::vector<int64_t> sub_count(nint + 1, 0);
        for (size_t t = 0; t < ntask; t++) {
            if (!task_mirror[t])
                continue;
            const int r = R.task_range[t], j0 = R.off[r] + r_shift;
            for (int I = j0 / IR; I <= (j0 + R.len[r] - 1) / IR; I++)
                sub_count[I + 1]++;
        }
        for (int I = 0; I < nint; I++)
            sub_count[I + 1] += sub_count[I];
        sub_ptr            = sub_count;
        const int64_t nsub = sub_count[nint];
        sub_task.assign(nsub, 0);
        sub_row0.assign(nsub, 0);
        sub_nrows.assign(nsub, 0);
        sub_dst.assign(nsub, 0);
        std::vector<double> int_work(nint, 0.0);
        std::vector<int64_t> pos(sub_count.begin(), sub_count.end() - 1);
        for (size_t t = 0; t < ntask; t++) {
            if (!task_mirror[t])
                continue;
            const int r = R.task_range[t], ch = R.task_chunk[t], cw = R.cw[r];
            const int w = std::min(cw, R.cols[r] - ch * cw);
            const int j0 = R.off[r] + r_shift, j1 = j0 + R.len[r];
            for (int I = j0 / IR; I <= (j1 - 1) / IR; I++) {
                const int lo = std::max(j0, I * IR), hi = std::min(j1, (I + 1) * IR);
                const int64_t q = pos[I]++;
                sub_task[q]  = (int32_t)t;
                sub_row0[q]  = lo - j0;
                sub_nrows[q] = hi - lo;
                sub_dst[q]   = lo - I * IR;
                int_work[I] += (double)(hi - lo) * w + 256;
            }
        }
        int_order.resize(nint);
        std::iota(int_order.begin(), int_order.end(), 0);
        if (H.opt.i(HMX_OPT_TASK_ORDER) == 3) { // intervals over the same rows gather the same a' (see xcd_deal)
            const int unit_rows = std::max(IR, H.opt.i(HMX_OPT_XCD_UNIT_ROWS));
            std::vector<int64_t> unit(nint), wk(nint);
            for (int I = 0; I < nint; I++) {
                unit[I] = (int64_t)I * IR / unit_rows;
                wk[I]   = (int64_t)int_work[I];
            }
            int_order = xcd_deal(unit, wk, per_group);
        } else
            std::stable_sort(int_order.begin(), int_order.end(), [&](int a, int b) { return int_work[a] > int_work[b]; });
        return nint;
    };
    int nint = 0;
    if (!bad)
        nint = build_intervals(SYM_IR, s_sub_ptr, s_sub_task, s_sub_row0, s_sub_nrows, s_sub_dst, s_int_order, 1);
    // the same for the multi-RHS form of the second sweep (rowsym_mfma16_kernel: one wave per 64 rows; rowsym_mu_kernel: one workgroup)
    H.s64_nint = 0;
    if (!bad)
        H.s64_nint = build_intervals(SYM_IR_MU, p64, t64, r64, n64, d64, o64, HMX_ROWSYM_WAVES); // (a wave per interval in rowsym_mfma16_kernel)
    phase_nosync("  sym: tasks");
    // dense mirrored columns: contributions per output row, numbered in layout order ("levels")
    for (size_t p = 0; p < ed_b.size() && !bad; p++) {
        const hmx_leaf &l = XL[ed_b[p]];
        if (!(tmode || l.mirror))
            continue;
        const int j0 = l.s_offset - d_base;
        if (j0 < 0 || j0 + l.s_size > nOut) {
            bad = true;
            break;
        }
        for (int j = 0; j < l.s_size; j++)
            s_cnt[j0 + j]++;
    }
    if (bad) {
        set_error("symmetric storage needs the mirrored leaves' source clusters inside the target rows of the operator");
        return HMX_ERR_UNSUPPORTED;
    }
    s_total = RWBASE;
    for (int32_t c : s_cnt)
        H.s_kmax = std::max(H.s_kmax, (int)c);
    if (s_total >= (int64_t(1) << 31) - 1 || (int64_t)H.s_kmax * nOut >= (int64_t(1) << 40)) {
        set_error("operator too large for 32-bit slots of the fused symmetric product (HMX_SYM_EXPANDED=1 selects the expanded layout)");
        return HMX_ERR_UNSUPPORTED;
    }
    phase_nosync("  sym: dense count");
    s_fidx_n = (size_t)H.s_kmax * nOut;
    s_fidx.reset(new int32_t[std::max<size_t>(s_fidx_n, 1)]);
    std::vector<int32_t> fill(nOut, 0);
    // every thread owns an interval of the mirrored columns and walks ALL pairs (leaf-major), clipped to its interval: the levels of a
    // column are numbered in the pairs' order, as the one-thread loop numbers them
    parallel_for((size_t)nOut, [&](size_t clo, size_t chi) {
        for (size_t p = 0; p < ed_b.size(); p++) {
            const int b = ed_b[p], r = ed_r[p];
            const hmx_leaf &l = XL[b];
            if (!(tmode || l.mirror))
                continue;
            const int j0 = l.s_offset - d_base;
            const int ja = std::max(0, (int)clo - j0), jb = std::min((int)l.s_size, (int)chi - j0);
            if (ja >= jb)
                continue;
            const int64_t base = EWBASE + epad[r] + ed_c[p];
            int32_t *dst       = s_mdst.data() + E.colbase[r] + ed_c[p];
            for (int j = ja; j < jb; j++) {
                dst[j]                                               = (int32_t)(base + j);
                s_fidx[(size_t)(fill[j0 + j]++) * nOut + (j0 + j)] = (int32_t)(base + j);
            }
        }
    });
    H.s_nint = nint;
    phase_nosync("  sym: fidx fill");
    H.n_sym_combine = (int)s_cd.size();
    phase_nosync("fused symmetric slots");

    // ---- uploads (the first one waits for whatever is queued on the null stream: the pack kernels of build_streams) ----
    HMX_HIP(H.s_mdst.upload(s_mdst));
    HMX_HIP(H.s_coef.upload(s_coef));
    HMX_HIP(H.s_count.upload(s_cnt));
    HMX_HIP(H.s_sub_ptr.upload(s_sub_ptr));
    HMX_HIP(H.s_sub_task.upload(s_sub_task));
    HMX_HIP(H.s_sub_row0.upload(s_sub_row0));
    HMX_HIP(H.s_sub_nrows.upload(s_sub_nrows));
    HMX_HIP(H.s_sub_dst.upload(s_sub_dst));
    HMX_HIP(H.s_int_order.upload(s_int_order));
    HMX_HIP(H.sc_dst.upload(s_cd));
    HMX_HIP(H.sc_lp.upload(s_clp));
    HMX_HIP(H.sc_count.upload(s_cc));
    HMX_HIP(H.sc_k.upload(s_ck));
    HMX_HIP(H.s_list.upload(s_list));
    HMX_HIP(H.s_fidx.alloc(std::max<size_t>(s_fidx_n, 1)));
    if (s_fidx_n)
        HMX_HIP(hipMemcpy(H.s_fidx.d, s_fidx.get(), s_fidx_n * sizeof(int32_t), hipMemcpyHostToDevice));
    HMX_HIP(H.SW.alloc(s_total + 1));
    H.s_slots = s_total;
    H.SW16.release();
    if (H.s64_nint > 0) {
        HMX_HIP(H.s64_sub_ptr.upload(p64));
        HMX_HIP(H.s64_sub_task.upload(t64));
        HMX_HIP(H.s64_sub_row0.upload(r64));
        HMX_HIP(H.s64_sub_nrows.upload(n64));
        HMX_HIP(H.s64_sub_dst.upload(d64));
        HMX_HIP(H.s64_int_order.upload(o64));
    }
    return HMX_OK;
}

static int build_streams(HMat &H) {
    Timer tim;
    const bool phase_timing = H.opt.i(HMX_OPT_BUILD_TIMING) != 0;
    double phase_last       = 0;
    auto phase_nosync       = [&](const char *name) { // host phases that run while the pack kernels are in flight
        if (!phase_timing)
            return;
        const double t = tim.s();
        fprintf(stderr, "[hmx build]   layout: %-20s %8.1f ms\n", name, 1e3 * (t - phase_last));
        phase_last = t;
    };
    auto phase = [&](const char *name) {
        if (phase_timing)
            (void)hipDeviceSynchronize();
        phase_nosync(name);
    };
    const int64_t nb_real = (int64_t)H.leaves.size();
    constexpr int TR_MAX = 64;
    const int SR_MAX     = std::max(64, H.opt.i(HMX_OPT_R_PIECE_ROWS));
    // Symmetric / Hermitian storage ('S' / 'H', 'L' / 'U'): the streams hold the STORED TRIANGLE and the product is fused -- each stored
    // coefficient of a dense leaf and of a U factor is read once, V factors twice (expand_sym_kernel, rowsym_kernel): half the HBM footprint and
    // 0.7 x the traffic of the expanded layout.  HMX_OPT_SYM_STORAGE = 1 lays every leaf of leaves_for_symmetry out ALSO as its (conjugate)
    // transpose (same crosses, roles of U and V exchanged; same dense generator): the full operator, one untransposed pass.
    const bool want_expanded = H.opt.i(HMX_OPT_SYM_STORAGE) == 1;
    const int herm           = H.symmetry_for_leaves == 'H' ? 1 : 0; // 'H': the mirrored leaf is the CONJUGATE transpose
    H.sym_expanded           = H.has_mirror && want_expanded;
    H.sym_fused              = H.has_mirror && !H.sym_expanded && !H.view_of;
    if (H.view_of && H.has_mirror)
        H.sym_expanded = true; // a transposed view is only ever built from an expanded layout
    // a transposed view borrows crosses, staged blocks and generator from its owner
    const HMat &SRC = H.view_of ? *H.view_of : H;
    const bool tv   = H.view_of != nullptr && H.view_transposed;
    H.chunk_plan_n = 0;
    if (!H.view_of) { // the layout changes: views built earlier are stale
        H.T_op.reset();
        H.X_op.reset();
        H.T_op_failed = H.X_op_failed = H.trans_tables_failed = false;
    }
    std::vector<hmx_leaf> XL = H.leaves;
    std::vector<int> XK      = H.kind;
    std::vector<int64_t> xcolptr = H.colptr, xstaged = H.staged_off;
    std::vector<int32_t> xswapped = H.swapped, xtransposed(nb_real, tv ? 1 : 0), xconj(nb_real, 0);
    xcolptr.resize(nb_real, 0);
    xstaged.resize(nb_real, -1);
    xswapped.resize(nb_real, 0);
    if (tv)
        for (auto &v : xswapped)
            v = v ? 0 : 1; // U and V exchange roles
    if (H.sym_expanded)
        for (int64_t b = 0; b < nb_real; b++) {
            if (!H.leaves[b].mirror)
                continue;
            hmx_leaf v = H.leaves[b];
            std::swap(v.t_offset, v.s_offset);
            std::swap(v.t_size, v.s_size);
            v.mirror = 0;
            XL.push_back(v);
            XK.push_back(H.kind[b]);
            xcolptr.push_back(xcolptr[b]);
            xstaged.push_back(xstaged[b]);
            xswapped.push_back(xswapped[b] ? 0 : 1);
            xtransposed.push_back(tv ? 0 : 1);
            xconj.push_back(herm); // Hermitian storage: the mirrored copy is the conjugate (transpose)
        }
    const int64_t nb = (int64_t)XL.size();
    // ---- ranges ---------------------------------------------------------------------------------
    // E ranges partition the local rows at every block boundary (each output row has exactly one owner).
    // R ranges are per DISTINCT source cluster of the low-rank leaves (cut into pieces of <= SR_MAX rows): a
    // block is reduced over ceil(n/SR_MAX) pieces of its own cluster instead of over every leaf cluster below
    // it, so blocks up to SR_MAX columns need no partial sums at all and the largest ones a few dozen.
    // row breakpoints: marks over the local rows, read back in order (no sort of 2 x leaves numbers); the distinct source clusters of the
    // low-rank leaves: sorted + deduplicated per slice of the leaf list on a few threads, then once more over the survivors
    std::vector<int> tbp;
    std::vector<std::pair<int, int>> sclusters;
    {
        std::vector<char> mark((size_t)H.nT + 1, 0);
        mark[0] = mark[H.nT] = 1;
        std::vector<int> outside;
        const size_t NS = std::min<size_t>({(size_t)16, (size_t)host_cores(), (size_t)nb / 32768 + 1});
        std::vector<std::vector<std::pair<int, int>>> sc(NS);
        std::vector<std::thread> th;
        auto slice = [&](size_t t) {
            auto &v = sc[t];
            for (int64_t b = nb * (int64_t)t / (int64_t)NS; b < nb * (int64_t)(t + 1) / (int64_t)NS; b++) {
                const hmx_leaf &l = XL[b];
                if (XK[b] == LK_LOWRANK && l.rank > 0)
                    v.emplace_back(l.s_offset, l.s_size);
            }
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
        };
        for (size_t t = 1; t < NS; t++)
            th.emplace_back(slice, t);
        for (int64_t b = 0; b < nb; b++) { // meanwhile, on this thread
            const hmx_leaf &l = XL[b];
            const int64_t lo = (int64_t)l.t_offset - H.T0, hi = lo + l.t_size;
            if (lo >= 0 && hi <= H.nT)
                mark[lo] = mark[hi] = 1;
            else {
                outside.push_back(l.t_offset);
                outside.push_back(l.t_offset + l.t_size);
            }
        }
        slice(0);
        for (auto &x : th)
            x.join();
        for (int64_t i = 0; i <= H.nT; i++)
            if (mark[i])
                tbp.push_back(H.T0 + (int)i);
        tbp.insert(tbp.end(), outside.begin(), outside.end());
        for (auto &v : sc)
            sclusters.insert(sclusters.end(), v.begin(), v.end());
    }
    phase("  copies, breakpoints");
    std::sort(sclusters.begin(), sclusters.end());
    sclusters.erase(std::unique(sclusters.begin(), sclusters.end()), sclusters.end());
    StreamSet &E = H.E, &R = H.R;
    make_ranges(tbp, TR_MAX, H.T0, E.off, E.len);
    phase("  source clusters, row ranges");
    R.off.clear();
    R.len.clear();
    std::vector<int32_t> scluster_first(sclusters.size() + 1, 0);
    // Pieces of a source cluster larger than SR_MAX: its descendants of at most SR_MAX rows in the source cluster tree (when the
    // tree is known and cuts reasonably: binary trees halve, so pieces are SR_MAX / 2 ... SR_MAX rows), otherwise steps of SR_MAX rows
    // from the cluster's start.  Along the tree the pieces of every cluster level nest inside the same windows (below).
    const bool want_tree_pieces = H.opt.i(HMX_OPT_R_TREE_PIECES) != 0;
    std::map<std::pair<int, int>, int> node_of;
    if (want_tree_pieces)
        for (size_t v = 0; v < H.tree_s.size(); v++)
            node_of[{H.tree_s[v].off, H.tree_s[v].size}] = (int)v; // same (offset, size) more than once (single-child root): the deepest
    auto cut_pieces = [&](bool along_tree) -> bool { // false: the tree cannot be used (then called again without it)
        R.off.clear();
        R.len.clear();
        int64_t n_cut = 0, n_pieces = 0, rows = 0;
        for (size_t c = 0; c < sclusters.size(); c++) {
            std::vector<int32_t> o, ln;
            if (along_tree && sclusters[c].second > SR_MAX) {
                auto it = node_of.find({sclusters[c].first, sclusters[c].second});
                if (it == node_of.end())
                    return false; // a source cluster that is no node of the tree: not the tree the blocks came from
                std::vector<int> stack{it->second};
                while (!stack.empty()) { // depth first, children in order: pieces come out by increasing offset
                    const HMat::TreeNode nd = H.tree_s[stack.back()];
                    stack.pop_back();
                    if (nd.size <= SR_MAX) {
                        o.push_back(nd.off - H.S0);
                        ln.push_back(nd.size);
                    } else if (nd.n_children == 0) { // a leaf cluster larger than SR_MAX (maximal_leaf_size > SR_MAX): steps of SR_MAX rows
                        std::vector<int> bp{nd.off, nd.off + nd.size};
                        std::vector<int32_t> o2, l2;
                        make_ranges(bp, SR_MAX, H.S0, o2, l2);
                        o.insert(o.end(), o2.begin(), o2.end());
                        ln.insert(ln.end(), l2.begin(), l2.end());
                    } else {
                        for (int k = nd.n_children - 1; k >= 0; k--)
                            stack.push_back(nd.first_child + k);
                    }
                }
                n_cut++;
                n_pieces += (int64_t)o.size();
                rows += sclusters[c].second;
            } else {
                std::vector<int> bp{sclusters[c].first, sclusters[c].first + sclusters[c].second};
                make_ranges(bp, SR_MAX, H.S0, o, ln);
            }
            scluster_first[c] = (int32_t)R.off.size();
            R.off.insert(R.off.end(), o.begin(), o.end());
            R.len.insert(R.len.end(), ln.begin(), ln.end());
        }
        // trees with many children per node cut into slivers: then the fixed steps are the better pieces (and there are no windows)
        return !(along_tree && n_cut > 0 && (double)rows / (double)n_pieces < 0.35 * SR_MAX);
    };
    bool tree_pieces = want_tree_pieces && !H.tree_s.empty() && cut_pieces(true);
    if (!tree_pieces)
        (void)cut_pieces(false);
    scluster_first[sclusters.size()] = (int32_t)R.off.size();
    phase("  pieces");
    // position -> range lookup
    std::vector<int32_t> t_pos2range(H.nT + 1, -1);
    for (int r = 0; r < E.nranges(); r++)
        t_pos2range[E.off[r]] = r;
    auto range_span = [](const std::vector<int32_t> &pos2range, const StreamSet &S, int lo, int hi, int &ra, int &rb) {
        ra = pos2range[lo];
        rb = ra;
        while (rb < S.nranges() && S.off[rb] < hi)
            rb++;
    };
    // ---- columns per range, pair lists, a / partial offsets ----------------------------------------
    E.cols.assign(E.nranges(), 0);
    E.cw.assign(E.nranges(), 0);
    R.cols.assign(R.nranges(), 0);
    std::vector<int32_t> elr_b, elr_r, elr_c, ed_b, ed_r, ed_c, rlr_b, rlr_r, rlr_c;
    std::vector<int64_t> aoff(nb, -1), poff(nb, -1);
    std::vector<int32_t> ns_of(nb, 0), s_first(nb, 0);
    int64_t A_total = 0, P_total = 0;
    H.stats = hmx_stats{};
    H.stats.rank_min = 1 << 30;
    double rank_sum  = 0;
    // columns are given out leaf by leaf in the leaves' own order
    // Two passes over the leaves (in their own order), each split over a few threads: pass 1 counts, per thread and per range, the
    // columns its leaves add (and the pairs, ranks and partial slots); a prefix over the threads turns the counts into each thread's
    // starting column per range and starting position in the pair lists; pass 2 writes the pairs.  The result is what the one-thread
    // loop gives (columns in leaf order, pair lists leaf-major) -- 58 ms of a 320 ms build at N = 1e6 before.
    {
        const int nre = E.nranges(), nrr = R.nranges();
        const size_t NT = H.opt.i(HMX_OPT_LAYOUT_THREADS) > 0 ? (size_t)H.opt.i(HMX_OPT_LAYOUT_THREADS) : std::min<size_t>({(size_t)16, (size_t)host_cores(), (size_t)nb / 16384 + 1});
        struct Part {
            std::vector<int32_t> ecnt, rcnt; // columns this part adds to every E range / R piece
            int64_t n_elr = 0, n_ed = 0, n_rlr = 0, a = 0, p = 0;
            int64_t n_lowrank = 0, n_dense = 0, cgen_lr = 0, cgen_d = 0;
            int rank_min = 1 << 30, rank_max = 0;
            double rank_sum = 0;
        };
        std::vector<Part> part(NT);
        auto leaf_spans = [&](int64_t b, bool &skip, bool &lr, int &ncols, int &ra, int &rb, int &sa, int &sb) {
            const hmx_leaf &l = XL[b];
            lr   = XK[b] == LK_LOWRANK;
            skip = lr && l.rank <= 0; // rank-0 low-rank block: contributes nothing (add_lrmat_vector_product.hpp:11)
            if (skip)
                return;
            ncols = lr ? l.rank : l.s_size;
            range_span(t_pos2range, E, l.t_offset - H.T0, l.t_offset - H.T0 + l.t_size, ra, rb);
            sa = sb = 0;
            if (lr) {
                const size_t sc = std::lower_bound(sclusters.begin(), sclusters.end(), std::make_pair((int)l.s_offset, (int)l.s_size)) - sclusters.begin();
                sa = scluster_first[sc], sb = scluster_first[sc + 1];
            }
        };
        auto run_parts = [&](auto &&fn) {
            if (NT == 1) {
                fn((size_t)0);
                return;
            }
            std::vector<std::thread> th;
            for (size_t t = 0; t < NT; t++)
                th.emplace_back([&, t] { fn(t); });
            for (auto &x : th)
                x.join();
        };
        run_parts([&](size_t t) {
            Part &P = part[t];
            P.ecnt.assign(nre, 0);
            P.rcnt.assign(nrr, 0);
            for (int64_t ib = nb * (int64_t)t / (int64_t)NT; ib < nb * (int64_t)(t + 1) / (int64_t)NT; ib++) {
                const int64_t b = ib;
                bool skip, lr;
                int ncols, ra, rb, sa, sb;
                leaf_spans(b, skip, lr, ncols, ra, rb, sa, sb);
                if (skip)
                    continue;
                const hmx_leaf &l = XL[b];
                for (int r = ra; r < rb; r++)
                    P.ecnt[r] += ncols;
                (lr ? P.n_elr : P.n_ed) += rb - ra;
                if (lr) {
                    P.a += l.rank;
                    if (sb - sa > 1)
                        P.p += (int64_t)(sb - sa) * l.rank;
                    for (int r = sa; r < sb; r++)
                        P.rcnt[r] += l.rank;
                    P.n_rlr += sb - sa;
                    if (b < nb_real) { // statistics describe the stored leaves (htool's definitions), not the mirrored copies
                        P.n_lowrank++;
                        P.cgen_lr += (int64_t)l.rank * (l.t_size + l.s_size);
                        P.rank_min = std::min(P.rank_min, (int)l.rank);
                        P.rank_max = std::max(P.rank_max, (int)l.rank);
                        P.rank_sum += l.rank;
                    }
                } else if (b < nb_real) {
                    P.n_dense++;
                    P.cgen_d += (int64_t)l.t_size * l.s_size;
                }
            }
        });
        phase("  pair pass 1");
        // exclusive prefix over the parts, per range: ecnt / rcnt become each part's first column
        parallel_for((size_t)nre, [&](size_t lo, size_t hi) {
            for (size_t r = lo; r < hi; r++) {
                int32_t run = 0;
                for (size_t t = 0; t < NT; t++) {
                    const int32_t c = part[t].ecnt[r];
                    part[t].ecnt[r] = run;
                    run += c;
                }
                E.cols[r] = run;
            }
        });
        parallel_for((size_t)nrr, [&](size_t lo, size_t hi) {
            for (size_t r = lo; r < hi; r++) {
                int32_t run = 0;
                for (size_t t = 0; t < NT; t++) {
                    const int32_t c = part[t].rcnt[r];
                    part[t].rcnt[r] = run;
                    run += c;
                }
                R.cols[r] = run;
            }
        });
        std::vector<int64_t> o_elr(NT + 1, 0), o_ed(NT + 1, 0), o_rlr(NT + 1, 0), o_a(NT + 1, 0), o_p(NT + 1, 0);
        for (size_t t = 0; t < NT; t++) {
            o_elr[t + 1] = o_elr[t] + part[t].n_elr;
            o_ed[t + 1]  = o_ed[t] + part[t].n_ed;
            o_rlr[t + 1] = o_rlr[t] + part[t].n_rlr;
            o_a[t + 1]   = o_a[t] + part[t].a;
            o_p[t + 1]   = o_p[t] + part[t].p;
            H.stats.n_lowrank += part[t].n_lowrank;
            H.stats.n_dense += part[t].n_dense;
            H.stats.cgen_lowrank += part[t].cgen_lr;
            H.stats.cgen_dense += part[t].cgen_d;
            H.stats.rank_min = std::min(H.stats.rank_min, part[t].rank_min);
            H.stats.rank_max = std::max(H.stats.rank_max, part[t].rank_max);
            rank_sum += part[t].rank_sum;
        }
        A_total = o_a[NT];
        P_total = o_p[NT];
        elr_b.resize(o_elr[NT]), elr_r.resize(o_elr[NT]), elr_c.resize(o_elr[NT]);
        ed_b.resize(o_ed[NT]), ed_r.resize(o_ed[NT]), ed_c.resize(o_ed[NT]);
        rlr_b.resize(o_rlr[NT]), rlr_r.resize(o_rlr[NT]), rlr_c.resize(o_rlr[NT]);
        phase("  prefix, resize");
        run_parts([&](size_t t) {
            Part &P = part[t];
            int64_t q_elr = o_elr[t], q_ed = o_ed[t], q_rlr = o_rlr[t], a_run = o_a[t], p_run = o_p[t];
            for (int64_t ib = nb * (int64_t)t / (int64_t)NT; ib < nb * (int64_t)(t + 1) / (int64_t)NT; ib++) {
                const int64_t b = ib;
                bool skip, lr;
                int ncols, ra, rb, sa, sb;
                leaf_spans(b, skip, lr, ncols, ra, rb, sa, sb);
                if (skip)
                    continue;
                const hmx_leaf &l = XL[b];
                for (int r = ra; r < rb; r++) {
                    int64_t &q = lr ? q_elr : q_ed;
                    (lr ? elr_b : ed_b)[q] = (int32_t)b;
                    (lr ? elr_r : ed_r)[q] = r;
                    (lr ? elr_c : ed_c)[q] = P.ecnt[r];
                    q++;
                    P.ecnt[r] += ncols;
                }
                if (lr) {
                    aoff[b] = a_run;
                    a_run += l.rank;
                    ns_of[b]   = sb - sa;
                    s_first[b] = sa;
                    if (sb - sa > 1) {
                        poff[b] = p_run;
                        p_run += (int64_t)(sb - sa) * l.rank;
                    }
                    for (int r = sa; r < sb; r++) {
                        rlr_b[q_rlr] = (int32_t)b;
                        rlr_r[q_rlr] = r;
                        rlr_c[q_rlr] = P.rcnt[r];
                        q_rlr++;
                        P.rcnt[r] += l.rank;
                    }
                }
            }
        });
    }
    if (H.stats.n_lowrank == 0)
        H.stats.rank_min = 0;
    H.stats.rank_mean = H.stats.n_lowrank ? rank_sum / H.stats.n_lowrank : 0;
    H.A_total = A_total;
    H.P_total = P_total;
    phase("ranges, pair lists");
    // ---- bases ------------------------------------------------------------------------------------
    E.base.assign(E.nranges(), 0);
    E.colbase.assign(E.nranges(), 0);
    E.elems = E.total_cols = 0;
    for (int r = 0; r < E.nranges(); r++) {
        E.base[r]    = E.elems;
        E.colbase[r] = E.total_cols;
        E.elems += (int64_t)E.len[r] * E.cols[r];
        E.total_cols += E.cols[r];
    }
    E.task_range.resize(E.nranges());
    std::iota(E.task_range.begin(), E.task_range.end(), 0);
    // launch order: heaviest first (shorter tail); HMX_SORT_TASKS=2: heaviest first only across power-of-two weight classes,
    // address order inside a class (neighbouring workgroups stream neighbouring memory); 3: heaviest UNIT first, a unit = the tasks of
    // `xcd_unit_rows` consecutive rows (they gather the same operand rows), kept on one XCD one after the other (xcd_deal)
    const int sort_mode = H.opt.i(HMX_OPT_TASK_ORDER);
    const int unit_rows = std::max(1, H.opt.i(HMX_OPT_XCD_UNIT_ROWS));
    auto weight_class = [](int64_t w) { int c = 0; while (w > 1) { w >>= 1; c++; } return c; };
    if (sort_mode == 3) {
        std::vector<int64_t> unit(E.nranges()), wk(E.nranges());
        for (int r = 0; r < E.nranges(); r++) {
            unit[r] = E.off[r] / unit_rows;
            wk[r]   = (int64_t)E.len[r] * E.cols[r];
        }
        E.task_range = xcd_deal(unit, wk, 1);
    }
    if (sort_mode == 1 || sort_mode == 2) {
        std::vector<int64_t> wk(E.nranges());
        for (int r = 0; r < E.nranges(); r++)
            wk[r] = sort_mode == 2 ? (int64_t)weight_class((int64_t)E.len[r] * E.cols[r]) : (int64_t)E.len[r] * E.cols[r];
        std::stable_sort(E.task_range.begin(), E.task_range.end(), [&](int a, int b) { return wk[a] > wk[b]; });
    }
    E.task_chunk.clear();
    R.base.assign(R.nranges(), 0);
    R.colbase.assign(R.nranges(), 0);
    R.elems = R.total_cols = 0;
    R.task_range.clear();
    R.task_chunk.clear();
    R.cw.assign(R.nranges(), 2);
    for (int r = 0; r < R.nranges(); r++) {
        R.base[r]    = R.elems;
        R.colbase[r] = R.total_cols;
        const int C = R.cols[r], nch = (C + 127) / 128;
        if (C > 0) { // balanced chunks: nch chunks of width cw (even), the last one takes what is left
            const int cw = hmx_wp((C + nch - 1) / nch);
            R.cw[r]      = cw;
            const int wlast = C - (nch - 1) * cw;
            R.elems += (int64_t)R.len[r] * ((int64_t)(nch - 1) * cw + hmx_wp(wlast));
        }
        R.total_cols += C;
        for (int c = 0; c < nch; c++) {
            R.task_range.push_back(r);
            R.task_chunk.push_back(c);
        }
    }
    if (phase_timing) { // where the R-stream's coefficients sit, by chunk width (narrow chunks: few coefficients per row of the chunk)
        int64_t by_width[5] = {0, 0, 0, 0, 0}, tasks[5] = {0, 0, 0, 0, 0};
        for (size_t t = 0; t < R.task_range.size(); t++) {
            const int r = R.task_range[t], w = std::min<int>(R.cols[r] - R.task_chunk[t] * R.cw[r], R.cw[r]);
            const int k = w <= 8 ? 0 : (w <= 16 ? 1 : (w <= 32 ? 2 : (w <= 64 ? 3 : 4)));
            by_width[k] += (int64_t)R.len[r] * w;
            tasks[k]++;
        }
        fprintf(stderr, "[hmx build]   R-stream coefficients by chunk width <= 8 / 16 / 32 / 64 / 128: %.1f / %.1f / %.1f / %.1f / %.1f %% (%lld / %lld / %lld / %lld / %lld tasks)\n",
                100.0 * by_width[0] / std::max<int64_t>(R.elems, 1), 100.0 * by_width[1] / std::max<int64_t>(R.elems, 1), 100.0 * by_width[2] / std::max<int64_t>(R.elems, 1),
                100.0 * by_width[3] / std::max<int64_t>(R.elems, 1), 100.0 * by_width[4] / std::max<int64_t>(R.elems, 1), (long long)tasks[0], (long long)tasks[1], (long long)tasks[2],
                (long long)tasks[3], (long long)tasks[4]);
    }
    if (sort_mode) { // longest tasks first: shorter kernel tail (-8 % on reduce_kernel)
        std::vector<int> ord(R.task_range.size());
        std::iota(ord.begin(), ord.end(), 0);
        auto work = [&](int t) {
            const int r = R.task_range[t], c = R.task_chunk[t];
            int w = R.cols[r] - c * R.cw[r];
            w     = std::min(w, (int)R.cw[r]);
            return (int64_t)R.len[r] * w;
        };
        std::vector<int64_t> wk(ord.size()); // the key once per task, not once per comparison
        for (size_t t = 0; t < ord.size(); t++)
            wk[t] = sort_mode == 2 ? (int64_t)weight_class(work((int)t)) : work((int)t);
        if (sort_mode == 3) { // unit = the pieces (of every level of the source tree) over the same `xcd_unit_rows` rows of x
            std::vector<int64_t> unit(ord.size());
            for (size_t t = 0; t < ord.size(); t++)
                unit[t] = R.off[R.task_range[t]] / unit_rows;
            const std::vector<int32_t> o = xcd_deal(unit, wk, 1);
            ord.assign(o.begin(), o.end());
        } else
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return wk[a] > wk[b]; });
        std::vector<int32_t> tr(ord.size()), tc(ord.size());
        for (size_t k = 0; k < ord.size(); k++) {
            tr[k] = R.task_range[ord[k]];
            tc[k] = R.task_chunk[ord[k]];
        }
        R.task_range.swap(tr);
        R.task_chunk.swap(tc);
    }
    if (E.total_cols >= (int64_t(1) << 31) || R.total_cols >= (int64_t(1) << 31) || (int64_t)H.nS + A_total + P_total + 2 >= (int64_t(1) << 31)) {
        set_error("operator too large for 32-bit column indices");
        return HMX_ERR_UNSUPPORTED;
    }
    // ---- index arrays -------------------------------------------------------------------------------
    phase("bases, task order");
    // ---- upload metadata, allocate streams ------------------------------------------------------------
    // Order since round 3: what the pack kernels need goes first, the pack kernels are launched, and the HOST work they do not depend on
    // (the index arrays of the product kernels, the slots of the fused symmetric product) runs while they fill the streams.
    if (!H.sym_fused) {
        for (auto *a : {&H.s_mdst, &H.s_coef, &H.s_count, &H.sc_dst, &H.sc_lp, &H.sc_count, &H.sc_k, &H.s_list, &H.s_fidx})
            a->release();
        for (auto *a : {&H.s_sub_task, &H.s_sub_row0, &H.s_sub_nrows, &H.s_sub_dst, &H.s_int_order})
            a->release();
        H.s_sub_ptr.release();
        H.SW.release();
    }
    HMX_HIP(E.upload_meta());
    HMX_HIP(R.upload_meta());
    HMX_HIP(E.stream.alloc(std::max<int64_t>(E.elems, 1)));
    HMX_HIP(R.stream.alloc(std::max<int64_t>(R.elems, 1)));
    HMX_HIP(R.stream.zero()); // padded odd-width chunks keep a zero column
    // ---- pack ---------------------------------------------------------------------------------------------
    std::vector<int32_t> ranks(nb), symu(nb, 0);
    for (int64_t b = 0; b < nb; b++) {
        ranks[b] = XL[b].rank;
        if (XL[b].symmetric && XK[b] == LK_DENSE_STAGED && !SRC.dense_stage.d) // uploaded symmetric leaf: one triangle is valid
            symu[b] = H.uplo_for_leaves == 'L' ? 1 : (H.uplo_for_leaves == 'U' ? 2 : 0);
    }
    HMX_HIP(H.d_rank.upload(ranks));
    HMX_HIP(H.d_sym_uplo.upload(symu));
    {
        std::vector<int32_t> a(nb), bb(nb), c(nb), d(nb);
        for (int64_t i = 0; i < nb; i++) {
            a[i]  = XL[i].t_offset;
            bb[i] = XL[i].t_size;
            c[i]  = XL[i].s_offset;
            d[i]  = XL[i].s_size;
        }
        HMX_HIP(H.d_t_off.upload(a));
        HMX_HIP(H.d_t_size.upload(bb));
        HMX_HIP(H.d_s_off.upload(c));
        HMX_HIP(H.d_s_size.upload(d));
        HMX_HIP(H.d_colptr.upload(xcolptr));
        HMX_HIP(H.d_swapped.upload(xswapped));
        HMX_HIP(H.d_staged_off.upload(xstaged));
        HMX_HIP(H.d_transposed.upload(xtransposed));
        HMX_HIP(H.d_conj.upload(xconj));
    }
    phase("uploads, allocations");
    DEvent e0, e1;
    DArr<int32_t> pk[9]; // pair lists of the three launches: all uploaded BEFORE the first launch (a blocking copy waits for the kernels
                         // already queued on its stream), alive until the kernels are done
    if (!elr_b.empty()) {
        HMX_HIP(pk[0].upload(elr_b));
        HMX_HIP(pk[1].upload(elr_r));
        HMX_HIP(pk[2].upload(elr_c));
    }
    if (!rlr_b.empty()) {
        HMX_HIP(pk[3].upload(rlr_b));
        HMX_HIP(pk[4].upload(rlr_r));
        HMX_HIP(pk[5].upload(rlr_c));
    }
    if (!ed_b.empty()) {
        HMX_HIP(pk[6].upload(ed_b));
        HMX_HIP(pk[7].upload(ed_r));
        HMX_HIP(pk[8].upload(ed_c));
    }
    HMX_HIP(hipEventRecord(e0, 0));
    if (!elr_b.empty()) {
        PackLrArgs P{SRC.pool.d, SRC.d_cross_off.d, H.d_colptr.d, H.d_rank.d, H.d_swapped.d, H.d_t_off.d, H.d_t_size.d, H.d_s_off.d, H.d_s_size.d,
                     pk[0].d, pk[1].d, pk[2].d, E.d_off.d, E.d_len.d, E.d_base.d, E.d_cols.d, E.d_cw.d, E.stream.d, H.T0, H.d_conj.d};
        hipLaunchKernelGGL(pack_lr_expand_kernel, dim3((unsigned)elr_b.size()), dim3(256), 0, 0, P, (int64_t)elr_b.size());
        HMX_HIP(hipGetLastError());
    }
    if (!rlr_b.empty()) {
        PackLrArgs P{SRC.pool.d, SRC.d_cross_off.d, H.d_colptr.d, H.d_rank.d, H.d_swapped.d, H.d_t_off.d, H.d_t_size.d, H.d_s_off.d, H.d_s_size.d,
                     pk[3].d, pk[4].d, pk[5].d, R.d_off.d, R.d_len.d, R.d_base.d, R.d_cols.d, R.d_cw.d, R.stream.d, H.S0, H.d_conj.d};
        hipLaunchKernelGGL(pack_lr_reduce_kernel, dim3((unsigned)rlr_b.size()), dim3(256), 0, 0, P, (int64_t)rlr_b.size());
        HMX_HIP(hipGetLastError());
    }
    if (!ed_b.empty()) {
        // row / column coordinates of THIS layout: a transposed view's rows are the owner's source points
        const DArr<double> &rx = tv ? SRC.sx : SRC.tx, &ry = tv ? SRC.sy : SRC.ty, &rz = tv ? SRC.sz : SRC.tz;
        const DArr<double> &cx = tv ? SRC.tx : SRC.sx, &cy = tv ? SRC.ty : SRC.sy, &cz = tv ? SRC.tz : SRC.sz;
        PackDenseArgs P{SRC.ks, rx.d, ry.d, rz.d, cx.d, cy.d, cz.d, pk[6].d, pk[7].d, pk[8].d, E.d_off.d, E.d_len.d, E.d_base.d,
                        H.d_t_off.d, H.d_t_size.d, H.d_s_off.d, H.d_s_size.d, H.d_staged_off.d, H.d_sym_uplo.d, H.d_transposed.d, H.d_conj.d,
                        SRC.dense_stage.d ? SRC.dense_stage.d : SRC.pool.d, E.stream.d, H.T0, herm};
        hipLaunchKernelGGL(pack_dense_kernel, dim3((unsigned)ed_b.size()), dim3(256), 0, 0, P, (int64_t)ed_b.size());
        HMX_HIP(hipGetLastError());
    }
    HMX_HIP(hipEventRecord(e1, 0));
    // The pack kernels now fill the streams while the host builds the index arrays.  Whatever makes this function return before they
    // are waited for (a failed upload, an operator the fused symmetric layout cannot hold) must not leave them writing into arrays the
    // caller is about to release, nor an operator that looks built: every early exit waits for the device and marks H unbuilt.
    struct PackGuard {
        HMat &H;
        bool armed = true;
        ~PackGuard() {
            if (armed) {
                (void)hipDeviceSynchronize();
                (void)hipGetLastError();
                H.finalized = false;
            }
        }
    } pack_guard{H};
    phase_nosync("pack kernels launched");
    const int64_t zA = H.nS, zP = H.nS + A_total;
    H.zero_slot      = H.nS + A_total + P_total;
    H.h_e_zidx.resize(E.total_cols); // every column belongs to exactly one (leaf, range) pair: written completely below
    auto fill_e = [&](const std::vector<int32_t> &pb, const std::vector<int32_t> &pr, const std::vector<int32_t> &pc, bool lr) {
        parallel_for(pb.size(), [&](size_t lo, size_t hi) { // every (leaf, range) pair owns its own columns
            for (size_t p = lo; p < hi; p++) {
                const int b = pb[p], r = pr[p];
                const hmx_leaf &l = XL[b];
                const int ncols   = lr ? l.rank : l.s_size;
                const int64_t z0  = lr ? zA + aoff[b] : (int64_t)(l.s_offset - H.S0);
                int32_t *dst      = H.h_e_zidx.data() + E.colbase[r] + pc[p];
                for (int j = 0; j < ncols; j++)
                    dst[j] = (int32_t)(z0 + j);
            }
        });
    };
    fill_e(elr_b, elr_r, elr_c, true);
    fill_e(ed_b, ed_r, ed_c, false);
    { // where the dense leaves' slices sit in the E-streams (bulk download: api_get_blocks); leaf-major, the stored leaves only
        H.dp_leaf.clear();
        H.dp_range.clear();
        H.dp_col.clear();
        for (size_t q = 0; q < ed_b.size(); q++)
            if (ed_b[q] < nb_real) {
                H.dp_leaf.push_back(ed_b[q]);
                H.dp_range.push_back(ed_r[q]);
                H.dp_col.push_back(ed_c[q]);
            }
    }
    phase_nosync("  e index");
    hvec32 h_outidx(R.total_cols);
    parallel_for(rlr_b.size(), [&](size_t lo, size_t hi) {
        for (size_t p = lo; p < hi; p++) {
            const int b = rlr_b[p], r = rlr_r[p];
            const hmx_leaf &l = XL[b];
            const int64_t cb  = R.colbase[r] + rlr_c[p];
            for (int k = 0; k < l.rank; k++) {
                h_outidx[cb + k]   = ns_of[b] == 1 ? (int32_t)(zA + aoff[b] + k) : (int32_t)(zP + poff[b] + (int64_t)(r - s_first[b]) * l.rank + k);
            }
        }
    });
    phase_nosync("  r index");
    std::vector<int32_t> cd, cs, cst, cc;
    for (int64_t b = 0; b < nb; b++)
        if (poff[b] >= 0)
            for (int k = 0; k < XL[b].rank; k++) {
                cd.push_back((int32_t)(zA + aoff[b] + k));
                cs.push_back((int32_t)(zP + poff[b] + k));
                cst.push_back(XL[b].rank);
                cc.push_back(ns_of[b]);
            }
    H.n_combine = (int)cd.size();
    // ---- fused symmetric product: slots of the mirrored partial results (build_mirror_tables) -------------------------------------------
    H.n_sym_combine = 0;
    H.s_kmax        = 0;
    H.trans_fused   = false; // the tables of the stored-data transposed product belonged to the layout that is being replaced
    if (H.sym_fused) {
        MirrorCtx M{XL, XK, nb, elr_b, elr_r, elr_c, ed_b, ed_r, ed_c, rlr_b, rlr_r, rlr_c, aoff, A_total, false, [&](const char *n) { phase_nosync(n); }};
        const int rcm = build_mirror_tables(H, M);
        if (rcm != HMX_OK)
            return rcm;
    }

    phase_nosync("index arrays");
    // ---- uploads of the index arrays (the first one waits for the pack kernels: same stream) ----------------------------------------
    HMX_HIP(H.e_zidx.upload(H.h_e_zidx));
    HMX_HIP(H.r_outidx.upload(h_outidx));
    HMX_HIP(H.c_dst.upload(cd));
    HMX_HIP(H.c_src.upload(cs));
    HMX_HIP(H.c_stride.upload(cst));
    HMX_HIP(H.c_count.upload(cc));
    HMX_HIP(H.Z.alloc(H.zero_slot + 1));
    HMX_HIP(H.Z.zero());
    HMX_HIP(hipEventSynchronize(e1));
    HMX_HIP(hipDeviceSynchronize()); // an error of the pack kernels surfaces here
    pack_guard.armed = false;
    float ms         = 0;
    HMX_HIP(hipEventElapsedTime(&ms, e0, e1));
    for (auto &a : pk)
        a.release();
    phase("uploads of the index arrays");
    H.stats.t_pack_s     = tim.s();
    H.stats.t_assemble_s = ms * 1e-3;
    H.stats.stream_bytes = (E.elems + R.elems) * (int64_t)sizeof(scalar);
    H.stats.expand_coeffs = E.elems;
    H.stats.a_total       = A_total;
    H.stats.reduce_coeffs = 0;
    for (int64_t b = 0; b < nb; b++)
        if (XK[b] == LK_LOWRANK && XL[b].rank > 0)
            H.stats.reduce_coeffs += (int64_t)XL[b].rank * XL[b].s_size;
    H.finalized          = true;
    return HMX_OK;
}

static void prof_mark(HMat &H, hipStream_t st, const char *name) {
    if (!H.profiling)
        return;
    hipEvent_t e;
    if (H.ev.size() <= H.ev_names.size()) {
        (void)hipEventCreate(&e);
        H.ev.push_back(e);
    }
    e = H.ev[H.ev_names.size()];
    (void)hipEventRecord(e, st);
    H.ev_names.push_back(name);
}

// forward pass on device pointers: y = alpha * (sum over leaves) x + beta * y using the fast kernels
// zidx: coefficient index array of the E-streams (all leaves, or mirror leaves only)
static int ensure_expand_chunks(HMat &H, int nchunks) {
    const StreamSet &E = H.E;
    const int nr       = E.nranges();
    nchunks            = std::max(1, std::min(nchunks, std::max(nr, 1)));
    if (H.chunk_plan_n == nchunks && H.d_chunk_order.d)
        return HMX_OK;
    auto work = [&](int r) { return (double)E.len[r] * E.cols[r] + 64; }; // + a constant: an empty range still costs a workgroup
    double total = 0;
    for (int r = 0; r < nr; r++)
        total += work(r);
    H.chunk_first.assign(nchunks, 0);
    H.chunk_count.assign(nchunks, 0);
    H.chunk_row_lo.assign(nchunks, 0);
    H.chunk_row_hi.assign(nchunks, 0);
    std::vector<int32_t> order(std::max(nr, 1), 0);
    int r = 0;
    double acc = 0;
    for (int c = 0; c < nchunks; c++) { // the ranges are in row order: chunk c takes them up to the (c + 1)-th share of the work
        const int first   = r;
        const double upto = total * (c + 1) / nchunks;
        if (c == nchunks - 1)
            r = nr;
        else
            while (r < nr && nr - r > nchunks - 1 - c && (r == first || acc + 0.5 * work(r) <= upto)) {
                acc += work(r);
                r++;
            }
        H.chunk_first[c] = first;
        H.chunk_count[c] = r - first;
        for (int k = first; k < r; k++)
            order[k] = k;
        std::stable_sort(order.begin() + first, order.begin() + r, [&](int a, int b) { return (int64_t)E.len[a] * E.cols[a] > (int64_t)E.len[b] * E.cols[b]; });
    }
    for (int c = 0; c < nchunks; c++) { // the ranges partition the local rows: chunk c owns the rows from its first range to the next chunk's
        H.chunk_row_lo[c] = c == 0 ? 0 : (H.chunk_first[c] < nr ? E.off[H.chunk_first[c]] : H.nT);
        if (c > 0)
            H.chunk_row_hi[c - 1] = H.chunk_row_lo[c];
    }
    H.chunk_row_hi[nchunks - 1] = H.nT;
    HMX_HIP(H.d_chunk_order.upload(order));
    H.chunk_plan_n = nchunks;
    return HMX_OK;
}

static int run_forward(HMat &H, const int32_t *zidx, const scalar *x_src, scalar alpha, scalar beta, scalar *y, hipStream_t st, bool sym_fused = false,
                       int nchunks = 0, after_chunk_fn after_chunk = nullptr, void *after_user = nullptr) {
    const scalar *xin = x_src; // both stages read the caller's vector directly, nothing is copied into Z's x region
    const int nx      = H.nS;
    const int RW      = H.opt.i(HMX_OPT_REDUCE_WAVES) == 4 ? 4 : 1; // 1 (default: one wave per workgroup frees its slot as soon as its task ends) or 4
    // expand: 4 waves per row range; when there are too few ranges to fill the chip more than once (<= 4096: the per-rank share
    // of an 8-GPU run, or N ~ 1e5) 8 waves per range shorten the tail of the heavy ranges (-5 %), at full size they cost 2 %
    const int EW = H.opt.i(HMX_OPT_EXPAND_WAVES) ? H.opt.i(HMX_OPT_EXPAND_WAVES) : (H.E.nranges() <= 4096 ? 8 : 4);
    const int ntasks = (int)H.R.task_range.size();
    if (ntasks > 0) {
        ReduceArgs A{H.R.stream.d, H.R.d_task_range.d, H.R.d_task_chunk.d, H.R.d_off.d, H.R.d_len.d, H.R.d_cols.d, H.R.d_cw.d, H.R.d_base.d, H.R.d_colbase.d,
                     H.r_outidx.d, xin, H.Z.d, ntasks};
        switch (RW) {
        case 4: hipLaunchKernelGGL(reduce_kernel<4>, dim3((ntasks + 3) / 4), dim3(256), 0, st, A); break;
        default: hipLaunchKernelGGL(reduce_kernel<1>, dim3(ntasks), dim3(64), 0, st, A); break;
        }
        prof_mark(H, st, "reduce_kernel");
    }
    if (H.n_combine > 0) {
        CombineArgs C{H.c_dst.d, H.c_src.d, H.c_stride.d, H.c_count.d, H.Z.d, H.n_combine};
        hipLaunchKernelGGL(combine_kernel, dim3((H.n_combine + 255) / 256), dim3(256), 0, st, C);
        prof_mark(H, st, "combine_kernel");
    }
    if (sym_fused) {
        // compact symmetric storage: forward product and mirrored column sums in one sweep over the E-streams, then a' is
        // folded, the R-streams are swept a second time (y_s += V^T a') and the output levels are added in their fixed order
        if (H.E.nranges() > 0) {
            ExpandSymArgs X{{H.E.stream.d, H.E.d_task_range.d, H.E.d_off.d, H.E.d_len.d, H.E.d_cols.d, H.E.d_base.d, H.E.d_colbase.d, zidx, H.Z.d, y, alpha, beta, H.E.nranges(), xin, nx},
                            H.s_mdst.d, H.SW.d, x_src + (H.T0 - H.S0), H.symmetry_for_leaves == 'H' ? 1 : 0};
            const size_t lds = 0;
            switch (EW) {
            case 8: hipLaunchKernelGGL(expand_sym_kernel<8>, dim3(H.E.nranges()), dim3(512), lds, st, X); break;
            default: hipLaunchKernelGGL(expand_sym_kernel<4>, dim3(H.E.nranges()), dim3(256), lds, st, X); break;
            }
            prof_mark(H, st, "expand_sym_kernel");
        }
        if (H.n_sym_combine > 0) {
            const int nw = H.n_sym_combine_wave, nt = H.n_sym_combine - nw;
            if (nw > 0) {
                CombineListArgs C{H.sc_dst.d, H.sc_lp.d, H.sc_count.d, H.sc_k.d, H.s_list.d, H.SW.d, nw};
                hipLaunchKernelGGL(combine_list_wave_kernel, dim3((nw + 3) / 4), dim3(256), 0, st, C);
            }
            if (nt > 0) {
                CombineListArgs C{H.sc_dst.d + nw, H.sc_lp.d + nw, H.sc_count.d + nw, H.sc_k.d + nw, H.s_list.d, H.SW.d, nt};
                hipLaunchKernelGGL(combine_list_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, C);
            }
            prof_mark(H, st, "combine_sym_kernel");
        }
        if (H.s_nint > 0) {
            RowSymArgs A{H.R.stream.d, H.R.d_task_range.d, H.R.d_task_chunk.d, H.R.d_len.d, H.R.d_cols.d, H.R.d_cw.d, H.R.d_base.d, H.R.d_colbase.d, H.s_coef.d,
                         H.s_int_order.d, H.s_sub_ptr.d, H.s_sub_task.d, H.s_sub_row0.d, H.s_sub_nrows.d, H.s_sub_dst.d,
                         H.SW.d, H.s_fidx.d, H.s_count.d, y, alpha, H.nT, H.symmetry_for_leaves == 'H' ? 1 : 0, scalar(0), 1};
            hipLaunchKernelGGL(rowsym_kernel<SYM_WAVES>, dim3(H.s_nint), dim3(SYM_WAVES * 64), 0, st, A);
            prof_mark(H, st, "rowsym_kernel");
        }
    } else if (H.E.nranges() > 0 && nchunks > 1) {
        // the same kernel over contiguous groups of row ranges: after group c its rows of y are final and `after_chunk` may start
        // sending them while group c + 1 computes
        const int rc = ensure_expand_chunks(H, nchunks);
        if (rc != HMX_OK)
            return rc;
        for (int c = 0; c < H.chunk_plan_n; c++) {
            const int cnt = H.chunk_count[c];
            if (cnt > 0) {
                ExpandArgs X{H.E.stream.d, H.d_chunk_order.d + H.chunk_first[c], H.E.d_off.d, H.E.d_len.d, H.E.d_cols.d, H.E.d_base.d, H.E.d_colbase.d, zidx, H.Z.d, y, alpha, beta, cnt, xin, nx};
                switch (EW) {
                case 8: hipLaunchKernelGGL(expand_kernel<8>, dim3(cnt), dim3(512), 0, st, X); break;
                default: hipLaunchKernelGGL(expand_kernel<4>, dim3(cnt), dim3(256), 0, st, X); break;
                }
            }
            if (after_chunk)
                after_chunk(after_user, c, H.chunk_row_lo[c], H.chunk_row_hi[c]);
        }
        prof_mark(H, st, "expand_kernel");
    } else if (H.E.nranges() > 0) {
        ExpandArgs X{H.E.stream.d, H.E.d_task_range.d, H.E.d_off.d, H.E.d_len.d, H.E.d_cols.d, H.E.d_base.d, H.E.d_colbase.d, zidx, H.Z.d, y, alpha, beta, H.E.nranges(), xin, nx};
        switch (EW) {
        case 8: hipLaunchKernelGGL(expand_kernel<8>, dim3(H.E.nranges()), dim3(512), 0, st, X); break;
        default: hipLaunchKernelGGL(expand_kernel<4>, dim3(H.E.nranges()), dim3(256), 0, st, X); break;
        }
        prof_mark(H, st, "expand_kernel");
    }
    HMX_HIP(hipGetLastError());
    return HMX_OK;
}

// ---- the transposed product on the STORED data (round 4) ------------------------------------------------------------------------------
// y_s = alpha A^T x_t + beta y_s of an ordinary ('N') operator without a second layout and without atomics: the reference swaps the cluster
// roles on the same leaves (hmatrix/linalg/add_hmatrix_vector_product.hpp:74-81); here the machinery of the fused symmetric product runs with
// every leaf mirrored and nothing applied forward:
//   expand_sym_kernel<W, false>   one sweep over the E-streams: column sums E^T x_t per row range -- a slice of a' = U^T x_t for a low-rank leaf,
//                                 the leaf's contribution to an output row for a dense one -- into their slots
//   combine_list_kernel           a' of the leaves that span several row ranges, fixed order
//   rowsym_kernel                 one sweep over the R-streams, owner-computes: y_s = alpha (V^T a' + dense contributions) + beta y_s
// The tables (slot per E column, coefficient slot per R column, the intervals' sub-task lists, the dense contributions per output row: about
// 3 % of the operator's bytes) are built on demand by build_trans_tables -- hmx_hmatrix_prepare(H, 'T', ...) or the first such product.
// Every stored coefficient is read once, as in the forward product; fixed summation order: bit-reproducible.
static int build_trans_tables(HMat &H) {
    if (H.trans_fused)
        return HMX_OK;
    if (!H.finalized || H.has_mirror || H.view_of)
        return HMX_ERR_UNSUPPORTED;
    HMX_HIP(hipSetDevice(H.device));
    const StreamSet &E = H.E, &R = H.R;
    const int64_t nb   = (int64_t)H.leaves.size();
    const int nre = E.nranges(), nrr = R.nranges();
    // the (leaf, range, first column) pairs of the layout: columns were given out leaf by leaf in the leaves' own order (build_streams)
    std::vector<int32_t> t_pos2range((size_t)H.nT + 1, -1);
    for (int r = 0; r < nre; r++)
        t_pos2range[E.off[r]] = r;
    std::vector<std::pair<int, int>> sclusters;
    for (int64_t b = 0; b < nb; b++)
        if (H.kind[b] == LK_LOWRANK && H.leaves[b].rank > 0)
            sclusters.emplace_back(H.leaves[b].s_offset, H.leaves[b].s_size);
    std::sort(sclusters.begin(), sclusters.end());
    sclusters.erase(std::unique(sclusters.begin(), sclusters.end()), sclusters.end());
    std::vector<int32_t> first(sclusters.size() + 1, 0); // the pieces of the distinct source clusters follow one another in the clusters' order and partition them
    int piece = 0;
    for (size_t c = 0; c < sclusters.size(); c++) {
        first[c] = piece;
        for (int covered = 0; covered < sclusters[c].second && piece < nrr; piece++)
            covered += R.len[piece];
    }
    first[sclusters.size()] = piece;
    std::vector<int32_t> ecnt(nre, 0), rcnt(nrr, 0), elr_b, elr_r, elr_c, ed_b, ed_r, ed_c, rlr_b, rlr_r, rlr_c;
    std::vector<int64_t> aoff(nb, -1);
    int64_t A_total = 0;
    bool ok         = piece == nrr;
    for (int64_t b = 0; b < nb && ok; b++) {
        const hmx_leaf &l = H.leaves[b];
        const bool lr     = H.kind[b] == LK_LOWRANK;
        if (lr && l.rank <= 0)
            continue;
        const int ncols = lr ? l.rank : l.s_size, lo = l.t_offset - H.T0, hi = lo + l.t_size;
        if (lo < 0 || hi > H.nT || t_pos2range[lo] < 0) {
            ok = false;
            break;
        }
        for (int r = t_pos2range[lo]; r < nre && E.off[r] < hi; r++) {
            (lr ? elr_b : ed_b).push_back((int32_t)b);
            (lr ? elr_r : ed_r).push_back(r);
            (lr ? elr_c : ed_c).push_back(ecnt[r]);
            ecnt[r] += ncols;
        }
        if (lr) {
            aoff[b] = A_total;
            A_total += l.rank;
            const size_t sc = std::lower_bound(sclusters.begin(), sclusters.end(), std::make_pair((int)l.s_offset, (int)l.s_size)) - sclusters.begin();
            for (int r = first[sc]; r < first[sc + 1]; r++) {
                rlr_b.push_back((int32_t)b);
                rlr_r.push_back(r);
                rlr_c.push_back(rcnt[r]);
                rcnt[r] += l.rank;
            }
        }
    }
    for (int r = 0; r < nre && ok; r++)
        ok = ecnt[r] == E.cols[r];
    for (int r = 0; r < nrr && ok; r++)
        ok = rcnt[r] == R.cols[r];
    if (!ok || A_total != H.A_total) {
        set_error("transposed product on the stored data: the layout could not be retraced (internal error)");
        return HMX_ERR_STATE;
    }
    MirrorCtx M{H.leaves, H.kind, nb, elr_b, elr_r, elr_c, ed_b, ed_r, ed_c, rlr_b, rlr_r, rlr_c, aoff, A_total, true, nullptr};
    const int rc = build_mirror_tables(H, M);
    if (rc != HMX_OK) {
        for (auto *a : {&H.s_mdst, &H.s_coef, &H.s_count, &H.sc_dst, &H.sc_lp, &H.sc_count, &H.sc_k, &H.s_list, &H.s_fidx, &H.s_sub_task, &H.s_sub_row0, &H.s_sub_nrows, &H.s_sub_dst, &H.s_int_order})
            a->release();
        H.s_sub_ptr.release();
        H.SW.release();
        return rc;
    }
    H.trans_fused = true;
    return HMX_OK;
}

static int run_transposed_fused(HMat &H, const scalar *in, scalar alpha, scalar beta, scalar *out, hipStream_t st) {
    const int EW = H.opt.i(HMX_OPT_EXPAND_WAVES) ? H.opt.i(HMX_OPT_EXPAND_WAVES) : (H.E.nranges() <= 4096 ? 8 : 4);
    if (H.E.nranges() > 0) {
        ExpandSymArgs X{{H.E.stream.d, H.E.d_task_range.d, H.E.d_off.d, H.E.d_len.d, H.E.d_cols.d, H.E.d_base.d, H.E.d_colbase.d, nullptr, H.Z.d, nullptr, alpha, beta, H.E.nranges(), nullptr, 0},
                        H.s_mdst.d, H.SW.d, in, 0};
        switch (EW) {
        case 8: hipLaunchKernelGGL((expand_sym_kernel<8, false>), dim3(H.E.nranges()), dim3(512), 0, st, X); break;
        default: hipLaunchKernelGGL((expand_sym_kernel<4, false>), dim3(H.E.nranges()), dim3(256), 0, st, X); break;
        }
        prof_mark(H, st, "expand_colsum_kernel");
    }
    if (H.n_sym_combine > 0) {
        const int nw = H.n_sym_combine_wave, nt = H.n_sym_combine - nw;
        if (nw > 0) {
            CombineListArgs C{H.sc_dst.d, H.sc_lp.d, H.sc_count.d, H.sc_k.d, H.s_list.d, H.SW.d, nw};
            hipLaunchKernelGGL(combine_list_wave_kernel, dim3((nw + 3) / 4), dim3(256), 0, st, C);
        }
        if (nt > 0) {
            CombineListArgs C{H.sc_dst.d + nw, H.sc_lp.d + nw, H.sc_count.d + nw, H.sc_k.d + nw, H.s_list.d, H.SW.d, nt};
            hipLaunchKernelGGL(combine_list_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, C);
        }
        prof_mark(H, st, "combine_sym_kernel");
    }
    if (H.s_nint > 0) {
        RowSymArgs A{H.R.stream.d, H.R.d_task_range.d, H.R.d_task_chunk.d, H.R.d_len.d, H.R.d_cols.d, H.R.d_cw.d, H.R.d_base.d, H.R.d_colbase.d, H.s_coef.d,
                     H.s_int_order.d, H.s_sub_ptr.d, H.s_sub_task.d, H.s_sub_row0.d, H.s_sub_nrows.d, H.s_sub_dst.d,
                     H.SW.d, H.s_fidx.d, H.s_count.d, out, alpha, H.nS, 0, beta, 0};
        hipLaunchKernelGGL(rowsym_kernel<SYM_WAVES>, dim3(H.s_nint), dim3(SYM_WAVES * 64), 0, st, A);
        prof_mark(H, st, "rowsym_kernel");
    }
    HMX_HIP(hipGetLastError());
    return HMX_OK;
}

// Wave-uniform operand of the multi-RHS VALU reduce kernel through the scalar cache instead of LDS (real coefficient types).
// Measured at N=1e6, mu=16, fp32: reduce 0.85 ms (scalar) vs 0.98 ms (LDS); the same trick in the expand stage lost (1.22 vs 1.14 ms: the
// gathered coefficient rows miss the scalar cache) and was removed.  Default: fp32 only.  HMX_MU_SCALAR=0 / 1: never / also for fp64.
static bool mu_scalar_operands(const HMat &H) {
    const int v = H.opt.i(HMX_OPT_SCALAR_OPERANDS);
    return v < 0 ? sizeof(scalar) == 4 : v != 0;
}
template <int MU>
static void launch_mu(HMat &H, ReduceArgs &RA, int mu, int cbase, hipStream_t st) {
    constexpr int RW = 4;
#if !HMX_COMPLEX
    if (MU >= 4 && mu_scalar_operands(H)) {
        if constexpr (MU >= 4)
            if (RA.ntasks > 0)
                hipLaunchKernelGGL((reduce_mus_kernel<RW, MU>), dim3((RA.ntasks + RW - 1) / RW), dim3(RW * 64), 0, st, RA, mu, cbase);
        prof_mark(H, st, "reduce_mus_kernel");
        return;
    }
#endif
    if (RA.ntasks > 0)
        hipLaunchKernelGGL((reduce_mu_kernel<RW, MU>), dim3((RA.ntasks + RW - 1) / RW), dim3(RW * 64), 0, st, RA, mu, cbase);
    prof_mark(H, st, "reduce_mu_kernel");
}
template <int MU>
static void launch_mu_expand(HMat &H, ExpandArgs &XA, int mu, int cbase, hipStream_t st) {
    constexpr int EW = 4;
    if (XA.nranges > 0)
        hipLaunchKernelGGL((expand_mu_kernel<EW, MU>), dim3(XA.nranges), dim3(EW * 64), 0, st, XA, mu, cbase);
    prof_mark(H, st, "expand_mu_kernel");
}

// Fused multi-RHS forward pass (trans='N', no mirror leaves): Y = alpha * H * X + beta * Y, X and Y row-major.
static int run_forward_mu(HMat &H, const scalar *X, scalar alpha, scalar beta, scalar *Y, int mu, hipStream_t st, int nchunks = 0, after_chunk_fn after_chunk = nullptr,
                          void *after_user = nullptr) {
    const size_t need = (size_t)(H.zero_slot + 1) * mu;
    if (H.Zmu.n < need)
        HMX_HIP(H.Zmu.alloc(need));
    // the x region of Zmu is never filled: both stages read the caller's X directly
    ReduceArgs RA{H.R.stream.d, H.R.d_task_range.d, H.R.d_task_chunk.d, H.R.d_off.d, H.R.d_len.d, H.R.d_cols.d, H.R.d_cw.d, H.R.d_base.d, H.R.d_colbase.d,
                  H.r_outidx.d, X, H.Zmu.d, (int)H.R.task_range.size()};
    ExpandArgs XA{H.E.stream.d, H.E.d_task_range.d, H.E.d_off.d, H.E.d_len.d, H.E.d_cols.d, H.E.d_base.d, H.E.d_colbase.d, H.e_zidx.d, H.Zmu.d, Y, alpha, beta, H.E.nranges(), X, H.nS};
    // Groups of right-hand sides, one sweep over the streams each.  Real coefficients: groups of 16 and, beyond 16, of up to 32 run on the
    // matrix cores with the stream tiles staged through LDS (*_mfma16s / *_mfma32s); complex: groups of 8 / up to 16 (*_zmfma8s / *_zmfma16s).
    // Those kernels take RAGGED groups (missing right-hand sides are operands nobody stores the results of): 9 ... 15 real right-hand
    // sides are one group of 16 instead of 8 + 4 + 2 + 1 (four sweeps), 3 and 5 ... 7 likewise.  Exact groups of 8, 4, 2, 1 run the VALU
    // kernels.  HMX_NO_MFMA=1: VALU kernels throughout (A/B comparison; fp32: HMX_MFMA_F32=0), HMX_MFMA_WIDE=0: no sweeps of 32 (complex: 16).
    const bool no_mfma = H.opt.i(HMX_OPT_MATRIX_CORES) == 0;
#if HMX_COMPLEX
    const bool use_mfma = !no_mfma;
    constexpr int GMAX  = 8; // widest VALU kernel
#else
    const bool use_mfma = !no_mfma && (sizeof(scalar) == 8 || H.opt.i(HMX_OPT_MATRIX_CORES_F32) != 0);
    constexpr int GMAX  = 16;
#endif
    const int wide = H.opt.i(HMX_OPT_WIDE_SWEEPS);
    // fn(kernel width, first column, right-hand sides in the group); the two stages need not cut the right-hand sides into the same
    // groups (stage 2 starts when all of stage 1 is done), but they do
    auto for_groups = [&](auto &&fn) {
        int c = 0;
        while (c < mu) {
            const int left = mu - c;
            int g          = (left >= 16 && GMAX >= 16) ? 16 : (left >= 8 ? 8 : (left >= 4 ? 4 : (left >= 2 ? 2 : 1)));
            int n          = g;
            const bool odd_tail = left == 3 || (left >= 5 && left < 8);
#if HMX_COMPLEX
            if (use_mfma && odd_tail)
                g = 8, n = left;
            if (use_mfma && wide && left > 8)
                g = 16, n = left < 16 ? left : 16;
#else
            if (use_mfma && ((left >= 9 && left < 16) || odd_tail))
                g = 16, n = left;
            if (use_mfma && wide && left > 16)
                g = 32, n = left < 32 ? left : 32;
#endif
            fn(g, c, n);
            c += n;
        }
    };
    // stage 1 for every group of right-hand sides, then the partial sums, then stage 2
    for_groups([&](int g, int c, int nrhs) {
        constexpr int W = 4; // tasks (= waves) per workgroup
        const dim3 grid((unsigned)((RA.ntasks + W - 1) / W)), wg(W * 64);
#if HMX_COMPLEX
        if (use_mfma && (g == 16 || g == 8)) {
            if (RA.ntasks > 0) {
                if (g == 16)
                    hipLaunchKernelGGL((reduce_zmfma16s_kernel<W>), grid, wg, 0, st, RA, mu, c, nrhs);
                else
                    hipLaunchKernelGGL((reduce_zmfma8s_kernel<W>), grid, wg, 0, st, RA, mu, c, nrhs);
            }
            prof_mark(H, st, g == 16 ? "reduce_zmfma16s_kernel" : "reduce_zmfma8s_kernel");
            return;
        }
#else
        if (use_mfma && (g == 32 || g == 16)) {
            if (RA.ntasks > 0) {
                if (g == 32)
                    hipLaunchKernelGGL((reduce_mfma32s_kernel<W>), grid, wg, 0, st, RA, mu, c, nrhs);
                else
                    hipLaunchKernelGGL((reduce_mfma16s_kernel<W>), grid, wg, 0, st, RA, mu, c, nrhs);
            }
            prof_mark(H, st, g == 32 ? "reduce_mfma32s_kernel" : "reduce_mfma16s_kernel");
            return;
        }
#endif
        switch (g) {
#if !HMX_COMPLEX
        case 16: launch_mu<16>(H, RA, mu, c, st); break;
#endif
        case 8: launch_mu<8>(H, RA, mu, c, st); break;
        case 4: launch_mu<4>(H, RA, mu, c, st); break;
        case 2: launch_mu<2>(H, RA, mu, c, st); break;
        default: launch_mu<1>(H, RA, mu, c, st); break;
        }
    });
    if (H.n_combine > 0) {
        CombineArgs C{H.c_dst.d, H.c_src.d, H.c_stride.d, H.c_count.d, H.Zmu.d, H.n_combine};
        const int64_t tot = (int64_t)H.n_combine * mu;
        hipLaunchKernelGGL(combine_mu_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, C, mu);
        prof_mark(H, st, "combine_mu_kernel");
    }
    auto expand_group = [&](int g, int c, int nrhs) {
        constexpr int W = 4; // waves per row range
        const dim3 grid((unsigned)XA.nranges), wg(W * 64);
#if HMX_COMPLEX
        if (use_mfma && (g == 16 || g == 8)) {
            if (XA.nranges > 0) {
                if (g == 16)
                    hipLaunchKernelGGL((expand_zmfma16s_kernel<W>), grid, wg, 0, st, XA, mu, c, nrhs);
                else
                    hipLaunchKernelGGL((expand_zmfma8s_kernel<W>), grid, wg, 0, st, XA, mu, c, nrhs);
            }
            prof_mark(H, st, g == 16 ? "expand_zmfma16s_kernel" : "expand_zmfma8s_kernel");
            return;
        }
#else
        if (use_mfma && (g == 32 || g == 16)) {
            if (XA.nranges > 0) {
                if (g == 32)
                    hipLaunchKernelGGL((expand_mfma32s_kernel<W>), grid, wg, 0, st, XA, mu, c, nrhs);
                else
                    hipLaunchKernelGGL((expand_mfma16s_kernel<W>), grid, wg, 0, st, XA, mu, c, nrhs);
            }
            prof_mark(H, st, g == 32 ? "expand_mfma32s_kernel" : "expand_mfma16s_kernel");
            return;
        }
#endif
        switch (g) {
#if !HMX_COMPLEX
        case 16: launch_mu_expand<16>(H, XA, mu, c, st); break;
#endif
        case 8: launch_mu_expand<8>(H, XA, mu, c, st); break;
        case 4: launch_mu_expand<4>(H, XA, mu, c, st); break;
        case 2: launch_mu_expand<2>(H, XA, mu, c, st); break;
        default: launch_mu_expand<1>(H, XA, mu, c, st); break;
        }
    };
    if (nchunks > 1 && H.E.nranges() > 1) {
        // the expand stage over contiguous groups of row ranges, all groups of right-hand sides per chunk: after chunk c its rows of Y are
        // final and `after_chunk` may start sending them while chunk c + 1 computes (as run_forward does for one vector)
        const int rc = ensure_expand_chunks(H, nchunks);
        if (rc != HMX_OK)
            return rc;
        for (int c = 0; c < H.chunk_plan_n; c++) {
            XA.order   = H.d_chunk_order.d + H.chunk_first[c];
            XA.nranges = H.chunk_count[c];
            if (XA.nranges > 0)
                for_groups(expand_group);
            if (after_chunk)
                after_chunk(after_user, c, H.chunk_row_lo[c], H.chunk_row_hi[c]);
        }
    } else {
        for_groups(expand_group);
        if (after_chunk)
            after_chunk(after_user, 0, 0, H.nT);
    }
    HMX_HIP(hipGetLastError());
    return HMX_OK;
}

// Several right-hand sides on the STORED TRIANGLE of a symmetric / Hermitian operator (kernels_body.hpp, "Several right-hand sides on the
// stored data"): sweeps of up to SWW right-hand sides (16 real, 8 complex); per sweep the reduce stage, the fused pass over the E-streams, the
// fold of a' and the second pass over the R-streams.  No second layout of the operator: what is added to the compact operator is SW16, SWW
// partial sums per slot of the single-vector product (N = 1e6 fp64: 1.6 GB next to 9.4 GB of streams; the expanded view: 18.6 GB).
// Real coefficients run on the matrix cores (expand_sym_mfma16_kernel, rowsym_mfma16_kernel), complex ones on the VALU (expand_sym_mu_kernel,
// rowsym_mu_kernel; round 5 -- before, complex operators without room for the view ran one single-vector product per right-hand side).
static bool sym_mu_fused(const HMat &H) {
    // HMX_OPT_SYM_MULTI_RHS = 1: always the stored triangle; 0: always the expanded view; -1: the expanded view while HBM has room for it
    // (the faster of the two today: N = 4e6 fp32, 16 right-hand sides, one MI355X: 15.9 ms on 42 + 83 GB against 18 ms on 42 + 6 GB),
    // the stored triangle when it has not, or when the factors the view is built from were released
    const int mode = H.opt.i(HMX_OPT_SYM_MULTI_RHS);
    if (!(H.sym_fused && H.s64_nint > 0) || mode == 0)
        return false;
    if (mode > 0)
        return true;
#if HMX_COMPLEX
    // complex coefficients: the stored triangle is as fast as the view (N = 1e6 Hermitian complex double, 8 right-hand sides: 17.0 ms on 54.8 GB
    // against 18.2 ms on 54.8 + 108.5 GB; complex symmetric 6.4 against 6.6 ms) -- the mirrored product packs both planes of 8 columns into one
    // MFMA per k-step -- so nothing is built unless it is asked for
    return true;
#endif
    if (H.X_op)
        return false;
    if (H.X_op_failed || H.factors_released || H.opt.i(HMX_OPT_SYM_NO_VIEW) != 0)
        return true;
    size_t free_b = 0, total_b = 0;
    return hmx_mem_info(&free_b, &total_b) != hipSuccess || (double)free_b < 2.3 * (double)H.stats.stream_bytes; // ensure_expanded_view's own admission test
}
static int ensure_sw16(HMat &H, hipStream_t st) {
    const size_t need16 = (size_t)(H.s_slots + 1) * SWW;
    if (H.SW16.n < need16) {
        HMX_HIP(H.SW16.alloc(need16));
        HMX_HIP(hipMemsetAsync(H.SW16.d, 0, need16 * sizeof(scalar), st)); // slot s_slots stays zero for ever: the operand of the columns that are no mirrored leaf's
    }
    return HMX_OK;
}
// the sweeps over E (forward + mirrored column sums, or -- fwd = false -- the column sums only), the folds of a' and the second sweep over R
// for the nrhs right-hand sides starting at column c.  herm: mirrored leaves are conjugate transposes.
static int sym_mu_sweeps(HMat &H, bool fwd, const scalar *X, const scalar *xrow, scalar alpha, scalar beta, scalar *Y, int nout, int accumulate, int herm, int mu, int c, int nrhs, hipStream_t st) {
    constexpr int W = 4;
    if (H.E.nranges() > 0) {
        ExpandSymArgs XS{{H.E.stream.d, H.E.d_task_range.d, H.E.d_off.d, H.E.d_len.d, H.E.d_cols.d, H.E.d_base.d, H.E.d_colbase.d, fwd ? H.e_zidx.d : nullptr, fwd ? H.Zmu.d : nullptr,
                          fwd ? Y : nullptr, alpha, beta, H.E.nranges(), fwd ? X : nullptr, fwd ? H.nS : 0},
                         H.s_mdst.d, H.SW16.d, xrow, herm};
        const dim3 grid((unsigned)H.E.nranges()), wg(W * 64);
#if HMX_COMPLEX
#define HMX_SYM_MU_E(MU)                                                                                      \
    do {                                                                                                      \
        if (fwd)                                                                                              \
            hipLaunchKernelGGL((expand_sym_mu_kernel<W, MU, true>), grid, wg, 0, st, XS, mu, c, nrhs);        \
        else                                                                                                  \
            hipLaunchKernelGGL((expand_sym_mu_kernel<W, MU, false>), grid, wg, 0, st, XS, mu, c, nrhs);       \
    } while (0)
        if (H.opt.i(HMX_OPT_MATRIX_CORES) != 0) { // groups of up to 8 on the matrix cores (ragged groups: operands nobody stores the results of)
            if (fwd)
                hipLaunchKernelGGL((expand_sym_zmfma8_kernel<W, true>), grid, wg, 0, st, XS, mu, c, nrhs);
            else
                hipLaunchKernelGGL((expand_sym_zmfma8_kernel<W, false>), grid, wg, 0, st, XS, mu, c, nrhs);
            prof_mark(H, st, fwd ? "expand_sym_zmfma8_kernel" : "expand_colsum_zmfma8_kernel");
        } else {
            if (nrhs <= 2)
                HMX_SYM_MU_E(2);
            else if (nrhs <= 4)
                HMX_SYM_MU_E(4);
            else
                HMX_SYM_MU_E(8);
            prof_mark(H, st, fwd ? "expand_sym_mu_kernel" : "expand_colsum_mu_kernel");
        }
#undef HMX_SYM_MU_E
#else
        if (fwd)
            hipLaunchKernelGGL((expand_sym_mfma16_kernel<W, true>), grid, wg, 0, st, XS, mu, c, nrhs);
        else
            hipLaunchKernelGGL((expand_sym_mfma16_kernel<W, false>), grid, wg, 0, st, XS, mu, c, nrhs);
        prof_mark(H, st, fwd ? "expand_sym_mfma16_kernel" : "expand_colsum_mfma16_kernel");
#endif
    }
    if (H.n_sym_combine > 0) {
        const int nw = H.n_sym_combine_wave, nt = H.n_sym_combine - nw; // the first nw entries fold >= 32 partial sums: one wave each
        if (nw > 0) {
            CombineListArgs C{H.sc_dst.d, H.sc_lp.d, H.sc_count.d, H.sc_k.d, H.s_list.d, H.SW16.d, nw};
            hipLaunchKernelGGL(combine_list_mu_wave_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, st, C);
        }
        if (nt > 0) {
            CombineListArgs C{H.sc_dst.d + nw, H.sc_lp.d + nw, H.sc_count.d + nw, H.sc_k.d + nw, H.s_list.d, H.SW16.d, nt};
            const int64_t tot = (int64_t)nt * SWW;
            hipLaunchKernelGGL(combine_list_mu_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, C);
        }
        prof_mark(H, st, "combine_sym_mu_kernel");
    }
    if (H.s64_nint > 0) {
        RowSymArgs RS{H.R.stream.d, H.R.d_task_range.d, H.R.d_task_chunk.d, H.R.d_len.d, H.R.d_cols.d, H.R.d_cw.d, H.R.d_base.d, H.R.d_colbase.d, H.s_coef.d, H.s64_int_order.d,
                      H.s64_sub_ptr.d, H.s64_sub_task.d, H.s64_sub_row0.d, H.s64_sub_nrows.d, H.s64_sub_dst.d, H.SW16.d, H.s_fidx.d, H.s_count.d, Y, alpha, nout, herm, beta, accumulate};
#if HMX_COMPLEX
        const dim3 grid((unsigned)H.s64_nint), wg(W * 64);
        if (H.opt.i(HMX_OPT_MATRIX_CORES) != 0) {
            RowSymZArgs PZ{RS, reinterpret_cast<const real *>(H.SW16.d), (int)H.s_slots, H.s64_nint};
            hipLaunchKernelGGL((rowsym_zmfma8_kernel<W>), dim3((unsigned)((H.s64_nint + W - 1) / W)), wg, 0, st, PZ, mu, c, nrhs);
            prof_mark(H, st, "rowsym_zmfma8_kernel");
        } else if (nrhs <= 2)
            hipLaunchKernelGGL((rowsym_mu_kernel<W, 2>), grid, wg, 0, st, RS, (const scalar *)H.SW16.d, mu, c, nrhs);
        else if (nrhs <= 4)
            hipLaunchKernelGGL((rowsym_mu_kernel<W, 4>), grid, wg, 0, st, RS, (const scalar *)H.SW16.d, mu, c, nrhs);
        else
            hipLaunchKernelGGL((rowsym_mu_kernel<W, 8>), grid, wg, 0, st, RS, (const scalar *)H.SW16.d, mu, c, nrhs);
        if (H.opt.i(HMX_OPT_MATRIX_CORES) == 0)
            prof_mark(H, st, "rowsym_mu_kernel");
#else
        RowSymMuArgs P{RS, H.SW16.d, (int)H.s_slots, H.s64_nint};
        constexpr int RWV = HMX_ROWSYM_WAVES; // intervals (= waves) per workgroup
        hipLaunchKernelGGL((rowsym_mfma16_kernel<RWV>), dim3((unsigned)((H.s64_nint + RWV - 1) / RWV)), dim3(RWV * 64), 0, st, P, mu, c, nrhs);
        prof_mark(H, st, "rowsym_mfma16_kernel");
#endif
    }
    return HMX_OK;
}
static int run_forward_mu_sym(HMat &H, const scalar *X, scalar alpha, scalar beta, scalar *Y, int mu, hipStream_t st) {
    const size_t need = (size_t)(H.zero_slot + 1) * mu;
    if (H.Zmu.n < need)
        HMX_HIP(H.Zmu.alloc(need));
    int rc = ensure_sw16(H, st);
    if (rc != HMX_OK)
        return rc;
    ReduceArgs RA{H.R.stream.d, H.R.d_task_range.d, H.R.d_task_chunk.d, H.R.d_off.d, H.R.d_len.d, H.R.d_cols.d, H.R.d_cw.d, H.R.d_base.d, H.R.d_colbase.d,
                  H.r_outidx.d, X, H.Zmu.d, (int)H.R.task_range.size()};
    constexpr int W = 4;
    for (int c = 0; c < mu; c += SWW) { // a = V X_s, every sweep
        const int nrhs = std::min(SWW, mu - c);
        if (RA.ntasks > 0) {
#if HMX_COMPLEX
            hipLaunchKernelGGL((reduce_zmfma8s_kernel<W>), dim3((unsigned)((RA.ntasks + W - 1) / W)), dim3(W * 64), 0, st, RA, mu, c, nrhs);
#else
            hipLaunchKernelGGL((reduce_mfma16s_kernel<W>), dim3((unsigned)((RA.ntasks + W - 1) / W)), dim3(W * 64), 0, st, RA, mu, c, nrhs);
#endif
        }
        prof_mark(H, st, HMX_COMPLEX ? "reduce_zmfma8s_kernel" : "reduce_mfma16s_kernel");
    }
    if (H.n_combine > 0) {
        CombineArgs C{H.c_dst.d, H.c_src.d, H.c_stride.d, H.c_count.d, H.Zmu.d, H.n_combine};
        const int64_t tot = (int64_t)H.n_combine * mu;
        hipLaunchKernelGGL(combine_mu_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, C, mu);
        prof_mark(H, st, "combine_mu_kernel");
    }
    const int herm = H.symmetry_for_leaves == 'H' ? 1 : 0;
    for (int c = 0; c < mu; c += SWW) {
        rc = sym_mu_sweeps(H, true, X, X + (int64_t)(H.T0 - H.S0) * mu, alpha, beta, Y, H.nT, 1, herm, mu, c, std::min(SWW, mu - c), st);
        if (rc != HMX_OK)
            return rc;
    }
    HMX_HIP(hipGetLastError());
    return HMX_OK;
}

// Several right-hand sides of the transposed product on the STORED data (run_transposed_fused for groups of SWW): the kernels of the
// stored-triangle product with every leaf mirrored and nothing applied forward.  Runs when HBM has no room for the transposed stream layout
// the fused multi-RHS kernels prefer (until round 4: one single-vector product per right-hand side then; complex types until round 5).
static int run_transposed_fused_mu(HMat &H, const scalar *X, scalar alpha, scalar beta, scalar *Y, int mu, hipStream_t st) {
    int rc = ensure_sw16(H, st);
    if (rc != HMX_OK)
        return rc;
    for (int c = 0; c < mu; c += SWW) {
        rc = sym_mu_sweeps(H, false, nullptr, X, alpha, beta, Y, H.nS, 0, 0, mu, c, std::min(SWW, mu - c), st);
        if (rc != HMX_OK)
            return rc;
    }
    HMX_HIP(hipGetLastError());
    return HMX_OK;
}

// trans = 'T' at the speed of trans = 'N': the transposed operator gets its own E-/R-streams (same crosses with the roles of
// U and V exchanged, dense leaves regenerated / read transposed), built on the first transposed product.  Costs a second
// copy of the streams in HBM.  nullptr: not possible (HMX_OPT_TRANSPOSED_LAYOUT = 0, factors released, no room) -- the callers then run
// on the stored data (run_transposed_fused, run_transposed_fused_mu) or, for a row-restricted symmetric operator, report the reason.
static int build_streams(HMat &H);
static HMat *ensure_transposed_operator(HMat &H) {
    if (H.T_op)
        return H.T_op.get();
    // (a fused symmetric owner is fine: the view lays the mirrored leaves out explicitly, see build_streams)
    if (H.factors_released || H.T_op_failed || H.view_of || H.opt.i(HMX_OPT_TRANSPOSED_LAYOUT) == 0)
        return nullptr;
    size_t free_b = 0, total_b = 0;
    // a fused symmetric owner holds the stored triangle only, its transposed view the whole operator
    if (hmx_mem_info(&free_b, &total_b) != hipSuccess || (double)free_b < (H.sym_fused ? 2.3 : 1.15) * (double)H.stats.stream_bytes) {
        H.T_op_failed = true; // not enough HBM for a second layout
        return nullptr;
    }
    std::unique_ptr<HMat> T(new HMat());
    T->device  = H.device;
    T->opt     = H.opt;
    T->view_of = &H;
    T->leaves  = H.leaves;
    for (auto &l : T->leaves) {
        std::swap(l.t_offset, l.s_offset);
        std::swap(l.t_size, l.s_size);
    }
    T->kind = H.kind;
    T->T0 = H.S0, T->nT = H.nS, T->S0 = H.T0, T->nS = H.nT;
    T->nT_total = H.nS_total, T->nS_total = H.nT_total;
    T->tree_t = H.tree_s, T->tree_s = H.tree_t;
    T->symmetry_for_leaves = H.symmetry_for_leaves;
    T->uplo_for_leaves     = H.uplo_for_leaves == 'L' ? 'U' : (H.uplo_for_leaves == 'U' ? 'L' : 'N');
    T->build_epsilon       = H.build_epsilon;
    T->has_mirror          = H.has_mirror;
    T->colptr              = H.colptr;
    T->swapped             = H.swapped;
    T->staged_off          = H.staged_off;
    T->profiling           = H.profiling;
    const hmx_stats keep   = H.stats;
    const int rc           = build_streams(*T);
    (void)keep;
    if (rc != HMX_OK) {
        H.T_op_failed = true;
        (void)hipGetLastError();
        return nullptr;
    }
    H.T_op = std::move(T);
    return H.T_op.get();
}

// Multi-RHS products on compact symmetric storage.  With mu right-hand sides every mirrored column of a 64-row range yields mu
// partial sums: for mu = 16 the partial results would be a quarter of the streamed bytes, written and read again -- more traffic
// than the mirrored copies save.  So the fused multi-RHS kernels run on an expanded layout of the same operator, built from the
// same crosses when the first multi-RHS product arrives (HBM permitting; otherwise one fused single-vector product per column).
static HMat *ensure_expanded_view(HMat &H) {
    if (H.X_op)
        return H.X_op.get();
    if (!H.sym_fused || H.factors_released || H.X_op_failed || H.view_of || H.opt.i(HMX_OPT_SYM_NO_VIEW) != 0)
        return nullptr;
    size_t free_b = 0, total_b = 0;
    if (hmx_mem_info(&free_b, &total_b) != hipSuccess || (double)free_b < 2.3 * (double)H.stats.stream_bytes) {
        H.X_op_failed = true;
        return nullptr;
    }
    std::unique_ptr<HMat> X(new HMat());
    X->device          = H.device;
    X->opt             = H.opt;
    X->view_of         = &H;
    X->view_transposed = false;
    X->leaves          = H.leaves;
    X->kind            = H.kind;
    X->T0 = H.T0, X->nT = H.nT, X->S0 = H.S0, X->nS = H.nS;
    X->nT_total = H.nT_total, X->nS_total = H.nS_total;
    X->tree_t = H.tree_t, X->tree_s = H.tree_s; // symmetric storage: one cluster tree on both sides
    X->symmetry_for_leaves = H.symmetry_for_leaves;
    X->uplo_for_leaves     = H.uplo_for_leaves;
    X->build_epsilon       = H.build_epsilon;
    X->has_mirror          = H.has_mirror;
    X->colptr              = H.colptr;
    X->swapped             = H.swapped;
    X->staged_off          = H.staged_off;
    X->profiling           = H.profiling;
    if (build_streams(*X) != HMX_OK) {
        H.X_op_failed = true;
        (void)hipGetLastError();
        return nullptr;
    }
    H.X_op = std::move(X);
    return H.X_op.get();
}

static int matvec_device(HMat &H, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out, hipStream_t st, bool inner = false) {
    if (!H.finalized) {
        set_error("hmx_hmatrix_matvec: operator not built (call hmx_hmatrix_compress or hmx_hmatrix_finalize first)");
        return HMX_ERR_STATE;
    }
#if HMX_COMPLEX
    // add_hmatrix_vector_product.hpp:59-62: trans='T' with 'H' leaves and trans='C' with 'S' leaves are refused
    if (!inner && ((trans == 'T' && H.symmetry_for_leaves == 'H') || (trans == 'C' && H.symmetry_for_leaves == 'S'))) {
        set_error(std::string("hmx_hmatrix_matvec: operation is not supported (trans=") + trans + " with " + H.symmetry_for_leaves + " leaves)");
        return HMX_ERR_INVALID;
    }
    if (trans == 'C' && H.symmetry_for_leaves == 'H' && H.has_mirror && H.T0 == H.S0 && H.nT == H.nS)
        trans = 'N'; // a square Hermitian operator is its own conjugate transpose
    if (trans == 'C') { // alpha A^H x + beta y = conj( conj(alpha) A^T conj(x) + conj(beta) conj(y) )
        const int nin = H.nT, nout = H.nS;
        if ((int64_t)H.conj_in.n < nin)
            HMX_HIP(H.conj_in.alloc(std::max(nin, 1)));
        hipLaunchKernelGGL(conj_kernel, dim3((nin + 255) / 256), dim3(256), 0, st, (int64_t)nin, in, H.conj_in.d);
        if (!hmx_is_zero(beta))
            hipLaunchKernelGGL(conj_kernel, dim3((nout + 255) / 256), dim3(256), 0, st, (int64_t)nout, (const scalar *)out, out);
        const int rc = matvec_device(H, 'T', hmx_conj(alpha), H.conj_in.d, hmx_conj(beta), out, st, true);
        if (rc != HMX_OK)
            return rc;
        hipLaunchKernelGGL(conj_kernel, dim3((nout + 255) / 256), dim3(256), 0, st, (int64_t)nout, (const scalar *)out, out);
        HMX_HIP(hipGetLastError());
        return HMX_OK;
    }
#endif
#if !HMX_COMPLEX
    if (trans == 'C') { // real coefficients: the conjugate transpose is the transpose (BLAS gemv 'C'); the reference still
                        // refuses 'C' on 'S' leaves (add_hmatrix_vector_product.hpp:59-62)
        if (!inner && H.symmetry_for_leaves == 'S') {
            set_error("hmx_hmatrix_matvec: operation is not supported (trans=C with S leaves)");
            return HMX_ERR_INVALID;
        }
        trans = 'T';
    }
#endif
    if (trans != 'N' && trans != 'T') {
        set_error("hmx_hmatrix_matvec: trans must be 'N', 'T' or 'C'");
        return HMX_ERR_INVALID;
    }
    if (H.has_mirror && !H.sym_expanded && (H.S0 > H.T0 || H.S0 + H.nS < H.T0 + H.nT)) {
        set_error("symmetric storage needs the target rows to be a sub-range of the source columns");
        return HMX_ERR_UNSUPPORTED;
    }
    H.ev_names.clear();
    prof_mark(H, st, "begin");
    int rc;
    // a square operator stored symmetrically IS its own transpose ('S') / conjugate transpose ('H', handled above as 'C')
    if (trans == 'T' && !inner && H.symmetry_for_leaves == 'S' && H.has_mirror && H.T0 == H.S0 && H.nT == H.nS)
        trans = 'N';
    bool done = false;
    if (trans == 'T') {
        // An ordinary operator: on the STORED data (run_transposed_fused; tables of ~3 % of the operator built on first use or by
        // hmx_hmatrix_prepare) unless a transposed layout exists already (a multi-RHS 'T' product builds one, HBM permitting) or
        // HMX_OPT_TRANSPOSED_LAYOUT = 1 asks for it: a second copy of the streams is then the price of the last 10 % of speed.
        // A row-restricted symmetric operator (mirrored leaves among ordinary ones) always runs on its transposed view.
        const int want_streams = H.opt.i(HMX_OPT_TRANSPOSED_LAYOUT);
        if (!H.has_mirror && !H.view_of && !H.T_op && want_streams != 1) {
            if (!H.trans_fused && !H.trans_tables_failed && build_trans_tables(H) != HMX_OK) {
                H.trans_tables_failed = true;
                (void)hipGetLastError();
            }
            if (H.trans_fused) {
                rc   = run_transposed_fused(H, in, alpha, beta, out, st);
                done = true;
            }
        }
        if (!done) {
            HMat *T = ensure_transposed_operator(H);
            if (!T) {
                set_error(std::string("hmx_hmatrix_matvec: the transposed product of this operator needs its transposed stream layout, which cannot be built (") +
                          (H.factors_released ? "the factors were released: call hmx_hmatrix_release_factors with bit 0 of with_transposed set"
                                              : (H.opt.i(HMX_OPT_TRANSPOSED_LAYOUT) == 0 ? "HMX_OPT_TRANSPOSED_LAYOUT is 0" : "not enough free device memory")) +
                          ")");
                return HMX_ERR_UNSUPPORTED;
            }
            T->profiling = H.profiling;
            rc           = matvec_device(*T, 'N', alpha, in, beta, out, st, true);
            if (rc == HMX_OK && H.profiling) {
                H.last_ms    = T->last_ms;
                H.last_names = T->last_names;
            }
            return rc;
        }
    } else {
        rc = run_forward(H, H.e_zidx.d, in, alpha, beta, out, st, H.sym_fused);
    }
    if (rc != HMX_OK)
        return rc;
    if (H.profiling) {
        HMX_HIP(hipStreamSynchronize(st));
        H.last_ms.clear();
        H.last_names.clear();
        for (size_t k = 1; k < H.ev_names.size(); k++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, H.ev[k - 1], H.ev[k]);
            H.last_ms.push_back(ms);
            H.last_names.push_back(H.ev_names[k]);
        }
    }
    return HMX_OK;
}


int api_create(const hmx_block_tree *bt, int device_id, HMat **out) {
    if (!bt || !out) {
        set_error("hmx_hmatrix_create: NULL argument");
        return HMX_ERR_INVALID;
    }
    int rc = ensure_device(device_id);
    if (rc != HMX_OK)
        return rc;
    auto *H   = new HMat();
    H->device = device_id;
    H->leaves = bt->leaves;
    H->kind.assign(H->leaves.size(), LK_PENDING);
    H->T0 = bt->root_t_offset;
    H->nT = bt->root_t_size;
    H->S0 = bt->root_s_offset;
    H->nS = bt->root_s_size;
    H->nT_total            = bt->target->n;
    H->nS_total            = bt->source->n;
    H->symmetry_for_leaves = bt->symmetry_for_leaves;
    H->uplo_for_leaves     = bt->uplo_for_leaves;
    H->perm_t              = bt->target->perm;
    H->perm_s              = bt->source->perm;
    auto copy_tree = [](const hmx_cluster_tree &T, std::vector<HMat::TreeNode> &out) {
        out.resize(T.nodes.size());
        for (size_t v = 0; v < T.nodes.size(); v++)
            out[v] = HMat::TreeNode{T.nodes[v].offset, T.nodes[v].size, T.nodes[v].first_child, T.nodes[v].n_children};
    };
    copy_tree(*bt->target, H->tree_t);
    copy_tree(*bt->source, H->tree_s);
    H->t_root_is_tree_root = (H->T0 == 0 && H->nT == bt->target->n);
    H->perm_local          = bt->target->permutation_is_local;
    for (auto &l : H->leaves)
        H->has_mirror = H->has_mirror || l.mirror;
    const size_t nb = H->leaves.size();
    std::vector<int32_t> a(nb), b(nb), c(nb), d(nb);
    for (size_t i = 0; i < nb; i++) {
        a[i] = H->leaves[i].t_offset;
        b[i] = H->leaves[i].t_size;
        c[i] = H->leaves[i].s_offset;
        d[i] = H->leaves[i].s_size;
    }
    // leaf offsets on the device are GLOBAL cluster positions (they index coordinates); stream ranges are
    // root-local, the pack kernels add the origin back.
    if (H->d_t_off.upload(a) != hipSuccess || H->d_t_size.upload(b) != hipSuccess || H->d_s_off.upload(c) != hipSuccess || H->d_s_size.upload(d) != hipSuccess) {
        set_error("hmx_hmatrix_create: device allocation failed");
        delete H;
        return HMX_ERR_HIP;
    }
    // staged_U / V / D (one std::vector per leaf, for blocks uploaded through set_block_*) are sized on first use: ensure_staged
    *out = H;
    return HMX_OK;
}


// hmx_hmatrix_set_option / get_option (include/hmx.h: hmx_option).  Layout options are fixed once the streams exist, build options once the
// blocks are compressed; product options may change between any two products.
int api_set_option(HMat *H, int option, double value) {
    const OptionSpec *sp = H ? Options::spec(option) : nullptr;
    if (!sp) {
        set_error("hmx_hmatrix_set_option: unknown option " + std::to_string(option));
        return HMX_ERR_INVALID;
    }
    if (!(value >= sp->lo && value <= sp->hi)) {
        set_error(std::string("hmx_hmatrix_set_option: value out of range for ") + sp->env + " [" + std::to_string(sp->lo) + ", " + std::to_string(sp->hi) + "]");
        return HMX_ERR_INVALID;
    }
    if (sp->when != OPT_PRODUCT && H->finalized && H->opt.v[option] != value) {
        set_error(std::string("hmx_hmatrix_set_option: ") + sp->env + " is a " + (sp->when == OPT_LAYOUT ? "layout" : "build") + " option: set it before hmx_hmatrix_compress / hmx_hmatrix_finalize");
        return HMX_ERR_STATE;
    }
    H->opt.v[option] = value;
    if (H->T_op)
        H->T_op->opt.v[option] = value;
    if (H->X_op)
        H->X_op->opt.v[option] = value;
    return HMX_OK;
}
int api_get_option(const HMat *H, int option, double *value) {
    const OptionSpec *sp = (H && value) ? Options::spec(option) : nullptr;
    if (!sp) {
        set_error("hmx_hmatrix_get_option: unknown option or NULL argument");
        return HMX_ERR_INVALID;
    }
    *value = H->opt.v[option];
    return HMX_OK;
}
int api_set_callback_threads(HMat *H, int threads) {
    if (!H || threads < 0) {
        set_error("hmx_hmatrix_set_callback_threads: invalid arguments");
        return HMX_ERR_INVALID;
    }
    H->callback_threads = threads;
    return HMX_OK;
}
int api_set_kernel(HMat *H, int kernel, const double *params, int nparams, int dim, const double *tc, const double *sc) {
    const int need_params = kernel == HMX_KERNEL_INV_DIST ? 2 : (kernel == HMX_KERNEL_HELMHOLTZ ? 3 : (kernel == HMX_KERNEL_LAPLACE_SL ? 1 : 1 << 30));
    if (!H || !params || !tc || !sc || nparams < need_params || (dim != 2 && dim != 3)) {
        set_error("hmx_hmatrix_set_kernel: invalid arguments (unknown kernel, too few parameters, or a dimension other than 2 / 3)");
        return HMX_ERR_INVALID;
    }
    HMX_HIP(hipSetDevice(H->device));
    if (kernel == HMX_KERNEL_INV_DIST) // params: delta, scale [, cre, cim, hermitian] -- the last three only matter for complex coefficients
        H->ks = KernelSpec{KS_INV_DIST, dim, params[0], params[1], nparams > 2 ? params[2] : 1.0, nparams > 3 ? params[3] : 0.0, (nparams > 4 && params[4] != 0.0) ? 1 : 0, 0.0};
    else if (kernel == HMX_KERNEL_HELMHOLTZ) // params: delta, scale, wavenumber
        H->ks = KernelSpec{KS_HELMHOLTZ, dim, params[0], params[1], 1.0, 0.0, 0, params[2]};
    else // HMX_KERNEL_LAPLACE_SL: delta [, cre, cim]
        H->ks = KernelSpec{KS_LAPLACE_SL, dim, params[0], 1.0, nparams > 1 ? params[1] : 1.0, nparams > 2 ? params[2] : 0.0, 0, 0.0};
    // coordinates permuted once into cluster order so block rows / columns are contiguous
    auto soa = [&](const double *xyz, const std::vector<int32_t> &perm, DArr<double> &X, DArr<double> &Y, DArr<double> &Zc) -> hipError_t {
        const size_t n = perm.size();
        std::vector<double> x(n), y(n), z(n, 0.0);
        parallel_for(n, [&](size_t lo, size_t hi) { // a gather through the permutation: cache misses, spread over a few threads
            for (size_t i = lo; i < hi; i++) {
                const double *p = xyz + (size_t)dim * perm[i];
                x[i]            = p[0];
                y[i]            = p[1];
                if (dim == 3)
                    z[i] = p[2];
            }
        });
        hipError_t e;
        if ((e = X.upload(x)) != hipSuccess) return e;
        if ((e = Y.upload(y)) != hipSuccess) return e;
        return Zc.upload(z);
    };
    HMX_HIP(soa(tc, H->perm_t, H->tx, H->ty, H->tz));
    HMX_HIP(soa(sc, H->perm_s, H->sx, H->sy, H->sz));
    H->has_kernel = true;
    return HMX_OK;
}

int api_set_callback(HMat *H, void (*fn)(void *, int, int, const int32_t *, const int32_t *, scalar *), void *user) {
    if (!H || !fn) {
        set_error("hmx_hmatrix_set_callback: invalid arguments");
        return HMX_ERR_INVALID;
    }
    H->callback      = fn;
    H->callback_user = user;
    H->has_kernel    = false;
    return HMX_OK;
}


// ---------------------------------------------------------------------------------------------
// Host generator on all cores.  The reference compresses the admissible blocks and assembles the dense ones from an OpenMP
// `parallel for` (HMatrixTreeBuilder::openmp_compute_blocks, hmatrix/tree_builder/tree_builder.hpp:603-648), i.e. the user's
// VirtualGenerator::copy_submatrix runs on every core unless HTOOL_WITH_PYTHON_INTERFACE is defined (:606).  Here:
//   * a few DRIVER threads (lanes; at most 8: more threads inside the HIP runtime cost more than they bring -- measured at N = 1e6:
//     16 lanes 0.68 s, 64 lanes 2.2 s, 256 lanes 21 s for the same work) own a HIP stream and two slots each (a range of one pinned
//     host buffer + device buffer + an event), so that a lane prepares one slot while the other slot's upload, kernel and (for the ACA)
//     packed result copy are in flight;
//   * ALL generator threads (the drivers and the remaining cores as workers) evaluate: a driver cuts the lines of its slot's phase into
//     chunks of ~16 K entries and shares them out (CbLanes::parallel), helping itself until its own chunks are done.
//   * ACA: the admissible blocks are cut into batches (largest blocks first); a slot takes a batch and runs the lock-step iteration on
//     it -- one line per active block evaluated into pinned memory, one H2D copy, one aca_cb_*_kernel launch over the batch, one packed
//     D2H copy of (status, I1, I2) -- until the batch is done, then takes the next batch.  Batches progress independently.
//   * dense leaves / assembled blocks: panels of whole columns, evaluated into pinned memory and copied to their place.
// hmx_hmatrix_set_callback_threads(H, 1) (or HMX_CALLBACK_THREADS=1) keeps every call on the calling thread.
// ---------------------------------------------------------------------------------------------
struct CbSlot {
    scalar *h_buf = nullptr, *d_buf = nullptr;
    size_t cap = 0; // entries
    CbItem *h_items = nullptr, *d_items = nullptr;
    CbResult *h_res = nullptr, *d_res = nullptr;
    size_t cap_blocks = 0;
    hipEvent_t done = nullptr;
    bool pending = false;
    // the batch in progress (ACA)
    std::vector<int32_t> active, I1, I2;
    std::vector<size_t> chunk_first;
    bool row_phase = true;
    hipError_t wait() {
        if (!pending)
            return hipSuccess;
        pending = false;
        return hipEventSynchronize(done);
    }
};
struct CbLane {
    hipStream_t st = nullptr;
    CbSlot slot[2];
};
struct CbLanes {
    int device = 0;
    std::vector<CbLane> lanes; // the drivers
    int nworkers = 0;          // generator threads besides the drivers
    std::mutex mu;
    std::string error; // first failure of any lane
    std::atomic<bool> failed{false};
    // one pinned and one device allocation for all slots
    scalar *h_all = nullptr, *d_all = nullptr;
    char *h_meta = nullptr, *d_meta = nullptr;
    // shared evaluation: jobs = chunked loops published by the drivers
    struct Job {
        const std::function<void(size_t)> *body;
        size_t n;
        std::atomic<size_t> next{0}, done{0};
    };
    std::mutex job_mu;
    std::condition_variable job_cv;
    std::deque<Job *> jobs;
    bool stop = false;
    std::vector<std::thread> workers;

    CbLanes(int dev, int threads, int max_drivers) : device(dev) {
        threads = std::max(1, threads);
        const int nd = std::max(1, std::min(threads, max_drivers));
        lanes.resize((size_t)nd);
        nworkers = threads - nd;
        for (int w = 0; w < nworkers; w++)
            workers.emplace_back([this] { worker_loop(); });
    }
    CbLanes(const CbLanes &)            = delete;
    CbLanes &operator=(const CbLanes &) = delete;
    ~CbLanes() {
        {
            std::lock_guard<std::mutex> lock(job_mu);
            stop = true;
        }
        job_cv.notify_all();
        for (auto &w : workers)
            w.join();
        (void)hipSetDevice(device);
        for (auto &L : lanes) {
            if (L.st)
                (void)hipStreamSynchronize(L.st);
            for (auto &S : L.slot)
                if (S.done)
                    (void)hipEventDestroy(S.done);
            if (L.st)
                (void)hipStreamDestroy(L.st);
        }
        if (h_all)
            (void)hipHostFree(h_all);
        if (d_all)
            (void)hipFree(d_all);
        if (h_meta)
            (void)hipHostFree(h_meta);
        if (d_meta)
            (void)hipFree(d_meta);
    }
    size_t nslots() const { return 2 * lanes.size(); }
    // every slot gets room for `entries` evaluated entries and `blocks` launch positions (called before run())
    hipError_t reserve(size_t entries, size_t blocks) {
        hipError_t e;
        entries = (entries + 63) / 64 * 64;
        blocks  = std::max<size_t>(blocks, 1);
        if (lanes[0].slot[0].cap < entries) {
            if (h_all)
                (void)hipHostFree(h_all);
            if (d_all)
                (void)hipFree(d_all);
            h_all = d_all = nullptr;
            if ((e = hipHostMalloc((void **)&h_all, nslots() * entries * sizeof(scalar), hipHostMallocDefault)) != hipSuccess)
                return e;
            if ((e = hipMalloc((void **)&d_all, nslots() * entries * sizeof(scalar))) != hipSuccess)
                return e;
            size_t k = 0;
            for (auto &L : lanes)
                for (auto &S : L.slot) {
                    S.h_buf = h_all + k * entries;
                    S.d_buf = d_all + k * entries;
                    S.cap   = entries;
                    k++;
                }
        }
        if (lanes[0].slot[0].cap_blocks < blocks) {
            if (h_meta)
                (void)hipHostFree(h_meta);
            if (d_meta)
                (void)hipFree(d_meta);
            h_meta = d_meta = nullptr;
            const size_t per = blocks * (sizeof(CbItem) + sizeof(CbResult)); // both 16 bytes per position
            if ((e = hipHostMalloc((void **)&h_meta, nslots() * per, hipHostMallocDefault)) != hipSuccess)
                return e;
            if ((e = hipMalloc((void **)&d_meta, nslots() * per)) != hipSuccess)
                return e;
            size_t k = 0;
            for (auto &L : lanes)
                for (auto &S : L.slot) {
                    S.h_items    = reinterpret_cast<CbItem *>(h_meta + k * per);
                    S.d_items    = reinterpret_cast<CbItem *>(d_meta + k * per);
                    S.h_res      = reinterpret_cast<CbResult *>(S.h_items + blocks);
                    S.d_res      = reinterpret_cast<CbResult *>(S.d_items + blocks);
                    S.cap_blocks = blocks;
                    k++;
                }
        }
        for (auto &L : lanes)
            for (auto &S : L.slot)
                if (!S.done && (e = hipEventCreateWithFlags(&S.done, hipEventDisableTiming | hipEventBlockingSync)) != hipSuccess)
                    return e;
        return hipSuccess;
    }
    void fail(const std::string &what) {
        std::lock_guard<std::mutex> lock(mu);
        if (error.empty())
            error = what;
        failed = true;
    }
    void worker_loop() {
        for (;;) {
            Job *j   = nullptr;
            size_t i = 0;
            {
                std::unique_lock<std::mutex> lock(job_mu);
                job_cv.wait(lock, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) {
                    if (stop)
                        return;
                    continue;
                }
                j = jobs.front();
                i = j->next.fetch_add(1);
                if (i >= j->n) { // exhausted: nobody may find it any more
                    jobs.pop_front();
                    continue;
                }
            }
            try {
                (*j->body)(i);
            } catch (...) {
                fail("exception in the generator");
            }
            j->done.fetch_add(1, std::memory_order_release);
        }
    }
    // body(0) ... body(n - 1) on all generator threads; returns when every call has returned.  The caller (a driver) takes part.
    void parallel(size_t n, const std::function<void(size_t)> &body) {
        if (workers.empty() || n <= 1) {
            for (size_t i = 0; i < n; i++)
                body(i);
            return;
        }
        Job job;
        job.body = &body;
        job.n    = n;
        {
            std::lock_guard<std::mutex> lock(job_mu);
            jobs.push_back(&job);
        }
        job_cv.notify_all();
        // `job` lives on this frame and the workers hold a pointer to it: whatever body() does on this thread -- the user's generator may
        // throw -- the job leaves the queue and every chunk a worker is still inside has returned before the frame is unwound
        std::exception_ptr thrown;
        size_t claimed = 0; // chunks this thread took and did not finish (an exception: at most one)
        for (;;) {
            const size_t i = job.next.fetch_add(1);
            if (i >= n)
                break;
            try {
                body(i);
            } catch (...) {
                thrown = std::current_exception();
                fail("exception in the generator");
                claimed = 1;
                break;
            }
            job.done.fetch_add(1, std::memory_order_release);
        }
        {
            std::lock_guard<std::mutex> lock(job_mu);
            auto it = std::find(jobs.begin(), jobs.end(), &job);
            if (it != jobs.end())
                jobs.erase(it);
        }
        // chunks nobody has claimed yet will never run now that the job is off the queue: after an exception only the claimed ones are waited for
        const size_t taken = std::min(n, job.next.load());
        while (job.done.load(std::memory_order_acquire) + claimed < (thrown ? taken : n))
            std::this_thread::yield();
        if (thrown)
            std::rethrow_exception(thrown);
    }
    // fn(lane index) on every driver, each on its own thread (one driver: the calling thread); false when anything reported an error
    template <typename F>
    bool run(F &&fn) {
        auto body = [&](int t) {
            try {
                if (hipSetDevice(device) != hipSuccess) {
                    fail("hipSetDevice failed in a generator thread");
                    return;
                }
                if (!lanes[t].st && hipStreamCreateWithFlags(&lanes[t].st, hipStreamNonBlocking) != hipSuccess) {
                    fail("hipStreamCreate failed in a generator thread");
                    return;
                }
                fn(t);
            } catch (const std::exception &e) {
                fail(std::string("exception in a generator thread: ") + e.what());
            } catch (...) {
                fail("exception in a generator thread");
            }
        };
        if (lanes.size() == 1) {
            body(0);
        } else {
            std::vector<std::thread> th;
            for (size_t t = 0; t < lanes.size(); t++)
                th.emplace_back(body, (int)t);
            for (auto &x : th)
                x.join();
        }
        return !failed;
    }
};
#define HMX_LANE_HIP(LN, call)                                                                                          \
    do {                                                                                                                \
        const hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                         \
            (LN).fail(std::string(#call) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
            return;                                                                                                     \
        }                                                                                                               \
    } while (0)
// entries of one shared-out piece of a phase (~50 us of a simple generator)
constexpr int64_t CB_CHUNK_ENTRIES = 16384;

static int callback_thread_count(const HMat &H) {
    int n = H.callback_threads; // hmx_hmatrix_set_callback_threads: an explicit count wins
    if (n <= 0)
        n = H.opt.i(HMX_OPT_CALLBACK_THREADS) > 0 ? H.opt.i(HMX_OPT_CALLBACK_THREADS) : std::min(64, host_cores());
    return std::max(1, std::min(n, 256));
}

// Blocks assembled by the host generator into device memory: block `blocks[k]` (M x N, column-major, HMatrix::compute_dense_data's
// layout, hmatrix/hmatrix.hpp:222-226) goes to dest + dst_off[blocks[k]]; the offsets must be the running total of the block sizes in
// the order of `blocks`, so that whatever a slot evaluated in one go is one contiguous copy.  Large blocks are cut into panels of whole
// columns (a panel of a column-major block is contiguous).
static int cb_fill_blocks(HMat &H, CbLanes &LN, const std::vector<int32_t> &blocks, const std::vector<int64_t> &dst_off, scalar *dest) {
    struct Unit {
        int32_t b, c0, nc;
        int64_t dst;
    };
    int64_t max_rows = 1;
    for (int32_t b : blocks)
        max_rows = std::max<int64_t>(max_rows, H.leaves[b].t_size);
    // entries per slot fill: 8 MiB, at least one column of the tallest block
    const int64_t GROUP = std::max<int64_t>((int64_t(8) << 20) / (int64_t)sizeof(scalar), max_rows);
    std::vector<Unit> units;
    std::vector<size_t> group_first{0}; // groups of consecutive units of at most GROUP entries
    int64_t in_group = 0;
    for (int32_t b : blocks) {
        const hmx_leaf &l = H.leaves[b];
        const int64_t M = l.t_size, N = l.s_size;
        const int64_t step = M * N <= CB_CHUNK_ENTRIES ? N : std::max<int64_t>(1, CB_CHUNK_ENTRIES / M); // a unit is what one thread evaluates in one call
        for (int64_t c0 = 0; c0 < N; c0 += step) {
            const int64_t nc = std::min(step, N - c0), ent = M * nc;
            if (in_group > 0 && in_group + ent > GROUP) {
                group_first.push_back(units.size());
                in_group = 0;
            }
            units.push_back(Unit{b, (int32_t)c0, (int32_t)nc, dst_off[b] + M * c0});
            in_group += ent;
        }
    }
    group_first.push_back(units.size());
    const size_t ngroups = group_first.size() - 1;
    if (units.empty())
        return HMX_OK;
    HMX_HIP(LN.reserve((size_t)GROUP, 1));
    std::atomic<size_t> next{0};
    const bool ok = LN.run([&](int t) {
        CbLane &L = LN.lanes[t];
        for (int s = 0;; s ^= 1) {
            if (LN.failed)
                break;
            const size_t g = next.fetch_add(1);
            if (g >= ngroups)
                break;
            CbSlot &S = L.slot[s];
            HMX_LANE_HIP(LN, S.wait()); // the copy that last read this slot's pinned buffer
            const size_t u0 = group_first[g], u1 = group_first[g + 1];
            if (u0 == u1)
                continue;
            const int64_t base = units[u0].dst;
            const Unit &last   = units[u1 - 1];
            const int64_t tot  = last.dst + (int64_t)H.leaves[last.b].t_size * last.nc - base;
            // units of ~CB_CHUNK_ENTRIES entries; small leaves are bundled so that a shared-out piece is worth the hand-over
            std::vector<size_t> piece{u0};
            int64_t acc = 0;
            for (size_t u = u0; u < u1; u++) {
                acc += (int64_t)H.leaves[units[u].b].t_size * units[u].nc;
                if (acc >= CB_CHUNK_ENTRIES && u + 1 < u1) {
                    piece.push_back(u + 1);
                    acc = 0;
                }
            }
            piece.push_back(u1);
            const std::function<void(size_t)> body = [&](size_t p) {
                for (size_t u = piece[p]; u < piece[p + 1]; u++) {
                    const Unit &U     = units[u];
                    const hmx_leaf &l = H.leaves[U.b];
                    H.callback(H.callback_user, l.t_size, U.nc, H.perm_t.data() + l.t_offset, H.perm_s.data() + l.s_offset + U.c0, S.h_buf + (U.dst - base));
                }
            };
            LN.parallel(piece.size() - 1, body);
            HMX_LANE_HIP(LN, hipMemcpyAsync(dest + base, S.h_buf, (size_t)tot * sizeof(scalar), hipMemcpyHostToDevice, L.st));
            HMX_LANE_HIP(LN, hipEventRecord(S.done, L.st));
            S.pending = true;
        }
        for (auto &S : L.slot)
            HMX_LANE_HIP(LN, S.wait());
    });
    if (!ok) {
        set_error("hmx_hmatrix_compress (host generator): " + LN.error);
        return HMX_ERR_HIP;
    }
    return HMX_OK;
}

static int api_compress_impl(HMat *Hp, int compressor, double epsilon, int reqrank, bool full_pool);
int api_compress(HMat *Hp, int compressor, double epsilon, int reqrank) {
    // The cross pool is first sized from a rank estimate (allocations beyond a few tens of GB take seconds on this platform:
    // tools/malloc_timing.hip) and GROWS when blocks run out of it: the ACA variants suspend / park such blocks and continue them, fullACA
    // and SVD compress them again -- nothing else is repeated (until round 4 those two and the host-generator ACA repeated the whole build).
    return api_compress_impl(Hp, compressor, epsilon, reqrank, false);
}
static int api_compress_impl(HMat *Hp, int compressor, double epsilon, int reqrank, bool full_pool) {
    if (!Hp) {
        set_error("hmx_hmatrix_compress: NULL handle");
        return HMX_ERR_INVALID;
    }
    HMat &H = *Hp;
    if (!H.has_kernel && !H.callback) {
        set_error("hmx_hmatrix_compress: no generator set (hmx_hmatrix_set_kernel or hmx_hmatrix_set_callback)");
        return HMX_ERR_STATE;
    }
    // HMX_BUILD_TIMING=1: wall-clock of the build phases on stderr (tools/build_timing.py)
    const bool phase_timing = H.opt.i(HMX_OPT_BUILD_TIMING) != 0;
    auto phase_t0           = std::chrono::steady_clock::now();
    auto phase              = [&](const char *name) {
        if (!phase_timing)
            return;
        (void)hipDeviceSynchronize();
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[hmx build] %-28s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(t - phase_t0).count());
        phase_t0 = t;
    };
    const bool use_cb = H.callback != nullptr && !H.has_kernel;
    // evaluate one sub-block through the host generator: rows/cols are cluster positions, mapped to user numbers
    auto gen = [&](int M, int N, int row_pos, int col_pos, scalar *out) {
        H.callback(H.callback_user, M, N, H.perm_t.data() + row_pos, H.perm_s.data() + col_pos, out);
    };
    if (compressor < HMX_PARTIAL_ACA || compressor > HMX_SVD) {
        set_error("hmx_hmatrix_compress: unknown compressor");
        return HMX_ERR_INVALID;
    }
    const bool assembled = (compressor == HMX_FULL_ACA || compressor == HMX_SVD); // works on the assembled block
    if (reqrank == 0)
        reqrank = -1;
    H.build_epsilon = epsilon;
    HMX_HIP(hipSetDevice(H.device));
    const size_t nb = H.leaves.size();
    // ---- scratch for the admissible leaves ---------------------------------------------------------
    std::vector<int32_t> order;
    H.colptr.assign(nb, 0);
    std::vector<int32_t> colcap(nb, 0);
    std::vector<int64_t> visptr(nb, 0);
    int64_t ncross = 0, nvis = 0;
    double need = 0, estimate = 0;
    // slots per block = the largest rank the reference itself accepts, q (M + N) <= M N (partialACA.hpp:84): no other cap, so a block fails
    // into a dense leaf exactly when the reference's does (the slot table costs 8 bytes per possible cross: ~2 GB at N = 1e6)
    constexpr int64_t RANK_CAP = INT32_MAX;
    // expected rank of an admissible block: grows like log(1/eps) for the asymptotically smooth kernels H-matrices are for
    const bool rank_guess_given = H.opt.d(HMX_OPT_POOL_RANK_GUESS) > 0;
    const double rank_guess     = rank_guess_given ? H.opt.d(HMX_OPT_POOL_RANK_GUESS)
                              : (reqrank > 0 ? (double)reqrank : std::max(16.0, 8.0 + 3.0 * std::log10(1.0 / std::max(epsilon, 1e-16))));
    for (size_t b = 0; b < nb; b++) {
        const hmx_leaf &l = H.leaves[b];
        if (!l.admissible) {
            H.kind[b] = LK_DENSE_GEN;
            continue;
        }
        order.push_back((int32_t)b);
        const int64_t M = l.t_size, N = l.s_size;
        int64_t qmax = (M * N) / (M + N);
        if (reqrank > 0)
            qmax = compressor == HMX_SVD ? std::min<int64_t>(reqrank, std::min(M, N)) // SVD.hpp:64-92: no advantage test
                                         : std::min<int64_t>(qmax, std::min<int64_t>(reqrank, std::min(M, N)));
        qmax        = std::max<int64_t>(1, std::min<int64_t>(qmax, RANK_CAP));
        H.colptr[b] = ncross;
        colcap[b]   = (int32_t)qmax;
        ncross += qmax;
        visptr[b] = nvis;
        nvis += M + N;
        need += (double)qmax * (double)(M + N);
        estimate += std::min((double)qmax, rank_guess) * (double)(M + N);
    }
    { // largest blocks (rows + columns) first, leaf order inside a size: `order` is in leaf order and the sizes take a few dozen distinct
      // values (two per level of the trees), so one counting pass per distinct size class replaces the comparison sort
        std::map<int64_t, int64_t, std::greater<int64_t>> count;
        std::vector<int64_t> size_of(order.size());
        for (size_t k = 0; k < order.size(); k++)
            size_of[k] = (int64_t)H.leaves[order[k]].t_size + H.leaves[order[k]].s_size;
        if (order.size() > 0) {
            // consecutive leaves mostly share their size: the map is only consulted where a run ends
            int64_t last = -1;
            int64_t *slot = nullptr;
            for (size_t k = 0; k < order.size(); k++) {
                if (size_of[k] != last) {
                    last = size_of[k];
                    slot = &count[last];
                }
                ++*slot;
            }
            int64_t run = 0;
            for (auto &kv : count) {
                const int64_t c = kv.second;
                kv.second       = run;
                run += c;
            }
            std::vector<int32_t> sorted(order.size());
            last = -1;
            for (size_t k = 0; k < order.size(); k++) {
                if (size_of[k] != last) {
                    last = size_of[k];
                    slot = &count[last];
                }
                sorted[(*slot)++] = order[k];
            }
            order.swap(sorted);
        }
    }
    size_t free_b = 0, total_b = 0;
    HMX_HIP(hmx_mem_info(&free_b, &total_b));
    size_t largest_b = 0;
    HMX_HIP(hmx_mem_largest(&largest_b)); // the pool is ONE array: it must fit the driver's free memory or one hole of a reserved slab
    const double budget        = std::min(0.40 * (double)free_b, 0.95 * (double)largest_b) / sizeof(scalar);
    // (the host-generator ACA parks the blocks that find the pool exhausted and continues them after a growth step, so it starts from half
    // the pessimistic estimate: 16.6 instead of 41.5 GB at N = 1e6, where 14.2 GB are used)
    unsigned long long cap = (unsigned long long)std::max(1024.0, std::min(std::min(need, budget), full_pool ? need : (use_cb && !assembled ? 0.5 : 1.25) * estimate));
    phase("host scratch tables");
    DArr<unsigned long long> head;
    HMX_HIP(head.alloc(1));
    HMX_HIP(head.zero());
    DArr<unsigned char> visited;
    HMX_HIP(visited.alloc(std::max<int64_t>(nvis, 1)));
    HMX_HIP(visited.zero());
    DArr<int64_t> d_visptr;
    DArr<int32_t> d_order, d_colcap;
    HMX_HIP(d_visptr.upload(visptr));
    HMX_HIP(d_order.upload(order));
    HMX_HIP(d_colcap.upload(colcap));
    HMX_HIP(H.d_colptr.upload(H.colptr));
    HMX_HIP(H.d_cross_off.alloc(std::max<int64_t>(ncross, 1)));
    HMX_HIP(H.d_rank.alloc(std::max<size_t>(nb, 1)));
    HMX_HIP(H.d_rank.zero());
    HMX_HIP(H.d_swapped.alloc(std::max<size_t>(nb, 1)));
    HMX_HIP(H.d_swapped.zero());
    H.staged_off.assign(nb, -1);
    HMX_HIP(H.d_staged_off.upload(H.staged_off));

    DArr<int32_t> st_q, st_I1, st_I2; // state of suspended blocks (aca_kernel)
    DArr<real> st_frob, st_aux;
    if (!assembled && !use_cb) {
        HMX_HIP(st_q.alloc(std::max<size_t>(nb, 1)));
        HMX_HIP(st_I1.alloc(std::max<size_t>(nb, 1)));
        HMX_HIP(st_I2.alloc(std::max<size_t>(nb, 1)));
        HMX_HIP(st_frob.alloc(std::max<size_t>(nb, 1)));
        HMX_HIP(st_aux.alloc(std::max<size_t>(nb, 1)));
        for (auto *a : {&st_q, &st_I1, &st_I2})
            HMX_HIP(a->zero());
        for (auto *a : {&st_frob, &st_aux})
            HMX_HIP(a->zero());
    }
    phase("scratch upload");
    auto aca_args = [&](scalar *pool, unsigned long long pool_cap, const int32_t *order_dev) {
        AcaArgs A{};
        A.ks = H.ks;
        A.tx = H.tx.d; A.ty = H.ty.d; A.tz = H.tz.d;
        A.sx = H.sx.d; A.sy = H.sy.d; A.sz = H.sz.d;
        A.order  = order_dev;
        A.t_off  = H.d_t_off.d; A.t_size = H.d_t_size.d; A.s_off = H.d_s_off.d; A.s_size = H.d_s_size.d;
        A.symmetric_pivoting = compressor == HMX_SYMPARTIAL_ACA;
        A.epsilon   = epsilon;
        A.reqrank   = reqrank;
        A.pool      = pool;
        A.pool_head = head.d;
        A.pool_cap  = pool_cap;
        A.colptr    = H.d_colptr.d;
        A.colcap    = d_colcap.d;
        A.cross_off = H.d_cross_off.d;
        A.visited   = visited.d;
        A.vis_ptr   = d_visptr.d;
        A.rank_out  = H.d_rank.d;
        A.swapped_out = H.d_swapped.d;
        A.st_q = st_q.d; A.st_I1 = st_I1.d; A.st_I2 = st_I2.d; A.st_frob = st_frob.d; A.st_aux = st_aux.d;
        return A;
    };
    // blocks with both sides <= wave_max points are compressed by one wave each (aca_wave_kernel), the others by one workgroup each
    const int wave_max = (!assembled && !use_cb) ? std::min(H.opt.i(HMX_OPT_ACA_WAVE_MAX), 64 * ACA_WAVE_KR) : 0;
    // Pool sizing from a SAMPLE of the blocks.  The a-priori rank guess has to be pessimistic (it decides whether the
    // compression must be repeated) and is 3-4 times the ranks smooth kernels really give; large allocations cost
    // seconds on some boxes (tools/malloc_timing.hip) and the pool competes with the streams for HBM.  So every K-th
    // block of the size-sorted list (<= ~4000 blocks) is compressed first into a small pool, and the full pool is sized
    // at 1.3 x (measured / guessed) of the estimate.  The pool grows if the sample misled (grow_pool).  HMX_POOL_SAMPLE=0: off.
    if (!assembled && !use_cb && !full_pool && order.size() >= 20000 && (double)cap * sizeof(scalar) >= 4e9 && reqrank < 0 &&
        H.opt.i(HMX_OPT_POOL_SAMPLE) != 0 && !rank_guess_given) {
        const size_t K = std::max<size_t>(1, order.size() / 4096);
        std::vector<int32_t> sample;
        double guess_s = 0;
        for (size_t i = 0; i < order.size(); i += K) {
            const int32_t b = order[i];
            sample.push_back(b);
            guess_s += std::min((double)colcap[b], rank_guess) * (double)(H.leaves[b].t_size + H.leaves[b].s_size);
        }
        DArr<int32_t> d_sample;
        DArr<scalar> sample_pool;
        const unsigned long long scap = (unsigned long long)(1.25 * guess_s) + 1024;
        if (d_sample.upload(sample) == hipSuccess && sample_pool.alloc(scap) == hipSuccess) {
            AcaArgs S = aca_args(sample_pool.d, scap, d_sample.d);
            hipLaunchKernelGGL(aca_kernel<256>, dim3((unsigned)sample.size()), dim3(256), 0, 0, S);
            HMX_HIP(hipGetLastError());
            std::vector<int32_t> r(nb, 0);
            HMX_HIP(hipMemcpy(r.data(), H.d_rank.d, nb * 4, hipMemcpyDeviceToHost));
            double used_s = 0;
            bool overflow = false;
            for (int32_t b : sample) {
                overflow = overflow || r[b] == -2;
                int crosses = std::max(r[b], 1); // failed blocks still take one cross
                if (std::max(H.leaves[b].t_size, H.leaves[b].s_size) <= wave_max) // aca_wave_kernel takes its pool space ACA_WAVE_CHUNK crosses at a time
                    crosses = (crosses + ACA_WAVE_CHUNK - 1) / ACA_WAVE_CHUNK * ACA_WAVE_CHUNK;
                used_s += (double)crosses * (double)(H.leaves[b].t_size + H.leaves[b].s_size);
            }
            if (!overflow && guess_s > 0) {
                const double ratio = std::min(1.0, 1.3 * used_s / guess_s + 0.02);
                cap                = (unsigned long long)std::max(1024.0, std::min((double)cap, ratio * estimate)); // below the 1.5 x that triggers the shrink copy
            }
            // leave no trace of the sample run
            HMX_HIP(head.zero());
            HMX_HIP(visited.zero());
            HMX_HIP(H.d_rank.zero());
            HMX_HIP(H.d_swapped.zero());
            for (auto *a : {&st_q, &st_I1, &st_I2})
                HMX_HIP(a->zero());
            for (auto *a : {&st_frob, &st_aux})
                HMX_HIP(a->zero());
        } else {
            (void)hipGetLastError();
        }
        phase("pool sizing sample");
    }
    HMX_HIP(H.pool.alloc(cap));
    phase("pool allocation");
    // a zero row pivot and every growth round leave a grant unused: some slack over the exact need
    const unsigned long long maxcap = (unsigned long long)std::max(1024.0, std::min(need + (2.0 + ACA_WAVE_CHUNK) * (double)nvis + 64.0 * 1048576.0, budget));
    // grow_pool(): doubled (the ACA variants: their suspended / parked blocks CONTINUE, nothing granted so far is lost).  grow_pool(extra): room for
    // `extra` more elements beyond what is granted -- fullACA / SVD compress a block that ran out AGAIN from scratch, its first grants are lost, so
    // doubling rounds would spend the pool on abandoned crosses; with the failed blocks' full need added they all finish in the next round.
    auto grow_pool = [&](unsigned long long extra = 0) -> int { // HMX_OK: grown; 1: the budget is used up
        const unsigned long long limit = extra ? (unsigned long long)std::min((double)cap + (double)extra + 1024.0, budget) : maxcap;
        if (cap >= limit)
            return 1;
        const unsigned long long newcap = extra ? limit : std::min<unsigned long long>(maxcap, std::max<unsigned long long>(2 * cap, cap + 1024));
        DArr<scalar> bigger;
        if (bigger.alloc(newcap) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        HMX_HIP(hipMemcpy(bigger.d, H.pool.d, (size_t)cap * sizeof(scalar), hipMemcpyDeviceToDevice));
        std::swap(bigger.d, H.pool.d);
        std::swap(bigger.n, H.pool.n);
        std::swap(bigger.cap_, H.pool.cap_);
        std::swap(bigger.dev_, H.pool.dev_);
        bigger.release();
        const unsigned long long old = cap; // the grants that failed pushed the head beyond the old capacity: restart it there
        HMX_HIP(hipMemcpy(head.d, &old, 8, hipMemcpyHostToDevice));
        cap = newcap;
        return HMX_OK;
    };
    // the host generator's threads (lanes: stream + two pinned / device slot pairs each), shared by the ACA and the assembly of dense blocks
    std::unique_ptr<CbLanes> cb_lanes;
    if (use_cb)
        cb_lanes.reset(new CbLanes(H.device, callback_thread_count(H), std::max(1, H.opt.i(HMX_OPT_CALLBACK_DRIVERS))));
    const auto wall0 = std::chrono::steady_clock::now();
    DEvent e0, e1;
    HMX_HIP(hipEventRecord(e0, 0));
    if (!order.empty() && assembled) {
        // fullACA / SVD need the whole block: process the admissible leaves in batches that fit a scratch slab
        std::vector<int64_t> need_elems(nb, 0);
        int64_t largest = 0;
        for (int32_t b : order) {
            const int64_t M = H.leaves[b].t_size, N = H.leaves[b].s_size, m = std::max(M, N), n = std::min(M, N);
            if (M * N >= (int64_t(1) << 31)) {
                set_error("hmx_hmatrix_compress: fullACA/SVD need M*N < 2^31 per block (use a minimal block depth, as the reference must)");
                return HMX_ERR_UNSUPPORTED;
            }
            need_elems[b] = compressor == HMX_FULL_ACA ? M * N : m * n + n * n + 2 * n;
            largest       = std::max(largest, need_elems[b]);
        }
        size_t free2 = 0, total2 = 0;
        HMX_HIP(hmx_mem_info(&free2, &total2));
        {
            size_t one = 0;
            HMX_HIP(hmx_mem_largest(&one));
            free2 = std::min(free2, (size_t)(1.9 * (double)one)); // the scratch slab (0.5 * free2 below) is one array
        }
        int64_t total_need = 0;
        for (int32_t b : order)
            total_need += need_elems[b];
        const int64_t slab = std::max<int64_t>(largest, std::min<int64_t>(total_need, (int64_t)(0.5 * (double)free2 / sizeof(scalar))));
        if ((double)largest * sizeof(scalar) > 0.9 * (double)free2) {
            set_error("hmx_hmatrix_compress: an admissible block does not fit in HBM for fullACA/SVD");
            return HMX_ERR_HIP;
        }
        DArr<scalar> scratch;
        HMX_HIP(scratch.alloc(slab));
        std::vector<int64_t> soff(nb, 0);
        DArr<int64_t> d_soff;
        // Rounds: all blocks first; the blocks that found the pool exhausted (rank -2) are compressed again -- they only -- after the pool has
        // grown (a block of these compressors is assembled and compressed from scratch in one go, so "again" costs that block, not the build)
        std::vector<int32_t> todo = order;
        DArr<int32_t> d_todo;
        for (int round = 0;; round++) {
        const int32_t *todo_dev = d_order.d;
        if (round > 0) {
            HMX_HIP(d_todo.upload(todo));
            todo_dev = d_todo.d;
        }
        size_t pos = 0;
        while (pos < todo.size()) {
            int64_t used = 0;
            size_t end   = pos;
            while (end < todo.size() && used + need_elems[todo[end]] <= slab) {
                soff[todo[end]] = used;
                used += need_elems[todo[end]];
                end++;
            }
            HMX_HIP(d_soff.upload(soff));
            DenseCompressArgs D{};
            DArr<scalar> pre;
            DArr<int64_t> d_preoff;
            if (use_cb) { // the host generator assembles the blocks of this batch
                std::vector<int64_t> preoff(nb, 0);
                int64_t tot = 0;
                for (size_t k = pos; k < end; k++) {
                    preoff[todo[k]] = tot;
                    tot += (int64_t)H.leaves[todo[k]].t_size * H.leaves[todo[k]].s_size;
                }
                HMX_HIP(pre.alloc(std::max<int64_t>(tot, 1)));
                HMX_HIP(hipDeviceSynchronize());
                const int rcf = cb_fill_blocks(H, *cb_lanes, std::vector<int32_t>(todo.begin() + pos, todo.begin() + end), preoff, pre.d);
                if (rcf != HMX_OK)
                    return rcf;
                HMX_HIP(d_preoff.upload(preoff));
                D.pre     = pre.d;
                D.pre_off = d_preoff.d;
            }
            D.ks = H.ks;
            D.tx = H.tx.d; D.ty = H.ty.d; D.tz = H.tz.d;
            D.sx = H.sx.d; D.sy = H.sy.d; D.sz = H.sz.d;
            D.order = todo_dev + pos;
            D.t_off = H.d_t_off.d; D.t_size = H.d_t_size.d; D.s_off = H.d_s_off.d; D.s_size = H.d_s_size.d;
            D.scratch_off = d_soff.d;
            D.scratch     = scratch.d;
            D.epsilon     = epsilon;
            D.reqrank     = reqrank;
            D.pool        = H.pool.d;
            D.pool_head   = head.d;
            D.pool_cap    = cap;
            D.colptr      = H.d_colptr.d;
            D.colcap      = d_colcap.d;
            D.cross_off   = H.d_cross_off.d;
            D.rank_out    = H.d_rank.d;
            if (compressor == HMX_FULL_ACA)
                hipLaunchKernelGGL(fullaca_kernel<256>, dim3((unsigned)(end - pos)), dim3(256), 0, 0, D);
            else
                hipLaunchKernelGGL(svd_kernel<256>, dim3((unsigned)(end - pos)), dim3(256), 0, 0, D);
            HMX_HIP(hipGetLastError());
            HMX_HIP(hipDeviceSynchronize());
            pos = end;
        }
        std::vector<int32_t> rr(nb, 0);
        HMX_HIP(hipMemcpy(rr.data(), H.d_rank.d, nb * 4, hipMemcpyDeviceToHost));
        std::vector<int32_t> failed;
        for (int32_t b : todo)
            if (rr[b] == -2)
                failed.push_back(b);
        if (failed.empty())
            break;
        if (phase_timing)
            fprintf(stderr, "[hmx build]   round %d: %zu of %zu blocks found the pool of %.2f GB exhausted\n", round, failed.size(), todo.size(), (double)cap * sizeof(scalar) / 1e9);
        unsigned long long extra = 0; // everything the failed blocks can ask for
        for (int32_t b : failed)
            extra += (unsigned long long)(colcap[b] + 1) * (unsigned long long)(H.leaves[b].t_size + H.leaves[b].s_size); // (+1: a grant may precede the "not advantageous" test)
        const int rcg = grow_pool(std::max<unsigned long long>(extra, 1));
        if (rcg == 1)
            break; // reported below as an exhausted pool
        if (rcg != HMX_OK)
            return rcg;
        todo.swap(failed);
        }
    } else if (!order.empty() && use_cb) {
        // lock-step ACA: the generator runs on the host (on all cores: "Host generator on all cores" above), everything else on the
        // device (aca_cb_*_kernel).  A block that finds the pool exhausted is parked with its row pivot; when the lanes have drained, the
        // pool grows and the parked blocks continue with that row -- nothing is computed twice.
        DArr<int32_t> dI1, dI2, dq;
        DArr<real> dfrob, daux;
        DArr<scalar> dgamma;
        DArr<unsigned long long> dcur;
        for (auto *a : {&dI1, &dI2, &dq}) {
            HMX_HIP(a->alloc(nb));
            HMX_HIP(a->zero());
        }
        for (auto *a : {&dfrob, &daux}) {
            HMX_HIP(a->alloc(nb));
            HMX_HIP(a->zero());
        }
        HMX_HIP(dgamma.alloc(nb));
        HMX_HIP(dgamma.zero());
        HMX_HIP(dcur.alloc(nb));
        HMX_HIP(hipDeviceSynchronize()); // the lanes' streams do not wait for the null stream
        const bool sympiv = compressor == HMX_SYMPARTIAL_ACA;
        // entries of the longer side of a block: what one phase of the iteration evaluates at most
        auto line_len = [&](int32_t b) { return (int64_t)std::max(H.leaves[b].t_size, H.leaves[b].s_size); };
        // batch size: small enough that ~4 batches per slot exist (the tail of a lane is one batch), large enough that a phase is worth
        // a launch; at most CB_BATCH_BLOCKS blocks and CB_BATCH_ENTRIES entries per phase
        constexpr size_t CB_BATCH_BLOCKS   = 16384;
        const int64_t CB_BATCH_ENTRIES     = (int64_t(16) << 20) / (int64_t)sizeof(scalar);
        struct Todo {
            int32_t b, I1;
        };
        std::vector<Todo> todo;
        todo.reserve(order.size());
        for (int32_t b : order)
            todo.push_back(Todo{b, 0});
        std::vector<Todo> parked;
        std::mutex parked_mu;
        for (int round = 0;; round++) {
            int64_t total_entries = 0;
            for (const Todo &t : todo)
                total_entries += line_len(t.b);
            int64_t longest = 1;
            for (const Todo &t : todo)
                longest = std::max(longest, line_len(t.b));
            const int64_t per_batch = std::max<int64_t>(1, std::min<int64_t>(CB_BATCH_ENTRIES, total_entries / (4 * (int64_t)cb_lanes->nslots()) + 1));
            HMX_HIP(cb_lanes->reserve((size_t)std::max(per_batch, longest), CB_BATCH_BLOCKS));
            std::vector<size_t> batch_first{0};
            {
                int64_t ent = 0;
                size_t cnt  = 0;
                for (size_t k = 0; k < todo.size(); k++) {
                    const int64_t e = line_len(todo[k].b);
                    if (cnt > 0 && (ent + e > per_batch || cnt >= CB_BATCH_BLOCKS)) {
                        batch_first.push_back(k);
                        ent = 0;
                        cnt = 0;
                    }
                    ent += e;
                    cnt++;
                }
                batch_first.push_back(todo.size());
            }
            const size_t nbatches = batch_first.size() - 1;
            std::atomic<size_t> next_batch{0};
            AcaCbArgs A0{};
            A0.t_off = H.d_t_off.d; A0.t_size = H.d_t_size.d; A0.s_off = H.d_s_off.d; A0.s_size = H.d_s_size.d;
            A0.symmetric_pivoting = sympiv;
            A0.epsilon = epsilon; A0.reqrank = reqrank;
            A0.pool = H.pool.d; A0.pool_head = head.d; A0.pool_cap = cap;
            A0.colptr = H.d_colptr.d; A0.colcap = d_colcap.d; A0.cross_off = H.d_cross_off.d;
            A0.visited = visited.d; A0.vis_ptr = d_visptr.d;
            A0.I1 = dI1.d; A0.I2 = dI2.d; A0.q = dq.d;
            A0.frob = dfrob.d; A0.aux = daux.d; A0.gamma = dgamma.d; A0.cur_off = dcur.d;
            A0.rank_out = H.d_rank.d; A0.swapped_out = H.d_swapped.d;
            CbLanes &LN = *cb_lanes;
            std::atomic<long long> ns_gen{0}, ns_wait{0}, ns_enqueue{0}, n_phases{0}, n_entries{0};
            auto now_ns = [] { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            const bool ok = LN.run([&](int t) {
                CbLane &L = LN.lanes[t];
                long long l_gen = 0, l_wait = 0, l_enq = 0, l_ph = 0, l_ent = 0;
                struct Flush {
                    std::function<void()> f;
                    ~Flush() { f(); }
                } flush{[&] { ns_gen += l_gen; ns_wait += l_wait; ns_enqueue += l_enq; n_phases += l_ph; n_entries += l_ent; }};
                for (auto &S : L.slot) {
                    S.active.clear();
                    S.pending = false;
                }
                for (;;) {
                    bool any = false;
                    for (auto &S : L.slot) {
                        if (LN.failed)
                            return;
                        if (S.pending) { // digest the phase that was in flight
                            const long long tw = now_ns();
                            HMX_LANE_HIP(LN, S.wait());
                            l_wait += now_ns() - tw;
                            size_t w = 0;
                            for (size_t i = 0; i < S.active.size(); i++) {
                                const CbResult r = S.h_res[i];
                                if (r.status == CB_ACTIVE) {
                                    S.active[w] = S.active[i];
                                    S.I1[w]     = r.I1;
                                    S.I2[w]     = r.I2;
                                    w++;
                                } else if (r.status == CB_SUSPENDED) {
                                    std::lock_guard<std::mutex> lock(parked_mu);
                                    parked.push_back(Todo{S.active[i], S.I1[i]});
                                }
                            }
                            S.active.resize(w);
                            S.I1.resize(w);
                            S.I2.resize(w);
                            S.row_phase = !S.row_phase;
                        }
                        if (S.active.empty()) { // next batch
                            const size_t k = next_batch.fetch_add(1);
                            if (k >= nbatches)
                                continue;
                            const size_t k0 = batch_first[k], k1 = batch_first[k + 1];
                            S.active.resize(k1 - k0);
                            S.I1.resize(k1 - k0);
                            S.I2.assign(k1 - k0, 0);
                            for (size_t i = k0; i < k1; i++) {
                                S.active[i - k0] = todo[i].b;
                                S.I1[i - k0]     = todo[i].I1;
                            }
                            S.row_phase = true;
                        }
                        // evaluate this phase's lines (shared out in chunks to all generator threads), then upload + kernel + result copy on
                        // the lane's stream
                        const long long tg = now_ns();
                        int64_t tot = 0, in_chunk = 0;
                        S.chunk_first.assign(1, 0);
                        for (size_t i = 0; i < S.active.size(); i++) {
                            const int32_t b   = S.active[i];
                            const hmx_leaf &l = H.leaves[b];
                            const bool sw     = sympiv && !(l.t_offset >= l.s_offset);
                            const int64_t len = (S.row_phase != sw) ? l.s_size : l.t_size; // row phase: index 2 runs over the source side unless swapped
                            S.h_items[i]      = CbItem{tot, b, 0};
                            tot += len;
                            in_chunk += len;
                            if (in_chunk >= CB_CHUNK_ENTRIES && i + 1 < S.active.size()) {
                                S.chunk_first.push_back(i + 1);
                                in_chunk = 0;
                            }
                        }
                        S.chunk_first.push_back(S.active.size());
                        const std::function<void(size_t)> body = [&](size_t c) {
                            for (size_t i = S.chunk_first[c]; i < S.chunk_first[c + 1]; i++) {
                                const hmx_leaf &l = H.leaves[S.active[i]];
                                const bool sw     = sympiv && !(l.t_offset >= l.s_offset);
                                scalar *out       = S.h_buf + S.h_items[i].off;
                                if (S.row_phase) { // entries (I1, k), k over index 2
                                    if (!sw)
                                        gen(1, l.s_size, l.t_offset + S.I1[i], l.s_offset, out);
                                    else
                                        gen(l.t_size, 1, l.t_offset, l.s_offset + S.I1[i], out);
                                } else { // entries (k, I2), k over index 1
                                    if (!sw)
                                        gen(l.t_size, 1, l.t_offset, l.s_offset + S.I2[i], out);
                                    else
                                        gen(1, l.s_size, l.t_offset + S.I2[i], l.s_offset, out);
                                }
                            }
                        };
                        LN.parallel(S.chunk_first.size() - 1, body);
                        const size_t na = S.active.size();
                        const long long tq = now_ns();
                        l_gen += tq - tg;
                        l_ph++;
                        l_ent += tot;
                        HMX_LANE_HIP(LN, hipMemcpyAsync(S.d_buf, S.h_buf, (size_t)tot * sizeof(scalar), hipMemcpyHostToDevice, L.st));
                        HMX_LANE_HIP(LN, hipMemcpyAsync(S.d_items, S.h_items, na * sizeof(CbItem), hipMemcpyHostToDevice, L.st));
                        AcaCbArgs A = A0;
                        A.items     = S.d_items;
                        A.res       = S.d_res;
                        A.buf       = S.d_buf;
                        if (S.row_phase)
                            hipLaunchKernelGGL(aca_cb_row_kernel<256>, dim3((unsigned)na), dim3(256), 0, L.st, A);
                        else
                            hipLaunchKernelGGL(aca_cb_col_kernel<256>, dim3((unsigned)na), dim3(256), 0, L.st, A);
                        HMX_LANE_HIP(LN, hipGetLastError());
                        HMX_LANE_HIP(LN, hipMemcpyAsync(S.h_res, S.d_res, na * sizeof(CbResult), hipMemcpyDeviceToHost, L.st));
                        HMX_LANE_HIP(LN, hipEventRecord(S.done, L.st));
                        l_enq += now_ns() - tq;
                        S.pending = true;
                        any       = true;
                    }
                    if (!any)
                        break;
                }
            });
            if (!ok) {
                set_error("hmx_hmatrix_compress (host generator): " + LN.error);
                return HMX_ERR_HIP;
            }
            if (phase_timing)
                fprintf(stderr, "[hmx build]   round %d: %zu blocks in %zu batches, %zu drivers + %d workers, %zu parked at a pool of %.2f GB; %lld phases, %.3e entries; "
                                "driver-seconds: evaluation %.2f, waiting for the device %.2f, enqueue %.2f\n", round, todo.size(), nbatches,
                        LN.lanes.size(), LN.nworkers, parked.size(), (double)cap * sizeof(scalar) / 1e9, (long long)n_phases, (double)n_entries, ns_gen * 1e-9, ns_wait * 1e-9, ns_enqueue * 1e-9);
            if (parked.empty())
                break;
            const int rcg = grow_pool();
            if (rcg == 1)
                break; // reported below as an exhausted pool
            if (rcg != HMX_OK)
                return rcg;
            // (largest first again: the order the batches are cut in)
            std::sort(parked.begin(), parked.end(), [&](const Todo &a, const Todo &b) {
                const int64_t sa = (int64_t)H.leaves[a.b].t_size + H.leaves[a.b].s_size, sb = (int64_t)H.leaves[b.b].t_size + H.leaves[b.b].s_size;
                return sa != sb ? sa > sb : a.b < b.b;
            });
            todo.swap(parked);
            parked.clear();
        }
    } else if (!order.empty()) {
        // Rounds: all blocks first; a block that finds the rank-estimated pool exhausted suspends with its state (aca_kernel), the pool
        // grows (new allocation + device copy of the crosses written so far) and the suspended blocks continue where they stopped --
        // nothing is computed twice, the blocks that had finished keep their crosses.
        // Large blocks whose rank keeps growing leave the one-workgroup kernel after team_q iterations and continue with several workgroups
        // each (aca_team_*_kernel, three launches per iteration over all such blocks).
        std::vector<int32_t> active     = order; // `order` is sorted by n1 + n2, largest first; so is every later list
        DArr<int32_t> d_active;
        std::vector<int32_t> round_ranks(nb, 0);
        DArr<int32_t> t_status, t_need;
        DArr<scalar> t_gamma;
        DArr<unsigned long long> t_off;
        DArr<unsigned int> t_counter;
        DArr<real> t_paux;
        auto since_phase = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - phase_t0).count(); };
        // entries of a line per workgroup: 1024 while the launch has workgroups enough to fill the GPU, 256 when few blocks are left (a
        // workgroup walks the whole history whatever its share, 16 loads in flight per thread either way: N=1e6 Hermitian case, team phase
        // of the second round 1.63 s with 1024 throughout, 1.53 s with 256 throughout -- but the first round 0.92 instead of 0.75 s)
        const int team_slice_env = H.opt.i(HMX_OPT_ACA_TEAM_SLICE) > 0 ? std::max(64, H.opt.i(HMX_OPT_ACA_TEAM_SLICE)) : 0;
        auto run_team = [&](const std::vector<int32_t> &blocks, int round) -> int {
            if (!t_status.d) {
                HMX_HIP(t_status.alloc(nb));
                HMX_HIP(t_need.alloc(nb));
                HMX_HIP(t_gamma.alloc(nb));
                HMX_HIP(t_off.alloc(nb));
                HMX_HIP(t_counter.alloc(nb));
                HMX_HIP(t_paux.alloc(nb));
                HMX_HIP(t_need.zero());
                HMX_HIP(t_counter.zero());
            }
            HMX_HIP(t_status.zero()); // every block in `blocks` is active (again); the entries of other blocks are not looked at
            std::vector<int32_t> cur = blocks, st(nb);
            int launches = 0;
            double t_wait = 0, t_copy = 0;
            // Tables for at most 64 workgroups per block, allocated once: hipFree waits for the whole device, and the side stream is busy with
            // the small blocks meanwhile.  (Dealing the teams out to 2 / 4 / 8 streams so that launches overlap was measured on the N=1e6
            // Hermitian case: 1.65 / 2.3 / 2.8 s for the team phase against 1.65 s on one stream -- the launches themselves become the cost.)
            DArr<int32_t> d_block, d_G, d_wg0, d_wgteam, d_pidx;
            DArr<real> d_pval;
            DArr<scalar> d_pfrob;
            HMX_HIP(d_block.alloc(cur.size()));
            HMX_HIP(d_G.alloc(cur.size()));
            HMX_HIP(d_wg0.alloc(cur.size()));
            HMX_HIP(d_wgteam.alloc(64 * cur.size()));
            HMX_HIP(d_pidx.alloc(64 * cur.size()));
            HMX_HIP(d_pval.alloc(64 * cur.size()));
            HMX_HIP(d_pfrob.alloc(64 * cur.size()));
            while (!cur.empty()) {
                int team_slice = team_slice_env;
                if (team_slice == 0) { // (the kernels have a one-entry and a four-entry path per thread: shares of 257-512 entries would idle half of the latter)
                    int64_t wgs = 0;
                    for (int32_t b : cur)
                        wgs += (std::max(H.leaves[b].t_size, H.leaves[b].s_size) + 1023) / 1024;
                    team_slice = wgs >= 1536 ? 1024 : 256;
                }
                std::vector<int32_t> team_G(cur.size()), team_wg0(cur.size()), wg_team;
                for (size_t t = 0; t < cur.size(); t++) {
                    const hmx_leaf &l = H.leaves[cur[t]];
                    team_G[t]         = std::max(1, std::min(64, (std::max(l.t_size, l.s_size) + team_slice - 1) / team_slice));
                    team_wg0[t]       = (int32_t)wg_team.size();
                    wg_team.insert(wg_team.end(), (size_t)team_G[t], (int32_t)t);
                }
                HMX_HIP(hipMemcpy(d_block.d, cur.data(), cur.size() * 4, hipMemcpyHostToDevice));
                HMX_HIP(hipMemcpy(d_G.d, team_G.data(), cur.size() * 4, hipMemcpyHostToDevice));
                HMX_HIP(hipMemcpy(d_wg0.d, team_wg0.data(), cur.size() * 4, hipMemcpyHostToDevice));
                HMX_HIP(hipMemcpy(d_wgteam.d, wg_team.data(), wg_team.size() * 4, hipMemcpyHostToDevice));
                AcaTeamArgs T{};
                T.A = aca_args(H.pool.d, cap, d_order.d);
                T.wg_team = d_wgteam.d; T.team_block = d_block.d; T.team_wg0 = d_wg0.d; T.team_G = d_G.d;
                T.status = t_status.d; T.need_dots = t_need.d; T.gamma = t_gamma.d; T.off = t_off.d; T.counter = t_counter.d;
                T.pval = d_pval.d; T.pidx = d_pidx.d; T.pfrob = d_pfrob.d; T.paux = t_paux.d;
                const dim3 grid((unsigned)wg_team.size()), wg(256);
                for (;;) {
                    for (int it = 0; it < 16; it++) {
                        hipLaunchKernelGGL(aca_team_control_kernel<256>, grid, wg, 0, 0, T);
                        hipLaunchKernelGGL(aca_team_row_kernel<256>, grid, wg, 0, 0, T);
                        hipLaunchKernelGGL(aca_team_col_kernel<256>, grid, wg, 0, 0, T);
                        launches += 3;
                    }
                    HMX_HIP(hipGetLastError());
                    const auto tq0 = std::chrono::steady_clock::now();
                    HMX_HIP(hipStreamSynchronize(0));
                    const auto tq1 = std::chrono::steady_clock::now();
                    HMX_HIP(hipMemcpy(st.data(), t_status.d, nb * 4, hipMemcpyDeviceToHost));
                    const auto tq2 = std::chrono::steady_clock::now();
                    t_wait += std::chrono::duration<double, std::milli>(tq1 - tq0).count();
                    t_copy += std::chrono::duration<double, std::milli>(tq2 - tq1).count();
                    std::vector<int32_t> still;
                    for (int32_t b : cur)
                        if (st[b] == 0)
                            still.push_back(b);
                    if (still.size() * 2 <= cur.size()) { // fewer, smaller launches for the blocks that go on
                        cur.swap(still);
                        break;
                    }
                }
            }
            if (phase_timing)
                fprintf(stderr, "[hmx build]   round %d (%.0f ms): %zu blocks continued by workgroup teams, %d launches (host waited %.0f ms for the kernels, %.0f ms for status copies)\n", round, since_phase(), blocks.size(), launches, t_wait, t_copy);
            return HMX_OK;
        };
        const bool team_ok = reqrank < 0 && H.opt.i(HMX_OPT_ACA_TEAMS) != 0;
        const int team_min = team_ok ? H.opt.i(HMX_OPT_ACA_TEAM_MIN) : 0;
        const int team_q   = H.opt.i(HMX_OPT_ACA_TEAM_AFTER);
        hipStream_t side = nullptr; // the blocks below team_min, concurrently with the large ones and their teams
        struct SideGuard {
            hipStream_t &s;
            ~SideGuard() {
                if (s)
                    (void)hipStreamDestroy(s);
            }
        } side_guard{side};
        HMX_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        // ... and one stream per size class of the one-wave kernel: the classes take 4-8 ms each when alone on the GPU, behind one another on ONE
        // stream the last two only started when the first -- sharing the GPU with the workgroup kernels -- had finished (profiles/r5_aca_trace.log)
        hipStream_t wave_stream[3] = {nullptr, nullptr, nullptr};
        SideGuard wave_guard0{wave_stream[0]}, wave_guard1{wave_stream[1]}, wave_guard2{wave_stream[2]};
        for (auto &ws : wave_stream)
            HMX_HIP(hipStreamCreateWithFlags(&ws, hipStreamNonBlocking));
#ifdef HMX_ACA_SERIAL // measurement only: every compression kernel on the null stream, one after the other (tools/aca_trace.sh)
        (void)hipStreamDestroy(side);
        side = nullptr;
        for (auto &ws : wave_stream) {
            (void)hipStreamDestroy(ws);
            ws = nullptr;
        }
#endif
        DArr<int32_t> d_medium, d_small;
        for (int round = 0;; round++) {
            AcaArgs A  = aca_args(H.pool.d, cap, round == 0 ? d_order.d : d_active.d);
            A.team_min = team_min;
            A.team_q   = team_q;
            // `active` is sorted by n1 + n2, largest first: the blocks that may hand over to teams are a prefix.  They run on the null
            // stream and their teams follow at once; everything smaller runs on the side stream meanwhile (one workgroup per block,
            // dominated by its few high-rank blocks: the two overlap well).
            size_t nbig = 0;
            while (team_min > 0 && nbig < active.size() && (int64_t)H.leaves[active[nbig]].t_size + H.leaves[active[nbig]].s_size >= team_min)
                nbig++;
            // ... and of those, the blocks with both sides <= wave_max points go to aca_wave_kernel (one wave per block) on a stream of their own
            // (three size classes, largest first: 4, 2 or 1 entries of a line per lane)
            std::vector<int32_t> medium, small, small_class[3];
            for (size_t i = nbig; i < active.size(); i++) {
                const int side_max = std::max(H.leaves[active[i]].t_size, H.leaves[active[i]].s_size);
                if (side_max <= wave_max)
                    small_class[side_max <= 64 ? 2 : (side_max <= 128 ? 1 : 0)].push_back(active[i]);
                else
                    medium.push_back(active[i]);
            }
            for (const auto &c : small_class)
                small.insert(small.end(), c.begin(), c.end());
            if (!small.empty()) {
                HMX_HIP(d_medium.upload(medium));
                HMX_HIP(d_small.upload(small));
            }
            HMX_HIP(hipDeviceSynchronize()); // uploads, pool growth and state resets on the null stream, before the side streams read them
            if (nbig > 0) // the longest launch (few blocks, the highest ranks) first
                hipLaunchKernelGGL(aca_kernel<256>, dim3((unsigned)nbig), dim3(256), 0, 0, A);
            if (!small.empty()) {
                constexpr int WV = 4;
                AcaArgs W = A;
                W.order   = d_small.d;
                auto grid = [](size_t n) { return dim3((unsigned)((n + WV - 1) / WV)); };
                if (!small_class[0].empty())
                    hipLaunchKernelGGL((aca_wave_kernel<WV, 4>), grid(small_class[0].size()), dim3(WV * 64), 0, wave_stream[0], W, (int)small_class[0].size());
                W.order += small_class[0].size();
                if (!small_class[1].empty())
                    hipLaunchKernelGGL((aca_wave_kernel<WV, 2>), grid(small_class[1].size()), dim3(WV * 64), 0, wave_stream[1], W, (int)small_class[1].size());
                W.order += small_class[1].size();
                if (!small_class[2].empty())
                    hipLaunchKernelGGL((aca_wave_kernel<WV, 1>), grid(small_class[2].size()), dim3(WV * 64), 0, wave_stream[2], W, (int)small_class[2].size());
                if (!medium.empty()) {
                    AcaArgs S = A;
                    S.order   = d_medium.d;
                    hipLaunchKernelGGL(aca_kernel<256>, dim3((unsigned)medium.size()), dim3(256), 0, side, S);
                }
            } else if (active.size() > nbig) {
                AcaArgs S = A;
                S.order += nbig;
                hipLaunchKernelGGL(aca_kernel<256>, dim3((unsigned)(active.size() - nbig)), dim3(256), 0, side, S);
            }
            if (nbig > 0) {
                HMX_HIP(hipGetLastError());
                HMX_HIP(hipMemcpy(round_ranks.data(), H.d_rank.d, nb * 4, hipMemcpyDeviceToHost));
                std::vector<int32_t> handed;
                for (size_t i = 0; i < nbig; i++)
                    if (round_ranks[active[i]] == -3)
                        handed.push_back(active[i]);
                if (phase_timing)
                    fprintf(stderr, "[hmx build]   round %d (%.0f ms): first %d iterations of the %zu blocks of %d rows + columns or more, pool %.2f GB\n", round, since_phase(), team_q, nbig,
                            team_min, (double)cap * sizeof(scalar) / 1e9);
                if (!handed.empty()) {
                    const int rct = run_team(handed, round);
                    if (rct != HMX_OK)
                        return rct;
                }
            }
            HMX_HIP(hipStreamSynchronize(side));
            for (auto &ws : wave_stream)
                HMX_HIP(hipStreamSynchronize(ws));
            HMX_HIP(hipGetLastError());
            HMX_HIP(hipMemcpy(round_ranks.data(), H.d_rank.d, nb * 4, hipMemcpyDeviceToHost));
            if (phase_timing)
                fprintf(stderr, "[hmx build]   round %d (%.0f ms): one-workgroup kernel over %zu blocks, one-wave kernel over %zu blocks done\n", round, since_phase(), medium.size(), small.size());
            std::vector<int32_t> suspended;
            for (int32_t b : active)
                if (round_ranks[b] == -2)
                    suspended.push_back(b);
            if (suspended.empty())
                break;
            if (phase_timing)
                fprintf(stderr, "[hmx build]   round %d: %zu of %zu blocks suspended at a pool of %.2f GB\n", round, suspended.size(), active.size(), (double)cap * sizeof(scalar) / 1e9);
            const int rcg = grow_pool();
            if (rcg == 1)
                break; // reported below as an exhausted pool
            if (rcg != HMX_OK)
                return rcg;
            active.swap(suspended);
            HMX_HIP(d_active.upload(active));
        }
    }
    HMX_HIP(hipEventRecord(e1, 0));
    HMX_HIP(hipEventSynchronize(e1));
    float ms = 0;
    HMX_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (use_cb) // the lanes run on their own streams: wall time of the compression (generator included)
        ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    std::vector<int32_t> ranks(nb, 0);
    H.swapped.assign(nb, 0);
    if (nb) {
        HMX_HIP(hipMemcpy(ranks.data(), H.d_rank.d, nb * 4, hipMemcpyDeviceToHost));
        HMX_HIP(hipMemcpy(H.swapped.data(), H.d_swapped.d, nb * 4, hipMemcpyDeviceToHost));
    }
    HMX_HIP(hipMemcpy(&H.pool_used, head.d, 8, hipMemcpyDeviceToHost));
    phase("compression kernels");
    int64_t false_pos = 0;
    for (int32_t b : order) {
        if (ranks[b] == -2) {
            set_error("hmx_hmatrix_compress: compression pool exhausted (not enough free HBM)");
            return HMX_ERR_HIP;
        }
        if (ranks[b] > 0) {
            H.kind[b]        = LK_LOWRANK;
            H.leaves[b].rank = ranks[b];
        } else { // compressor failed -> dense block (tree_builder.hpp:572-577)
            H.kind[b]        = LK_DENSE_GEN;
            H.leaves[b].rank = -1;
            false_pos++;
        }
    }
    for (size_t b = 0; b < nb; b++)
        if (H.kind[b] != LK_LOWRANK)
            H.leaves[b].rank = -1;
    if ((double)H.pool.n > 1.5 * (double)H.pool_used + 1024) { // give the unused part of the pool back
        DArr<scalar> exact;
        if (exact.alloc(std::max<size_t>((size_t)H.pool_used, 1)) == hipSuccess) {
            HMX_HIP(hipMemcpy(exact.d, H.pool.d, (size_t)H.pool_used * sizeof(scalar), hipMemcpyDeviceToDevice));
            std::swap(exact.d, H.pool.d);
            std::swap(exact.n, H.pool.n);
            std::swap(exact.cap_, H.pool.cap_);
        } else {
            (void)hipGetLastError();
        }
    }
    phase("pool shrink");
    if (use_cb) { // dense leaves (and failed admissible ones): HMatrix::compute_dense_data through the host generator, on all its threads
        int64_t tot = 0;
        std::vector<int32_t> dense_blocks;
        for (size_t b = 0; b < nb; b++)
            if (H.kind[b] != LK_LOWRANK) {
                H.staged_off[b] = tot;
                tot += (int64_t)H.leaves[b].t_size * H.leaves[b].s_size;
                dense_blocks.push_back((int32_t)b);
                H.kind[b] = LK_DENSE_STAGED;
            }
        HMX_HIP(H.dense_stage.alloc(std::max<int64_t>(tot, 1)));
        HMX_HIP(hipDeviceSynchronize()); // the lanes' streams do not wait for the null stream
        const int rcf = cb_fill_blocks(H, *cb_lanes, dense_blocks, H.staged_off, H.dense_stage.d);
        if (rcf != HMX_OK)
            return rcf;
        cb_lanes.reset();
        phase("dense blocks (host generator)");
    } else {
        H.dense_stage.release();
    }
    int rc = build_streams(H);
    phase("stream layout + packing");
    if (rc != HMX_OK)
        return rc;
    H.stats.n_false_positive = false_pos;
    H.stats.t_compress_s     = ms * 1e-3;
    return HMX_OK;
}

// recompression(hmatrix) (hmatrix/utils/recompression.hpp:8-31): SVD recompression of every low-rank leaf with the
// accuracy the operator was built with (LowRankMatrix::get_epsilon), then the streams are laid out again.
int api_recompress(HMat *Hp, double epsilon) {
    if (!Hp) {
        set_error("hmx_hmatrix_recompress: NULL handle");
        return HMX_ERR_INVALID;
    }
    HMat &H = *Hp;
    if (!H.finalized || H.pool.n == 0 || H.factors_released) {
        set_error("hmx_hmatrix_recompress: operator not built, or its factors were released");
        return HMX_ERR_STATE;
    }
    HMX_HIP(hipSetDevice(H.device));
    if (epsilon <= 0)
        epsilon = H.build_epsilon;
    const size_t nb = H.leaves.size();
    std::vector<int32_t> order, ranks(nb, 0);
    std::vector<int64_t> need(nb, 0);
    int64_t largest = 0;
    for (size_t b = 0; b < nb; b++) {
        ranks[b] = H.leaves[b].rank;
        if (H.kind[b] != LK_LOWRANK || H.leaves[b].rank <= 0)
            continue;
        const int64_t M = H.leaves[b].t_size, N = H.leaves[b].s_size, r = H.leaves[b].rank;
        need[b] = (M + N) * r + 4 * r * r + 4 * r;
        largest = std::max(largest, need[b]);
        order.push_back((int32_t)b);
    }
    if (order.empty())
        return HMX_OK;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return need[a] != need[b] ? need[a] > need[b] : a < b; });
    size_t free_b = 0, total_b = 0;
    HMX_HIP(hmx_mem_info(&free_b, &total_b));
    {
        size_t one = 0;
        HMX_HIP(hmx_mem_largest(&one));
        free_b = std::min(free_b, (size_t)(1.9 * (double)one)); // the scratch slab (0.5 * free_b below) is one array
    }
    int64_t total_need = 0;
    for (int32_t b : order)
        total_need += need[b];
    // never more scratch than all blocks together need: giant allocations take seconds (tools/malloc_timing.hip)
    const int64_t slab = std::max<int64_t>(largest, std::min<int64_t>(total_need, (int64_t)(0.5 * (double)free_b / sizeof(scalar))));
    if ((double)largest * sizeof(scalar) > 0.9 * (double)free_b) {
        set_error("hmx_hmatrix_recompress: a block does not fit in HBM scratch");
        return HMX_ERR_HIP;
    }
    DArr<scalar> scratch;
    HMX_HIP(scratch.alloc(slab));
    DArr<int32_t> d_order, d_ts, d_ss, d_sw;
    DArr<int64_t> d_soff, d_colptr;
    std::vector<int32_t> ts(nb), ss(nb), sw(nb, 0);
    for (size_t b = 0; b < nb; b++) {
        ts[b] = H.leaves[b].t_size;
        ss[b] = H.leaves[b].s_size;
        sw[b] = b < H.swapped.size() ? H.swapped[b] : 0;
    }
    std::vector<int64_t> colptr = H.colptr;
    colptr.resize(nb, 0);
    HMX_HIP(d_order.upload(order));
    HMX_HIP(d_ts.upload(ts));
    HMX_HIP(d_ss.upload(ss));
    HMX_HIP(d_sw.upload(sw));
    HMX_HIP(d_colptr.upload(colptr));
    HMX_HIP(H.d_rank.upload(ranks));
    std::vector<int64_t> soff(nb, 0);
    size_t pos = 0;
    while (pos < order.size()) {
        int64_t used = 0;
        size_t end   = pos;
        while (end < order.size() && used + need[order[end]] <= slab) {
            soff[order[end]] = used;
            used += need[order[end]];
            end++;
        }
        HMX_HIP(d_soff.upload(soff));
        RecompressArgs A{d_order.d + pos, d_ts.d, d_ss.d, d_sw.d, d_soff.d, scratch.d, epsilon, H.pool.d, d_colptr.d, H.d_cross_off.d, H.d_rank.d};
        hipLaunchKernelGGL(recompress_kernel<256>, dim3((unsigned)(end - pos)), dim3(256), 0, 0, A);
        HMX_HIP(hipGetLastError());
        HMX_HIP(hipDeviceSynchronize());
        pos = end;
    }
    HMX_HIP(hipMemcpy(ranks.data(), H.d_rank.d, nb * 4, hipMemcpyDeviceToHost));
    for (int32_t b : order)
        H.leaves[b].rank = ranks[b];
    const hmx_stats keep = H.stats;
    const int rc         = build_streams(H);
    H.stats.n_false_positive = keep.n_false_positive;
    H.stats.t_compress_s     = keep.t_compress_s;
    return rc;
}

static void ensure_staged(HMat &H) {
    const size_t nb = H.leaves.size();
    if (H.staged_U.size() != nb) {
        H.staged_U.resize(nb);
        H.staged_V.resize(nb);
        H.staged_D.resize(nb);
    }
}
int api_set_block_lowrank(HMat *H, int64_t leaf, int rank, const scalar *U, const scalar *V) {
    if (!H || leaf < 0 || leaf >= (int64_t)H->leaves.size() || rank < 0 || (rank > 0 && (!U || !V))) {
        set_error("hmx_hmatrix_set_block_lowrank: invalid arguments");
        return HMX_ERR_INVALID;
    }
    const hmx_leaf &l = H->leaves[leaf];
    ensure_staged(*H);
    H->staged_U[leaf].assign(U, U + (size_t)l.t_size * rank);
    // V arrives r x N column-major; keep it k-major (row k contiguous) like a cross
    H->staged_V[leaf].resize((size_t)l.s_size * rank);
    for (int k = 0; k < rank; k++)
        for (int j = 0; j < l.s_size; j++)
            H->staged_V[leaf][(size_t)k * l.s_size + j] = V[k + (size_t)rank * j];
    H->staged_D[leaf].clear();
    H->kind[leaf]        = LK_LOWRANK;
    H->leaves[leaf].rank = rank;
    H->finalized         = false;
    return HMX_OK;
}
int api_set_block_dense(HMat *H, int64_t leaf, const scalar *D) {
    if (!H || leaf < 0 || leaf >= (int64_t)H->leaves.size() || !D) {
        set_error("hmx_hmatrix_set_block_dense: invalid arguments");
        return HMX_ERR_INVALID;
    }
    const hmx_leaf &l = H->leaves[leaf];
    ensure_staged(*H);
    H->staged_D[leaf].assign(D, D + (size_t)l.t_size * l.s_size);
    H->staged_U[leaf].clear();
    H->staged_V[leaf].clear();
    H->kind[leaf]        = LK_DENSE_STAGED;
    H->leaves[leaf].rank = -1;
    H->finalized         = false;
    return HMX_OK;
}
int api_finalize(HMat *Hp) {
    if (!Hp)
        return HMX_ERR_INVALID;
    HMat &H = *Hp;
    HMX_HIP(hipSetDevice(H.device));
    const size_t nb = H.leaves.size();
    // every leaf needs a payload
    int64_t total = 0, ncross = 0;
    ensure_staged(H);
    for (size_t b = 0; b < nb; b++) {
        if (H.kind[b] == LK_PENDING || H.kind[b] == LK_DENSE_GEN) {
            if (H.has_kernel && !H.leaves[b].admissible) { // dense leaves may be left to the device generator
                H.kind[b] = LK_DENSE_GEN;
            } else {
                set_error("hmx_hmatrix_finalize: leaf " + std::to_string(b) + " has no uploaded payload");
                return HMX_ERR_STATE;
            }
        }
        total += (int64_t)H.staged_U[b].size() + H.staged_V[b].size() + H.staged_D[b].size();
        if (H.kind[b] == LK_LOWRANK)
            ncross += H.leaves[b].rank;
    }
    std::vector<scalar> host(std::max<int64_t>(total, 1));
    std::vector<int64_t> cross(std::max<int64_t>(ncross, 1)), staged(nb, -1);
    H.colptr.assign(nb, 0);
    H.swapped.assign(nb, 0);
    int64_t pos = 0, cpos = 0;
    for (size_t b = 0; b < nb; b++) {
        const hmx_leaf &l = H.leaves[b];
        if (H.kind[b] == LK_LOWRANK) {
            H.colptr[b] = cpos;
            for (int k = 0; k < l.rank; k++) { // cross k = [U(:,k) | V(k,:)]
                cross[cpos++] = pos;
                std::copy_n(H.staged_U[b].data() + (size_t)k * l.t_size, l.t_size, host.data() + pos);
                pos += l.t_size;
                std::copy_n(H.staged_V[b].data() + (size_t)k * l.s_size, l.s_size, host.data() + pos);
                pos += l.s_size;
            }
        } else if (H.kind[b] == LK_DENSE_STAGED) {
            staged[b] = pos;
            std::copy(H.staged_D[b].begin(), H.staged_D[b].end(), host.begin() + pos);
            pos += (int64_t)H.staged_D[b].size();
        }
    }
    HMX_HIP(H.pool.upload(host));
    H.pool_used = (unsigned long long)pos;
    HMX_HIP(H.d_cross_off.upload(cross));
    HMX_HIP(H.d_colptr.upload(H.colptr));
    HMX_HIP(H.d_swapped.upload(H.swapped));
    HMX_HIP(H.d_staged_off.upload(staged));
    H.staged_off = staged;
    if (!H.has_kernel) { // pack_dense never evaluates the generator on this path, but needs valid pointers
        H.ks = KernelSpec{0, 3, 0, 0, 1, 0, 0, 0};
    }
    return build_streams(H);
}

int api_leaf_ranks(const HMat *H, int32_t *rank) {
    if (!H || !rank)
        return HMX_ERR_INVALID;
    for (size_t b = 0; b < H->leaves.size(); b++)
        rank[b] = H->leaves[b].rank;
    return HMX_OK;
}

int api_get_block(const HMat *Hc, int64_t leaf, scalar *U_or_D, scalar *V) {
    HMat *H = const_cast<HMat *>(Hc);
    if (!H || leaf < 0 || leaf >= (int64_t)H->leaves.size() || !U_or_D) {
        set_error("hmx_hmatrix_get_block: invalid arguments");
        return HMX_ERR_INVALID;
    }
    if (!H->finalized) {
        set_error("hmx_hmatrix_get_block: operator not built");
        return HMX_ERR_STATE;
    }
    HMX_HIP(hipSetDevice(H->device));
    const hmx_leaf &l = H->leaves[leaf];
    const int M = l.t_size, N = l.s_size;
    if (H->kind[leaf] == LK_LOWRANK) {
        if (!V)
            return HMX_ERR_INVALID;
        if (H->factors_released) {
            set_error("hmx_hmatrix_get_block: the low-rank factors were released (hmx_hmatrix_release_factors)");
            return HMX_ERR_STATE;
        }
        const int r  = l.rank;
        const bool sw = H->swapped[leaf] != 0;
        const int n1 = sw ? N : M, n2 = sw ? M : N;
        std::vector<int64_t> cross(std::max(r, 1));
        HMX_HIP(hipMemcpy(cross.data(), H->d_cross_off.d + H->colptr[leaf], (size_t)r * 8, hipMemcpyDeviceToHost));
        std::vector<scalar> buf((size_t)n1 + n2);
        for (int k = 0; k < r; k++) {
            HMX_HIP(hipMemcpy(buf.data(), H->pool.d + cross[k], buf.size() * sizeof(scalar), hipMemcpyDeviceToHost));
            const scalar *ucol = sw ? buf.data() + n1 : buf.data();
            const scalar *vrow = sw ? buf.data() : buf.data() + n1;
            std::copy_n(ucol, M, U_or_D + (size_t)k * M);
            for (int j = 0; j < N; j++)
                V[k + (size_t)r * j] = vrow[j];
        }
        return HMX_OK;
    }
    // dense: gather the slices back out of the E-streams
    const StreamSet &E = H->E;
    int r0 = (int)(std::lower_bound(E.off.begin(), E.off.end(), l.t_offset - H->T0) - E.off.begin());
    for (int r = r0; r < E.nranges() && E.off[r] < l.t_offset - H->T0 + M; r++) {
        // find this block's first column in range r: scan the z index of the range for its x position
        const int64_t cb = E.colbase[r];
        int col          = -1;
        for (int c = 0; c < E.cols[r]; c++)
            if (H->h_e_zidx[cb + c] == l.s_offset - H->S0) { // a dense column (index below nS) starting at this block's first source point
                col = c;
                break;
            }
        if (col < 0) {
            set_error("hmx_hmatrix_get_block: internal lookup failed");
            return HMX_ERR_STATE;
        }
        const int len = E.len[r], rel = E.off[r] - (l.t_offset - H->T0);
        std::vector<scalar> buf((size_t)len * N);
        HMX_HIP(hipMemcpy(buf.data(), E.stream.d + E.base[r] + (int64_t)col * len, buf.size() * sizeof(scalar), hipMemcpyDeviceToHost));
        for (int j = 0; j < N; j++)
            for (int i = 0; i < len; i++)
                U_or_D[(size_t)(rel + i) + (size_t)M * j] = buf[(size_t)j * len + i];
    }
    return HMX_OK;
}

// Bulk download: `count` blocks in a few large device-to-host copies instead of one blocking copy per cross / per slice (what a loop over
// hmx_hmatrix_get_block costs: 468 754 leaves at N = 1e6).  The blocks are gathered on the device into a staging array in htool's own
// layouts (get_lr_blocks_kernel / get_dense_blocks_kernel), the staging array crosses PCIe into pinned memory in pieces of 256 MiB, and
// the host threads copy every block to the caller's pointer while the next piece is in flight.  V[k] may be NULL for dense leaves.
int api_get_blocks(const HMat *Hc, int64_t count, const int64_t *leaves, scalar *const *U_or_D, scalar *const *V) {
    HMat *H = const_cast<HMat *>(Hc);
    if (!H || count < 0 || (count > 0 && (!leaves || !U_or_D))) {
        set_error("hmx_hmatrix_get_blocks: invalid arguments");
        return HMX_ERR_INVALID;
    }
    if (!H->finalized) {
        set_error("hmx_hmatrix_get_blocks: operator not built");
        return HMX_ERR_STATE;
    }
    if (count == 0)
        return HMX_OK;
    HMX_HIP(hipSetDevice(H->device));
    std::vector<GetItem> items((size_t)count);
    std::vector<int64_t> entries((size_t)count);
    int64_t largest = 1;
    bool any_lr     = false;
    for (int64_t k = 0; k < count; k++) {
        const int64_t b = leaves[k];
        if (b < 0 || b >= (int64_t)H->leaves.size() || !U_or_D[k]) {
            set_error("hmx_hmatrix_get_blocks: leaf index out of range or NULL destination");
            return HMX_ERR_INVALID;
        }
        const hmx_leaf &l = H->leaves[b];
        const bool lr     = H->kind[b] == LK_LOWRANK;
        if (lr && (!V || !V[k])) {
            set_error("hmx_hmatrix_get_blocks: a low-rank leaf needs a destination for V");
            return HMX_ERR_INVALID;
        }
        any_lr     = any_lr || lr;
        items[k]   = GetItem{0, lr ? H->colptr[b] : 0, (int32_t)b, lr ? l.rank : -1, l.t_size, l.s_size, lr ? H->swapped[b] : 0, l.t_offset - H->T0};
        entries[k] = lr ? (int64_t)l.rank * ((int64_t)l.t_size + l.s_size) : (int64_t)l.t_size * l.s_size;
        largest    = std::max(largest, entries[k]);
    }
    if (any_lr && H->factors_released) {
        set_error("hmx_hmatrix_get_blocks: the low-rank factors were released (hmx_hmatrix_release_factors)");
        return HMX_ERR_STATE;
    }
    const int64_t CAP = std::max<int64_t>((int64_t(256) << 20) / (int64_t)sizeof(scalar), largest);
    struct Piece {
        scalar *h = nullptr;
        DArr<scalar> d;
        DArr<GetItem> d_items;
        DArr<int32_t> d_pi, d_pr, d_pc;
        hipEvent_t ev = nullptr;
        int64_t k0 = 0, k1 = 0;
        ~Piece() {
            if (h)
                (void)hipHostFree(h);
            if (ev)
                (void)hipEventDestroy(ev);
        }
    } piece[2];
    hipStream_t st = nullptr;
    struct StreamGuard {
        hipStream_t &s;
        ~StreamGuard() {
            if (s) {
                (void)hipStreamSynchronize(s);
                (void)hipStreamDestroy(s);
            }
        }
    } guard{st};
    HMX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int64_t total = std::accumulate(entries.begin(), entries.end(), (int64_t)0);
    const int npieces_needed = total > CAP ? 2 : 1;
    const int64_t cap_eff     = std::min(CAP, std::max<int64_t>(total, 1));
    for (int s = 0; s < npieces_needed; s++) {
        HMX_HIP(hipHostMalloc((void **)&piece[s].h, (size_t)cap_eff * sizeof(scalar), hipHostMallocDefault));
        HMX_HIP(piece[s].d.alloc((size_t)cap_eff));
        HMX_HIP(hipEventCreateWithFlags(&piece[s].ev, hipEventDisableTiming | hipEventBlockingSync));
    }
    HMX_HIP(hipDeviceSynchronize()); // whatever built or last used the operator
    auto scatter = [&](Piece &P) { // staging (pinned) -> the caller's blocks, on the host cores
        const int64_t n = P.k1 - P.k0;
        const size_t nt = (size_t)std::max<int64_t>(1, std::min<int64_t>({(int64_t)host_cores(), (int64_t)32, n}));
        std::atomic<int64_t> next{P.k0};
        auto work = [&] {
            for (;;) {
                const int64_t k = next.fetch_add(1);
                if (k >= P.k1)
                    break;
                const GetItem &it = items[k];
                const scalar *src = P.h + it.dst;
                if (it.rank >= 0) {
                    std::memcpy(U_or_D[k], src, (size_t)it.M * it.rank * sizeof(scalar));
                    std::memcpy(V[k], src + (int64_t)it.M * it.rank, (size_t)it.rank * it.N * sizeof(scalar));
                } else {
                    std::memcpy(U_or_D[k], src, (size_t)it.M * it.N * sizeof(scalar));
                }
            }
        };
        if (nt == 1) {
            work();
            return;
        }
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; t++)
            th.emplace_back(work);
        for (auto &x : th)
            x.join();
    };
    int64_t k = 0;
    int cur   = 0;
    bool have_prev = false;
    while (k < count) {
        Piece &P = piece[cur];
        // the leaves of this piece
        int64_t used = 0, k1 = k;
        std::vector<int32_t> pi, pr, pc;
        bool lr_here = false;
        while (k1 < count && used + entries[k1] <= cap_eff) {
            items[k1].dst = used;
            used += entries[k1];
            if (items[k1].rank >= 0) {
                lr_here = true;
            } else { // its slices in the E-streams
                const int32_t b = items[k1].leaf;
                auto lo = std::lower_bound(H->dp_leaf.begin(), H->dp_leaf.end(), b), hi = std::upper_bound(lo, H->dp_leaf.end(), b);
                int64_t rows = 0;
                for (auto itp = lo; itp != hi; ++itp) {
                    const size_t q = (size_t)(itp - H->dp_leaf.begin());
                    pi.push_back((int32_t)(k1 - k));
                    pr.push_back(H->dp_range[q]);
                    pc.push_back(H->dp_col[q]);
                    rows += H->E.len[H->dp_range[q]];
                }
                if (rows != items[k1].M) {
                    set_error("hmx_hmatrix_get_blocks: internal lookup failed (dense leaf not found in the streams)");
                    return HMX_ERR_STATE;
                }
            }
            k1++;
        }
        P.k0 = k, P.k1 = k1;
        HMX_HIP(P.d_items.alloc((size_t)(k1 - k)));
        HMX_HIP(hipMemcpyAsync(P.d_items.d, items.data() + k, (size_t)(k1 - k) * sizeof(GetItem), hipMemcpyHostToDevice, st));
        if (lr_here)
            hipLaunchKernelGGL(get_lr_blocks_kernel, dim3((unsigned)(k1 - k), 4), dim3(256), 0, st, (const GetItem *)P.d_items.d, (const scalar *)H->pool.d, (const int64_t *)H->d_cross_off.d, P.d.d);
        if (!pi.empty()) {
            HMX_HIP(P.d_pi.alloc(pi.size()));
            HMX_HIP(P.d_pr.alloc(pi.size()));
            HMX_HIP(P.d_pc.alloc(pi.size()));
            HMX_HIP(hipMemcpyAsync(P.d_pi.d, pi.data(), pi.size() * 4, hipMemcpyHostToDevice, st));
            HMX_HIP(hipMemcpyAsync(P.d_pr.d, pr.data(), pi.size() * 4, hipMemcpyHostToDevice, st));
            HMX_HIP(hipMemcpyAsync(P.d_pc.d, pc.data(), pi.size() * 4, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(get_dense_blocks_kernel, dim3((unsigned)pi.size()), dim3(256), 0, st, (const GetItem *)P.d_items.d, (const int32_t *)P.d_pi.d, (const int32_t *)P.d_pr.d,
                               (const int32_t *)P.d_pc.d, (const scalar *)H->E.stream.d, (const int64_t *)H->E.d_base.d, (const int32_t *)H->E.d_off.d, (const int32_t *)H->E.d_len.d, P.d.d);
        }
        HMX_HIP(hipGetLastError());
        HMX_HIP(hipMemcpyAsync(P.h, P.d.d, (size_t)used * sizeof(scalar), hipMemcpyDeviceToHost, st));
        HMX_HIP(hipEventRecord(P.ev, st));
        if (have_prev) // the previous piece is complete in pinned memory: the host threads hand it out while this one is gathered and copied
            scatter(piece[cur ^ 1]);
        HMX_HIP(hipStreamSynchronize(st)); // (the small host vectors pi / pr / pc and the item slice must outlive their copies)
        have_prev = true;
        k         = k1;
        cur ^= 1;
        if (npieces_needed == 1 && k < count) { // (cannot happen: one piece holds everything)
            set_error("hmx_hmatrix_get_blocks: internal staging error");
            return HMX_ERR_STATE;
        }
    }
    if (have_prev)
        scatter(piece[cur ^ 1]);
    return HMX_OK;
}

// ---- binary dump of the compressed operator: HmxFileHeader and the layout are described in engine_common.hpp ----------------

int api_save(const HMat *Hc, const char *path) {
    HMat *H = const_cast<HMat *>(Hc);
    if (!H || !path) {
        set_error("hmx_hmatrix_save: invalid arguments");
        return HMX_ERR_INVALID;
    }
    if (!H->finalized || H->factors_released) {
        set_error("hmx_hmatrix_save: operator not built, or its factors were released");
        return HMX_ERR_STATE;
    }
    HMX_HIP(hipSetDevice(H->device));
    FILE *f = fopen(path, "wb");
    if (!f) {
        set_error(std::string("hmx_hmatrix_save: cannot create ") + path);
        return HMX_ERR_INVALID;
    }
    HmxFileHeader hd{};
    std::memcpy(hd.magic, HMX_FILE_MAGIC, 8);
    hd.elem_size = (int32_t)sizeof(scalar);
    hd.reserved  = HMX_COMPLEX; // 1: complex coefficients (tells a complex<float> file from a double one)
    hd.nleaves   = (int64_t)H->leaves.size();
    hd.T0 = H->T0, hd.nT = H->nT, hd.S0 = H->S0, hd.nS = H->nS;
    hd.symmetry = H->symmetry_for_leaves, hd.uplo = H->uplo_for_leaves;
    hd.epsilon  = H->build_epsilon;
    bool ok     = fwrite(&hd, sizeof hd, 1, f) == 1;
    ok          = ok && (H->leaves.empty() || fwrite(H->leaves.data(), sizeof(hmx_leaf), H->leaves.size(), f) == H->leaves.size());
    // the crosses of every low-rank leaf in one transfer
    std::vector<scalar> pool(std::max<size_t>((size_t)H->pool_used, 1));
    std::vector<int64_t> cross(std::max<size_t>(H->d_cross_off.n, 1));
    if (H->pool_used)
        HMX_HIP(hipMemcpy(pool.data(), H->pool.d, (size_t)H->pool_used * sizeof(scalar), hipMemcpyDeviceToHost));
    if (H->d_cross_off.n)
        HMX_HIP(hipMemcpy(cross.data(), H->d_cross_off.d, H->d_cross_off.n * sizeof(int64_t), hipMemcpyDeviceToHost));
    std::vector<scalar> buf;
    for (size_t b = 0; ok && b < H->leaves.size(); b++) {
        const hmx_leaf &l = H->leaves[b];
        const int M = l.t_size, N = l.s_size;
        if (H->kind[b] == LK_LOWRANK) {
            const int r   = l.rank;
            const bool sw = H->swapped[b] != 0;
            const int n1  = sw ? N : M;
            buf.assign((size_t)r * (M + N), scalar(0));
            scalar *U = buf.data(), *V = buf.data() + (size_t)r * M;
            for (int k = 0; k < r; k++) {
                const scalar *c    = pool.data() + cross[H->colptr[b] + k];
                const scalar *ucol = sw ? c + n1 : c, *vrow = sw ? c : c + n1;
                std::copy_n(ucol, M, U + (size_t)k * M);
                for (int j = 0; j < N; j++)
                    V[k + (size_t)r * j] = vrow[j];
            }
        } else {
            buf.assign((size_t)M * N, scalar(0));
            const int rc = api_get_block(H, (int64_t)b, buf.data(), nullptr);
            if (rc != HMX_OK) {
                fclose(f);
                return rc;
            }
        }
        ok = buf.empty() || fwrite(buf.data(), sizeof(scalar), buf.size(), f) == buf.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        set_error(std::string("hmx_hmatrix_save: write to ") + path + " failed");
        return HMX_ERR_INVALID;
    }
    return HMX_OK;
}

// `f` is positioned just behind the header (engine.hip reads it to pick the precision)
int api_load(const hmx_block_tree *bt, int device_id, FILE *f, const HmxFileHeader &hd, HMat **out) {
    HMat *H = nullptr;
    int rc  = api_create(bt, device_id, &H);
    if (rc != HMX_OK)
        return rc;
    auto fail = [&](const std::string &why) {
        set_error("hmx_hmatrix_load: " + why);
        delete H;
        return HMX_ERR_INVALID;
    };
    if (hd.nleaves != (int64_t)H->leaves.size() || hd.T0 != H->T0 || hd.nT != H->nT || hd.S0 != H->S0 || hd.nS != H->nS)
        return fail("the file was written for a different block tree");
    std::vector<hmx_leaf> fl((size_t)hd.nleaves);
    if (hd.nleaves && fread(fl.data(), sizeof(hmx_leaf), fl.size(), f) != fl.size())
        return fail("truncated file");
    std::vector<scalar> buf;
    for (size_t b = 0; b < fl.size(); b++) {
        const hmx_leaf &a = fl[b], &l = H->leaves[b];
        if (a.t_offset != l.t_offset || a.t_size != l.t_size || a.s_offset != l.s_offset || a.s_size != l.s_size || a.mirror != l.mirror)
            return fail("leaf " + std::to_string(b) + " does not match the block tree");
        if (a.rank < -1 || a.rank > std::min(a.t_size, a.s_size)) // a corrupt rank would size the buffers below
            return fail("leaf " + std::to_string(b) + " has an impossible rank");
        const size_t count = a.rank >= 0 ? (size_t)a.rank * (a.t_size + a.s_size) : (size_t)a.t_size * a.s_size;
        buf.resize(std::max<size_t>(count, 1));
        if (count && fread(buf.data(), sizeof(scalar), count, f) != count)
            return fail("truncated file");
        rc = a.rank >= 0 ? api_set_block_lowrank(H, (int64_t)b, a.rank, buf.data(), buf.data() + (size_t)a.rank * a.t_size) : api_set_block_dense(H, (int64_t)b, buf.data());
        if (rc != HMX_OK) {
            delete H;
            return rc;
        }
    }
    H->build_epsilon = hd.epsilon;
    rc               = api_finalize(H);
    if (rc != HMX_OK) {
        delete H;
        return rc;
    }
    *out = H;
    return HMX_OK;
}

// Give the compression pool (the ACA crosses / uploaded blocks the streams were packed from) back: products only need the
// streams.  Afterwards low-rank blocks can no longer be downloaded, saved or recompressed, and no second layout can be built any more:
// transposed products run on the stored data (a row-restricted symmetric operator needs its transposed view: bit 0 of with_transposed
// builds it first), multi-RHS products of symmetric operators on the stored triangle (bit 1 builds the expanded view first).
int api_release_factors(HMat *Hp, int with_transposed) {
    if (!Hp || !Hp->finalized) {
        set_error("hmx_hmatrix_release_factors: operator not built");
        return HMX_ERR_STATE;
    }
    HMat &H = *Hp;
    HMX_HIP(hipSetDevice(H.device));
    if (with_transposed & 1)
        (void)ensure_transposed_operator(H);
    if (with_transposed & 2) // the expanded view multi-RHS products on compact symmetric storage run on
        (void)ensure_expanded_view(H);
    if (H.dense_stage.d) // host-generated dense leaves live only in the streams from now on
        H.dense_stage.release();
    H.pool.release();
    H.d_cross_off.release();
    H.pool_used        = 0;
    H.factors_released = true;
    DeviceCache::get().trim();
    return HMX_OK;
}

int api_stats(const HMat *H, hmx_stats *out) {
    if (!H || !out)
        return HMX_ERR_INVALID;
    *out                  = H->stats;
    out->transposed_bytes = H->T_op ? H->T_op->stats.stream_bytes : 0;
    if (H->trans_fused) // the tables of the transposed product on the stored data
        out->transposed_bytes += (int64_t)((H->s_mdst.n + H->s_coef.n + H->s_count.n + H->sc_dst.n + H->sc_lp.n + H->sc_count.n + H->sc_k.n + H->s_list.n + H->s_fidx.n + H->s_sub_task.n +
                                            H->s_sub_row0.n + H->s_sub_nrows.n + H->s_sub_dst.n + H->s_int_order.n) * sizeof(int32_t) + H->s_sub_ptr.n * sizeof(int64_t) + H->SW.n * sizeof(scalar));
    out->expanded_bytes   = H->X_op ? H->X_op->stats.stream_bytes : 0;
    return HMX_OK;
}

static int with_buffers(HMat &H, char trans, const scalar *in, scalar *out, int mu, int mem, hipStream_t st, scalar beta,
                        const scalar **din, scalar **dout, bool &staged) {
    const size_t nin = (size_t)(trans == 'N' ? H.nS : H.nT) * mu, nout = (size_t)(trans == 'N' ? H.nT : H.nS) * mu;
    staged = (mem == HMX_MEM_HOST);
    if (!staged) {
        *din  = in;
        *dout = out;
        return HMX_OK;
    }
    if (H.tmp_in.n < nin)
        HMX_HIP(H.tmp_in.alloc(nin));
    if (H.tmp_out.n < nout)
        HMX_HIP(H.tmp_out.alloc(nout));
    HMX_HIP(hipMemcpyAsync(H.tmp_in.d, in, nin * sizeof(scalar), hipMemcpyHostToDevice, st));
    if (!hmx_is_zero(beta))
        HMX_HIP(hipMemcpyAsync(H.tmp_out.d, out, nout * sizeof(scalar), hipMemcpyHostToDevice, st));
    *din  = H.tmp_in.d;
    *dout = H.tmp_out.d;
    return HMX_OK;
}

int api_matvec(HMat *Hp, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out, int mem, void *stream) {
    if (!Hp || !in || !out) {
        set_error("hmx_hmatrix_matvec: NULL argument");
        return HMX_ERR_INVALID;
    }
    HMat &H = *Hp;
    HMX_HIP(hipSetDevice(H.device));
    hipStream_t st = (hipStream_t)stream;
    const scalar *din;
    scalar *dout;
    bool staged;
    int rc = with_buffers(H, trans, in, out, 1, mem, st, beta, &din, &dout, staged);
    if (rc != HMX_OK)
        return rc;
    rc = matvec_device(H, trans, alpha, din, beta, dout, st);
    if (rc != HMX_OK)
        return rc;
    if (staged) {
        const size_t nout = (size_t)(trans == 'N' ? H.nT : H.nS);
        HMX_HIP(hipMemcpyAsync(out, dout, nout * sizeof(scalar), hipMemcpyDeviceToHost, st));
        HMX_HIP(hipStreamSynchronize(st));
    }
    return HMX_OK;
}

// trans = 'N' product on device pointers with the expand stage in `nchunks` row chunks; after_chunk(user, c, row_lo, row_hi) is called on
// the host right after chunk c was LAUNCHED on `stream`: rows [row_lo, row_hi) of `out` are final once the stream reaches that point.
// Returns the number of chunks used through *used (1: the operator could not be chunked -- fused symmetric storage adds to rows after
// the expand stage -- and after_chunk was called once, for all rows, after the whole product).
int api_matvec_chunked(HMat *Hp, scalar alpha, const scalar *in, scalar beta, scalar *out, void *stream, int nchunks, after_chunk_fn after_chunk, void *user, int *used) {
    if (!Hp || !in || !out) {
        set_error("hmx_hmatrix_matvec (chunked): NULL argument");
        return HMX_ERR_INVALID;
    }
    HMat &H = *Hp;
    HMX_HIP(hipSetDevice(H.device));
    hipStream_t st = (hipStream_t)stream;
    if (!H.finalized) {
        set_error("hmx_hmatrix_matvec: operator not built (call hmx_hmatrix_compress or hmx_hmatrix_finalize first)");
        return HMX_ERR_STATE;
    }
    const bool chunkable = nchunks > 1 && !(H.has_mirror && !H.sym_expanded) && H.E.nranges() > 1;
    if (!chunkable) {
        const int rc = matvec_device(H, 'N', alpha, in, beta, out, st);
        if (rc != HMX_OK)
            return rc;
        if (after_chunk)
            after_chunk(user, 0, 0, H.nT);
        if (used)
            *used = 1;
        return HMX_OK;
    }
    H.ev_names.clear();
    const bool prof = H.profiling; // per-kernel events make no sense with interleaved collectives
    H.profiling     = false;
    const int rc    = run_forward(H, H.e_zidx.d, in, alpha, beta, out, st, false, nchunks, after_chunk, user);
    H.profiling     = prof;
    if (used)
        *used = H.chunk_plan_n;
    return rc;
}
// row bounds of the chunks api_matvec_chunked will use (bounds[0..n]; n returned through *n_out; n = 1 when the operator is not chunkable)
int api_chunk_bounds(HMat *Hp, int nchunks, int *n_out, int32_t *bounds) {
    if (!Hp || !n_out || !bounds)
        return HMX_ERR_INVALID;
    HMat &H = *Hp;
    HMX_HIP(hipSetDevice(H.device));
    const bool chunkable = nchunks > 1 && H.finalized && !(H.has_mirror && !H.sym_expanded) && H.E.nranges() > 1;
    if (!chunkable) {
        *n_out    = 1;
        bounds[0] = 0;
        bounds[1] = H.nT;
        return HMX_OK;
    }
    const int rc = ensure_expand_chunks(H, nchunks);
    if (rc != HMX_OK)
        return rc;
    *n_out = H.chunk_plan_n;
    for (int c = 0; c < H.chunk_plan_n; c++)
        bounds[c] = H.chunk_row_lo[c];
    bounds[H.chunk_plan_n] = H.nT;
    return HMX_OK;
}

static int matmat_device(HMat &H, char trans, scalar alpha, const scalar *din, scalar beta, scalar *dout, int mu, hipStream_t st);
// The layout a trans = 'N' product with several right-hand sides runs on: the operator's own streams, or (compact symmetric storage) its
// expanded view -- same rows, its own row ranges.  nullptr: no fused multi-RHS path (one pass per right-hand side).
static HMat *matmat_layout_n(HMat &H) {
    if (!(H.finalized && H.opt.i(HMX_OPT_MULTI_RHS_FUSED) != 0))
        return nullptr;
    if (!H.sym_fused)
        return &H;
    if (sym_mu_fused(H)) // the product runs on the stored triangle (rows receive mirrored contributions after the E pass): single exchange, and no view is built
        return nullptr;
    return ensure_expanded_view(H);
}
// api_matvec_chunked for mu right-hand sides (row-major, device pointers, trans = 'N'): after_chunk(user, c, row_lo, row_hi) is called on the
// host right after the expand kernels of row chunk c (all groups of right-hand sides) were launched on `stream`.
int api_matmat_chunked(HMat *Hp, scalar alpha, const scalar *in, scalar beta, scalar *out, int mu, void *stream, int nchunks, after_chunk_fn after_chunk, void *user, int *used) {
    if (!Hp || !in || !out || mu < 1) {
        set_error("hmx_hmatrix_matmat_row_major (chunked): invalid arguments");
        return HMX_ERR_INVALID;
    }
    HMat &H = *Hp;
    HMX_HIP(hipSetDevice(H.device));
    hipStream_t st = (hipStream_t)stream;
    if (!H.finalized) {
        set_error("hmx_hmatrix_matmat_row_major: operator not built (call hmx_hmatrix_compress or hmx_hmatrix_finalize first)");
        return HMX_ERR_STATE;
    }
    HMat *F = nchunks > 1 ? matmat_layout_n(H) : nullptr;
    if (!F || F->E.nranges() <= 1) {
        const int rc = matmat_device(H, 'N', alpha, in, beta, out, mu, st);
        if (rc != HMX_OK)
            return rc;
        if (after_chunk)
            after_chunk(user, 0, 0, H.nT);
        if (used)
            *used = 1;
        return HMX_OK;
    }
    F->ev_names.clear();
    const bool prof = F->profiling;
    F->profiling    = false; // per-kernel events make no sense with interleaved collectives
    const int rc    = run_forward_mu(*F, in, alpha, beta, out, mu, st, nchunks, after_chunk, user);
    F->profiling    = prof;
    if (used)
        *used = F->chunk_plan_n;
    return rc;
}
// row bounds of the chunks api_matmat_chunked will use (they differ from the single-vector ones when the product runs on the expanded view)
int api_chunk_bounds_mu(HMat *Hp, int nchunks, int *n_out, int32_t *bounds) {
    if (!Hp || !n_out || !bounds)
        return HMX_ERR_INVALID;
    HMat &H = *Hp;
    HMX_HIP(hipSetDevice(H.device));
    HMat *F = (nchunks > 1 && H.finalized) ? matmat_layout_n(H) : nullptr;
    if (!F || F->E.nranges() <= 1) {
        *n_out    = 1;
        bounds[0] = 0;
        bounds[1] = H.nT;
        return HMX_OK;
    }
    const int rc = ensure_expand_chunks(*F, nchunks);
    if (rc != HMX_OK)
        return rc;
    *n_out = F->chunk_plan_n;
    for (int c = 0; c < F->chunk_plan_n; c++)
        bounds[c] = F->chunk_row_lo[c];
    bounds[F->chunk_plan_n] = H.nT;
    return HMX_OK;
}

int api_matvec_user(HMat *Hp, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out, int mem, void *stream) {
    if (!Hp || !in || !out) {
        set_error("hmx_hmatrix_matvec_user: NULL argument");
        return HMX_ERR_INVALID;
    }
    HMat &H = *Hp;
    // cluster_to_user / user_to_cluster are only stable for a root cluster or a local permutation
    // (clustering/cluster_node.hpp:152-157)
    if (!(H.t_root_is_tree_root || H.perm_local) || !(H.S0 == 0 && H.nS == H.nS_total)) {
        set_error("hmx_hmatrix_matvec_user: cluster is neither root nor local, permutation is not stable");
        return HMX_ERR_INVALID;
    }
    HMX_HIP(hipSetDevice(H.device));
    hipStream_t st = (hipStream_t)stream;
    const scalar *din;
    scalar *dout;
    bool staged;
    int rc = with_buffers(H, trans, in, out, 1, mem, st, beta, &din, &dout, staged);
    if (rc != HMX_OK)
        return rc;
    if (!H.d_perm_t.d) {
        HMX_HIP(H.d_perm_t.upload(H.perm_t));
        HMX_HIP(H.d_perm_s.upload(H.perm_s));
    }
    const int nin = trans == 'N' ? H.nS : H.nT, nout = trans == 'N' ? H.nT : H.nS;
    const int32_t *pin = trans == 'N' ? H.d_perm_s.d + H.S0 : H.d_perm_t.d + H.T0, *pout = trans == 'N' ? H.d_perm_t.d + H.T0 : H.d_perm_s.d + H.S0;
    const int bin = trans == 'N' ? H.S0 : H.T0, bout = trans == 'N' ? H.T0 : H.S0;
    if (H.tmp_in2.n < (size_t)nin)
        HMX_HIP(H.tmp_in2.alloc(nin));
    if (H.tmp_out2.n < (size_t)nout)
        HMX_HIP(H.tmp_out2.alloc(nout));
    hipLaunchKernelGGL(gather_kernel, dim3((nin + 255) / 256), dim3(256), 0, st, nin, pin, bin, din, H.tmp_in2.d, 1);
    if (!hmx_is_zero(beta))
        hipLaunchKernelGGL(gather_kernel, dim3((nout + 255) / 256), dim3(256), 0, st, nout, pout, bout, (const scalar *)dout, H.tmp_out2.d, 1);
    rc = matvec_device(H, trans, alpha, H.tmp_in2.d, beta, H.tmp_out2.d, st);
    if (rc != HMX_OK)
        return rc;
    hipLaunchKernelGGL(scatter_kernel, dim3((nout + 255) / 256), dim3(256), 0, st, nout, pout, bout, (const scalar *)H.tmp_out2.d, dout, 1);
    HMX_HIP(hipGetLastError());
    if (staged) {
        HMX_HIP(hipMemcpyAsync(out, dout, (size_t)nout * sizeof(scalar), hipMemcpyDeviceToHost, st));
        HMX_HIP(hipStreamSynchronize(st));
    }
    return HMX_OK;
}

// row-major multi-RHS product on device pointers (the body shared by the row-major and the column-major / user-numbering entry points)
static int matmat_device(HMat &H, char trans, scalar alpha, const scalar *din, scalar beta, scalar *dout, int mu, hipStream_t st) {
    int rc;
    const int nin = trans == 'N' ? H.nS : H.nT, nout = trans == 'N' ? H.nT : H.nS;
#if !HMX_COMPLEX
    if (trans == 'C' && H.symmetry_for_leaves != 'S')
        trans = 'T'; // real coefficients
#endif
    const bool fused_ok   = H.finalized && H.opt.i(HMX_OPT_MULTI_RHS_FUSED) != 0;
    const bool square_sym = H.has_mirror && H.T0 == H.S0 && H.nT == H.nS;
    // a square symmetric ('S') operator is its own transpose, a square Hermitian one its own conjugate transpose
    const bool as_n = trans == 'N' || (trans == 'T' && square_sym && H.symmetry_for_leaves == 'S') || (trans == 'C' && square_sym && H.symmetry_for_leaves == 'H');
    auto collect_times = [&](HMat &F) -> int {
        if (H.profiling) {
            HMX_HIP(hipStreamSynchronize(st));
            H.last_ms.clear();
            H.last_names.clear();
            for (size_t k = 1; k < F.ev_names.size(); k++) {
                float ms = 0;
                (void)hipEventElapsedTime(&ms, F.ev[k - 1], F.ev[k]);
                H.last_ms.push_back(ms);
                H.last_names.push_back(F.ev_names[k]);
            }
        }
        return HMX_OK;
    };
    // 1. symmetric / Hermitian storage, untransposed: the stored triangle itself (sym_mu_fused decides between it and the expanded view)
    if (fused_ok && mu > 1 && as_n && sym_mu_fused(H)) {
        H.ev_names.clear();
        prof_mark(H, st, "begin");
        rc = run_forward_mu_sym(H, din, alpha, beta, dout, mu, st);
        return rc != HMX_OK ? rc : collect_times(H);
    }
    // 2. a layout the fused multi-RHS kernels run on: the operator's own streams, its expanded view, or its transposed layout
    HMat *F        = nullptr;
    bool conj_wrap = false;
    if (fused_ok && (!H.has_mirror || H.sym_expanded || H.sym_fused)) {
        if (as_n)
            F = H.sym_fused ? ensure_expanded_view(H) : &H;
        else if (trans == 'T' && !(HMX_COMPLEX && H.symmetry_for_leaves == 'H'))
            F = ensure_transposed_operator(H);
#if HMX_COMPLEX
        else if (trans == 'C' && H.symmetry_for_leaves != 'S') { // conj o 'T' o conj on all right-hand sides at once
            F         = ensure_transposed_operator(H);
            conj_wrap = true;
        }
#endif
    }
    // 3. no such layout (no room in HBM, HMX_OPT_TRANSPOSED_LAYOUT = 0, factors released): the stored data -- the stored triangle of a
    //    symmetric operator whatever the option says, the transposed product of an ordinary operator through its mirrored sweeps.  Nothing
    //    falls back to one product per right-hand side any more.
    const bool stored_sym   = !F && fused_ok && mu > 1 && as_n && H.sym_fused && H.s64_nint > 0;
    const bool stored_trans = !F && fused_ok && mu > 1 && !as_n && (trans == 'T' || conj_wrap) && !H.has_mirror && !H.view_of;
    if (stored_trans && !H.trans_fused && !H.trans_tables_failed && build_trans_tables(H) != HMX_OK) {
        H.trans_tables_failed = true;
        (void)hipGetLastError();
    }
    const bool use_stored_trans = stored_trans && H.trans_fused && H.s64_nint > 0;
    if (!(F || stored_sym || use_stored_trans))
        conj_wrap = false;
#if HMX_COMPLEX
    if (conj_wrap) {
        const int64_t tin = (int64_t)nin * mu, tout = (int64_t)nout * mu;
        if ((int64_t)H.conj_in.n < tin)
            HMX_HIP(H.conj_in.alloc(tin));
        hipLaunchKernelGGL(conj_kernel, dim3((unsigned)((tin + 255) / 256)), dim3(256), 0, st, tin, din, H.conj_in.d);
        if (!hmx_is_zero(beta))
            hipLaunchKernelGGL(conj_kernel, dim3((unsigned)((tout + 255) / 256)), dim3(256), 0, st, tout, (const scalar *)dout, dout);
        din   = H.conj_in.d;
        alpha = hmx_conj(alpha);
        beta  = hmx_conj(beta);
    }
#endif
    auto conj_back = [&]() {
#if HMX_COMPLEX
        if (conj_wrap) {
            const int64_t tout = (int64_t)nout * mu;
            hipLaunchKernelGGL(conj_kernel, dim3((unsigned)((tout + 255) / 256)), dim3(256), 0, st, tout, (const scalar *)dout, dout);
        }
#endif
    };
    if (F) {
        F->profiling = H.profiling;
        F->ev_names.clear();
        prof_mark(*F, st, "begin");
        rc = run_forward_mu(*F, din, alpha, beta, dout, mu, st);
        if (rc != HMX_OK)
            return rc;
        rc = collect_times(*F);
        conj_back();
        return rc;
    }
    if (stored_sym || use_stored_trans) {
        H.ev_names.clear();
        prof_mark(H, st, "begin");
        rc = stored_sym ? run_forward_mu_sym(H, din, alpha, beta, dout, mu, st) : run_transposed_fused_mu(H, din, alpha, beta, dout, mu, st);
        if (rc != HMX_OK)
            return rc;
        rc = collect_times(H);
        conj_back();
        return rc;
    }
    // what is left: fused products switched off (HMX_OPT_MULTI_RHS_FUSED = 0), or a row-restricted symmetric operator's transposed product
    // without room for its transposed view -- one product per right-hand side (the second reports the missing view itself)
    if (H.tmp_in2.n < (size_t)nin)
        HMX_HIP(H.tmp_in2.alloc(nin));
    if (H.tmp_out2.n < (size_t)nout)
        HMX_HIP(H.tmp_out2.alloc(nout));
    for (int c = 0; c < mu; c++) {
        hipLaunchKernelGGL(col_extract_kernel, dim3((nin + 255) / 256), dim3(256), 0, st, nin, mu, c, din, H.tmp_in2.d);
        if (!hmx_is_zero(beta))
            hipLaunchKernelGGL(col_extract_kernel, dim3((nout + 255) / 256), dim3(256), 0, st, nout, mu, c, (const scalar *)dout, H.tmp_out2.d);
        rc = matvec_device(H, trans, alpha, H.tmp_in2.d, beta, H.tmp_out2.d, st);
        if (rc != HMX_OK)
            return rc;
        hipLaunchKernelGGL(col_insert_kernel, dim3((nout + 255) / 256), dim3(256), 0, st, nout, mu, c, (const scalar *)H.tmp_out2.d, dout);
    }
    HMX_HIP(hipGetLastError());
    return HMX_OK;
}


int api_matmat_row_major(HMat *Hp, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out, int mu, int mem, void *stream) {
    if (!Hp || !in || !out || mu < 1) {
        set_error("hmx_hmatrix_matmat_row_major: invalid arguments");
        return HMX_ERR_INVALID;
    }
    HMat &H = *Hp;
    HMX_HIP(hipSetDevice(H.device));
    hipStream_t st = (hipStream_t)stream;
    const scalar *din;
    scalar *dout;
    bool staged;
    int rc = with_buffers(H, trans, in, out, mu, mem, st, beta, &din, &dout, staged);
    if (rc != HMX_OK)
        return rc;
    rc = matmat_device(H, trans, alpha, din, beta, dout, mu, st);
    if (rc != HMX_OK)
        return rc;
    if (staged) {
        const size_t nout = (size_t)(trans == 'N' ? H.nT : H.nS);
        HMX_HIP(hipMemcpyAsync(out, dout, nout * mu * sizeof(scalar), hipMemcpyDeviceToHost, st));
        HMX_HIP(hipStreamSynchronize(st));
    }
    return HMX_OK;
}

// add_hmatrix_matrix_product (hmatrix/linalg/add_hmatrix_matrix_product.hpp:26-77,176-205): column-major B (n x mu) and C (m x mu)
// in USER numbering; every column is permuted to cluster numbering and the operands are transposed to row-major (one gather
// kernel each way), the fused row-major product runs, the result is transposed and permuted back.
int api_matmat_user(HMat *Hp, char trans, scalar alpha, const scalar *in, scalar beta, scalar *out, int mu, int mem, void *stream) {
    if (!Hp || !in || !out || mu < 1) {
        set_error("hmx_hmatrix_matmat_user: invalid arguments");
        return HMX_ERR_INVALID;
    }
    HMat &H = *Hp;
    if (!(H.t_root_is_tree_root || H.perm_local) || !(H.S0 == 0 && H.nS == H.nS_total)) {
        set_error("hmx_hmatrix_matmat_user: cluster is neither root nor local, permutation is not stable");
        return HMX_ERR_INVALID;
    }
    HMX_HIP(hipSetDevice(H.device));
    hipStream_t st = (hipStream_t)stream;
    const scalar *din;
    scalar *dout;
    bool staged;
    int rc = with_buffers(H, trans, in, out, mu, mem, st, beta, &din, &dout, staged);
    if (rc != HMX_OK)
        return rc;
    if (!H.d_perm_t.d) {
        HMX_HIP(H.d_perm_t.upload(H.perm_t));
        HMX_HIP(H.d_perm_s.upload(H.perm_s));
    }
    const int nin = trans == 'N' ? H.nS : H.nT, nout = trans == 'N' ? H.nT : H.nS;
    const int32_t *pin = trans == 'N' ? H.d_perm_s.d + H.S0 : H.d_perm_t.d + H.T0, *pout = trans == 'N' ? H.d_perm_t.d + H.T0 : H.d_perm_s.d + H.S0;
    const int bin = trans == 'N' ? H.S0 : H.T0, bout = trans == 'N' ? H.T0 : H.S0;
    const int64_t tin = (int64_t)nin * mu, tout = (int64_t)nout * mu;
    if ((int64_t)H.mm_in.n < tin)
        HMX_HIP(H.mm_in.alloc(tin));
    if ((int64_t)H.mm_out.n < tout)
        HMX_HIP(H.mm_out.alloc(tout));
    hipLaunchKernelGGL(gather_cm_kernel, dim3((unsigned)((tin + 255) / 256)), dim3(256), 0, st, nin, mu, pin, bin, din, H.mm_in.d);
    if (!hmx_is_zero(beta))
        hipLaunchKernelGGL(gather_cm_kernel, dim3((unsigned)((tout + 255) / 256)), dim3(256), 0, st, nout, mu, pout, bout, (const scalar *)dout, H.mm_out.d);
    rc = matmat_device(H, trans, alpha, H.mm_in.d, beta, H.mm_out.d, mu, st);
    if (rc != HMX_OK)
        return rc;
    hipLaunchKernelGGL(scatter_cm_kernel, dim3((unsigned)((tout + 255) / 256)), dim3(256), 0, st, nout, mu, pout, bout, (const scalar *)H.mm_out.d, dout);
    HMX_HIP(hipGetLastError());
    if (staged) {
        HMX_HIP(hipMemcpyAsync(out, dout, (size_t)tout * sizeof(scalar), hipMemcpyDeviceToHost, st));
        HMX_HIP(hipStreamSynchronize(st));
    }
    return HMX_OK;
}

int api_set_profiling(HMat *H, int enabled) {
    if (!H)
        return HMX_ERR_INVALID;
    H->profiling = enabled != 0;
    return HMX_OK;
}
int api_last_kernel_times(const HMat *H, int max, const char **names, float *ms) {
    if (!H)
        return 0;
    int n = std::min<int>(max, (int)H->last_ms.size());
    for (int k = 0; k < n; k++) {
        names[k] = H->last_names[k];
        ms[k]    = H->last_ms[k];
    }
    return n;
}



// hmx_hmatrix_prepare: everything a product with this `trans` and this many right-hand sides needs beyond the operator itself is built
// NOW -- the transposed stream layout of a 'T' / 'C' product (ensure_transposed_operator), the expanded view multi-RHS products on
// compact symmetric storage run on (ensure_expanded_view), work vectors, permutation and staging buffers -- by running one product
// of that shape on zero operands through each entry point (cluster numbering, user numbering).  Afterwards products of that shape
// allocate nothing: no latency cliff and no out-of-memory surprise in the middle of a Krylov solve.
int api_prepare(HMat *Hp, char trans, int mu) {
    if (!Hp || mu < 1) {
        set_error("hmx_hmatrix_prepare: invalid arguments");
        return HMX_ERR_INVALID;
    }
    HMat &H = *Hp;
    if (!H.finalized) {
        set_error("hmx_hmatrix_prepare: operator not built");
        return HMX_ERR_STATE;
    }
    HMX_HIP(hipSetDevice(H.device));
    const bool n_form = trans == 'N';
    const size_t nin = (size_t)(n_form ? H.nS : H.nT) * mu, nout = (size_t)(n_form ? H.nT : H.nS) * mu;
    DArr<scalar> in, out;
    HMX_HIP(in.alloc(std::max<size_t>(nin, 1)));
    HMX_HIP(out.alloc(std::max<size_t>(nout, 1)));
    HMX_HIP(in.zero());
    HMX_HIP(out.zero());
    int rc = mu == 1 ? api_matvec(Hp, trans, scalar(1), in.d, scalar(0), out.d, HMX_MEM_DEVICE, nullptr)
                     : api_matmat_row_major(Hp, trans, scalar(1), in.d, scalar(0), out.d, mu, HMX_MEM_DEVICE, nullptr);
    if (rc != HMX_OK)
        return rc;
    // the user-numbering front ends exist for this operator (the predicate api_matvec_user / api_matmat_user apply: a stable permutation AND
    // the whole source cluster -- a block-diagonal / local-to-local operator on a local-permutation tree has the first, not the second):
    // their staging buffers too
    if ((H.t_root_is_tree_root || H.perm_local) && H.S0 == 0 && H.nS == H.nS_total) {
        rc = mu == 1 ? api_matvec_user(Hp, trans, scalar(1), in.d, scalar(0), out.d, HMX_MEM_DEVICE, nullptr)
                     : api_matmat_user(Hp, trans, scalar(1), in.d, scalar(0), out.d, mu, HMX_MEM_DEVICE, nullptr);
        if (rc != HMX_OK)
            return rc;
    }
    HMX_HIP(hipDeviceSynchronize());
    return HMX_OK;
}

int api_device_of(const HMat *H) { return H ? H->device : -1; }
int api_root(const HMat *H, int32_t *t_off_size_s_off_size) { // root block of the operator, global cluster numbering
    if (!H || !t_off_size_s_off_size)
        return HMX_ERR_INVALID;
    t_off_size_s_off_size[0] = H->T0, t_off_size_s_off_size[1] = H->nT, t_off_size_s_off_size[2] = H->S0, t_off_size_s_off_size[3] = H->nS;
    return HMX_OK;
}
void api_destroy(HMat *H) { delete H; }
void api_axpby(int64_t n, const scalar *w, scalar beta, scalar *y, hipStream_t st) {
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (int)n, scalar(1), w, beta, y);
}
