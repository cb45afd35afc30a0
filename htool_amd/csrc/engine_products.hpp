// engine_products.hpp -- the sweeps of a product on the device: one vector, several right-hand sides, symmetric / transposed forms, second layouts.
// Part of the engine's host code: included by engine_body.hpp inside namespace hmx::{f64,f32,z64,c32}.  No include guard on purpose.

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

// the mirrored sweeps keep their group's accumulators in dynamic LDS (beyond the 64 KB a kernel may use without saying so)
#define HMX_LAUNCH_GROUPED(kernel, grid, block, lds_bytes, st, ...)                                                                    \
    do {                                                                                                                                \
        static std::atomic<size_t> allowed_[16]; /* per device: the attribute belongs to the function ON a device */                     \
        int dev_ = 0;                                                                                                                   \
        (void)hipGetDevice(&dev_);                                                                                                      \
        dev_ &= 15;                                                                                                                     \
        if ((size_t)(lds_bytes) > allowed_[dev_].load()) {                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bytes)); \
            (void)hipGetLastError();                                                                                                    \
            allowed_[dev_].store((size_t)(lds_bytes));                                                                                  \
        }                                                                                                                               \
        hipLaunchKernelGGL(kernel, grid, block, lds_bytes, st, __VA_ARGS__);                                                            \
    } while (0)

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
            ExpandSymArgs X{{H.E.stream.d, H.s_grp_order.d, H.E.d_off.d, H.E.d_len.d, H.E.d_cols.d, H.E.d_base.d, H.E.d_colbase.d, zidx, H.Z.d, y, alpha, beta, H.E.nranges(), xin, nx},
                            H.s_mdst.d, H.SW.d, x_src + (H.T0 - H.S0), H.symmetry_for_leaves == 'H' ? 1 : 0, H.s_grp_flush.d, H.s_grp_na.d, H.s_group};
            const size_t lds = (size_t)H.s_gcap * sizeof(scalar);
            switch (EW) {
            case 8: HMX_LAUNCH_GROUPED(expand_sym_kernel<8>, dim3(H.s_ngroups), dim3(512), lds, st, X); break;
            default: HMX_LAUNCH_GROUPED(expand_sym_kernel<4>, dim3(H.s_ngroups), dim3(256), lds, st, X); break;
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
            RowSymArgs A{H.R.stream.d, H.s_coef.d, H.s_int_order.d, H.s_sub_ptr.d, H.s_sub_src.d, H.s_sub_cb.d, H.s_sub_w.d, H.s_sub_nrows.d, H.s_sub_dst.d,
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
        for (auto *a : {&H.s_mdst, &H.s_coef, &H.s_count, &H.sc_dst, &H.sc_lp, &H.sc_count, &H.sc_k, &H.s_list, &H.s_fidx, &H.s_sub_w, &H.s_sub_nrows, &H.s_sub_dst, &H.s_int_order})
            a->release();
        for (auto *a : {&H.s_sub_ptr, &H.s_sub_src, &H.s_sub_cb})
            a->release();
        H.SW.release();
        return rc;
    }
    H.trans_fused = true;
    return HMX_OK;
}

static int run_transposed_fused(HMat &H, const scalar *in, scalar alpha, scalar beta, scalar *out, hipStream_t st) {
    const int EW = H.opt.i(HMX_OPT_EXPAND_WAVES) ? H.opt.i(HMX_OPT_EXPAND_WAVES) : (H.E.nranges() <= 4096 ? 8 : 4);
    if (H.E.nranges() > 0) {
        ExpandSymArgs X{{H.E.stream.d, H.s_grp_order.d, H.E.d_off.d, H.E.d_len.d, H.E.d_cols.d, H.E.d_base.d, H.E.d_colbase.d, nullptr, H.Z.d, nullptr, alpha, beta, H.E.nranges(), nullptr, 0},
                        H.s_mdst.d, H.SW.d, in, 0, H.s_grp_flush.d, H.s_grp_na.d, H.s_group};
        const size_t lds = (size_t)H.s_gcap * sizeof(scalar);
        switch (EW) {
        case 8: HMX_LAUNCH_GROUPED((expand_sym_kernel<8, false>), dim3(H.s_ngroups), dim3(512), lds, st, X); break;
        default: HMX_LAUNCH_GROUPED((expand_sym_kernel<4, false>), dim3(H.s_ngroups), dim3(256), lds, st, X); break;
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
        RowSymArgs A{H.R.stream.d, H.s_coef.d, H.s_int_order.d, H.s_sub_ptr.d, H.s_sub_src.d, H.s_sub_cb.d, H.s_sub_w.d, H.s_sub_nrows.d, H.s_sub_dst.d,
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
        HMX_HIP(place_array(H, H.Zmu, need, 1, st, &H.placed_zmu));
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
    // HMX_OPT_SYM_MULTI_RHS = 1 or -1 (the default): the stored triangle; 0: an expanded view of the operator (twice the memory, twice the
    // traffic by construction).  Until round 5 the view was the default for real coefficients while HBM had room for it -- it was the faster of
    // the two (N = 1e6 fp64, 16 right-hand sides: 3.6 against 3.7-4.2 ms).  Round 6 (column sums folded per group of row ranges in LDS, the second
    // sweep over the R-streams as one pipeline per interval, two / three waves per SIMD): 3.3 ms on the stored triangle against 3.8 on the view,
    // one rank's operator of BASELINE configs[4] 2.2 ms either way at 7.6 instead of 18.2 GB -- nothing is built unless it is asked for.
    const int mode = H.opt.i(HMX_OPT_SYM_MULTI_RHS);
    return H.sym_fused && H.s64_nint > 0 && mode != 0;
}
static int ensure_sw16(HMat &H, hipStream_t st) {
    const size_t need16 = (size_t)(H.s_slots + 1) * SWW;
    if (H.SW16.n < need16) {
        HMX_HIP(place_array(H, H.SW16, need16, 0, st, &H.placed_sw16)); // (zero-filled: slot s_slots stays zero for ever -- the operand of the columns that are no mirrored leaf's)
    }
    return HMX_OK;
}
// the sweeps over E (forward + mirrored column sums, or -- fwd = false -- the column sums only), the folds of a' and the second sweep over R
// for the nrhs right-hand sides starting at column c.  herm: mirrored leaves are conjugate transposes.
static int sym_mu_sweeps(HMat &H, bool fwd, const scalar *X, const scalar *xrow, scalar alpha, scalar beta, scalar *Y, int nout, int accumulate, int herm, int mu, int c, int nrhs, hipStream_t st) {
    constexpr int W = 4;
    if (H.E.nranges() > 0) {
        ExpandSymArgs XS{{H.E.stream.d, H.s_grp_order.d, H.E.d_off.d, H.E.d_len.d, H.E.d_cols.d, H.E.d_base.d, H.E.d_colbase.d, fwd ? H.e_zidx.d : nullptr, fwd ? H.Zmu.d : nullptr,
                          fwd ? Y : nullptr, alpha, beta, H.E.nranges(), fwd ? X : nullptr, fwd ? H.nS : 0},
                         H.s_mdst.d, H.SW16.d, xrow, herm, H.s_grp_flush.d, H.s_grp_na.d, H.s_group};
        const dim3 grid((unsigned)H.s_ngroups), wg(W * 64);
        const size_t lds = (size_t)H.s_gcap * SWW * sizeof(scalar); // the group's accumulators
#if HMX_COMPLEX
#define HMX_SYM_MU_E(MU)                                                                                      \
    do {                                                                                                      \
        if (fwd)                                                                                              \
            HMX_LAUNCH_GROUPED((expand_sym_mu_kernel<W, MU, true>), grid, wg, lds, st, XS, mu, c, nrhs);       \
        else                                                                                                  \
            HMX_LAUNCH_GROUPED((expand_sym_mu_kernel<W, MU, false>), grid, wg, lds, st, XS, mu, c, nrhs);      \
    } while (0)
        if (H.opt.i(HMX_OPT_MATRIX_CORES) != 0) { // groups of up to 8 on the matrix cores (ragged groups: operands nobody stores the results of)
            if (fwd)
                HMX_LAUNCH_GROUPED((expand_sym_zmfma8_kernel<W, true>), grid, wg, lds, st, XS, mu, c, nrhs);
            else
                HMX_LAUNCH_GROUPED((expand_sym_zmfma8_kernel<W, false>), grid, wg, lds, st, XS, mu, c, nrhs);
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
            HMX_LAUNCH_GROUPED((expand_sym_mfma16_kernel<W, true>), grid, wg, lds, st, XS, mu, c, nrhs);
        else
            HMX_LAUNCH_GROUPED((expand_sym_mfma16_kernel<W, false>), grid, wg, lds, st, XS, mu, c, nrhs);
        prof_mark(H, st, fwd ? "expand_sym_mfma16_kernel" : "expand_colsum_mfma16_kernel");
#endif
    }
    if (H.n_sym_combine > 0) {
        const int nw = H.n_sym_combine_wave, nt = H.n_sym_combine - nw; // the first nw entries fold >= 32 partial sums: one wave each; both kinds in one launch
        CombineListArgs CW{H.sc_dst.d, H.sc_lp.d, H.sc_count.d, H.sc_k.d, H.s_list.d, H.SW16.d, nw};
        CombineListArgs CT{H.sc_dst.d + nw, H.sc_lp.d + nw, H.sc_count.d + nw, H.sc_k.d + nw, H.s_list.d, H.SW16.d, nt};
        const int wave_blocks = (nw + 3) / 4;
        const int64_t thread_blocks = ((int64_t)nt * SWW + 255) / 256;
        hipLaunchKernelGGL(combine_list_mu_both_kernel, dim3((unsigned)(wave_blocks + thread_blocks)), dim3(256), 0, st, CW, CT, wave_blocks);
        prof_mark(H, st, "combine_sym_mu_kernel");
    }
    if (H.s64_nint > 0) {
        RowSegArgs RS{H.R.stream.d, H.s64_int_order.d, H.s64_int_off.d, H.s64_seg_ptr.d, H.s64_seg_src.d, H.s64_seg_cb.d, H.s64_seg_wp.d, H.s64_seg_w.d, H.s_coef.d,
                      H.SW16.d, (int)H.s_slots, H.s64_nint, H.s_fidx.d, H.s_count.d, Y, alpha, beta, nout, herm, accumulate};
#if HMX_COMPLEX
        const dim3 grid((unsigned)H.s64_nint), wg(W * 64);
        if (H.opt.i(HMX_OPT_MATRIX_CORES) != 0) {
            hipLaunchKernelGGL((rowsym_zmfma8_kernel<W>), dim3((unsigned)((H.s64_nint + W - 1) / W)), wg, 0, st, RS, mu, c, nrhs);
            prof_mark(H, st, "rowsym_zmfma8_kernel");
        } else if (nrhs <= 2)
            hipLaunchKernelGGL((rowsym_mu_kernel<W, 2>), grid, wg, 0, st, RS, mu, c, nrhs);
        else if (nrhs <= 4)
            hipLaunchKernelGGL((rowsym_mu_kernel<W, 4>), grid, wg, 0, st, RS, mu, c, nrhs);
        else
            hipLaunchKernelGGL((rowsym_mu_kernel<W, 8>), grid, wg, 0, st, RS, mu, c, nrhs);
        if (H.opt.i(HMX_OPT_MATRIX_CORES) == 0)
            prof_mark(H, st, "rowsym_mu_kernel");
#else
        constexpr int RWV = HMX_ROWSYM_WAVES; // intervals (= waves) per workgroup
        hipLaunchKernelGGL((rowsym_mfma16_kernel<RWV>), dim3((unsigned)((H.s64_nint + RWV - 1) / RWV)), dim3(RWV * 64), 0, st, RS, mu, c, nrhs);
        prof_mark(H, st, "rowsym_mfma16_kernel");
#endif
    }
    return HMX_OK;
}
static int run_forward_mu_sym(HMat &H, const scalar *X, scalar alpha, scalar beta, scalar *Y, int mu, hipStream_t st) {
    const size_t need = (size_t)(H.zero_slot + 1) * mu;
    if (H.Zmu.n < need)
        HMX_HIP(place_array(H, H.Zmu, need, 1, st, &H.placed_zmu));
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
    T->may_probe           = H.may_probe; // a view built inside hmx_hmatrix_prepare may measure; one built by a product call may not
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
    X->may_probe           = H.may_probe; // (see ensure_transposed_operator)
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
