// engine_layout.hpp -- from compressed blocks to E- / R-streams: index tables of the mirrored (symmetric, transposed) products, build_streams.
// Part of the engine's host code: included by engine_body.hpp inside namespace hmx::{f64,f32,z64,c32}.  No include guard on purpose.

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
    hipEvent_t after = nullptr; // recorded behind the work a placement probe must not overlap with (the pack kernels of a build)
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
    std::vector<int64_t> s_sub_ptr, s_sub_src, s_sub_cb;
    std::vector<int32_t> s_sub_w, s_sub_nrows, s_sub_dst, s_int_order;
    std::vector<int32_t> o64; // launch order of the intervals of the multi-RHS form of the second sweep (below)
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
    const int64_t EWBASE = (A_total + 15) & ~int64_t(15), FLBASE = EWBASE + EWN;
    // ---- groups of row ranges (round 6) -----------------------------------------------------------------------------------------------
    // One workgroup of the mirrored sweep takes a GROUP of G consecutive row ranges, one after the other.  A mirrored column whose sum
    // belongs with the sums of columns of OTHER ranges of the group -- the r columns of a low-rank leaf that spans several of them (partial
    // sums of the same a'), the columns of dense leaves of the group that share their source cluster (contributions to the same output rows)
    // -- is not written out per range: it is added to an accumulator in LDS and the group writes the folded sums once, as one contiguous run
    // (the FLUSH slots of the group).  With 16 right-hand sides the partial sums were a quarter of the bytes the sweep reads, written and
    // then read again by the folding kernels; at N = 1e6, G = 4, it is 40 % of that.  mdst >= 0: slot in W, written per range; <= -2:
    // accumulator -2 - mdst of the group.  The candidates that would save most go first while the group's LDS budget lasts (`gcap` accumulators);
    // the others keep their own slots.  G = 1: no accumulators, the layout of rounds 2-5.  Fixed order of the additions (range by range; a
    // column is one lane's): bit-reproducible.
    const int nre = E.nranges();
    const int G   = std::max(1, H.opt.i(HMX_OPT_SYM_GROUP));
    // accumulators per group: what leaves the 16-RHS sweep its waves per SIMD -- LDS decides: tiles + accumulators of two workgroups per CU for
    // 8-byte reals (36.9 KB + 320 x 128 B each), of three for 4-byte reals (20.5 KB + 512 x 64 B) in 160 KB
    const int gcap_auto = sizeof(real) == 8 ? 320 : 512;
    const int gcap_max  = sizeof(real) == 8 ? 896 : 1024; // what ONE workgroup's tiles leave of a CU's 160 KB (41 KB of tiles + 896 x 128 B)
    const int gcap = G > 1 ? std::min(gcap_max, H.opt.i(HMX_OPT_SYM_GROUP_SLOTS) < 0 ? gcap_auto : H.opt.i(HMX_OPT_SYM_GROUP_SLOTS)) : 0;
    const int ng  = (nre + G - 1) / G;
    // pair -> accumulator of its group (-1: own slots); dense: and whether the pair is the first of its key (it alone counts as a contribution)
    std::vector<int32_t> lr_acc(elr_b.size(), -1), d_acc(ed_b.size(), -1);
    std::vector<char> d_first(ed_b.size(), 1);
    std::vector<int32_t> gna(ng, 0);
    if (gcap > 0) {
        struct Cand {
            int32_t g, count, width, kind; // kind 0: run of low-rank pairs [lo, hi) of elr_*, 1: dense pairs dk[lo .. hi)
            int64_t lo, hi;
        };
        std::vector<Cand> cand;
        for (size_t p = 0; p < elr_b.size();) { // the pairs of a leaf are consecutive (leaf-major) and cover consecutive ranges
            size_t q = p + 1;
            while (q < elr_b.size() && elr_b[q] == elr_b[p] && elr_r[q] / G == elr_r[p] / G)
                q++;
            if (q - p >= 2 && is_mir(elr_b[p]))
                cand.push_back({elr_r[p] / G, (int32_t)(q - p), (int32_t)XL[elr_b[p]].rank, 0, (int64_t)p, (int64_t)q});
            p = q;
        }
        std::vector<std::pair<int64_t, int32_t>> dk; // (group, source offset) -> dense pair
        for (size_t p = 0; p < ed_b.size(); p++)
            if (is_mir(ed_b[p]))
                dk.emplace_back(((int64_t)(ed_r[p] / G) << 32) | (uint32_t)XL[ed_b[p]].s_offset, (int32_t)p);
        std::sort(dk.begin(), dk.end());
        for (size_t i = 0; i < dk.size();) {
            size_t j = i + 1;
            bool same_size = true;
            while (j < dk.size() && dk[j].first == dk[i].first) {
                same_size = same_size && XL[ed_b[dk[j].second]].s_size == XL[ed_b[dk[i].second]].s_size;
                j++;
            }
            if (j - i >= 2 && same_size)
                cand.push_back({(int32_t)(dk[i].first >> 32), (int32_t)(j - i), (int32_t)XL[ed_b[dk[i].second]].s_size, 1, (int64_t)i, (int64_t)j});
            i = j;
        }
        std::sort(cand.begin(), cand.end(), [](const Cand &x, const Cand &y) {
            if (x.g != y.g) return x.g < y.g;
            if (x.count != y.count) return x.count > y.count; // saves (count - 1) writes per accumulator
            if (x.kind != y.kind) return x.kind < y.kind;
            return x.lo < y.lo;
        });
        for (const Cand &c : cand) {
            if (gna[c.g] + c.width > gcap)
                continue;
            const int32_t at = gna[c.g];
            gna[c.g] += c.width;
            if (c.kind == 0)
                for (int64_t p = c.lo; p < c.hi; p++)
                    lr_acc[p] = at;
            else
                for (int64_t i = c.lo; i < c.hi; i++) {
                    d_acc[dk[i].second]   = at;
                    d_first[dk[i].second] = i == c.lo;
                }
        }
    }
    std::vector<int64_t> gfl(ng + 1, 0); // flush slots of group g: FLBASE + gfl[g] ... (16-aligned runs)
    for (int g = 0; g < ng; g++)
        gfl[g + 1] = gfl[g] + ((gna[g] + 15) & ~15);
    const int64_t RWBASE = FLBASE + gfl[ng];
    phase_nosync("  sym: groups");
    // low-rank mirrored leaves: a leaf whose column sums end up in ONE place (one range, or one group's accumulators) is complete there (a'
    // is read from that slot), otherwise a list of the places of its partial sums feeds combine_list_kernel, which writes a'[aoff + k]
    std::vector<int32_t> nlist(nb, 0);
    for (size_t p = 0; p < elr_b.size(); p++)
        if (lr_acc[p] < 0 || p == 0 || elr_b[p - 1] != elr_b[p] || lr_acc[p - 1] != lr_acc[p] || elr_r[p - 1] / G != elr_r[p] / G)
            nlist[elr_b[p]]++;
    std::vector<int64_t> lptr(nb, -1);
    int64_t LN = 0;
    for (int64_t b = 0; b < nb; b++)
        if (is_mir(b) && XK[b] == LK_LOWRANK && XL[b].rank > 0 && nlist[b] > 1) {
            lptr[b] = LN;
            LN += nlist[b];
        }
    phase_nosync("  sym: setup");
    s_list.assign(LN, 0);
    std::vector<int64_t> single_slot(nb, -1);
    {
        // first pair of every leaf (the pairs of a leaf are consecutive): the threads below walk whole leaves
        std::vector<int64_t> leaf_first;
        for (size_t p = 0; p < elr_b.size(); p++)
            if (p == 0 || elr_b[p - 1] != elr_b[p])
                leaf_first.push_back((int64_t)p);
        leaf_first.push_back((int64_t)elr_b.size());
        parallel_for(leaf_first.size() - 1, [&](size_t lo, size_t hi) {
            for (size_t li = lo; li < hi; li++) {
                const int b = elr_b[leaf_first[li]];
                if (!is_mir(b))
                    continue;
                int64_t at = 0; // position in the leaf's list
                for (int64_t p = leaf_first[li]; p < leaf_first[li + 1]; p++) {
                    const int r = elr_r[p];
                    const bool acc = lr_acc[p] >= 0;
                    const bool first_of_run = !acc || p == leaf_first[li] || lr_acc[p - 1] != lr_acc[p] || elr_r[p - 1] / G != r / G;
                    const int64_t base = acc ? FLBASE + gfl[r / G] + lr_acc[p] : EWBASE + epad[r] + elr_c[p];
                    if (first_of_run) {
                        if (nlist[b] == 1)
                            single_slot[b] = base;
                        else
                            s_list[lptr[b] + at] = (int32_t)base;
                        at++;
                    }
                    int32_t *dst = s_mdst.data() + E.colbase[r] + elr_c[p];
                    for (int k = 0; k < XL[b].rank; k++)
                        dst[k] = acc ? (int32_t)(-2 - (lr_acc[p] + k)) : (int32_t)(base + k);
                }
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
            if (lptr[b] >= 0 && (nlist[b] >= 32) == (pass == 0)) {
                for (int k = 0; k < XL[b].rank; k++) {
                    s_cd.push_back((int32_t)(aoff[b] + k));
                    s_clp.push_back((int32_t)lptr[b]);
                    s_cc.push_back(nlist[b]);
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
    auto build_intervals = [&](int IR, std::vector<int64_t> &sub_ptr, std::vector<int64_t> &sub_src, std::vector<int64_t> &sub_cb, std::vector<int32_t> &sub_w, std::vector<int32_t> &sub_nrows,
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
        sub_src.assign(nsub, 0);
        sub_cb.assign(nsub, 0);
        sub_w.assign(nsub, 0);
        sub_nrows.assign(nsub, 0);
        sub_dst.assign(nsub, 0);
        std::vector<double> int_work(nint, 0.0);
        std::vector<int64_t> pos(sub_count.begin(), sub_count.end() - 1);
        for (size_t t = 0; t < ntask; t++) {
            if (!task_mirror[t])
                continue;
            const int r = R.task_range[t], ch = R.task_chunk[t], cw = R.cw[r];
            const int w = std::min(cw, R.cols[r] - ch * cw), wp = hmx_wp(w);
            const int j0 = R.off[r] + r_shift, j1 = j0 + R.len[r];
            for (int I = j0 / IR; I <= (j1 - 1) / IR; I++) {
                const int lo = std::max(j0, I * IR), hi = std::min(j1, (I + 1) * IR);
                const int64_t q = pos[I]++;
                sub_src[q]   = R.base[r] + (int64_t)ch * R.len[r] * cw + (int64_t)(lo - j0) * wp; // the sub-task's first row, the chunk's first column
                sub_cb[q]    = R.colbase[r] + (int64_t)ch * cw;
                sub_w[q]     = w;
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
        nint = build_intervals(SYM_IR, s_sub_ptr, s_sub_src, s_sub_cb, s_sub_w, s_sub_nrows, s_sub_dst, s_int_order, 1);
    // The multi-RHS form of the second sweep (rowsym_mfma16_kernel / rowsym_zmfma8_kernel: one WAVE per interval; rowsym_mu_kernel: one workgroup):
    // intervals of at most SYM_IR_MU rows cut AT the boundaries of the mirrored pieces, so that every (piece, chunk) task covers whole
    // intervals -- no sub-task of a few rows at the edge of a fixed 64-row grid (clusters of 61 rows against intervals of 64: every piece
    // met two intervals, a quarter more tiles than the stream holds).  Per interval the kernel walks SEGMENTS: one per (sub-task, 64-column half
    // of its chunk), each a ready record (first element, row pitch, columns, first entry of the coefficient-slot table) instead of the
    // chain sub-task -> task -> piece -> geometry of a dozen dependent loads.
    H.s64_nint = 0;
    std::vector<int32_t> i64_off;  // interval I = output rows [i64_off[I], i64_off[I + 1])
    std::vector<int64_t> g64_ptr;  // per interval: its segments [g64_ptr[I], g64_ptr[I + 1])
    std::vector<int64_t> g64_src, g64_cb;
    std::vector<int32_t> g64_wp, g64_w;
    if (!bad) {
        std::vector<char> cut((size_t)nOut + 1, 0);
        cut[0] = cut[nOut] = 1;
        for (size_t t = 0; t < ntask; t++)
            if (task_mirror[t]) {
                const int r = R.task_range[t], j0 = R.off[r] + r_shift;
                cut[j0] = cut[j0 + R.len[r]] = 1;
            }
        std::vector<int> bp;
        for (int j = 0; j <= nOut; j++)
            if (cut[j])
                bp.push_back(j);
        std::vector<int32_t> ilen;
        make_ranges(bp, SYM_IR_MU, 0, i64_off, ilen);
        const int nint = (int)i64_off.size();
        i64_off.push_back(nOut);
        std::vector<int32_t> row2int((size_t)nOut + 1, 0);
        for (int I = 0; I < nint; I++)
            for (int j = i64_off[I]; j < i64_off[I + 1]; j++)
                row2int[j] = I;
        row2int[nOut] = nint;
        auto halves = [&](size_t t) {
            const int r = R.task_range[t], ch = R.task_chunk[t], cw = R.cw[r];
            return (std::min(cw, R.cols[r] - ch * cw) + 63) / 64;
        };
        std::vector<int64_t> cnt((size_t)nint + 1, 0);
        for (size_t t = 0; t < ntask; t++)
            if (task_mirror[t]) {
                const int r = R.task_range[t], j0 = R.off[r] + r_shift;
                for (int I = row2int[j0]; I < row2int[j0 + R.len[r]]; I++)
                    cnt[I + 1] += halves(t);
            }
        for (int I = 0; I < nint; I++)
            cnt[I + 1] += cnt[I];
        g64_ptr = cnt;
        const int64_t nseg = cnt[nint];
        g64_src.assign(nseg, 0), g64_cb.assign(nseg, 0), g64_wp.assign(nseg, 0), g64_w.assign(nseg, 0);
        std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
        std::vector<double> int_work(nint, 0.0);
        for (size_t t = 0; t < ntask; t++) { // launch order of the tasks = order inside every interval's list, as for the single-vector form
            if (!task_mirror[t])
                continue;
            const int r = R.task_range[t], ch = R.task_chunk[t], cw = R.cw[r];
            const int w = std::min(cw, R.cols[r] - ch * cw), wp = hmx_wp(w);
            const int j0 = R.off[r] + r_shift;
            for (int I = row2int[j0]; I < row2int[j0 + R.len[r]]; I++)
                for (int c0 = 0; c0 < w; c0 += 64) {
                    const int64_t q = pos[I]++;
                    g64_src[q] = R.base[r] + (int64_t)ch * R.len[r] * cw + (int64_t)(i64_off[I] - j0) * wp + c0;
                    g64_cb[q]  = R.colbase[r] + (int64_t)ch * cw + c0;
                    g64_wp[q]  = wp;
                    g64_w[q]   = std::min(64, w - c0);
                    int_work[I] += (double)(i64_off[I + 1] - i64_off[I]) * std::min(64, w - c0) + 256;
                }
        }
        o64.resize(nint);
        std::iota(o64.begin(), o64.end(), 0);
        if (H.opt.i(HMX_OPT_TASK_ORDER) == 3) { // intervals over the same rows gather the same a' (see xcd_deal)
            const int unit_rows = std::max(SYM_IR_MU, H.opt.i(HMX_OPT_XCD_UNIT_ROWS));
            std::vector<int64_t> unit(nint), wk(nint);
            for (int I = 0; I < nint; I++) {
                unit[I] = (int64_t)i64_off[I] / unit_rows;
                wk[I]   = (int64_t)int_work[I];
            }
            o64 = xcd_deal(unit, wk, HMX_ROWSYM_WAVES);
        } else
            std::stable_sort(o64.begin(), o64.end(), [&](int a, int b) { return int_work[a] > int_work[b]; });
        H.s64_nint = nint;
    }
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
        if (!d_first[p])
            continue; // folded with an earlier pair of its group: one contribution for all of them
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
            const bool acc     = d_acc[p] >= 0;
            const int64_t base = acc ? FLBASE + gfl[r / G] + d_acc[p] : EWBASE + epad[r] + ed_c[p];
            int32_t *dst       = s_mdst.data() + E.colbase[r] + ed_c[p];
            for (int j = ja; j < jb; j++) {
                dst[j] = acc ? (int32_t)(-2 - (d_acc[p] + j)) : (int32_t)(base + j);
                if (d_first[p])
                    s_fidx[(size_t)(fill[j0 + j]++) * nOut + (j0 + j)] = (int32_t)(base + j);
            }
        }
    });
    H.s_nint = nint;
    phase_nosync("  sym: fidx fill");
    H.n_sym_combine = (int)s_cd.size();
    phase_nosync("fused symmetric slots");

    // ---- uploads (the first one waits for whatever is queued on the null stream: the pack kernels of build_streams) ----
    {   // launch order of the groups: heaviest first
        std::vector<int64_t> gw(ng, 0);
        for (int r = 0; r < nre; r++)
            gw[r / G] += (int64_t)E.len[r] * E.cols[r];
        std::vector<int32_t> gorder(ng), gflush(ng);
        std::iota(gorder.begin(), gorder.end(), 0);
        std::stable_sort(gorder.begin(), gorder.end(), [&](int x, int y) { return gw[x] > gw[y]; });
        for (int g = 0; g < ng; g++)
            gflush[g] = (int32_t)(FLBASE + gfl[g]);
        HMX_HIP(H.s_grp_order.upload(gorder));
        HMX_HIP(H.s_grp_flush.upload(gflush));
        HMX_HIP(H.s_grp_na.upload(gna));
        H.s_ngroups = ng;
        H.s_group   = G;
        H.s_gcap    = 0;
        for (int g = 0; g < ng; g++)
            H.s_gcap = std::max(H.s_gcap, (int)gna[g]);
        if (H.opt.i(HMX_OPT_BUILD_TIMING) != 0) { // how many partial sums a sweep still writes
            int64_t own = 0, folded = 0;
            for (int32_t v : s_mdst)
                (v >= 0 ? own : folded) += v != -1;
            fprintf(stderr, "[hmx build]   mirrored sweeps: groups of %d row ranges, %lld column sums written per range + %lld folded in LDS into %lld written per group (largest group: %d accumulators)\n",
                    G, (long long)own, (long long)folded, (long long)std::accumulate(gna.begin(), gna.end(), (int64_t)0), H.s_gcap);
        }
    }
    HMX_HIP(H.s_mdst.upload(s_mdst));
    HMX_HIP(H.s_coef.upload(s_coef));
    HMX_HIP(H.s_count.upload(s_cnt));
    HMX_HIP(H.s_sub_ptr.upload(s_sub_ptr));
    HMX_HIP(H.s_sub_src.upload(s_sub_src));
    HMX_HIP(H.s_sub_cb.upload(s_sub_cb));
    HMX_HIP(H.s_sub_w.upload(s_sub_w));
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
    HMX_HIP(place_array(H, H.SW, (size_t)s_total + 1, 0, nullptr, &H.placed_sw, M.after)); // column sums: written while E is read
    H.s_slots = s_total;
    H.SW16.release();
    if (H.s64_nint > 0) {
        HMX_HIP(H.s64_int_off.upload(i64_off));
        HMX_HIP(H.s64_seg_ptr.upload(g64_ptr));
        HMX_HIP(H.s64_seg_src.upload(g64_src));
        HMX_HIP(H.s64_seg_cb.upload(g64_cb));
        HMX_HIP(H.s64_seg_wp.upload(g64_wp));
        HMX_HIP(H.s64_seg_w.upload(g64_w));
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
    H.place_e_known = H.place_r_known = false; // new streams: what an earlier layout of this operator measured about places is void
    H.place_e = H.place_r = -1;
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
        for (auto *a : {&H.s_sub_w, &H.s_sub_nrows, &H.s_sub_dst, &H.s_int_order})
            a->release();
        for (auto *a : {&H.s_sub_ptr, &H.s_sub_src, &H.s_sub_cb})
            a->release();
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
    // per leaf / per pair first indices of the product kernels' index arrays (fill_index_kernel below): uploaded BEFORE the pack kernels are
    // launched, like the pair lists -- a blocking copy behind them would wait for them
    const int64_t zA = H.nS, zP = H.nS + A_total;
    DArr<int32_t> d_ncols, d_z0e, d_o0;
    {
        std::vector<int32_t> ncols(nb), z0(nb);
        for (int64_t b = 0; b < nb; b++) {
            const bool lr = XK[b] == LK_LOWRANK;
            ncols[b]      = lr ? XL[b].rank : XL[b].s_size;
            z0[b]         = lr ? (int32_t)(zA + aoff[b]) : (int32_t)(XL[b].s_offset - H.S0);
        }
        std::vector<int32_t> o0(rlr_b.size());
        parallel_for(rlr_b.size(), [&](size_t lo, size_t hi) {
            for (size_t p = lo; p < hi; p++) {
                const int b = rlr_b[p];
                o0[p] = ns_of[b] == 1 ? (int32_t)(zA + aoff[b]) : (int32_t)(zP + poff[b] + (int64_t)(rlr_r[p] - s_first[b]) * XL[b].rank);
            }
        });
        HMX_HIP(d_ncols.upload(ncols));
        HMX_HIP(d_z0e.upload(z0));
        HMX_HIP(d_o0.upload(o0));
        HMX_HIP(H.e_zidx.alloc(std::max<int64_t>(E.total_cols, 1)));
        HMX_HIP(H.r_outidx.alloc(std::max<int64_t>(R.total_cols, 1)));
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
    H.zero_slot      = H.nS + A_total + P_total;
    // e_zidx / r_outidx: filled on the device from the pair lists (fill_index_kernel), behind the pack kernels on the same stream
    {
        if (!elr_b.empty()) {
            FillIndexArgs F{pk[0].d, pk[1].d, pk[2].d, E.d_colbase.d, d_ncols.d, d_z0e.d, H.e_zidx.d, 0};
            hipLaunchKernelGGL(fill_index_kernel, dim3((unsigned)((elr_b.size() + 3) / 4)), dim3(256), 0, 0, F, (int64_t)elr_b.size());
        }
        if (!ed_b.empty()) {
            FillIndexArgs F{pk[6].d, pk[7].d, pk[8].d, E.d_colbase.d, d_ncols.d, d_z0e.d, H.e_zidx.d, 0};
            hipLaunchKernelGGL(fill_index_kernel, dim3((unsigned)((ed_b.size() + 3) / 4)), dim3(256), 0, 0, F, (int64_t)ed_b.size());
        }
        if (!rlr_b.empty()) {
            FillIndexArgs F{pk[3].d, pk[4].d, pk[5].d, R.d_colbase.d, d_ncols.d, d_o0.d, H.r_outidx.d, 1};
            hipLaunchKernelGGL(fill_index_kernel, dim3((unsigned)((rlr_b.size() + 3) / 4)), dim3(256), 0, 0, F, (int64_t)rlr_b.size());
        }
        HMX_HIP(hipGetLastError());
    }
    { // where the dense leaves' slices sit in the E-streams (hmx_hmatrix_get_block / bulk download: api_get_blocks); leaf-major, the stored leaves only
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
    phase_nosync("  e / r index (device)");
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
        MirrorCtx M{XL, XK, nb, elr_b, elr_r, elr_c, ed_b, ed_r, ed_c, rlr_b, rlr_r, rlr_c, aoff, A_total, false, [&](const char *n) { phase_nosync(n); }, e1};
        const int rcm = build_mirror_tables(H, M);
        if (rcm != HMX_OK)
            return rcm;
    }

    phase_nosync("index arrays");
    // ---- uploads of the index arrays (the first one waits for the pack kernels: same stream) ----------------------------------------
    HMX_HIP(H.c_dst.upload(cd));
    HMX_HIP(H.c_src.upload(cs));
    HMX_HIP(H.c_stride.upload(cst));
    HMX_HIP(H.c_count.upload(cc));
    // Z = [x | a | partial sums]: written by the reduce stage while the R-stream is read (place_written: where that pair runs fastest)
    HMX_HIP(place_array(H, H.Z, (size_t)H.zero_slot + 1, 1, nullptr, &H.placed_z, e1));
    if (phase_timing) // where the arrays of a product lie
        fprintf(stderr, "[hmx build]   arrays: E-stream %p (%.3f GB), R-stream %p (%.3f GB), Z %p (placement probe: R alone %.0f GB/s, with Z at first fit %.0f, where it stays %.0f; %d places tried%s)\n",
                (void *)E.stream.d, E.elems * sizeof(scalar) / 1e9, (void *)R.stream.d, R.elems * sizeof(scalar) / 1e9, (void *)H.Z.d, H.placed_z.read_only, H.placed_z.first, H.placed_z.chosen,
                H.placed_z.tried, H.placed_z.cached ? ", remembered from an earlier build at this place" : "");
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
