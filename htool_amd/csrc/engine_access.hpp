// engine_access.hpp -- reading an operator back: ranks, blocks, bulk download, save / load, released factors, statistics.
// Part of the engine's host code: included by engine_body.hpp inside namespace hmx::{f64,f32,z64,c32}.  No include guard on purpose.

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
    // the block's slices: (leaf, range, first column) triples, leaf-major (build_streams)
    const size_t q0 = (size_t)(std::lower_bound(H->dp_leaf.begin(), H->dp_leaf.end(), (int32_t)leaf) - H->dp_leaf.begin());
    if (q0 >= H->dp_leaf.size() || H->dp_leaf[q0] != (int32_t)leaf) {
        set_error("hmx_hmatrix_get_block: internal lookup failed");
        return HMX_ERR_STATE;
    }
    for (size_t q = q0; q < H->dp_leaf.size() && H->dp_leaf[q] == (int32_t)leaf; q++) {
        const int r = H->dp_range[q], col = H->dp_col[q];
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
        out->transposed_bytes += (int64_t)((H->s_mdst.n + H->s_coef.n + H->s_count.n + H->sc_dst.n + H->sc_lp.n + H->sc_count.n + H->sc_k.n + H->s_list.n + H->s_fidx.n + H->s_sub_w.n +
                                            H->s_sub_nrows.n + H->s_sub_dst.n + H->s_int_order.n) * sizeof(int32_t) + (H->s_sub_ptr.n + H->s_sub_src.n + H->s_sub_cb.n) * sizeof(int64_t) + H->SW.n * sizeof(scalar));
    out->expanded_bytes   = H->X_op ? H->X_op->stats.stream_bytes : 0;
    out->placed_read_gbps = H->placed_z.read_only, out->placed_first_gbps = H->placed_z.first, out->placed_gbps = H->placed_z.chosen, out->placed_tried = H->placed_z.tried;
    return HMX_OK;
}
