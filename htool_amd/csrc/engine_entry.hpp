// engine_entry.hpp -- the product entry points behind the C ABI: host / device vectors, user numbering, row chunks, several right-hand sides, prepare.
// Part of the engine's host code: included by engine_body.hpp inside namespace hmx::{f64,f32,z64,c32}.  No include guard on purpose.

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
    if (H.tmp_out.n < nout) // the product's output while the E-stream is read: the library's own buffer goes where that pair runs fastest (place_written)
        HMX_HIP(place_array(H, H.tmp_out, nout, 0, st));
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
    if (H.tmp_out2.n < (size_t)nout) // (the output in cluster numbering: see with_buffers)
        HMX_HIP(place_array(H, H.tmp_out2, (size_t)nout, 0, st));
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
        HMX_HIP(place_array(H, H.mm_out, tout, 0, st)); // (see with_buffers)
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
    ProbeScope probes(H); // the one place besides the builds where the written arrays' places may be measured (views built below inherit it)
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

// hmx_hmatrix_alloc_vector: an output vector of this operator's products, where its sweeps write fastest.  When `bytes` is the size of an output
// of this operator (rows x a number of right-hand sides) and a reserved slab offers places to choose from, the place is decided by the operator's
// OWN product: the product of that shape (zero operands) is timed with the vector at first fit and at nine places spread over the slab, and the
// vector stays where the product ran fastest (N = 1e6 fp64, 16 right-hand sides: the expand stage takes 2.03 or 2.30 ms depending on nothing
// else -- profiles/r6_modes_output_place.log; the synthetic probe of place_written, made for arrays of 1-2 % of a stream, picked a slow place in
// one process of three).  ~25 products, once, outside any product call of the caller.  Otherwise: place_array (probe allowed here), or plain.
int api_alloc_vector(HMat *Hp, char trans, int64_t bytes, void **ptr) {
    if (!Hp || !ptr || bytes <= 0 || !(trans == 'N' || trans == 'T' || trans == 'C')) {
        set_error("hmx_hmatrix_alloc_vector: invalid arguments");
        return HMX_ERR_INVALID;
    }
    HMat &H = *Hp;
    if (!H.finalized) {
        set_error("hmx_hmatrix_alloc_vector: operator not built");
        return HMX_ERR_STATE;
    }
    HMX_HIP(hipSetDevice(H.device));
    ProbeScope probes(H);
    std::unique_ptr<DArr<scalar>> a(new DArr<scalar>());
    const size_t count = ((size_t)bytes + sizeof(scalar) - 1) / sizeof(scalar);
    const size_t nout = (size_t)(trans == 'N' ? H.nT : H.nS), nin = (size_t)(trans == 'N' ? H.nS : H.nT);
    const int mu = (nout > 0 && count % nout == 0 && count / nout <= 4096) ? (int)(count / nout) : 0;
    bool placed  = false;
    if (mu >= 1 && bytes >= (int64_t(1) << 20) && H.opt.i(HMX_OPT_PLACE_WRITTEN) != 0 && DeviceSlabs::get().owns(H.E.stream.d)) {
        DArr<scalar> in;
        HMX_HIP(in.alloc(nin * (size_t)mu));
        HMX_HIP(in.zero());
        DEvent e0, e1;
        auto product = [&](scalar *out) {
            return mu == 1 ? matvec_device(H, trans, scalar(1), in.d, scalar(0), out, nullptr) : matmat_device(H, trans, scalar(1), in.d, scalar(0), out, mu, nullptr);
        };
        auto timed = [&](scalar *out, float *ms) -> int { // the second of two products
            int rc = product(out);
            if (rc != HMX_OK)
                return rc;
            HMX_HIP(hipEventRecord(e0, nullptr));
            rc = product(out);
            if (rc != HMX_OK)
                return rc;
            HMX_HIP(hipEventRecord(e1, nullptr));
            HMX_HIP(hipEventSynchronize(e1));
            HMX_HIP(hipEventElapsedTime(ms, e0, e1));
            return HMX_OK;
        };
        const bool was_profiling = H.profiling;
        H.profiling              = false;
        HMX_HIP(a->alloc(count));
        float best = 0;
        int rc     = product(a->d); // (whatever the first product of this shape builds and allocates is built now, not inside a timing)
        if (rc == HMX_OK)
            rc = timed(a->d, &best);
        if (rc == HMX_OK && DeviceSlabs::get().owns(a->d)) {
            for (double frac : {0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1.0}) {
                DArr<scalar> cand;
                if (cand.alloc_at(count, frac) != hipSuccess)
                    continue;
                float ms = 0;
                if (timed(cand.d, &ms) != HMX_OK)
                    break;
                if (ms < 0.985f * best) {
                    a->swap(cand);
                    best = ms;
                }
                HMX_HIP(hipDeviceSynchronize()); // before `cand` (the loser) goes back to the slab
            }
        }
        H.profiling = was_profiling;
        if (rc == HMX_OK) {
            HMX_HIP(hipMemset(a->d, 0, count * sizeof(scalar)));
            placed = true;
        } else { // a product this operator refuses (trans = 'C' on 'S' leaves, a transposed view that cannot be built): no reason to refuse the memory
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
            a->release();
        }
    }
    if (!placed) {
        // 'N': y is written by the expand stage (E-streams); transposed on the stored data: by the second sweep over the R-streams.  A transposed
        // product that runs on its own stream layout writes next to THAT layout's E-stream: its operator decides
        HMat &W        = (trans != 'N' && H.T_op) ? *H.T_op : H;
        const int pair = (trans == 'N' || &W != &H) ? 0 : 1;
        if (&W != &H)
            W.may_probe = true;
        const hipError_t e = place_array(W, *a, count, pair, nullptr);
        if (&W != &H)
            W.may_probe = false;
        HMX_HIP(e);
    }
    HMX_HIP(hipDeviceSynchronize());
    *ptr = a->d;
    H.user_vectors[a->d] = std::move(a);
    return HMX_OK;
}
int api_free_vector(HMat *Hp, void *ptr) {
    if (!Hp) {
        set_error("hmx_hmatrix_free_vector: NULL operator");
        return HMX_ERR_INVALID;
    }
    auto it = Hp->user_vectors.find(ptr);
    if (it == Hp->user_vectors.end()) {
        set_error("hmx_hmatrix_free_vector: not a vector of this operator");
        return HMX_ERR_INVALID;
    }
    HMX_HIP(hipSetDevice(Hp->device));
    Hp->user_vectors.erase(it);
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
