// engine_build.hpp -- creating an operator, its options and generator, compression (device kernel, host generator on all cores), recompression, uploaded blocks, finalize.
// Part of the engine's host code: included by engine_body.hpp inside namespace hmx::{f64,f32,z64,c32}.  No include guard on purpose.

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
    int rc;
    {
        ProbeScope probes(H);
        rc = build_streams(H);
    }
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
    ProbeScope probes(H);
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
    ProbeScope probes(H);
    return build_streams(H);
}
