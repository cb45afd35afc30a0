// engine.hip -- the C ABI (include/hmx.h) of the device side of libhmx and the DistributedOperator layer over RCCL.  The engine
// that owns the HBM-resident operator (compression, packing, H-matvec) is engine_body.hpp + kernels_body.hpp, compiled once per
// coefficient type in engine_inst.hip; this file dispatches on the type a handle holds.
//
// There is deliberately NO CPU fallback in this file: every entry point that computes needs a HIP
// device and reports HMX_ERR_NO_DEVICE / HMX_ERR_HIP otherwise.
#include "engine_common.hpp"
#include "kernels_common.hpp"

namespace hmx {
namespace f64 {
using scalar = double;
#include "engine_api.hpp"
} // namespace f64
namespace f32 {
using scalar = float;
#include "engine_api.hpp"
} // namespace f32
namespace z64 {
using scalar = cplx<double>;
#include "engine_api.hpp"
} // namespace z64
namespace c32 {
using scalar = cplx<float>;
#include "engine_api.hpp"
} // namespace c32
} // namespace hmx

using namespace hmx;

// bandwidth probes of hmx_device_copy_bandwidth / hmx_device_read_bandwidth (and the one-workgroup launch of hmx_device_init)
namespace hmx {
static __global__ __launch_bounds__(256) void copy16_kernel(const double2 *__restrict__ in, double2 *__restrict__ out, int64_t n) {
    // every workgroup copies its own contiguous chunk, four 16-byte non-temporal loads in flight per lane, then the four stores (the first
    // version moved one element per grid-stride step and reported 4.7-5.0 TB/s where the chip copies at 6 TB/s and more)
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    int64_t i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256) {
        const double2 a = stream_load(in + i), b = stream_load(in + i + 256), c = stream_load(in + i + 512), d = stream_load(in + i + 768);
        __builtin_nontemporal_store(a.x, &out[i].x), __builtin_nontemporal_store(a.y, &out[i].y);
        __builtin_nontemporal_store(b.x, &out[i + 256].x), __builtin_nontemporal_store(b.y, &out[i + 256].y);
        __builtin_nontemporal_store(c.x, &out[i + 512].x), __builtin_nontemporal_store(c.y, &out[i + 512].y);
        __builtin_nontemporal_store(d.x, &out[i + 768].x), __builtin_nontemporal_store(d.y, &out[i + 768].y);
    }
    for (; i < hi; i += 256)
        out[i] = in[i];
}

// read-only probe: every workgroup streams its own contiguous chunk with 16-byte non-temporal loads (four in flight per lane)
// and keeps running sums -- the access pattern of the stream kernels without any of their arithmetic or gathers
static __global__ __launch_bounds__(256) void read16_kernel(const double2 *__restrict__ in, double *__restrict__ out, int64_t n) {
    const int64_t per = n / gridDim.x;
    const double2 *p  = in + per * blockIdx.x;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t i = threadIdx.x; i + 3 * 256 < per; i += 4 * 256) {
        const double2 a = stream_load(p + i), b = stream_load(p + i + 256), c = stream_load(p + i + 512), d = stream_load(p + i + 768);
        s0 += a.x + a.y;
        s1 += b.x + b.y;
        s2 += c.x + c.y;
        s3 += d.x + d.y;
    }
    out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = (s0 + s1) + (s2 + s3);
}

// placement probe (hmx_placement_probe): the streaming read above over `nslices` slices spread evenly over a stream, while every wave
// stores 512 bytes to the candidate array every second step (1.6 % of the bytes it reads: the ratio of the write streams of the products)
static __global__ __launch_bounds__(256) void placement_probe_kernel(const double2 *__restrict__ in, int64_t n_total, int nslices, int64_t slice_elems, double *__restrict__ out,
                                                                     int64_t out_elems) {
    const int per_slice = gridDim.x / nslices, sl = blockIdx.x / per_slice, bl = blockIdx.x % per_slice;
    if (sl >= nslices)
        return;
    const int64_t slice0 = nslices > 1 ? (n_total - slice_elems) / (nslices - 1) * sl : 0;
    const int64_t per    = slice_elems / per_slice;
    const double2 *p     = in + slice0 + per * bl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t runs = out_elems / 64; // 512-byte runs of the candidate
    int64_t run        = ((int64_t)blockIdx.x * 4 + wave) * (runs / ((int64_t)gridDim.x * 4) > 0 ? runs / ((int64_t)gridDim.x * 4) : 1);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int64_t it = 0;
    for (int64_t i = threadIdx.x; i + 3 * 256 < per; i += 4 * 256, it++) {
        const double2 a = stream_load(p + i), b = stream_load(p + i + 256), c = stream_load(p + i + 512), d = stream_load(p + i + 768);
        s0 += a.x + a.y;
        s1 += b.x + b.y;
        s2 += c.x + c.y;
        s3 += d.x + d.y;
        if (out && (it & 1) == 0) {
            out[(run % runs) * 64 + lane] = (s0 + s1) + (s2 + s3);
            run++;
        }
    }
    if (s0 == 12345.678 && out)
        out[lane] = s1;
}

// GB/s of a streaming read of (a sample of) [stream, stream + stream_bytes) while a small write stream goes to [cand, cand + cand_bytes);
// cand = nullptr: the read alone.  On MI355X the pair runs 12-16 % faster when the two lie in different thirds of the physical memory
// (tools/placement_rw.hip) -- which third a virtual address belongs to is the driver's business, so it is measured.
// The candidate's contents are overwritten.
double placement_probe(const void *stream, size_t stream_bytes, void *cand, size_t cand_bytes, hipStream_t st, hipEvent_t after) {
    const int nslices = 8, grid = 2048;
    int64_t n_total   = (int64_t)(stream_bytes / 16);
    int64_t slice     = std::min<int64_t>(n_total / nslices, (int64_t)(256u << 20) / 16); // <= 8 x 256 MiB read per launch
    slice             = slice / (grid / nslices * 1024) * (grid / nslices * 1024);
    if (slice <= 0 || (cand && cand_bytes < (size_t)grid * 4 * 512))
        return 0.0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return 0.0;
    if (after)
        (void)hipStreamWaitEvent(st, after, 0);
    auto launch = [&]() {
        hipLaunchKernelGGL(placement_probe_kernel, dim3(grid), dim3(256), 0, st, (const double2 *)stream, n_total, nslices, slice, (double *)cand, (int64_t)(cand_bytes / 8));
    };
    launch();
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < 3; r++)
        launch();
    (void)hipEventRecord(e1, st);
    float ms = 0;
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ok ? 3.0 * (double)slice * nslices * 16 / (ms * 1e-3) / 1e9 : 0.0;
}

} // namespace hmx

// The opaque handle of the C ABI: one of the four instantiations
struct hmx_hmatrix {
    hmx::f64::HMat *d = nullptr;
    hmx::f32::HMat *s = nullptr;
    hmx::z64::HMat *z = nullptr;
    hmx::c32::HMat *c = nullptr;
    ~hmx_hmatrix() {
        hmx::f64::api_destroy(d);
        hmx::f32::api_destroy(s);
        hmx::z64::api_destroy(z);
        hmx::c32::api_destroy(c);
    }
};

// The C ABI never throws: host-side allocation failures (std::bad_alloc, std::length_error from a corrupt size) and anything
// else that escapes the engine become an error code + hmx_last_error().
#define HMX_GUARD(expr)                                                        \
    do {                                                                       \
        try {                                                                  \
            return (expr);                                                     \
        } catch (const std::bad_alloc &) {                                     \
            set_error("out of host memory");                                   \
            return HMX_ERR_INVALID;                                            \
        } catch (const std::exception &e_) {                                   \
            set_error(std::string("internal error: ") + e_.what());            \
            return HMX_ERR_INVALID;                                            \
        } catch (...) {                                                        \
            set_error("internal error");                                       \
            return HMX_ERR_INVALID;                                            \
        }                                                                      \
    } while (0)
// type-independent entry points: the same api_* function in whichever instantiation the handle holds
#define HMX_ALL(H, fn, ...)                                 \
    do {                                                    \
        if (!(H)) {                                         \
            set_error("NULL hmx_hmatrix handle");           \
            return HMX_ERR_INVALID;                         \
        }                                                   \
        if ((H)->d)                                         \
            HMX_GUARD(hmx::f64::fn((H)->d, ##__VA_ARGS__)); \
        if ((H)->s)                                         \
            HMX_GUARD(hmx::f32::fn((H)->s, ##__VA_ARGS__)); \
        if ((H)->z)                                         \
            HMX_GUARD(hmx::z64::fn((H)->z, ##__VA_ARGS__)); \
        HMX_GUARD(hmx::c32::fn((H)->c, ##__VA_ARGS__));     \
    } while (0)
#define HMX_NEED(H, member, what)                                                                     \
    do {                                                                                              \
        if (!(H) || !(H)->member) {                                                                   \
            set_error(std::string(what) + ": handle holds another coefficient type (plain entry points take double, *_s float, *_z complex double, *_c complex float)"); \
            return HMX_ERR_INVALID;                                                                   \
        }                                                                                             \
    } while (0)
// complex values cross the C ABI as interleaved (re, im) pairs
static inline cplx<double> zval(const double *p) { return cplx<double>(p[0], p[1]); }
static inline cplx<float> cval(const float *p) { return cplx<float>(p[0], p[1]); }
#define ZP(p) reinterpret_cast<const cplx<double> *>(p)
#define ZPM(p) reinterpret_cast<cplx<double> *>(p)
#define CP(p) reinterpret_cast<const cplx<float> *>(p)
#define CPM(p) reinterpret_cast<cplx<float> *>(p)

extern "C" {

int hmx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

// One-time cost of the first call into the device side of the library: HIP context creation and the load of libhmx's code object
// (every kernel of four coefficient types: ~0.2 s).  Callers that time operator builds call this first.
int hmx_device_init(int device_id) {
    const int rc = ensure_device(device_id);
    if (rc != HMX_OK)
        return rc;
    DArr<double2> a;
    HMX_HIP(a.alloc(64));
    hipLaunchKernelGGL(copy16_kernel, dim3(1), dim3(64), 0, 0, (const double2 *)a.d, a.d, (int64_t)0);
    HMX_HIP(hipGetLastError());
    HMX_HIP(hipDeviceSynchronize());
    return HMX_OK;
}

int hmx_hmatrix_create(const hmx_block_tree *bt, int device_id, hmx_hmatrix **out) {
    if (!out) {
        set_error("hmx_hmatrix_create: out is NULL");
        return HMX_ERR_INVALID;
    }
    hmx::f64::HMat *h = nullptr;
    int rc            = HMX_ERR_INVALID;
    try {
        rc = hmx::f64::api_create(bt, device_id, &h);
    } catch (...) {
        set_error("hmx_hmatrix_create: out of host memory");
    }
    if (rc != HMX_OK)
        return rc;
    *out      = new hmx_hmatrix();
    (*out)->d = h;
    return HMX_OK;
}
int hmx_hmatrix_create_s(const hmx_block_tree *bt, int device_id, hmx_hmatrix **out) {
    if (!out) {
        set_error("hmx_hmatrix_create_s: out is NULL");
        return HMX_ERR_INVALID;
    }
    hmx::f32::HMat *h = nullptr;
    int rc            = HMX_ERR_INVALID;
    try {
        rc = hmx::f32::api_create(bt, device_id, &h);
    } catch (...) {
        set_error("hmx_hmatrix_create: out of host memory");
    }
    if (rc != HMX_OK)
        return rc;
    *out      = new hmx_hmatrix();
    (*out)->s = h;
    return HMX_OK;
}
int hmx_hmatrix_create_z(const hmx_block_tree *bt, int device_id, hmx_hmatrix **out) {
    if (!out) {
        set_error("hmx_hmatrix_create_z: out is NULL");
        return HMX_ERR_INVALID;
    }
    hmx::z64::HMat *h = nullptr;
    int rc            = HMX_ERR_INVALID;
    try {
        rc = hmx::z64::api_create(bt, device_id, &h);
    } catch (...) {
        set_error("hmx_hmatrix_create: out of host memory");
    }
    if (rc != HMX_OK)
        return rc;
    *out      = new hmx_hmatrix();
    (*out)->z = h;
    return HMX_OK;
}
int hmx_hmatrix_create_c(const hmx_block_tree *bt, int device_id, hmx_hmatrix **out) {
    if (!out) {
        set_error("hmx_hmatrix_create_c: out is NULL");
        return HMX_ERR_INVALID;
    }
    hmx::c32::HMat *h = nullptr;
    int rc            = HMX_ERR_INVALID;
    try {
        rc = hmx::c32::api_create(bt, device_id, &h);
    } catch (...) {
        set_error("hmx_hmatrix_create: out of host memory");
    }
    if (rc != HMX_OK)
        return rc;
    *out      = new hmx_hmatrix();
    (*out)->c = h;
    return HMX_OK;
}
void hmx_hmatrix_destroy(hmx_hmatrix *H) {
    if (!H)
        return;
    // products are asynchronous on the caller's stream: nothing of this operator may be freed or recycled while one is in flight
    const int dev = H->d ? hmx::f64::api_device_of(H->d) : (H->s ? hmx::f32::api_device_of(H->s) : (H->z ? hmx::z64::api_device_of(H->z) : (H->c ? hmx::c32::api_device_of(H->c) : -1)));
    int cur       = 0;
    if (dev >= 0 && hipGetDevice(&cur) == hipSuccess) {
        if (cur != dev)
            (void)hipSetDevice(dev);
        (void)hipDeviceSynchronize();
        if (cur != dev)
            (void)hipSetDevice(cur);
    }
    delete H;
}
int hmx_hmatrix_is_f32(const hmx_hmatrix *H) { return H && H->s ? 1 : 0; }
int hmx_hmatrix_precision(const hmx_hmatrix *H) { return !H ? -1 : (H->d ? HMX_PREC_F64 : (H->s ? HMX_PREC_F32 : (H->z ? HMX_PREC_Z64 : HMX_PREC_C32))); }

int hmx_hmatrix_set_kernel(hmx_hmatrix *H, int kernel, const double *params, int nparams, int dim, const double *tc, const double *sc) {
    HMX_ALL(H, api_set_kernel, kernel, params, nparams, dim, tc, sc);
}
int hmx_hmatrix_set_callback(hmx_hmatrix *H, hmx_generator_fn fn, void *user) {
    HMX_NEED(H, d, "hmx_hmatrix_set_callback");
    HMX_GUARD(hmx::f64::api_set_callback(H->d, fn, user));
}
int hmx_hmatrix_set_callback_s(hmx_hmatrix *H, hmx_generator_fn_s fn, void *user) {
    HMX_NEED(H, s, "hmx_hmatrix_set_callback_s");
    HMX_GUARD(hmx::f32::api_set_callback(H->s, fn, user));
}
int hmx_hmatrix_set_callback_threads(hmx_hmatrix *H, int threads) { HMX_ALL(H, api_set_callback_threads, threads); }
int hmx_hmatrix_set_option(hmx_hmatrix *H, int option, double value) { HMX_ALL(H, api_set_option, option, value); }
int hmx_hmatrix_get_option(const hmx_hmatrix *H, int option, double *value) { HMX_ALL(H, api_get_option, option, value); }
int hmx_hmatrix_compress(hmx_hmatrix *H, int compressor, double epsilon, int reqrank) {
    HMX_ALL(H, api_compress, compressor, epsilon, reqrank);
}
int hmx_hmatrix_recompress(hmx_hmatrix *H, double epsilon) { HMX_ALL(H, api_recompress, epsilon); }
int hmx_hmatrix_finalize(hmx_hmatrix *H) { HMX_ALL(H, api_finalize); }
int hmx_hmatrix_leaf_ranks(const hmx_hmatrix *H, int32_t *rank) { HMX_ALL(H, api_leaf_ranks, rank); }
static int stats_full(const hmx_hmatrix *H, hmx_stats *out) { HMX_ALL(H, api_stats, out); }
int hmx_abi_version(void) { return HMX_ABI_VERSION; }
int hmx_hmatrix_stats_sized(const hmx_hmatrix *H, hmx_stats *out, size_t struct_size) {
    if (!out || struct_size == 0) {
        hmx::set_error("hmx_hmatrix_stats: NULL argument");
        return HMX_ERR_INVALID;
    }
    hmx_stats full{};
    const int rc = stats_full(H, &full);
    if (rc != HMX_OK)
        return rc;
    std::memcpy(out, &full, std::min(struct_size, sizeof(hmx_stats))); // a caller with an older, shorter struct gets its prefix
    return HMX_OK;
}
#undef hmx_hmatrix_stats
// binaries built before hmx_hmatrix_stats became a macro over the sized call: the struct as the last header without the macro declared it
// (everything before placed_read_gbps -- transposed_bytes and expanded_bytes included: that header had them)
int hmx_hmatrix_stats(const hmx_hmatrix *H, hmx_stats *out) { return hmx_hmatrix_stats_sized(H, out, offsetof(hmx_stats, placed_read_gbps)); }
int hmx_hmatrix_set_profiling(hmx_hmatrix *H, int enabled) { HMX_ALL(H, api_set_profiling, enabled); }
int hmx_hmatrix_last_kernel_times(const hmx_hmatrix *H, int max, const char **names, float *ms) {
    if (!H)
        return 0;
    HMX_ALL(H, api_last_kernel_times, max, names, ms);
}

// ---- fp64 coefficients ----------------------------------------------------------------------------------------
int hmx_hmatrix_set_block_lowrank(hmx_hmatrix *H, int64_t leaf, int rank, const double *U, const double *V) {
    HMX_NEED(H, d, "hmx_hmatrix_set_block_lowrank");
    HMX_GUARD(hmx::f64::api_set_block_lowrank(H->d, leaf, rank, U, V));
}
int hmx_hmatrix_set_block_dense(hmx_hmatrix *H, int64_t leaf, const double *D) {
    HMX_NEED(H, d, "hmx_hmatrix_set_block_dense");
    HMX_GUARD(hmx::f64::api_set_block_dense(H->d, leaf, D));
}
int hmx_hmatrix_get_block(const hmx_hmatrix *H, int64_t leaf, double *U_or_D, double *V) {
    HMX_NEED(H, d, "hmx_hmatrix_get_block");
    HMX_GUARD(hmx::f64::api_get_block(H->d, leaf, U_or_D, V));
}
int hmx_hmatrix_get_blocks(const hmx_hmatrix *H, int64_t count, const int64_t *leaves, double *const *U_or_D, double *const *V) {
    HMX_NEED(H, d, "hmx_hmatrix_get_blocks");
    HMX_GUARD(hmx::f64::api_get_blocks(H->d, count, leaves, U_or_D, V));
}
int hmx_hmatrix_get_blocks_s(const hmx_hmatrix *H, int64_t count, const int64_t *leaves, float *const *U_or_D, float *const *V) {
    HMX_NEED(H, s, "hmx_hmatrix_get_blocks_s");
    HMX_GUARD(hmx::f32::api_get_blocks(H->s, count, leaves, U_or_D, V));
}
int hmx_hmatrix_get_blocks_z(const hmx_hmatrix *H, int64_t count, const int64_t *leaves, double *const *U_or_D, double *const *V) {
    HMX_NEED(H, z, "hmx_hmatrix_get_blocks_z");
    HMX_GUARD(hmx::z64::api_get_blocks(H->z, count, leaves, reinterpret_cast<cplx<double> *const *>(U_or_D), reinterpret_cast<cplx<double> *const *>(V)));
}
int hmx_hmatrix_get_blocks_c(const hmx_hmatrix *H, int64_t count, const int64_t *leaves, float *const *U_or_D, float *const *V) {
    HMX_NEED(H, c, "hmx_hmatrix_get_blocks_c");
    HMX_GUARD(hmx::c32::api_get_blocks(H->c, count, leaves, reinterpret_cast<cplx<float> *const *>(U_or_D), reinterpret_cast<cplx<float> *const *>(V)));
}
int hmx_hmatrix_matvec(hmx_hmatrix *H, char trans, double alpha, const double *in, double beta, double *out, int mem, void *stream) {
    HMX_NEED(H, d, "hmx_hmatrix_matvec");
    HMX_GUARD(hmx::f64::api_matvec(H->d, trans, alpha, in, beta, out, mem, stream));
}
int hmx_hmatrix_matvec_user(hmx_hmatrix *H, char trans, double alpha, const double *in, double beta, double *out, int mem, void *stream) {
    HMX_NEED(H, d, "hmx_hmatrix_matvec_user");
    HMX_GUARD(hmx::f64::api_matvec_user(H->d, trans, alpha, in, beta, out, mem, stream));
}
int hmx_hmatrix_matmat_row_major(hmx_hmatrix *H, char trans, double alpha, const double *in, double beta, double *out, int mu, int mem, void *stream) {
    HMX_NEED(H, d, "hmx_hmatrix_matmat_row_major");
    HMX_GUARD(hmx::f64::api_matmat_row_major(H->d, trans, alpha, in, beta, out, mu, mem, stream));
}

int hmx_hmatrix_matmat_user(hmx_hmatrix *H, char trans, double alpha, const double *in, double beta, double *out, int mu, int mem, void *stream) {
    HMX_NEED(H, d, "hmx_hmatrix_matmat_user");
    HMX_GUARD(hmx::f64::api_matmat_user(H->d, trans, alpha, in, beta, out, mu, mem, stream));
}
int hmx_hmatrix_matmat_user_s(hmx_hmatrix *H, char trans, float alpha, const float *in, float beta, float *out, int mu, int mem, void *stream) {
    HMX_NEED(H, s, "hmx_hmatrix_matmat_user_s");
    HMX_GUARD(hmx::f32::api_matmat_user(H->s, trans, alpha, in, beta, out, mu, mem, stream));
}
int hmx_hmatrix_matmat_user_z(hmx_hmatrix *H, char trans, const double *alpha, const double *in, const double *beta, double *out, int mu, int mem, void *stream) {
    HMX_NEED(H, z, "hmx_hmatrix_matmat_user_z");
    HMX_GUARD(hmx::z64::api_matmat_user(H->z, trans, zval(alpha), ZP(in), zval(beta), ZPM(out), mu, mem, stream));
}
int hmx_hmatrix_matmat_user_c(hmx_hmatrix *H, char trans, const float *alpha, const float *in, const float *beta, float *out, int mu, int mem, void *stream) {
    HMX_NEED(H, c, "hmx_hmatrix_matmat_user_c");
    HMX_GUARD(hmx::c32::api_matmat_user(H->c, trans, cval(alpha), CP(in), cval(beta), CPM(out), mu, mem, stream));
}

// ---- fp32 coefficients (htool's HMatrix<float,double>) --------------------------------------------------------------
int hmx_hmatrix_set_block_lowrank_s(hmx_hmatrix *H, int64_t leaf, int rank, const float *U, const float *V) {
    HMX_NEED(H, s, "hmx_hmatrix_set_block_lowrank_s");
    HMX_GUARD(hmx::f32::api_set_block_lowrank(H->s, leaf, rank, U, V));
}
int hmx_hmatrix_set_block_dense_s(hmx_hmatrix *H, int64_t leaf, const float *D) {
    HMX_NEED(H, s, "hmx_hmatrix_set_block_dense_s");
    HMX_GUARD(hmx::f32::api_set_block_dense(H->s, leaf, D));
}
int hmx_hmatrix_get_block_s(const hmx_hmatrix *H, int64_t leaf, float *U_or_D, float *V) {
    HMX_NEED(H, s, "hmx_hmatrix_get_block_s");
    HMX_GUARD(hmx::f32::api_get_block(H->s, leaf, U_or_D, V));
}
int hmx_hmatrix_matvec_s(hmx_hmatrix *H, char trans, float alpha, const float *in, float beta, float *out, int mem, void *stream) {
    HMX_NEED(H, s, "hmx_hmatrix_matvec_s");
    HMX_GUARD(hmx::f32::api_matvec(H->s, trans, alpha, in, beta, out, mem, stream));
}
int hmx_hmatrix_matvec_user_s(hmx_hmatrix *H, char trans, float alpha, const float *in, float beta, float *out, int mem, void *stream) {
    HMX_NEED(H, s, "hmx_hmatrix_matvec_user_s");
    HMX_GUARD(hmx::f32::api_matvec_user(H->s, trans, alpha, in, beta, out, mem, stream));
}
int hmx_hmatrix_matmat_row_major_s(hmx_hmatrix *H, char trans, float alpha, const float *in, float beta, float *out, int mu, int mem, void *stream) {
    HMX_NEED(H, s, "hmx_hmatrix_matmat_row_major_s");
    HMX_GUARD(hmx::f32::api_matmat_row_major(H->s, trans, alpha, in, beta, out, mu, mem, stream));
}

// ---- complex coefficients (htool's HMatrix<std::complex<double>> / <std::complex<float>>): interleaved (re, im) ------------
int hmx_hmatrix_set_callback_z(hmx_hmatrix *H, hmx_generator_fn fn, void *user) {
    HMX_NEED(H, z, "hmx_hmatrix_set_callback_z");
    HMX_GUARD(hmx::z64::api_set_callback(H->z, reinterpret_cast<void (*)(void *, int, int, const int32_t *, const int32_t *, cplx<double> *)>(fn), user));
}
int hmx_hmatrix_set_callback_c(hmx_hmatrix *H, hmx_generator_fn_s fn, void *user) {
    HMX_NEED(H, c, "hmx_hmatrix_set_callback_c");
    HMX_GUARD(hmx::c32::api_set_callback(H->c, reinterpret_cast<void (*)(void *, int, int, const int32_t *, const int32_t *, cplx<float> *)>(fn), user));
}
int hmx_hmatrix_set_block_lowrank_z(hmx_hmatrix *H, int64_t leaf, int rank, const double *U, const double *V) {
    HMX_NEED(H, z, "hmx_hmatrix_set_block_lowrank_z");
    HMX_GUARD(hmx::z64::api_set_block_lowrank(H->z, leaf, rank, ZP(U), ZP(V)));
}
int hmx_hmatrix_set_block_dense_z(hmx_hmatrix *H, int64_t leaf, const double *D) {
    HMX_NEED(H, z, "hmx_hmatrix_set_block_dense_z");
    HMX_GUARD(hmx::z64::api_set_block_dense(H->z, leaf, ZP(D)));
}
int hmx_hmatrix_get_block_z(const hmx_hmatrix *H, int64_t leaf, double *U_or_D, double *V) {
    HMX_NEED(H, z, "hmx_hmatrix_get_block_z");
    HMX_GUARD(hmx::z64::api_get_block(H->z, leaf, ZPM(U_or_D), ZPM(V)));
}
int hmx_hmatrix_matvec_z(hmx_hmatrix *H, char trans, const double *alpha, const double *in, const double *beta, double *out, int mem, void *stream) {
    HMX_NEED(H, z, "hmx_hmatrix_matvec_z");
    HMX_GUARD(hmx::z64::api_matvec(H->z, trans, zval(alpha), ZP(in), zval(beta), ZPM(out), mem, stream));
}
int hmx_hmatrix_matvec_user_z(hmx_hmatrix *H, char trans, const double *alpha, const double *in, const double *beta, double *out, int mem, void *stream) {
    HMX_NEED(H, z, "hmx_hmatrix_matvec_user_z");
    HMX_GUARD(hmx::z64::api_matvec_user(H->z, trans, zval(alpha), ZP(in), zval(beta), ZPM(out), mem, stream));
}
int hmx_hmatrix_matmat_row_major_z(hmx_hmatrix *H, char trans, const double *alpha, const double *in, const double *beta, double *out, int mu, int mem, void *stream) {
    HMX_NEED(H, z, "hmx_hmatrix_matmat_row_major_z");
    HMX_GUARD(hmx::z64::api_matmat_row_major(H->z, trans, zval(alpha), ZP(in), zval(beta), ZPM(out), mu, mem, stream));
}
int hmx_hmatrix_set_block_lowrank_c(hmx_hmatrix *H, int64_t leaf, int rank, const float *U, const float *V) {
    HMX_NEED(H, c, "hmx_hmatrix_set_block_lowrank_c");
    HMX_GUARD(hmx::c32::api_set_block_lowrank(H->c, leaf, rank, CP(U), CP(V)));
}
int hmx_hmatrix_set_block_dense_c(hmx_hmatrix *H, int64_t leaf, const float *D) {
    HMX_NEED(H, c, "hmx_hmatrix_set_block_dense_c");
    HMX_GUARD(hmx::c32::api_set_block_dense(H->c, leaf, CP(D)));
}
int hmx_hmatrix_get_block_c(const hmx_hmatrix *H, int64_t leaf, float *U_or_D, float *V) {
    HMX_NEED(H, c, "hmx_hmatrix_get_block_c");
    HMX_GUARD(hmx::c32::api_get_block(H->c, leaf, CPM(U_or_D), CPM(V)));
}
int hmx_hmatrix_matvec_c(hmx_hmatrix *H, char trans, const float *alpha, const float *in, const float *beta, float *out, int mem, void *stream) {
    HMX_NEED(H, c, "hmx_hmatrix_matvec_c");
    HMX_GUARD(hmx::c32::api_matvec(H->c, trans, cval(alpha), CP(in), cval(beta), CPM(out), mem, stream));
}
int hmx_hmatrix_matvec_user_c(hmx_hmatrix *H, char trans, const float *alpha, const float *in, const float *beta, float *out, int mem, void *stream) {
    HMX_NEED(H, c, "hmx_hmatrix_matvec_user_c");
    HMX_GUARD(hmx::c32::api_matvec_user(H->c, trans, cval(alpha), CP(in), cval(beta), CPM(out), mem, stream));
}
int hmx_hmatrix_matmat_row_major_c(hmx_hmatrix *H, char trans, const float *alpha, const float *in, const float *beta, float *out, int mu, int mem, void *stream) {
    HMX_NEED(H, c, "hmx_hmatrix_matmat_row_major_c");
    HMX_GUARD(hmx::c32::api_matmat_row_major(H->c, trans, cval(alpha), CP(in), cval(beta), CPM(out), mu, mem, stream));
}

int hmx_hmatrix_prepare(hmx_hmatrix *H, char trans, int mu) { HMX_ALL(H, api_prepare, trans, mu); }
int hmx_hmatrix_alloc_vector(hmx_hmatrix *H, char trans, int64_t bytes, void **ptr) { HMX_ALL(H, api_alloc_vector, trans, bytes, ptr); }
int hmx_hmatrix_free_vector(hmx_hmatrix *H, void *ptr) { HMX_ALL(H, api_free_vector, ptr); }
int hmx_hmatrix_release_factors(hmx_hmatrix *H, int with_transposed) { HMX_ALL(H, api_release_factors, with_transposed); }
int hmx_hmatrix_save(const hmx_hmatrix *H, const char *path) { HMX_ALL(H, api_save, path); }
int hmx_hmatrix_load(const hmx_block_tree *bt, int device_id, const char *path, hmx_hmatrix **out) {
    if (!bt || !path || !out) {
        set_error("hmx_hmatrix_load: invalid arguments");
        return HMX_ERR_INVALID;
    }
    FILE *f = fopen(path, "rb");
    if (!f) {
        set_error(std::string("hmx_hmatrix_load: cannot open ") + path);
        return HMX_ERR_INVALID;
    }
    hmx::HmxFileHeader hd;
    if (fread(&hd, sizeof hd, 1, f) != 1 || std::memcmp(hd.magic, hmx::HMX_FILE_MAGIC, 8) != 0 || (hd.elem_size != 4 && hd.elem_size != 8 && hd.elem_size != 16)) {
        fclose(f);
        set_error(std::string("hmx_hmatrix_load: ") + path + " is not an hmx operator file");
        return HMX_ERR_INVALID;
    }
    hmx::f64::HMat *d = nullptr;
    hmx::f32::HMat *s = nullptr;
    hmx::z64::HMat *z = nullptr;
    hmx::c32::HMat *c = nullptr;
    int rc = HMX_ERR_INVALID;
    try {
    if (hd.elem_size == 8 && !hd.reserved) {
        rc = hmx::f64::api_load(bt, device_id, f, hd, &d);
    } else if (hd.elem_size == 4) {
        rc = hmx::f32::api_load(bt, device_id, f, hd, &s);
    } else if (hd.elem_size == 16) {
        rc = hmx::z64::api_load(bt, device_id, f, hd, &z);
    } else {
        rc = hmx::c32::api_load(bt, device_id, f, hd, &c);
    }
    } catch (...) { // a corrupt size field: std::bad_alloc / std::length_error from a host buffer
        set_error(std::string("hmx_hmatrix_load: ") + path + " is corrupt (allocation failed)");
        rc = HMX_ERR_INVALID;
    }
    fclose(f);
    if (rc != HMX_OK)
        return rc;
    *out      = new hmx_hmatrix();
    (*out)->d = d;
    (*out)->s = s;
    (*out)->z = z;
    (*out)->c = c;
    return HMX_OK;
}

int hmx_device_copy_bandwidth(int device_id, int64_t bytes, int reps, double *gbps) {
    int rc = ensure_device(device_id);
    if (rc != HMX_OK)
        return rc;
    DArr<double2> a, b;
    const int64_t n = bytes / 16;
    HMX_HIP(a.alloc(n));
    HMX_HIP(b.alloc(n));
    HMX_HIP(a.zero());
    DEvent e0, e1;
    hipLaunchKernelGGL(copy16_kernel, dim3(4096), dim3(256), 0, 0, (const double2 *)a.d, b.d, n);
    HMX_HIP(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(copy16_kernel, dim3(4096), dim3(256), 0, 0, (const double2 *)a.d, b.d, n);
    HMX_HIP(hipEventRecord(e1, 0));
    HMX_HIP(hipEventSynchronize(e1));
    float ms = 0;
    HMX_HIP(hipEventElapsedTime(&ms, e0, e1));
    *gbps = 2.0 * (double)n * 16 * reps / (ms * 1e-3) / 1e9;
    return HMX_OK;
}

// ---- DistributedOperator over RCCL ------------------------------------------------------------------------------------
} // extern "C" (reopened below)
#include <dlfcn.h>
struct hmx_dist {
    hmx_hmatrix *local = nullptr; // the global-to-local operator (the rank's block rows), may be absent
    hmx_hmatrix *diag  = nullptr; // a local-to-local operator (hmx_dist_add_local_to_local_operator: the block-diagonal H-matrix of partition `rank`)
    // DistributedOperator holds VECTORS of both kinds and loops over them (distributed_operator.hpp:47-53, global_to_global.hpp:63-72): the
    // operators registered after the first of each kind (hmx_dist_add_global_to_local_operator, further hmx_dist_add_local_to_local_operator calls)
    std::vector<hmx_hmatrix *> more_local, more_diag;
    bool one_operator() const { return local && !diag && more_local.empty() && more_diag.empty(); } // what the chunked / overlapped exchange is built for
    void *comm         = nullptr;
    int rank = 0, world = 1;
    hmx_rccl_api api{};
    std::vector<int> t_off, t_size, s_off, s_size; // partitions of the target / source cluster trees
    int nt = 0, ns = 0;
    // user numbering <-> partition numbering (PartitionFromCluster::global_to_partition_numbering / local_to_local_partition_numbering,
    // distributed_operator/implementations/partition_from_cluster.hpp:27-39): the trees' permutations, uploaded on first use
    std::vector<int32_t> t_perm, s_perm;
    bool t_perm_local = false, s_perm_local = false;
    DArr<int32_t> d_t_perm, d_s_perm;
    DArr<char> cm_in, cm_out; // row-major partition-numbering copies of the column-major front ends' operands
    size_t esz = 8;     // bytes per coefficient
    int dtype  = 8;     // ncclFloat64 / ncclFloat32 of the underlying real type
    int reals  = 1;     // real numbers per coefficient (2 for complex)
    DArr<char> work, work2;
    // hmx_dist_set_option (the environment variables of the same names give the initial values, read once in hmx_dist_create)
    bool force             = false; // HMX_DIST_OPT_FORCE_COLLECTIVES: call RCCL even with one rank (tests)
    bool no_allgather      = false; // HMX_DIST_OPT_NO_ALLGATHER: one grouped broadcast per rank even for equal parts
    bool no_reduce_scatter = false; // HMX_DIST_OPT_NO_REDUCE_SCATTER: all-reduce + slice in the transposed local-to-local product
    int (*reduce_scatter)(const void *, void *, size_t, int, int, void *, void *) = nullptr; // ncclReduceScatter, when available
    // point-to-point exchange of the output slices (hmx_dist_set_point_to_point): ncclSend / ncclRecv, and whether they are in use
    int (*send)(const void *, size_t, int, int, void *, void *) = nullptr;
    int (*recv)(void *, size_t, int, int, void *, void *)       = nullptr;
    bool p2p = false;
    // overlap of the output exchange with the expand stage (hmx_dist_set_overlap): row chunks, side stream, one event per chunk
    int overlap = 0;                  // requested number of chunks (0 / 1: off)
    int nchunks = 0;                  // chunks in use (the same on every rank)
    std::vector<int32_t> bounds;      // [rank][chunk]: first local row of chunk c on rank k; (nchunks + 1) entries per rank
    // the same for products with several right-hand sides: their expand stage may run on another layout (the expanded view of a compact
    // symmetric operator), whose row chunks are others; exchanged at the first such product after hmx_dist_set_overlap
    int nchunks_mu = 0;
    bool mu_bounds_known = false;
    std::vector<int32_t> bounds_mu;
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> chunk_ev;
    hipEvent_t join_ev = nullptr;
    hipStream_t cur_stream = nullptr; // the caller's stream of the product in progress (after_chunk callback)
    int cb_rc = 0;
    // trans = 'N' output exchange as the north star words it: ncclAllReduce of the zero-padded output vector (hmx_dist_set_output_collective)
    bool allreduce_out = false;
    // hmx_dist_set_profiling: events on the caller's stream around the part of a global-to-global product that is NOT local computation
    // -- start of the product, last local kernel enqueued, exchange complete -- so that the exposed exchange time is measured, not derived
    bool profiling = false;
    hipEvent_t prof_ev[3] = {nullptr, nullptr, nullptr};
    bool prof_valid = false;
    ~hmx_dist() {
        for (auto e : chunk_ev)
            (void)hipEventDestroy(e);
        if (join_ev)
            (void)hipEventDestroy(join_ev);
        for (auto e : prof_ev)
            if (e)
                (void)hipEventDestroy(e);
        if (side)
            (void)hipStreamDestroy(side);
    }
};
// beta == 0 for the handle's coefficient type: the old output values are never read, so they need not be copied to the work buffer
static bool dist_beta_is_zero(const hmx_dist &D, const void *beta) {
    if (D.dtype == 8) {
        const double *b = static_cast<const double *>(beta);
        return b[0] == 0.0 && (D.reals == 1 || b[1] == 0.0);
    }
    const float *b = static_cast<const float *>(beta);
    return b[0] == 0.f && (D.reals == 1 || b[1] == 0.f);
}
static int dist_api_from_library(hmx_rccl_api &api, void **reduce_scatter = nullptr, void **send = nullptr, void **recv = nullptr) {
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h)
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        set_error(std::string("hmx_dist_create: no RCCL function table given and librccl.so cannot be loaded: ") + dlerror());
        return HMX_ERR_UNSUPPORTED;
    }
    api.all_gather  = reinterpret_cast<decltype(api.all_gather)>(dlsym(h, "ncclAllGather"));
    api.all_reduce  = reinterpret_cast<decltype(api.all_reduce)>(dlsym(h, "ncclAllReduce"));
    api.broadcast   = reinterpret_cast<decltype(api.broadcast)>(dlsym(h, "ncclBroadcast"));
    api.group_start = reinterpret_cast<decltype(api.group_start)>(dlsym(h, "ncclGroupStart"));
    api.group_end   = reinterpret_cast<decltype(api.group_end)>(dlsym(h, "ncclGroupEnd"));
    if (reduce_scatter)
        *reduce_scatter = dlsym(h, "ncclReduceScatter");
    if (send)
        *send = dlsym(h, "ncclSend");
    if (recv)
        *recv = dlsym(h, "ncclRecv");
    if (!api.all_gather || !api.all_reduce || !api.broadcast || !api.group_start || !api.group_end) {
        set_error("hmx_dist_create: librccl.so lacks a collective entry point");
        return HMX_ERR_UNSUPPORTED;
    }
    return HMX_OK;
}
#define HMX_NCCL(call)                                                                  \
    do {                                                                                \
        const int r_ = (call);                                                          \
        if (r_ != 0) {                                                                  \
            set_error(std::string(#call) + ": RCCL error " + std::to_string(r_));      \
            return HMX_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)
// Point-to-point form of the slice exchange: every rank sends rows [lo_r, hi_r) of its slice straight to every peer and receives
// each peer's rows in place, one grouped ncclSend / ncclRecv per pair.  On the fully connected xGMI mesh of an 8-GPU node every
// pair has its own link, so a 1 MB slice crosses ONE link once, where a ring all-gather forwards it over seven hops in turn.
// `per` = elements per row (mu for the multi-RHS products).
static int dist_exchange_p2p(hmx_dist &D, const std::vector<int> &off, const int32_t *lo, const int32_t *hi, int stride, const char *local, char *out, size_t per, hipStream_t st) {
    const int r = D.rank;
    const int mylo = lo[(size_t)r * stride], myhi = hi[(size_t)r * stride];
    if (myhi > mylo)
        HMX_HIP(hipMemcpyAsync(out + (size_t)(off[r] + mylo) * per * D.esz, local + (size_t)mylo * per * D.esz, (size_t)(myhi - mylo) * per * D.esz, hipMemcpyDeviceToDevice, st));
    HMX_NCCL(D.api.group_start());
    for (int k = 0; k < D.world; k++) {
        if (k == r)
            continue;
        if (myhi > mylo)
            HMX_NCCL(D.send(local + (size_t)mylo * per * D.esz, (size_t)(myhi - mylo) * per * D.reals, D.dtype, k, D.comm, st));
        const int klo = lo[(size_t)k * stride], khi = hi[(size_t)k * stride];
        if (khi > klo)
            HMX_NCCL(D.recv(out + (size_t)(off[k] + klo) * per * D.esz, (size_t)(khi - klo) * per * D.reals, D.dtype, k, D.comm, st));
    }
    HMX_NCCL(D.api.group_end());
    return HMX_OK;
}
// out[off_k : off_k + size_k] = rank k's slice (MPI_Allgatherv); `per` elements per row
static int dist_gather_slices(hmx_dist &D, const std::vector<int> &off, const std::vector<int> &size, const char *local, char *out, hipStream_t st, size_t per = 1) {
    if (D.world == 1 && !D.force) {
        HMX_HIP(hipMemcpyAsync(out + (size_t)off[0] * per * D.esz, local, (size_t)size[0] * per * D.esz, hipMemcpyDeviceToDevice, st));
        return HMX_OK;
    }
    if (D.p2p) {
        std::vector<int32_t> lo(D.world, 0), hi(size.begin(), size.end());
        return dist_exchange_p2p(D, off, lo.data(), hi.data(), 1, local, out, per, st);
    }
    bool equal = off[0] == 0;
    for (int k = 1; k < D.world; k++)
        equal = equal && size[k] == size[0];
    if (equal && !D.no_allgather) {
        HMX_NCCL(D.api.all_gather(local, out, (size_t)size[0] * D.reals * per, D.dtype, D.comm, st));
        return HMX_OK;
    }
    HMX_NCCL(D.api.group_start());
    for (int k = 0; k < D.world; k++)
        HMX_NCCL(D.api.broadcast(local, out + (size_t)off[k] * per * D.esz, (size_t)size[k] * D.reals * per, D.dtype, k, D.comm, st));
    HMX_NCCL(D.api.group_end());
    return HMX_OK;
}
// one operator's product on device pointers, whatever its coefficient type
static int dist_op_product(hmx_hmatrix *H, char trans, const void *alpha, const void *in, const void *beta, void *out, hipStream_t st) {
    if (H->d)
        return hmx::f64::api_matvec(H->d, trans, *static_cast<const double *>(alpha), static_cast<const double *>(in), *static_cast<const double *>(beta), static_cast<double *>(out), HMX_MEM_DEVICE, st);
    if (H->s)
        return hmx::f32::api_matvec(H->s, trans, *static_cast<const float *>(alpha), static_cast<const float *>(in), *static_cast<const float *>(beta), static_cast<float *>(out), HMX_MEM_DEVICE, st);
    if (H->z)
        return hmx::z64::api_matvec(H->z, trans, zval(static_cast<const double *>(alpha)), ZP(in), zval(static_cast<const double *>(beta)), ZPM(out), HMX_MEM_DEVICE, st);
    HMX_GUARD(hmx::c32::api_matvec(H->c, trans, cval(static_cast<const float *>(alpha)), CP(in), cval(static_cast<const float *>(beta)), CPM(out), HMX_MEM_DEVICE, st));
}
static int dist_op_matmat(hmx_hmatrix *H, char trans, const void *alpha, const void *in, const void *beta, void *out, int mu, hipStream_t st) {
    if (H->d)
        return hmx::f64::api_matmat_row_major(H->d, trans, *static_cast<const double *>(alpha), static_cast<const double *>(in), *static_cast<const double *>(beta), static_cast<double *>(out), mu, HMX_MEM_DEVICE, st);
    if (H->s)
        return hmx::f32::api_matmat_row_major(H->s, trans, *static_cast<const float *>(alpha), static_cast<const float *>(in), *static_cast<const float *>(beta), static_cast<float *>(out), mu, HMX_MEM_DEVICE, st);
    if (H->z)
        return hmx::z64::api_matmat_row_major(H->z, trans, zval(static_cast<const double *>(alpha)), ZP(in), zval(static_cast<const double *>(beta)), ZPM(out), mu, HMX_MEM_DEVICE, st);
    return hmx::c32::api_matmat_row_major(H->c, trans, cval(static_cast<const float *>(alpha)), CP(in), cval(static_cast<const float *>(beta)), CPM(out), mu, HMX_MEM_DEVICE, st);
}
static const void *dist_one(const hmx_dist &D) {
    static const double one[2] = {1.0, 0.0};
    static const float onef[2] = {1.f, 0.f};
    return D.dtype == 8 ? (const void *)one : (const void *)onef;
}
// ALL operators of this rank (the loops of global_to_global.hpp:63-72 over the global-to-local and the local-to-local operators,
// beta for the first, 1 for the rest), mu right-hand sides (mu = 1: the vector kernels):
//   trans = 'N': `in` = the whole input in the source numbering, `out` = this rank's slice of the target numbering;
//   otherwise  : `in` = this rank's slice of the target numbering, `out` = the whole output in the source numbering
// -- the local-to-local operator sees the rank's slice of either.
static int dist_local_matmat(hmx_dist &D, char trans, const void *alpha, const void *in, const void *beta, void *out, int mu, hipStream_t st) {
    const size_t row = D.esz * (size_t)mu;
    bool first = true;
    auto g2l = [&](hmx_hmatrix *H) {
        const void *b = first ? beta : dist_one(D);
        first         = false;
        return mu == 1 ? dist_op_product(H, trans, alpha, in, b, out, st) : dist_op_matmat(H, trans, alpha, in, b, out, mu, st);
    };
    auto l2l = [&](hmx_hmatrix *H) {
        const void *in2 = trans == 'N' ? static_cast<const void *>(static_cast<const char *>(in) + (size_t)D.s_off[D.rank] * row) : in;
        void *out2      = trans == 'N' ? out : static_cast<void *>(static_cast<char *>(out) + (size_t)D.s_off[D.rank] * row);
        const void *b2  = first ? beta : dist_one(D);
        first           = false;
        return mu == 1 ? dist_op_product(H, trans, alpha, in2, b2, out2, st) : dist_op_matmat(H, trans, alpha, in2, b2, out2, mu, st);
    };
    // the order of global_to_global.hpp:63-72: every global-to-local operator, then every local-to-local operator; beta once
    if (D.local) {
        const int rc = g2l(D.local);
        if (rc != HMX_OK)
            return rc;
    }
    for (hmx_hmatrix *H : D.more_local) {
        const int rc = g2l(H);
        if (rc != HMX_OK)
            return rc;
    }
    if (D.diag) {
        const int rc = l2l(D.diag);
        if (rc != HMX_OK)
            return rc;
    }
    for (hmx_hmatrix *H : D.more_diag) {
        const int rc = l2l(H);
        if (rc != HMX_OK)
            return rc;
    }
    return HMX_OK;
}
static int dist_local_product(hmx_dist &D, char trans, const void *alpha, const void *in, const void *beta, void *out, hipStream_t st) {
    return dist_local_matmat(D, trans, alpha, in, beta, out, 1, st);
}
// chunked local product (trans = 'N'), whatever the coefficient type
static int dist_local_product_chunked(hmx_dist &D, const void *alpha, const void *in, const void *beta, void *out, hipStream_t st, int nchunks, void (*after)(void *, int, int, int), void *user, int *used) {
    hmx_hmatrix *H = D.local;
    if (H->d)
        return hmx::f64::api_matvec_chunked(H->d, *static_cast<const double *>(alpha), static_cast<const double *>(in), *static_cast<const double *>(beta), static_cast<double *>(out), st, nchunks, after, user, used);
    if (H->s)
        return hmx::f32::api_matvec_chunked(H->s, *static_cast<const float *>(alpha), static_cast<const float *>(in), *static_cast<const float *>(beta), static_cast<float *>(out), st, nchunks, after, user, used);
    if (H->z)
        return hmx::z64::api_matvec_chunked(H->z, zval(static_cast<const double *>(alpha)), ZP(in), zval(static_cast<const double *>(beta)), ZPM(out), st, nchunks, after, user, used);
    return hmx::c32::api_matvec_chunked(H->c, cval(static_cast<const float *>(alpha)), CP(in), cval(static_cast<const float *>(beta)), CPM(out), st, nchunks, after, user, used);
}
static int dist_chunk_bounds(hmx_dist &D, int nchunks, int *n, int32_t *b) {
    hmx_hmatrix *H = D.local;
    if (!H || !D.one_operator()) { // the chunked expand stage exists for ONE operator: with more registered the exchange stays whole
        *n   = 1;
        b[0] = 0;
        b[1] = D.t_size[D.rank];
        return HMX_OK;
    }
    if (H->d)
        return hmx::f64::api_chunk_bounds(H->d, nchunks, n, b);
    if (H->s)
        return hmx::f32::api_chunk_bounds(H->s, nchunks, n, b);
    if (H->z)
        return hmx::z64::api_chunk_bounds(H->z, nchunks, n, b);
    return hmx::c32::api_chunk_bounds(H->c, nchunks, n, b);
}
static int dist_chunk_bounds_mu(hmx_dist &D, int nchunks, int *n, int32_t *b) {
    hmx_hmatrix *H = D.local;
    if (!H || !D.one_operator()) {
        *n   = 1;
        b[0] = 0;
        b[1] = D.t_size[D.rank];
        return HMX_OK;
    }
    if (H->d)
        return hmx::f64::api_chunk_bounds_mu(H->d, nchunks, n, b);
    if (H->s)
        return hmx::f32::api_chunk_bounds_mu(H->s, nchunks, n, b);
    if (H->z)
        return hmx::z64::api_chunk_bounds_mu(H->z, nchunks, n, b);
    return hmx::c32::api_chunk_bounds_mu(H->c, nchunks, n, b);
}
static int dist_local_matmat_chunked(hmx_dist &D, const void *alpha, const void *in, const void *beta, void *out, int mu, hipStream_t st, int nchunks, void (*after)(void *, int, int, int),
                                     void *user, int *used) {
    hmx_hmatrix *H = D.local;
    if (H->d)
        return hmx::f64::api_matmat_chunked(H->d, *static_cast<const double *>(alpha), static_cast<const double *>(in), *static_cast<const double *>(beta), static_cast<double *>(out), mu, st, nchunks,
                                            after, user, used);
    if (H->s)
        return hmx::f32::api_matmat_chunked(H->s, *static_cast<const float *>(alpha), static_cast<const float *>(in), *static_cast<const float *>(beta), static_cast<float *>(out), mu, st, nchunks, after,
                                            user, used);
    if (H->z)
        return hmx::z64::api_matmat_chunked(H->z, zval(static_cast<const double *>(alpha)), ZP(in), zval(static_cast<const double *>(beta)), ZPM(out), mu, st, nchunks, after, user, used);
    return hmx::c32::api_matmat_chunked(H->c, cval(static_cast<const float *>(alpha)), CP(in), cval(static_cast<const float *>(beta)), CPM(out), mu, st, nchunks, after, user, used);
}
// rows [bounds[k][c], bounds[k][c + 1]) of every rank k's slice -> their place in `out`: one grouped broadcast per rank (the chunks
// of different ranks have different sizes and are not adjacent in `out`, so this is not an ncclAllGather).  `per`: right-hand sides
// (the rows of a row-major matrix are mu-interleaved: the same exchange with mu times the counts); `bounds`, `nchunks`: D.bounds / D.bounds_mu
static int dist_gather_chunk(hmx_dist &D, const std::vector<int32_t> &bounds, int nchunks, int c, const char *local, char *out, size_t per, hipStream_t st) {
    const int nb = nchunks + 1;
    if (D.p2p)
        return dist_exchange_p2p(D, D.t_off, bounds.data() + c, bounds.data() + c + 1, nb, local, out, per, st);
    HMX_NCCL(D.api.group_start());
    for (int k = 0; k < D.world; k++) {
        const int lo = bounds[(size_t)k * nb + c], hi = bounds[(size_t)k * nb + c + 1];
        if (hi <= lo)
            continue;
        HMX_NCCL(D.api.broadcast(local + (size_t)(k == D.rank ? lo : 0) * D.esz * per, out + (size_t)(D.t_off[k] + lo) * D.esz * per, (size_t)(hi - lo) * D.reals * per, D.dtype, k, D.comm, st));
    }
    HMX_NCCL(D.api.group_end());
    return HMX_OK;
}
struct DistChunkCtx {
    hmx_dist *D;
    const char *local;
    char *out;
    size_t per                         = 1;       // right-hand sides
    const std::vector<int32_t> *bounds = nullptr; // D.bounds or D.bounds_mu
    int nchunks                        = 0;
};
// called on the host right after chunk c of the expand stage was launched on the caller's stream: its exchange goes to the side stream
static void dist_after_chunk(void *user, int c, int row_lo, int row_hi) {
    DistChunkCtx &X = *static_cast<DistChunkCtx *>(user);
    hmx_dist &D     = *X.D;
    if (D.cb_rc != HMX_OK)
        return;
    // the rows this chunk really covered must be the ones every rank was told at hmx_dist_set_overlap: a re-laid-out local operator
    // (recompress, release_factors with a view, a new build) recomputes its chunk plan, and peers would receive stale row ranges
    const std::vector<int32_t> &B = *X.bounds;
    const int nb                  = X.nchunks + 1;
    if (c < 0 || c >= X.nchunks || row_lo != B[(size_t)D.rank * nb + c] || row_hi != B[(size_t)D.rank * nb + c + 1]) {
        set_error("hmx_dist (overlapped product): the local operator was re-laid out since hmx_dist_set_overlap (chunk rows differ); call it again");
        D.cb_rc = HMX_ERR_STATE;
        return;
    }
    if (hipEventRecord(D.chunk_ev[c], D.cur_stream) != hipSuccess || hipStreamWaitEvent(D.side, D.chunk_ev[c], 0) != hipSuccess) {
        D.cb_rc = HMX_ERR_HIP;
        return;
    }
    D.cb_rc = dist_gather_chunk(D, B, X.nchunks, c, X.local, X.out, X.per, D.side);
}


// ---- column-major / user-numbering front ends of the distributed products: pure data movement, one kernel per element size --------
// rm[i][c] = cm[(perm ? perm[i] - base : i) + ld * c]  (user_to_cluster per column + transpose,
// add_distributed_operator_matrix_product_global_to_global.hpp:158-171; `perm == nullptr`: partition numbering, transpose only)
template <typename E>
__global__ void dist_cm_to_rm_kernel(int n, int mu, const int32_t *perm, int base, const E *cm, int64_t ld, E *rm) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)n * mu) {
        const int i = (int)(e / mu), c = (int)(e - (int64_t)i * mu);
        rm[e]       = cm[(int64_t)(perm ? perm[i] - base : i) + ld * c];
    }
}
template <typename E>
__global__ void dist_rm_to_cm_kernel(int n, int mu, const int32_t *perm, int base, const E *rm, E *cm, int64_t ld) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)n * mu) {
        const int i = (int)(e / mu), c = (int)(e - (int64_t)i * mu);
        cm[(int64_t)(perm ? perm[i] - base : i) + ld * c] = rm[e];
    }
}
template <typename E>
__global__ void dist_zero_tail_kernel(E *out, int64_t ld, int first, int rows, int mu) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)rows * mu) {
        const int c = (int)(e / rows), i = (int)(e - (int64_t)c * rows);
        E z;
        memset(&z, 0, sizeof(E));
        out[(int64_t)c * ld + first + i] = z;
    }
}
static int dist_cm_to_rm(hmx_dist &D, int n, int mu, const int32_t *perm, int base, const void *cm, int64_t ld, void *rm, hipStream_t st) {
    const int64_t t = (int64_t)n * mu;
    if (t == 0)
        return HMX_OK;
    const dim3 g((unsigned)((t + 255) / 256)), b(256);
    if (D.esz == 4)
        hipLaunchKernelGGL(dist_cm_to_rm_kernel<float>, g, b, 0, st, n, mu, perm, base, (const float *)cm, ld, (float *)rm);
    else if (D.esz == 8)
        hipLaunchKernelGGL(dist_cm_to_rm_kernel<double>, g, b, 0, st, n, mu, perm, base, (const double *)cm, ld, (double *)rm);
    else
        hipLaunchKernelGGL(dist_cm_to_rm_kernel<double2>, g, b, 0, st, n, mu, perm, base, (const double2 *)cm, ld, (double2 *)rm);
    HMX_HIP(hipGetLastError());
    return HMX_OK;
}
static int dist_rm_to_cm(hmx_dist &D, int n, int mu, const int32_t *perm, int base, const void *rm, void *cm, int64_t ld, hipStream_t st) {
    const int64_t t = (int64_t)n * mu;
    if (t == 0)
        return HMX_OK;
    const dim3 g((unsigned)((t + 255) / 256)), b(256);
    if (D.esz == 4)
        hipLaunchKernelGGL(dist_rm_to_cm_kernel<float>, g, b, 0, st, n, mu, perm, base, (const float *)rm, (float *)cm, ld);
    else if (D.esz == 8)
        hipLaunchKernelGGL(dist_rm_to_cm_kernel<double>, g, b, 0, st, n, mu, perm, base, (const double *)rm, (double *)cm, ld);
    else
        hipLaunchKernelGGL(dist_rm_to_cm_kernel<double2>, g, b, 0, st, n, mu, perm, base, (const double2 *)rm, (double2 *)cm, ld);
    HMX_HIP(hipGetLastError());
    return HMX_OK;
}
// y[0..n) = beta * y + w, n coefficients of the operator's type (the epilogue of the transposed products)
static int dist_add_scaled(hmx_dist &D, int64_t n, const void *w, const void *beta, void *y, hipStream_t st) {
    if (n == 0)
        return HMX_OK;
    if (n > 0x7fffffff) {
        set_error("hmx_dist: more than 2^31 coefficients in one distributed multi-RHS vector");
        return HMX_ERR_UNSUPPORTED;
    }
    hmx_hmatrix *H = D.local ? D.local : D.diag;
    if (H->d)
        hmx::f64::api_axpby(n, (const double *)w, *static_cast<const double *>(beta), static_cast<double *>(y), st);
    else if (H->s)
        hmx::f32::api_axpby(n, (const float *)w, *static_cast<const float *>(beta), static_cast<float *>(y), st);
    else if (H->z)
        hmx::z64::api_axpby(n, ZP(w), zval(static_cast<const double *>(beta)), ZPM(y), st);
    else
        hmx::c32::api_axpby(n, CP(w), cval(static_cast<const float *>(beta)), CPM(y), st);
    HMX_HIP(hipGetLastError());
    return HMX_OK;
}
static int dist_device_perms(hmx_dist &D) {
    if (!D.d_t_perm.d && !D.t_perm.empty())
        HMX_HIP(D.d_t_perm.upload(D.t_perm));
    if (!D.d_s_perm.d && !D.s_perm.empty())
        HMX_HIP(D.d_s_perm.upload(D.s_perm));
    return HMX_OK;
}
static const void *dist_zero(const hmx_dist &D) {
    static const double zero[2] = {0.0, 0.0};
    static const float zerof[2] = {0.f, 0.f};
    return D.dtype == 8 ? (const void *)zero : (const void *)zerof;
}

extern "C" {
// Overlap of the output exchange with the computation (north star: "overlapped on a side HIP stream with the leaf GEMVs").  With
// chunks >= 2 the expand stage of a trans = 'N' global-to-global product runs in that many row chunks on the caller's stream; after each
// chunk an event hands its rows to a side stream, where they are exchanged (grouped ncclBroadcast, one per rank) while the next chunk
// computes; the caller's stream then waits for the last exchange.  COLLECTIVE: every rank calls it with the same value (the ranks
// exchange their chunk boundaries; an operator that cannot be chunked -- fused symmetric storage -- makes all ranks fall back to one
// exchange after the product).  chunks <= 1 switches it off (the default: hmx_dist_create is not collective, this function is).
// every rank's chunk bounds of one layout (multi = false: the single-vector product's, true: the multi-RHS product's), agreed on by all
// ranks: *nchunks_out = chunks when every rank can chunk its product that way, 0 otherwise (the collectives must match)
static int dist_exchange_bounds(hmx_dist &D, int chunks, bool multi, hipStream_t st, int *nchunks_out, std::vector<int32_t> &bounds) {
    const int nb = chunks + 1;
    std::vector<int32_t> mine(nb, 0);
    int n  = 1;
    const int rc_local = multi ? dist_chunk_bounds_mu(D, chunks, &n, mine.data()) : dist_chunk_bounds(D, chunks, &n, mine.data());
    if (rc_local != HMX_OK) { // this rank cannot chunk (e.g. the layout could not be built): it still takes part in the collective below and
        n = 0;                // reports "no chunks", so that ALL ranks fall back to the single exchange together instead of waiting for it
        std::fill(mine.begin(), mine.end(), 0);
    }
    // every rank sends (n, bounds[0..chunks]) as doubles (exact; the mock communicators of the tests only know float types)
    std::vector<double> send(nb + 1, 0.0), recv((size_t)(nb + 1) * D.world, 0.0);
    send[0] = n;
    for (int c = 0; c <= n; c++)
        send[1 + c] = mine[c];
    for (int c = n + 1; c < nb; c++)
        send[1 + c] = mine[n]; // unused chunks are empty
    DArr<double> ds, dr;
    HMX_HIP(ds.upload(send));
    HMX_HIP(dr.alloc(recv.size()));
    HMX_NCCL(D.api.all_gather(ds.d, dr.d, send.size(), 8 /* ncclFloat64 */, D.comm, st));
    HMX_HIP(hipStreamSynchronize(st));
    HMX_HIP(hipMemcpy(recv.data(), dr.d, recv.size() * sizeof(double), hipMemcpyDeviceToHost));
    bool all_chunked = true;
    for (int k = 0; k < D.world; k++)
        all_chunked = all_chunked && (int)recv[(size_t)k * (nb + 1)] == chunks;
    if (!all_chunked) { // some rank cannot chunk its product: nobody does
        *nchunks_out = 0;
        return HMX_OK;
    }
    *nchunks_out = chunks;
    bounds.assign((size_t)nb * D.world, 0);
    for (int k = 0; k < D.world; k++)
        for (int c = 0; c < nb; c++)
            bounds[(size_t)k * nb + c] = (int32_t)recv[(size_t)k * (nb + 1) + 1 + c];
    if (!D.side)
        HMX_HIP(hipStreamCreateWithFlags(&D.side, hipStreamNonBlocking));
    while ((int)D.chunk_ev.size() < chunks) {
        hipEvent_t e;
        HMX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        D.chunk_ev.push_back(e);
    }
    if (!D.join_ev)
        HMX_HIP(hipEventCreateWithFlags(&D.join_ev, hipEventDisableTiming));
    return HMX_OK;
}
int hmx_dist_set_overlap(hmx_dist *Dp, int chunks, void *stream) {
    if (!Dp) {
        set_error("hmx_dist_set_overlap: NULL handle");
        return HMX_ERR_INVALID;
    }
    hmx_dist &D       = *Dp;
    D.overlap         = chunks > 1 ? chunks : 0;
    D.nchunks         = 0;
    D.nchunks_mu      = 0;
    D.mu_bounds_known = false; // exchanged by the first product with several right-hand sides (its layout may have to be built first)
    if (!D.overlap || (D.world == 1 && !D.force))
        return HMX_OK;
    return dist_exchange_bounds(D, chunks, false, (hipStream_t)stream, &D.nchunks, D.bounds);
}
int hmx_dist_overlap_chunks(const hmx_dist *D) { return D ? D->nchunks : 0; }
int hmx_dist_overlap_chunks_multi(const hmx_dist *D) { return (D && D->mu_bounds_known) ? D->nchunks_mu : 0; }
/* ncclReduceScatter for the transposed local-to-local product when the communicator table was given by the caller */
int hmx_dist_set_reduce_scatter(hmx_dist *D, int (*fn)(const void *, void *, size_t, int, int, void *, void *)) {
    if (!D)
        return HMX_ERR_INVALID;
    D->reduce_scatter = fn;
    return HMX_OK;
}

/* Point-to-point exchange of the output slices (ncclSend / ncclRecv shapes); see include/hmx.h */
int hmx_dist_set_point_to_point(hmx_dist *D, int (*send)(const void *, size_t, int, int, void *, void *), int (*recv)(void *, size_t, int, int, void *, void *), int enable) {
    if (!D) {
        set_error("hmx_dist_set_point_to_point: NULL handle");
        return HMX_ERR_INVALID;
    }
    if (send && recv) {
        D->send = send;
        D->recv = recv;
    }
    if (enable && (!D->send || !D->recv)) {
        set_error("hmx_dist_set_point_to_point: no ncclSend / ncclRecv entry points (give them, or create the operator with a NULL collective table)");
        return HMX_ERR_UNSUPPORTED;
    }
    D->p2p = enable != 0;
    return HMX_OK;
}

/* trans = 'N' global-to-global products: how the disjoint output slices reach every rank.  0 (default): exchange of the slices
 * (MPI_Allgatherv of global_to_global.hpp:76: ncclAllGather / grouped broadcasts / pairwise send-recv); 1: ncclAllReduce of the
 * zero-padded length-N output vector, the north star's wording (p times the bytes; not combined with the chunked overlap). */
int hmx_dist_set_output_collective(hmx_dist *D, int all_reduce) {
    if (!D) {
        set_error("hmx_dist_set_output_collective: NULL handle");
        return HMX_ERR_INVALID;
    }
    D->allreduce_out = all_reduce != 0;
    return HMX_OK;
}
int hmx_dist_set_option(hmx_dist *D, int option, int value) {
    if (!D) {
        set_error("hmx_dist_set_option: NULL handle");
        return HMX_ERR_INVALID;
    }
    switch (option) {
    case HMX_DIST_OPT_FORCE_COLLECTIVES:
        if (value != 0 && !D->api.all_gather) { // created with one rank and no table: the collectives were never looked up
            void *rs = nullptr, *sd = nullptr, *rv = nullptr;
            const int rc = dist_api_from_library(D->api, &rs, &sd, &rv);
            if (rc != HMX_OK)
                return rc;
            D->reduce_scatter = reinterpret_cast<decltype(D->reduce_scatter)>(rs);
            D->send           = reinterpret_cast<decltype(D->send)>(sd);
            D->recv           = reinterpret_cast<decltype(D->recv)>(rv);
        }
        D->force = value != 0;
        break;
    case HMX_DIST_OPT_NO_ALLGATHER: D->no_allgather = value != 0; break;
    case HMX_DIST_OPT_NO_REDUCE_SCATTER: D->no_reduce_scatter = value != 0; break;
    default: set_error("hmx_dist_set_option: unknown option " + std::to_string(option)); return HMX_ERR_INVALID;
    }
    return HMX_OK;
}
/* Events on the caller's stream around the non-local part of hmx_dist_matvec_global_to_global (trans = 'N'). */
int hmx_dist_set_profiling(hmx_dist *D, int enabled) {
    if (!D) {
        set_error("hmx_dist_set_profiling: NULL handle");
        return HMX_ERR_INVALID;
    }
    D->profiling  = enabled != 0;
    D->prof_valid = false;
    if (D->profiling)
        for (auto &e : D->prof_ev)
            if (!e)
                HMX_HIP(hipEventCreate(&e));
    return HMX_OK;
}
/* local_ms: start of the last profiled product -> its last local kernel done; exposed_ms: from there until the caller's stream has
 * the whole result (what the exchange adds to the step once whatever overlapped with the local kernels is taken out).  Synchronises
 * on the last event. */
int hmx_dist_last_exchange_ms(hmx_dist *D, float *local_ms, float *exposed_ms) {
    if (!D || !local_ms || !exposed_ms) {
        set_error("hmx_dist_last_exchange_ms: NULL argument");
        return HMX_ERR_INVALID;
    }
    if (!D->prof_valid) {
        set_error("hmx_dist_last_exchange_ms: no profiled trans = 'N' global-to-global product yet (hmx_dist_set_profiling)");
        return HMX_ERR_STATE;
    }
    HMX_HIP(hipEventSynchronize(D->prof_ev[2]));
    HMX_HIP(hipEventElapsedTime(local_ms, D->prof_ev[0], D->prof_ev[1]));
    HMX_HIP(hipEventElapsedTime(exposed_ms, D->prof_ev[1], D->prof_ev[2]));
    return HMX_OK;
}

static void dist_set_precision(hmx_dist &D, int prec) {
    D.esz   = prec == HMX_PREC_F64 ? 8 : (prec == HMX_PREC_F32 ? 4 : (prec == HMX_PREC_Z64 ? 16 : 8));
    D.dtype = (prec == HMX_PREC_F64 || prec == HMX_PREC_Z64) ? 8 : 7; // ncclFloat64 : ncclFloat32
    D.reals = (prec == HMX_PREC_Z64 || prec == HMX_PREC_C32) ? 2 : 1;
}
int hmx_dist_create(hmx_hmatrix *local, const hmx_cluster_tree *target, const hmx_cluster_tree *source, void *nccl_comm, int rank, int world_size, const hmx_rccl_api *api, hmx_dist **out) {
    if (!target || !source || !out || world_size < 1 || rank < 0 || rank >= world_size || (world_size > 1 && !nccl_comm)) {
        set_error("hmx_dist_create: invalid arguments");
        return HMX_ERR_INVALID;
    }
    if ((int)target->on_partition.size() != world_size || (int)source->on_partition.size() != world_size) {
        set_error("hmx_dist_create: the cluster trees must have one partition per rank");
        return HMX_ERR_INVALID;
    }
    auto *D  = new hmx_dist();
    D->work.plain_ = D->work2.plain_ = true;
    D->local = local;
    D->comm  = nccl_comm;
    D->rank  = rank;
    D->world = world_size;
    D->force             = getenv("HMX_DIST_FORCE_COLLECTIVES") && atoi(getenv("HMX_DIST_FORCE_COLLECTIVES"));
    D->no_allgather      = getenv("HMX_DIST_NO_ALLGATHER") && atoi(getenv("HMX_DIST_NO_ALLGATHER"));
    D->no_reduce_scatter = getenv("HMX_DIST_NO_REDUCE_SCATTER") && atoi(getenv("HMX_DIST_NO_REDUCE_SCATTER"));
    if (api) {
        D->api = *api;
    } else if (world_size > 1 || D->force) {
        void *rs = nullptr, *sd = nullptr, *rv = nullptr;
        const int rc = dist_api_from_library(D->api, &rs, &sd, &rv);
        if (rc != HMX_OK) {
            delete D;
            return rc;
        }
        D->reduce_scatter = reinterpret_cast<decltype(D->reduce_scatter)>(rs);
        D->send           = reinterpret_cast<decltype(D->send)>(sd);
        D->recv           = reinterpret_cast<decltype(D->recv)>(rv);
    }
    for (int k = 0; k < world_size; k++) {
        D->t_off.push_back(target->nodes[target->on_partition[k]].offset);
        D->t_size.push_back(target->nodes[target->on_partition[k]].size);
        D->s_off.push_back(source->nodes[source->on_partition[k]].offset);
        D->s_size.push_back(source->nodes[source->on_partition[k]].size);
    }
    D->nt = target->n;
    D->ns = source->n;
    D->t_perm = target->perm, D->s_perm = source->perm;
    D->t_perm_local = target->permutation_is_local, D->s_perm_local = source->permutation_is_local;
    if (local)
        dist_set_precision(*D, hmx_hmatrix_precision(local));
    *out = D;
    return HMX_OK;
}
/* DistributedOperator::add_local_to_local_operator (distributed_operator/distributed_operator.hpp:50-53) with a LocalToLocalHMatrix
 * (implementations/local_to_local_operators/hmatrix.hpp:15-56): the H-matrix on (target partition rank) x (source partition rank), built
 * on a block tree from hmx_block_tree_create_local -- DefaultLocalApproximationBuilder's block-diagonal operator (utility.hpp:64-88).
 * Every product then adds its contribution on the rank's slices, after the global-to-local operator's (which may be absent: create the
 * hmx_dist with local = NULL).  May be called several times (the reference keeps a vector); same coefficient type. */
int hmx_dist_add_local_to_local_operator(hmx_dist *D, hmx_hmatrix *diag) {
    if (!D || !diag) {
        set_error("hmx_dist_add_local_to_local_operator: NULL argument");
        return HMX_ERR_INVALID;
    }
    const int prec = hmx_hmatrix_precision(diag);
    if (D->local && hmx_hmatrix_precision(D->local) != prec) {
        set_error("hmx_dist_add_local_to_local_operator: the operators of one DistributedOperator have one coefficient type");
        return HMX_ERR_INVALID;
    }
    { // the block-diagonal operator of THIS rank: (target partition rank) x (source partition rank), or its products run out of bounds
        int32_t r[4] = {0, 0, 0, 0};
        const int rc = diag->d ? hmx::f64::api_root(diag->d, r) : (diag->s ? hmx::f32::api_root(diag->s, r) : (diag->z ? hmx::z64::api_root(diag->z, r) : hmx::c32::api_root(diag->c, r)));
        if (rc != HMX_OK || r[0] != D->t_off[D->rank] || r[1] != D->t_size[D->rank] || r[2] != D->s_off[D->rank] || r[3] != D->s_size[D->rank]) {
            set_error("hmx_dist_add_local_to_local_operator: the operator is not the (target partition, source partition) block of this rank: rows [" + std::to_string(r[0]) + ", +" +
                      std::to_string(r[1]) + "), columns [" + std::to_string(r[2]) + ", +" + std::to_string(r[3]) + ") against partitions [" + std::to_string(D->t_off[D->rank]) + ", +" +
                      std::to_string(D->t_size[D->rank]) + ") x [" + std::to_string(D->s_off[D->rank]) + ", +" + std::to_string(D->s_size[D->rank]) + ")");
            return HMX_ERR_INVALID;
        }
    }
    if (D->diag)
        D->more_diag.push_back(diag); // (the reference keeps a vector: distributed_operator.hpp:50-53)
    else
        D->diag = diag;
    dist_set_precision(*D, prec);
    D->nchunks = 0; // see dist_chunk_bounds
    return HMX_OK;
}
/* DistributedOperator::add_global_to_local_operator (distributed_operator/distributed_operator.hpp:47-49): ANOTHER operator from the whole
 * source numbering to this rank's rows -- the reference keeps a vector of them and every product loops over it (global_to_global.hpp:63-66),
 * e.g. an H-matrix plus a correction.  Root block: (target partition rank) x (the whole source cluster); same coefficient type.  Kept by
 * reference.  With more than one operator the output exchange is the plain one (the chunked / overlapped forms are built for one operator). */
int hmx_dist_add_global_to_local_operator(hmx_dist *D, hmx_hmatrix *op) {
    if (!D || !op) {
        set_error("hmx_dist_add_global_to_local_operator: NULL argument");
        return HMX_ERR_INVALID;
    }
    const int prec       = hmx_hmatrix_precision(op);
    hmx_hmatrix *any_old = D->local ? D->local : D->diag;
    if (any_old && hmx_hmatrix_precision(any_old) != prec) {
        set_error("hmx_dist_add_global_to_local_operator: the operators of one DistributedOperator have one coefficient type");
        return HMX_ERR_INVALID;
    }
    int32_t r[4] = {0, 0, 0, 0};
    const int rc = op->d ? hmx::f64::api_root(op->d, r) : (op->s ? hmx::f32::api_root(op->s, r) : (op->z ? hmx::z64::api_root(op->z, r) : hmx::c32::api_root(op->c, r)));
    if (rc != HMX_OK || r[0] != D->t_off[D->rank] || r[1] != D->t_size[D->rank] || r[2] != 0 || r[3] != D->ns) {
        set_error("hmx_dist_add_global_to_local_operator: the operator is not the (target partition of this rank) x (whole source cluster) block: rows [" + std::to_string(r[0]) + ", +" +
                  std::to_string(r[1]) + "), columns [" + std::to_string(r[2]) + ", +" + std::to_string(r[3]) + ") against [" + std::to_string(D->t_off[D->rank]) + ", +" +
                  std::to_string(D->t_size[D->rank]) + ") x [0, +" + std::to_string(D->ns) + ")");
        return HMX_ERR_INVALID;
    }
    if (D->local)
        D->more_local.push_back(op);
    else
        D->local = op;
    dist_set_precision(*D, prec);
    D->nchunks = 0; // see dist_chunk_bounds
    return HMX_OK;
}
void hmx_dist_destroy(hmx_dist *D) { delete D; }

int hmx_dist_matvec_global_to_global(hmx_dist *Dp, char trans, const void *alpha, const void *x, const void *beta, void *y, void *stream) {
    if (!Dp || !alpha || !beta || !x || !y) {
        set_error("hmx_dist_matvec_global_to_global: invalid arguments");
        return HMX_ERR_INVALID;
    }
    hmx_dist &D    = *Dp;
    hipStream_t st = (hipStream_t)stream;
    if (!D.local && !D.diag) {
        set_error("hmx_dist: no operator registered");
        return HMX_ERR_STATE;
    }
    const size_t e = D.esz;
    char *yb       = static_cast<char *>(y);
    const char *xb = static_cast<const char *>(x);
    D.prof_valid = false;
    if (D.profiling && trans == 'N')
        HMX_HIP(hipEventRecord(D.prof_ev[0], st));
    if (trans == 'N' && D.allreduce_out && (D.world > 1 || D.force)) {
        // the north star's wording of the exchange: every rank puts alpha * A_loc x into its rows of a zeroed length-N vector, ncclAllReduce
        // sums the vectors (p times the bytes of the all-gather for the same result), then y = beta * y + that
        const int off = D.t_off[D.rank];
        const size_t bytes = (size_t)D.nt * e;
        if (D.work.n < bytes)
            HMX_HIP(D.work.alloc(bytes));
        HMX_HIP(hipMemsetAsync(D.work.d, 0, bytes, st));
        int rc = dist_local_product(D, 'N', alpha, x, dist_zero(D), D.work.d + (size_t)off * e, st);
        if (rc != HMX_OK)
            return rc;
        if (D.profiling)
            HMX_HIP(hipEventRecord(D.prof_ev[1], st));
        HMX_NCCL(D.api.all_reduce(D.work.d, D.work.d, (size_t)D.nt * D.reals, D.dtype, 0 /* ncclSum */, D.comm, st));
        rc = dist_add_scaled(D, D.nt, D.work.d, beta, y, st);
        if (rc == HMX_OK && D.profiling) {
            HMX_HIP(hipEventRecord(D.prof_ev[2], st));
            D.prof_valid = true;
        }
        return rc;
    }
    if (trans == 'N') { // local = beta * y_slice + alpha * A_loc x ; all-gather of the slices
        const int off = D.t_off[D.rank], n = D.t_size[D.rank];
        if (D.work.n < (size_t)n * e)
            HMX_HIP(D.work.alloc((size_t)n * e));
        if (!dist_beta_is_zero(D, beta))
            HMX_HIP(hipMemcpyAsync(D.work.d, yb + (size_t)off * e, (size_t)n * e, hipMemcpyDeviceToDevice, st));
        if (D.nchunks > 1 && (D.world > 1 || D.force)) {
            // chunked expand stage on `st`, each chunk's exchange on the side stream under the next chunk's kernel
            DistChunkCtx ctx{&D, D.work.d, yb, 1, &D.bounds, D.nchunks};
            D.cur_stream = st;
            D.cb_rc      = HMX_OK;
            HMX_HIP(hipEventRecord(D.join_ev, st)); // the side stream must not run ahead of what `st` did to y before this call
            HMX_HIP(hipStreamWaitEvent(D.side, D.join_ev, 0));
            int used = 0;
            int rc   = dist_local_product_chunked(D, alpha, x, beta, D.work.d, st, D.nchunks, dist_after_chunk, &ctx, &used);
            if (D.profiling)
                (void)hipEventRecord(D.prof_ev[1], st); // the last local kernel is enqueued: what `st` waits for from here on is exchange
            // whatever happened, exchanges already enqueued on the side stream read D.work and write y: the caller's stream joins them
            // before this function returns, so that a failed call leaves nothing running behind the caller's back
            const hipError_t j1 = hipEventRecord(D.join_ev, D.side), j2 = hipStreamWaitEvent(st, D.join_ev, 0);
            if (rc != HMX_OK)
                return rc;
            if (D.cb_rc != HMX_OK)
                return D.cb_rc;
            if (used != D.nchunks) {
                set_error("hmx_dist_matvec_global_to_global: the local operator changed since hmx_dist_set_overlap");
                return HMX_ERR_STATE;
            }
            HMX_HIP(j1);
            HMX_HIP(j2);
            if (D.profiling) {
                HMX_HIP(hipEventRecord(D.prof_ev[2], st));
                D.prof_valid = true;
            }
            return HMX_OK;
        }
        int rc = dist_local_product(D, 'N', alpha, x, beta, D.work.d, st);
        if (rc != HMX_OK)
            return rc;
        if (D.profiling)
            HMX_HIP(hipEventRecord(D.prof_ev[1], st));
        rc = dist_gather_slices(D, D.t_off, D.t_size, D.work.d, yb, st);
        if (rc == HMX_OK && D.profiling) {
            HMX_HIP(hipEventRecord(D.prof_ev[2], st));
            D.prof_valid = true;
        }
        return rc;
    }
    // transposed: every rank contributes alpha * A_loc^T x_slice to the whole vector; all-reduce; beta * y_old added once
    const int off = D.t_off[D.rank];
    const size_t bytes = (size_t)D.ns * e;
    if (D.work.n < bytes)
        HMX_HIP(D.work.alloc(bytes));
    HMX_HIP(hipMemsetAsync(D.work.d, 0, bytes, st));
    const double zero[2] = {0.0, 0.0};
    const float zerof[2] = {0.f, 0.f};
    const void *bz       = D.dtype == 8 ? (const void *)zero : (const void *)zerof;
    int rc = dist_local_product(D, trans, alpha, xb + (size_t)off * e, bz, D.work.d, st);
    if (rc != HMX_OK)
        return rc;
    if (D.world > 1 || D.force)
        HMX_NCCL(D.api.all_reduce(D.work.d, D.work.d, (size_t)D.ns * D.reals, D.dtype, 0 /* ncclSum */, D.comm, st));
    // y = beta * y + work
    return dist_add_scaled(D, D.ns, D.work.d, beta, y, st);
}

// add_distributed_operator_matrix_product_row_major_global_to_global.hpp:18-85: X (n x mu) and Y (m x mu) row-major (mu fastest), whole
// matrices replicated on every rank, partition numbering.  trans = 'N': the local slice of Y (its rows are contiguous: mu-interleaved),
// product, exchange of the slices (MPI_Allgatherv :76); transposed: all-reduce of the whole matrix (:78).
int hmx_dist_matmat_row_major_global_to_global(hmx_dist *Dp, char trans, const void *alpha, const void *X, const void *beta, void *Y, int mu, void *stream) {
    if (!Dp || !alpha || !beta || !X || !Y || mu < 1) {
        set_error("hmx_dist_matmat_row_major_global_to_global: invalid arguments");
        return HMX_ERR_INVALID;
    }
    hmx_dist &D    = *Dp;
    hipStream_t st = (hipStream_t)stream;
    if (!D.local && !D.diag) {
        set_error("hmx_dist: no operator registered");
        return HMX_ERR_STATE;
    }
    const size_t e = D.esz * (size_t)mu; // bytes per row
    char *yb       = static_cast<char *>(Y);
    const char *xb = static_cast<const char *>(X);
    if (trans == 'N') {
        const int off = D.t_off[D.rank], n = D.t_size[D.rank];
        if (D.work.n < (size_t)n * e)
            HMX_HIP(D.work.alloc((size_t)n * e));
        if (!dist_beta_is_zero(D, beta))
            HMX_HIP(hipMemcpyAsync(D.work.d, yb + (size_t)off * e, (size_t)n * e, hipMemcpyDeviceToDevice, st));
        if (D.overlap > 1 && mu > 1 && (D.world > 1 || D.force) && D.one_operator()) {
            // expand stage in row chunks on `st`, every chunk's (mu-interleaved) rows exchanged on the side stream under the next chunk's
            // kernels -- hmx_dist_matvec_global_to_global's scheme; the chunk rows of the multi-RHS layout are agreed on at the first call
            if (!D.mu_bounds_known) {
                const int rc0 = dist_exchange_bounds(D, D.overlap, true, st, &D.nchunks_mu, D.bounds_mu);
                if (rc0 != HMX_OK)
                    return rc0;
                D.mu_bounds_known = true;
            }
            if (D.nchunks_mu > 1) {
                DistChunkCtx ctx{&D, D.work.d, yb, (size_t)mu, &D.bounds_mu, D.nchunks_mu};
                D.cur_stream = st;
                D.cb_rc      = HMX_OK;
                HMX_HIP(hipEventRecord(D.join_ev, st));
                HMX_HIP(hipStreamWaitEvent(D.side, D.join_ev, 0));
                int used = 0;
                int rc   = dist_local_matmat_chunked(D, alpha, X, beta, D.work.d, mu, st, D.nchunks_mu, dist_after_chunk, &ctx, &used);
                // exchanges already enqueued on the side stream read D.work and write Y: the caller's stream joins them on every exit path
                const hipError_t j1 = hipEventRecord(D.join_ev, D.side), j2 = hipStreamWaitEvent(st, D.join_ev, 0);
                if (rc != HMX_OK)
                    return rc;
                if (D.cb_rc != HMX_OK)
                    return D.cb_rc;
                if (used != D.nchunks_mu) {
                    set_error("hmx_dist_matmat_row_major_global_to_global: the local operator changed since hmx_dist_set_overlap");
                    return HMX_ERR_STATE;
                }
                HMX_HIP(j1);
                HMX_HIP(j2);
                return HMX_OK;
            }
        }
        int rc = dist_local_matmat(D, 'N', alpha, X, beta, D.work.d, mu, st);
        if (rc != HMX_OK)
            return rc;
        // the slices are mu-interleaved rows: the same exchange with mu times the counts
        return dist_gather_slices(D, D.t_off, D.t_size, D.work.d, yb, st, (size_t)mu);
    }
    const int off      = D.t_off[D.rank];
    const size_t bytes = (size_t)D.ns * e;
    if (D.work.n < bytes)
        HMX_HIP(D.work.alloc(bytes));
    HMX_HIP(hipMemsetAsync(D.work.d, 0, bytes, st));
    const double zero[2] = {0.0, 0.0};
    const float zerof[2] = {0.f, 0.f};
    const void *bz       = D.dtype == 8 ? (const void *)zero : (const void *)zerof;
    int rc = dist_local_matmat(D, trans, alpha, xb + (size_t)off * e, bz, D.work.d, mu, st);
    if (rc != HMX_OK)
        return rc;
    if (D.world > 1 || D.force)
        HMX_NCCL(D.api.all_reduce(D.work.d, D.work.d, (size_t)D.ns * D.reals * mu, D.dtype, 0 /* ncclSum */, D.comm, st));
    return dist_add_scaled(D, (int64_t)D.ns * mu, D.work.d, beta, Y, st);
}

int hmx_dist_matvec_local_to_local(hmx_dist *Dp, char trans, const void *alpha, const void *x_local, const void *beta, void *y_local, void *stream) {
    if (!Dp || !alpha || !beta || !x_local || !y_local) {
        set_error("hmx_dist_matvec_local_to_local: invalid arguments");
        return HMX_ERR_INVALID;
    }
    hmx_dist &D    = *Dp;
    hipStream_t st = (hipStream_t)stream;
    const size_t e = D.esz;
    if (!D.local && !D.diag) {
        set_error("hmx_dist: no operator registered");
        return HMX_ERR_STATE;
    }
    if (!D.local && D.more_local.empty() && D.more_diag.empty()) // ONE local-to-local operator only (local_to_local.hpp:27-31): local slices in and out, nothing to exchange
        return dist_op_product(D.diag, trans, alpha, x_local, beta, y_local, st);
    if (trans == 'N') { // all-gather of x, then the local product straight into the local slice
        const size_t bytes = (size_t)D.ns * e;
        if (D.work2.n < bytes)
            HMX_HIP(D.work2.alloc(bytes));
        int rc = dist_gather_slices(D, D.s_off, D.s_size, static_cast<const char *>(x_local), D.work2.d, st);
        if (rc != HMX_OK)
            return rc;
        return dist_local_product(D, 'N', alpha, D.work2.d, beta, y_local, st);
    }
    // transposed: product into a zeroed global vector, all-reduce, y_local = beta * y_local + slice
    const size_t bytes = (size_t)D.ns * e;
    if (D.work2.n < bytes)
        HMX_HIP(D.work2.alloc(bytes));
    HMX_HIP(hipMemsetAsync(D.work2.d, 0, bytes, st));
    const double zero[2] = {0.0, 0.0};
    const float zerof[2] = {0.f, 0.f};
    const void *bz       = D.dtype == 8 ? (const void *)zero : (const void *)zerof;
    int rc = dist_local_product(D, trans, alpha, x_local, bz, D.work2.d, st);
    if (rc != HMX_OK)
        return rc;
    const int off = D.s_off[D.rank], n = D.s_size[D.rank];
    const char *w = D.work2.d + (size_t)off * e;
    if (D.world > 1 || D.force) {
        // the reference's MPI_Alltoallv + p axpys (local_to_local.hpp:77) is a reduce-scatter: with equal, contiguous partitions
        // ncclReduceScatter delivers exactly this rank's slice (N / p instead of N per rank); otherwise all-reduce + slice
        bool equal = D.s_off[0] == 0 && D.reduce_scatter != nullptr && !D.no_reduce_scatter;
        for (int k = 1; k < D.world && equal; k++)
            equal = D.s_size[k] == D.s_size[0] && D.s_off[k] == k * D.s_size[0];
        if (equal) {
            if (D.work.n < (size_t)n * e)
                HMX_HIP(D.work.alloc((size_t)n * e));
            HMX_NCCL(D.reduce_scatter(D.work2.d, D.work.d, (size_t)n * D.reals, D.dtype, 0, D.comm, st));
            w = D.work.d;
        } else {
            HMX_NCCL(D.api.all_reduce(D.work2.d, D.work2.d, (size_t)D.ns * D.reals, D.dtype, 0, D.comm, st));
        }
    }
    return dist_add_scaled(D, n, w, beta, y_local, st);
}


// internal_add_distributed_operator_matrix_product_row_major_local_to_local (distributed_operator/linalg/
// add_distributed_operator_matrix_product_row_major_local_to_local.hpp:19-95; what HPDDMOperator::GMV calls for mu != 1,
// wrappers/wrapper_hpddm.hpp:126): local row slices in and out, mu-interleaved rows, partition numbering.
// trans = 'N': local_to_global of X (MPI_Allgatherv, linalg/utility.hpp:11-28) then the local product straight into Y_local;
// transposed: the local product into a zeroed global matrix, then MPI_Alltoallv + p axpys (:64-93) = a reduce-scatter of
// mu * n rows (ncclReduceScatter on equal partitions, otherwise all-reduce + slice), Y_local = beta * Y_local + slice.
int hmx_dist_matmat_row_major_local_to_local(hmx_dist *Dp, char trans, const void *alpha, const void *X_local, const void *beta, void *Y_local, int mu, void *stream) {
    if (!Dp || !alpha || !beta || !X_local || !Y_local || mu < 1) {
        set_error("hmx_dist_matmat_row_major_local_to_local: invalid arguments");
        return HMX_ERR_INVALID;
    }
    hmx_dist &D    = *Dp;
    hipStream_t st = (hipStream_t)stream;
    if (!D.local && !D.diag) {
        set_error("hmx_dist: no operator registered");
        return HMX_ERR_STATE;
    }
    if (!D.local && D.more_local.empty() && D.more_diag.empty()) // ONE local-to-local operator only: nothing to exchange
        return dist_op_matmat(D.diag, trans, alpha, X_local, beta, Y_local, mu, st);
    const size_t e = D.esz * (size_t)mu; // bytes per row
    const size_t bytes = (size_t)D.ns * e;
    if (D.work2.n < bytes)
        HMX_HIP(D.work2.alloc(bytes));
    if (trans == 'N') {
        int rc = dist_gather_slices(D, D.s_off, D.s_size, static_cast<const char *>(X_local), D.work2.d, st, (size_t)mu);
        if (rc != HMX_OK)
            return rc;
        return dist_local_matmat(D, 'N', alpha, D.work2.d, beta, Y_local, mu, st);
    }
    HMX_HIP(hipMemsetAsync(D.work2.d, 0, bytes, st));
    int rc = dist_local_matmat(D, trans, alpha, X_local, dist_zero(D), D.work2.d, mu, st);
    if (rc != HMX_OK)
        return rc;
    const int off = D.s_off[D.rank], n = D.s_size[D.rank];
    const char *w = D.work2.d + (size_t)off * e;
    if (D.world > 1 || D.force) {
        bool equal = D.s_off[0] == 0 && D.reduce_scatter != nullptr && !D.no_reduce_scatter;
        for (int k = 1; k < D.world && equal; k++)
            equal = D.s_size[k] == D.s_size[0] && D.s_off[k] == k * D.s_size[0];
        if (equal) {
            if (D.work.n < (size_t)n * e)
                HMX_HIP(D.work.alloc((size_t)n * e));
            HMX_NCCL(D.reduce_scatter(D.work2.d, D.work.d, (size_t)n * D.reals * mu, D.dtype, 0, D.comm, st));
            w = D.work.d;
        } else {
            HMX_NCCL(D.api.all_reduce(D.work2.d, D.work2.d, (size_t)D.ns * D.reals * mu, D.dtype, 0, D.comm, st));
        }
    }
    return dist_add_scaled(D, (int64_t)n * mu, w, beta, Y_local, st);
}

// add_distributed_operator_matrix_product_global_to_global / internal_add_... (distributed_operator/linalg/
// add_distributed_operator_matrix_product_global_to_global.hpp:132-279 / :18-117): column-major X (n x mu) and Y (m x mu), whole
// matrices replicated on every rank; numbering = 1: USER numbering (every column through global_to_partition_numbering on the way in,
// partition_to_global_numbering on the way out, :158-171,263-276), 0: partition numbering (transposition only).  The operands are
// brought to the row-major layout by one kernel each way, then the row-major product above runs; for a transposed product only this
// rank's rows of X are converted (:165-169).  mu = 1 is the user-numbering vector product
// (add_distributed_operator_vector_product_global_to_global.hpp:97-118).
int hmx_dist_matmat_global_to_global(hmx_dist *Dp, char trans, const void *alpha, const void *X, const void *beta, void *Y, int mu, int numbering, void *stream) {
    if (!Dp || !alpha || !beta || !X || !Y || mu < 1 || (numbering != 0 && numbering != 1)) {
        set_error("hmx_dist_matmat_global_to_global: invalid arguments");
        return HMX_ERR_INVALID;
    }
    hmx_dist &D    = *Dp;
    hipStream_t st = (hipStream_t)stream;
    int rc         = numbering ? dist_device_perms(D) : HMX_OK;
    if (rc != HMX_OK)
        return rc;
    const bool N    = trans == 'N';
    const int nin = N ? D.ns : D.nt, nout = N ? D.nt : D.ns;
    const int32_t *pin = numbering ? (N ? D.d_s_perm.d : D.d_t_perm.d) : nullptr, *pout = numbering ? (N ? D.d_t_perm.d : D.d_s_perm.d) : nullptr;
    const size_t e = D.esz * (size_t)mu;
    if (D.cm_in.n < (size_t)nin * e)
        HMX_HIP(D.cm_in.alloc((size_t)nin * e));
    if (D.cm_out.n < (size_t)nout * e)
        HMX_HIP(D.cm_out.alloc((size_t)nout * e));
    // rows of X the product reads: all of them for 'N', this rank's slice of the target partition otherwise
    const int lo = N ? 0 : D.t_off[D.rank], cnt = N ? nin : D.t_size[D.rank];
    rc = dist_cm_to_rm(D, cnt, mu, pin ? pin + lo : nullptr, 0, numbering ? X : (const void *)(static_cast<const char *>(X) + (size_t)lo * D.esz), nin, D.cm_in.d + (size_t)lo * e, st);
    if (rc != HMX_OK)
        return rc;
    if (!dist_beta_is_zero(D, beta)) {
        rc = dist_cm_to_rm(D, nout, mu, pout, 0, Y, nout, D.cm_out.d, st);
        if (rc != HMX_OK)
            return rc;
    }
    rc = hmx_dist_matmat_row_major_global_to_global(Dp, trans, alpha, D.cm_in.d, beta, D.cm_out.d, mu, stream);
    if (rc != HMX_OK)
        return rc;
    return dist_rm_to_cm(D, nout, mu, pout, 0, D.cm_out.d, Y, nout, st);
}

// add_distributed_operator_matrix_product_local_to_local / internal_add_... (distributed_operator/linalg/
// add_distributed_operator_matrix_product_local_to_local.hpp:66-120 / :20-49): column-major local slices X_local (n_k x mu) and
// Y_local (m_k x mu); numbering = 1: the rank's LOCAL user numbering (local_to_local_partition_numbering /
// local_partition_to_local_numbering per column, :91-117 -- needs a cluster tree whose permutation is local to the partitions,
// clustering/cluster_node.hpp:124-146), 0: partition numbering.  mu = 1 with numbering = 1 is
// add_distributed_operator_vector_product_local_to_local (..._vector_product_local_to_local.hpp:99-125).
int hmx_dist_matmat_local_to_local(hmx_dist *Dp, char trans, const void *alpha, const void *X_local, const void *beta, void *Y_local, int mu, int numbering, void *stream) {
    if (!Dp || !alpha || !beta || !X_local || !Y_local || mu < 1 || (numbering != 0 && numbering != 1)) {
        set_error("hmx_dist_matmat_local_to_local: invalid arguments");
        return HMX_ERR_INVALID;
    }
    hmx_dist &D    = *Dp;
    hipStream_t st = (hipStream_t)stream;
    if (numbering && !(D.t_perm_local && D.s_perm_local)) {
        set_error("hmx_dist_matmat_local_to_local: permutation is not local to partition, local numbering cannot be used"); // cluster_node.hpp:126,138
        return HMX_ERR_INVALID;
    }
    int rc = numbering ? dist_device_perms(D) : HMX_OK;
    if (rc != HMX_OK)
        return rc;
    const bool N    = trans == 'N';
    const int in_off = N ? D.s_off[D.rank] : D.t_off[D.rank], nin = N ? D.s_size[D.rank] : D.t_size[D.rank];
    const int out_off = N ? D.t_off[D.rank] : D.s_off[D.rank], nout = N ? D.t_size[D.rank] : D.s_size[D.rank];
    const int32_t *pin = numbering ? (N ? D.d_s_perm.d : D.d_t_perm.d) + in_off : nullptr, *pout = numbering ? (N ? D.d_t_perm.d : D.d_s_perm.d) + out_off : nullptr;
    const size_t e = D.esz * (size_t)mu;
    if (D.cm_in.n < (size_t)nin * e)
        HMX_HIP(D.cm_in.alloc((size_t)nin * e));
    if (D.cm_out.n < (size_t)nout * e)
        HMX_HIP(D.cm_out.alloc((size_t)nout * e));
    rc = dist_cm_to_rm(D, nin, mu, pin, in_off, X_local, nin, D.cm_in.d, st);
    if (rc != HMX_OK)
        return rc;
    if (!dist_beta_is_zero(D, beta)) {
        rc = dist_cm_to_rm(D, nout, mu, pout, out_off, Y_local, nout, D.cm_out.d, st);
        if (rc != HMX_OK)
            return rc;
    }
    rc = mu == 1 ? hmx_dist_matvec_local_to_local(Dp, trans, alpha, D.cm_in.d, beta, D.cm_out.d, stream)
                 : hmx_dist_matmat_row_major_local_to_local(Dp, trans, alpha, D.cm_in.d, beta, D.cm_out.d, mu, stream);
    if (rc != HMX_OK)
        return rc;
    return dist_rm_to_cm(D, nout, mu, pout, out_off, D.cm_out.d, Y_local, nout, st);
}

// HPDDMOperator::GMV (wrappers/wrapper_hpddm.hpp:102-142) up to HPDDM's own overlap exchange: `in` and `out` are column-major with
// leading dimension `dof` (the local size INCLUDING the overlap HPDDM keeps behind the first local_size rows), mu columns.  As there:
// the first local_size rows of every column are brought to the row-major layout (:108-116), the local-to-local product with alpha = 1,
// beta = 0 runs (vector kernel for mu = 1, :118-125), the result is transposed back and rows [local_size, dof) of every column of `out`
// are set to zero (:128-138).  The caller then does what GMV does last: this->exchange(out, mu).
int hmx_dist_gmv(hmx_dist *Dp, const void *in, void *out, int mu, int dof, void *stream) {
    if (!Dp || !in || !out || mu < 1) {
        set_error("hmx_dist_gmv: invalid arguments");
        return HMX_ERR_INVALID;
    }
    hmx_dist &D    = *Dp;
    hipStream_t st = (hipStream_t)stream;
    const int n_in = D.s_size[D.rank], n_out = D.t_size[D.rank];
    if (n_in != n_out || dof < n_out) {
        set_error("hmx_dist_gmv: GMV works on one local size (square operator, same partition on both sides) and dof >= local size");
        return HMX_ERR_INVALID;
    }
    const size_t e = D.esz * (size_t)mu;
    if (D.cm_in.n < (size_t)n_in * e)
        HMX_HIP(D.cm_in.alloc((size_t)n_in * e));
    if (D.cm_out.n < (size_t)n_out * e)
        HMX_HIP(D.cm_out.alloc((size_t)n_out * e));
    int rc = HMX_OK;
    const void *xin = in;
    if (mu != 1) {
        rc = dist_cm_to_rm(D, n_in, mu, nullptr, 0, in, dof, D.cm_in.d, st);
        if (rc != HMX_OK)
            return rc;
        xin = D.cm_in.d;
    }
    rc = mu == 1 ? hmx_dist_matvec_local_to_local(Dp, 'N', dist_one(D), xin, dist_zero(D), D.cm_out.d, stream)
                 : hmx_dist_matmat_row_major_local_to_local(Dp, 'N', dist_one(D), xin, dist_zero(D), D.cm_out.d, mu, stream);
    if (rc != HMX_OK)
        return rc;
    rc = dist_rm_to_cm(D, n_out, mu, nullptr, 0, D.cm_out.d, out, dof, st);
    if (rc != HMX_OK)
        return rc;
    const int tail = dof - n_out;
    if (tail > 0) {
        const int64_t t = (int64_t)tail * mu;
        const dim3 g((unsigned)((t + 255) / 256)), b(256);
        if (D.esz == 4)
            hipLaunchKernelGGL(dist_zero_tail_kernel<float>, g, b, 0, st, (float *)out, (int64_t)dof, n_out, tail, mu);
        else if (D.esz == 8)
            hipLaunchKernelGGL(dist_zero_tail_kernel<double>, g, b, 0, st, (double *)out, (int64_t)dof, n_out, tail, mu);
        else
            hipLaunchKernelGGL(dist_zero_tail_kernel<double2>, g, b, 0, st, (double2 *)out, (int64_t)dof, n_out, tail, mu);
        HMX_HIP(hipGetLastError());
    }
    return HMX_OK;
}

double hmx_device_malloc_seconds(void) { return 1e-9 * (double)g_malloc_ns.load(); }
int64_t hmx_device_alloc_count(void) { return (int64_t)g_alloc_count.load(); }
int hmx_device_trim_cache(void) {
    DeviceCache::get().trim();
    (void)DeviceSlabs::get().release_idle();
    return HMX_OK;
}
int hmx_device_reserve(int device_id, int64_t bytes) {
    const int rc = ensure_device(device_id);
    if (rc != HMX_OK)
        return rc;
    if (bytes <= 0) {
        set_error("hmx_device_reserve: bytes must be positive");
        return HMX_ERR_INVALID;
    }
    if (DeviceSlabs::get().reserve(device_id, (size_t)bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("hmx_device_reserve: hipMalloc of the slab failed");
        return HMX_ERR_HIP;
    }
    return HMX_OK;
}

int hmx_device_slab_alloc_at(int device_id, int64_t bytes, double frac, void **ptr) {
    const int rc = ensure_device(device_id);
    if (rc != HMX_OK)
        return rc;
    if (!ptr || bytes <= 0 || !(frac >= 0.0 && frac <= 1.0)) {
        set_error("hmx_device_slab_alloc_at: invalid arguments");
        return HMX_ERR_INVALID;
    }
    size_t got = 0;
    void *p    = DeviceSlabs::get().take_at(device_id, (size_t)bytes, frac, &got);
    if (!p) {
        set_error("hmx_device_slab_alloc_at: no reserved slab has room (hmx_device_reserve)");
        return HMX_ERR_STATE;
    }
    *ptr = p;
    return HMX_OK;
}
int hmx_device_slab_free(int device_id, void *ptr, int64_t bytes) {
    const int rc = ensure_device(device_id);
    if (rc != HMX_OK)
        return rc;
    HMX_HIP(hipDeviceSynchronize());
    const size_t need = ((size_t)bytes + DeviceSlabs::GRAIN - 1) / DeviceSlabs::GRAIN * DeviceSlabs::GRAIN;
    if (!ptr || bytes <= 0 || !DeviceSlabs::get().give_back(ptr, need)) {
        set_error("hmx_device_slab_free: not a range of a reserved slab");
        return HMX_ERR_INVALID;
    }
    return HMX_OK;
}

int hmx_device_read_bandwidth(int device_id, int64_t bytes, int reps, double *gbps) {
    int rc = ensure_device(device_id);
    if (rc != HMX_OK)
        return rc;
    if (!gbps || bytes < 16 || reps < 1)
        return HMX_ERR_INVALID;
    DArr<double2> a;
    DArr<double> out;
    const int64_t n = bytes / 16;
    const int blocks = 1024, threads = 256; // 4 workgroups per CU, each on its own contiguous chunk (tools/read_bw.hip: the best of the variants)
    HMX_HIP(a.alloc(n));
    HMX_HIP(out.alloc((size_t)blocks * threads));
    HMX_HIP(a.zero());
    DEvent e0, e1;
    hipLaunchKernelGGL(read16_kernel, dim3(blocks), dim3(threads), 0, 0, (const double2 *)a.d, out.d, n);
    HMX_HIP(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(read16_kernel, dim3(blocks), dim3(threads), 0, 0, (const double2 *)a.d, out.d, n);
    HMX_HIP(hipEventRecord(e1, 0));
    HMX_HIP(hipEventSynchronize(e1));
    float ms = 0;
    HMX_HIP(hipEventElapsedTime(&ms, e0, e1));
    *gbps = (double)n * 16 * reps / (ms * 1e-3) / 1e9;
    return HMX_OK;
}

} // extern "C"
