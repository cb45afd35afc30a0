// kernels_common.hpp + kernels_body.hpp -- hand-written HIP kernels for gfx950 (CDNA4, wave64).  Compiled with -ffp-contract=off:
// every FMA in this file is an explicit __builtin_fma, so the compression kernels evaluate exactly the
// IEEE sequence the reference's scalar code spells out (SURVEY.md "Hard parts": ACA parity) while the
// matvec kernels still issue v_fma_f64.
//
// Data layout in HBM: the compressed operator is NOT kept as htool's per-block
// U (M x r) / V (r x N) / dense (M x N) matrices.  It is re-laid out as two sets of streams:
//   E-stream  (expand): for every target row range R (<= 64 rows, aligned to cluster boundaries) one
//             column-major len_R x C_R matrix holding, side by side, the slice of every block that
//             touches R: n_b columns for a dense block, r_b columns (its U slice) for a low-rank block.
//             y_R = E_R * z_R, z_R gathered from Z = [x | a].  lane = row, no cross-lane reduction.
//   R-stream  (reduce): for every source range S one row-major len_S x C_S matrix (128-column chunks)
//             holding the V slices of every low-rank block that touches S.  a_partial = x_S^T * R_S,
//             lane = column pair, no cross-lane reduction.
// Every stored coefficient is read exactly once per matvec, by fully coalesced wave loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hmx {

constexpr int WAVE = 64;

// lock-step ACA with a host generator (aca_cb_*_kernel): one entry per launch position
struct CbItem {
    int64_t off;   // first entry of the block's evaluated line in the launch's buffer
    int32_t block; // leaf index
    int32_t pad;
};
enum { CB_ACTIVE = 0, CB_FINISHED = 1, CB_SUSPENDED = 2 };
struct CbResult {
    int32_t status; // CB_ACTIVE: the block goes on with the next phase; CB_FINISHED; CB_SUSPENDED: the pool ran out before this row phase
    int32_t I1, I2; // next row pivot / column pivot of the iteration in progress (index-1 / index-2 side)
    int32_t pad;
};

// device-evaluable generator families (include/hmx.h: hmx_kernel)
enum { KS_INV_DIST = 0, KS_HELMHOLTZ = 1, KS_LAPLACE_SL = 2 };
struct KernelSpec {
    int kind;
    int dim;
    double p0, p1;
    // complex instantiations: (cre + i cim sgn) / (p0 + p1 |x-y|), sgn = 1 or, herm != 0, sign(x_t[0] - x_s[0])
    // (the complex symmetric / Hermitian forms of testing/generator_test.hpp:163-205)
    double cre, cim;
    int herm;
    double wavenumber; // KS_HELMHOLTZ
};

// |x - y|^2: squared differences accumulated left to right from 0 -- the order of examples/use_hmatrix.cpp:33 /
// testing/generator_test.hpp:159
__device__ __forceinline__ double eval_dist2(const KernelSpec &ks, double tx, double ty, double tz, double sx, double sy, double sz) {
    double s        = 0.0;
    const double d0 = tx - sx;
    s               = s + d0 * d0;
    const double d1 = ty - sy;
    s               = s + d1 * d1;
    if (ks.dim == 3) {
        const double d2 = tz - sz;
        s               = s + d2 * d2;
    }
    return s;
}
// K(x,y) = 1/(p0 + p1*|x-y|): one sqrt, one multiply, one add, one divide
__device__ __forceinline__ double eval_kernel(const KernelSpec &ks, double tx, double ty, double tz, double sx, double sy, double sz) {
    return 1.0 / (ks.p0 + ks.p1 * sqrt(eval_dist2(ks, tx, ty, tz, sx, sy, sz)));
}
// the denominator of the same family, for the complex instantiations
__device__ __forceinline__ double eval_kernel_den(const KernelSpec &ks, double tx, double ty, double tz, double sx, double sy, double sz) {
    return ks.p0 + ks.p1 * sqrt(eval_dist2(ks, tx, ty, tz, sx, sy, sz));
}

// sin and cos of the Helmholtz phase k r, written out (IEEE operations in a fixed order, no contraction) so that a host generator
// restating it (a user's own code, the test fixtures' reference driver) produces the same bits: x = n pi/2 + r by a three-constant Cody-Waite reduction
// (33-bit pieces of pi/2: n * piece is exact for |n| < 2^20, i.e. |x| < 1.6e6), then the classical minimax polynomials of sin and cos
// on [-pi/4, pi/4] (the coefficients of fdlibm's __kernel_sin / __kernel_cos).  About one ulp; no table, no large-argument path, no
// scratch memory -- the vendor's sincos carries a Payne-Hanek path whose registers every thread of aca_kernel would pay for.
__host__ __device__ __forceinline__ void hmx_sincos(double x, double &sn, double &cs) {
    const double fn = rint(x * 6.36619772367581382433e-01);
    double r        = x - fn * 1.57079632673412561417e+00;
    r               = r - fn * 6.07710050630396597660e-11;
    r               = r - fn * 2.02226624871116645580e-21;
    const double z  = r * r;
    const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s  = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
    const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c  = w + (((1.0 - w) - hz) + z * pc);
    const int q     = (int)((long long)fn & 3);
    sn              = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    cs              = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}
constexpr double HMX_FOUR_PI = 12.566370614359172; // 4 pi, the nearest double

// ---- complex coefficients: htool's HMatrix<std::complex<T>> ----------------------------------------------------------------
// Layout-compatible with std::complex<T> / C99 T _Complex (interleaved re, im).  operator* and operator/ are the plain
// unfused formulas (the compression kernels restate the reference's scalar sequence); the matvec kernels use hmx_fma.
template <typename T>
struct alignas(2 * sizeof(T)) cplx {
    T re, im;
    cplx() = default;
    __host__ __device__ constexpr cplx(T r, T i = T(0)) : re(r), im(i) {}
};
template <typename T> __host__ __device__ __forceinline__ cplx<T> operator+(cplx<T> a, cplx<T> b) { return cplx<T>(a.re + b.re, a.im + b.im); }
template <typename T> __host__ __device__ __forceinline__ cplx<T> operator-(cplx<T> a, cplx<T> b) { return cplx<T>(a.re - b.re, a.im - b.im); }
template <typename T> __host__ __device__ __forceinline__ cplx<T> operator-(cplx<T> a) { return cplx<T>(-a.re, -a.im); }
template <typename T> __host__ __device__ __forceinline__ cplx<T> operator*(cplx<T> a, cplx<T> b) { return cplx<T>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
template <typename T> __host__ __device__ __forceinline__ cplx<T> operator*(cplx<T> a, T b) { return cplx<T>(a.re * b, a.im * b); }
template <typename T> __host__ __device__ __forceinline__ cplx<T> operator*(T a, cplx<T> b) { return cplx<T>(a * b.re, a * b.im); }
template <typename T> __host__ __device__ __forceinline__ cplx<T> operator/(cplx<T> a, T b) { return cplx<T>(a.re / b, a.im / b); }
template <typename T> __host__ __device__ __forceinline__ cplx<T> operator/(cplx<T> x, cplx<T> y) { // Smith's algorithm, as libgcc's __divdc3 / __divsc3
    const T a = x.re, b = x.im, c = y.re, d = y.im;
    if ((c < 0 ? -c : c) < (d < 0 ? -d : d)) {
        const T ratio = c / d, denom = c * ratio + d;
        return cplx<T>((a * ratio + b) / denom, (b * ratio - a) / denom);
    }
    const T ratio = d / c, denom = d * ratio + c;
    return cplx<T>((b * ratio + a) / denom, (b - a * ratio) / denom);
}
template <typename T> __host__ __device__ __forceinline__ cplx<T> &operator+=(cplx<T> &a, cplx<T> b) { a.re += b.re; a.im += b.im; return a; }
template <typename T> __host__ __device__ __forceinline__ cplx<T> &operator-=(cplx<T> &a, cplx<T> b) { a.re -= b.re; a.im -= b.im; return a; }
template <typename T> __host__ __device__ __forceinline__ bool operator==(cplx<T> a, cplx<T> b) { return a.re == b.re && a.im == b.im; }
template <typename T> __host__ __device__ __forceinline__ bool operator!=(cplx<T> a, cplx<T> b) { return !(a == b); }
template <typename T> struct cplx2 { cplx<T> x, y; }; // two adjacent coefficients of an R-stream row

// ---- type-generic helpers shared by the instantiations of kernels_body.hpp ----------------------------------------
__device__ __forceinline__ double hmx_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float hmx_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ cplx<double> hmx_fma(cplx<double> a, cplx<double> b, cplx<double> c) {
    return cplx<double>(__builtin_fma(a.re, b.re, __builtin_fma(-a.im, b.im, c.re)), __builtin_fma(a.re, b.im, __builtin_fma(a.im, b.re, c.im)));
}
__device__ __forceinline__ cplx<float> hmx_fma(cplx<float> a, cplx<float> b, cplx<float> c) {
    return cplx<float>(__builtin_fmaf(a.re, b.re, __builtin_fmaf(-a.im, b.im, c.re)), __builtin_fmaf(a.re, b.im, __builtin_fmaf(a.im, b.re, c.im)));
}
__host__ __device__ __forceinline__ double hmx_conj(double v) { return v; }
__host__ __device__ __forceinline__ float hmx_conj(float v) { return v; }
template <typename T> __host__ __device__ __forceinline__ cplx<T> hmx_conj(cplx<T> v) { return cplx<T>(v.re, -v.im); }
__host__ __device__ __forceinline__ double hmx_re(double v) { return v; }
__host__ __device__ __forceinline__ float hmx_re(float v) { return v; }
template <typename T> __host__ __device__ __forceinline__ T hmx_re(cplx<T> v) { return v.re; }
__device__ __forceinline__ double hmx_abs(double v) { return fabs(v); }
__device__ __forceinline__ float hmx_abs(float v) { return fabsf(v); }
__device__ __forceinline__ double hmx_abs(cplx<double> v) { return hypot(v.re, v.im); } // std::abs(std::complex) is hypot
__device__ __forceinline__ float hmx_abs(cplx<float> v) { return hypotf(v.re, v.im); }
// |v|^2 the way matrix/utils/math.hpp:7-16 accumulates it: pow(abs(v), 2)
__device__ __forceinline__ double hmx_abs2(double v) { return v * v; }
__device__ __forceinline__ float hmx_abs2(float v) { return v * v; }
template <typename T> __device__ __forceinline__ T hmx_abs2(cplx<T> v) { const T a = hmx_abs(v); return a * a; }
__host__ __device__ __forceinline__ bool hmx_is_zero(double v) { return v == 0.0; }
__host__ __device__ __forceinline__ bool hmx_is_zero(float v) { return v == 0.0f; }
template <typename T> __host__ __device__ __forceinline__ bool hmx_is_zero(cplx<T> v) { return v.re == T(0) && v.im == T(0); }
// c ? a : b, component by component for complex values (registers, never an indexed private array)
__device__ __forceinline__ double hmx_select(bool c, double a, double b) { return c ? a : b; }
__device__ __forceinline__ float hmx_select(bool c, float a, float b) { return c ? a : b; }
template <typename T> __device__ __forceinline__ cplx<T> hmx_select(bool c, cplx<T> a, cplx<T> b) { return cplx<T>(c ? a.re : b.re, c ? a.im : b.im); }
__device__ __forceinline__ double hmx_shfl_xor(double v, int o) { return __shfl_xor(v, o, WAVE); }
__device__ __forceinline__ float hmx_shfl_xor(float v, int o) { return __shfl_xor(v, o, WAVE); }
__device__ __forceinline__ int hmx_shfl_xor(int v, int o) { return __shfl_xor(v, o, WAVE); }
template <typename T> __device__ __forceinline__ cplx<T> hmx_shfl_xor(cplx<T> v, int o) { return cplx<T>(__shfl_xor(v.re, o, WAVE), __shfl_xor(v.im, o, WAVE)); }
__device__ __forceinline__ double hmx_shfl(double v, int l) { return __shfl(v, l, WAVE); }
__device__ __forceinline__ float hmx_shfl(float v, int l) { return __shfl(v, l, WAVE); }
template <typename T> __device__ __forceinline__ cplx<T> hmx_shfl(cplx<T> v, int l) { return cplx<T>(__shfl(v.re, l, WAVE), __shfl(v.im, l, WAVE)); }
__device__ __forceinline__ double readlane_val(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_val(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
template <typename T> __device__ __forceinline__ cplx<T> readlane_val(cplx<T> v, int lane) { return cplx<T>(readlane_val(v.re, lane), readlane_val(v.im, lane)); }

// ---- cross-lane moves without LDS (gfx950): DPP row operations and v_permlane{16,32}_swap, applied to every 32-bit word of a value
template <int CTRL, typename T>
__device__ __forceinline__ T dpp_move(T v) { // CTRL: DPP control (0x00-0xFF quad_perm, 0x121-0x12F row_ror, 0x140 row_mirror, 0x141 row_half_mirror)
    constexpr int N = sizeof(T) / 4;
    int w[N];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (int i = 0; i < N; i++)
        w[i] = __builtin_amdgcn_update_dpp(0, w[i], CTRL, 0xf, 0xf, false);
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}
// a: lanes 32..63 <-> b: lanes 0..31 (v_permlane32_swap); afterwards a + b is, in lanes 0..31, the sum over the lane pair
// (l, l + 32) of the OLD a, and in lanes 32..63 the pair sum of the OLD b
template <typename T>
__device__ __forceinline__ void lane_swap32(T &a, T &b) {
    constexpr int N = sizeof(T) / 4;
    int wa[N], wb[N];
    __builtin_memcpy(wa, &a, sizeof(T));
    __builtin_memcpy(wb, &b, sizeof(T));
#pragma unroll
    for (int i = 0; i < N; i++) {
        const auto r = __builtin_amdgcn_permlane32_swap(wa[i], wb[i], false, false);
        wa[i]        = r[0];
        wb[i]        = r[1];
    }
    __builtin_memcpy(&a, wa, sizeof(T));
    __builtin_memcpy(&b, wb, sizeof(T));
}
// a: odd rows of 16 lanes <-> b: even rows (v_permlane16_swap): a + b is the pair sum (l, l ^ 16) of the old a in even rows, of the old b in odd rows
template <typename T>
__device__ __forceinline__ void lane_swap16(T &a, T &b) {
    constexpr int N = sizeof(T) / 4;
    int wa[N], wb[N];
    __builtin_memcpy(wa, &a, sizeof(T));
    __builtin_memcpy(wb, &b, sizeof(T));
#pragma unroll
    for (int i = 0; i < N; i++) {
        const auto r = __builtin_amdgcn_permlane16_swap(wa[i], wb[i], false, false);
        wa[i]        = r[0];
        wb[i]        = r[1];
    }
    __builtin_memcpy(&a, wa, sizeof(T));
    __builtin_memcpy(&b, wb, sizeof(T));
}

// Every coefficient is read exactly once per product, so the stream loads are marked non-temporal: they do
// not displace x / Z / index lines from L2 and the Infinity Cache.  Measured at N=1e6 (fp64): expand 1.88 -> 1.74 ms,
// reduce 1.23-1.33 -> 1.20 ms, and the run-to-run bimodality disappears.  -DHMX_NT=0 disables.
#ifndef HMX_NT
#define HMX_NT 1
#endif
typedef double hmx_d2 __attribute__((ext_vector_type(2)));
typedef double hmx_d2u __attribute__((ext_vector_type(2), aligned(8))); // two adjacent doubles at an 8-byte aligned address (global_load_dwordx4 needs no more)
typedef float hmx_f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float hmx_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double stream_load(const double *p) {
#if HMX_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ float stream_load(const float *p) {
#if HMX_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ double2 stream_load(const double2 *p) {
#if HMX_NT
    const hmx_d2 v = __builtin_nontemporal_load(reinterpret_cast<const hmx_d2 *>(p));
    return make_double2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ float2 stream_load(const float2 *p) {
#if HMX_NT
    const hmx_f2 v = __builtin_nontemporal_load(reinterpret_cast<const hmx_f2 *>(p));
    return make_float2(v.x, v.y);
#else
    return *p;
#endif
}

typedef float hmx_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cplx<double> stream_load(const cplx<double> *p) { // one 16-byte load
#if HMX_NT
    const hmx_d2 v = __builtin_nontemporal_load(reinterpret_cast<const hmx_d2 *>(p));
    return cplx<double>(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ cplx<float> stream_load(const cplx<float> *p) {
#if HMX_NT
    const hmx_f2 v = __builtin_nontemporal_load(reinterpret_cast<const hmx_f2 *>(p));
    return cplx<float>(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ cplx2<double> stream_load(const cplx2<double> *p) { // two 16-byte loads
    cplx2<double> r;
    r.x = stream_load(reinterpret_cast<const cplx<double> *>(p));
    r.y = stream_load(reinterpret_cast<const cplx<double> *>(p) + 1);
    return r;
}
__device__ __forceinline__ cplx2<float> stream_load(const cplx2<float> *p) { // one 16-byte load
    cplx2<float> r;
#if HMX_NT
    const hmx_f4v v = __builtin_nontemporal_load(reinterpret_cast<const hmx_f4v *>(p));
    r.x = cplx<float>(v.x, v.y);
    r.y = cplx<float>(v.z, v.w);
#else
    r = *p;
#endif
    return r;
}

// ---- MFMA 16x16x4 (fp64 and exact-fp32 forms), used by the 16-right-hand-side kernels ------------------------------
// A: lane l holds A[m = l & 15][k = l >> 4]; B: lane l holds B[k = l >> 4][n = l & 15] (one element per lane).
// C/D: 4 results per lane, column n = l & 15; row = (l >> 4) + 4*reg for f64, 4*(l >> 4) + reg for f32
// (cdna_hip_programming.md section 3: the f64 form does NOT use the f32 row map).
typedef double hmx_d4 __attribute__((ext_vector_type(4)));
typedef float hmx_f4 __attribute__((ext_vector_type(4)));
// two adjacent stream elements p[0], p[1] with one load when both are wanted (n = how many of them exist: 0, 1 or 2): halves the number of
// load instructions of the MFMA kernels, whose 16-row / 16-column tiles otherwise fetch 8 bytes per lane
__device__ __forceinline__ void stream_load2(const double *p, int n, double &a, double &b) {
    if (n >= 2) {
        const hmx_d2u v = __builtin_nontemporal_load(reinterpret_cast<const hmx_d2u *>(p));
        a = v.x, b = v.y;
    } else {
        a = n == 1 ? __builtin_nontemporal_load(p) : 0.0;
        b = 0.0;
    }
}
__device__ __forceinline__ void stream_load2(const float *p, int n, float &a, float &b) {
    if (n >= 2) {
        const hmx_f2u v = __builtin_nontemporal_load(reinterpret_cast<const hmx_f2u *>(p));
        a = v.x, b = v.y;
    } else {
        a = n == 1 ? __builtin_nontemporal_load(p) : 0.f;
        b = 0.f;
    }
}
template <typename T> struct Acc4;
template <> struct Acc4<double> { typedef hmx_d4 type; };
template <> struct Acc4<float> { typedef hmx_f4 type; };
// Nothing is scheduled across this point: the software-pipelined kernels use it to keep their load groups in program order (s_waitcnt vmcnt
// counts loads in the order they were issued: a gather the compiler sinks behind the next step's stream loads turns every wait for it into a
// drain of the whole queue)
#define HMX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// occupancy hints of the matrix-core kernels (waves per SIMD the register allocator must make room for); empty: the compiler's choice.
// Set per build for A/B runs: tools/variant.sh <name> "-DHMX_WPE_EXPAND_MFMA16S_KERNEL=__attribute__((amdgpu_waves_per_eu(3)))"
#ifndef HMX_WPE_EXPAND_MFMA16S_KERNEL
// fp64: 172 registers by the compiler's own choice = 2 waves per SIMD; asked for 3 (<= 168 registers, no scratch) the expand stage of 16 fp64
// right-hand sides takes 2.04 instead of 2.35 ms on the same box (round 5).  4-byte coefficients need ~100 registers either way.
#define HMX_WPE_EXPAND_MFMA16S_KERNEL __attribute__((amdgpu_waves_per_eu(3)))
#endif
#ifndef HMX_WPE_REDUCE_MFMA16S_KERNEL
#if HMX_INST == 0
// fp64: 192 registers by the compiler's own choice = 2 waves per SIMD; asked for 3 (<= 168): see the A/B in profiles/r6_variants.log
#define HMX_WPE_REDUCE_MFMA16S_KERNEL __attribute__((amdgpu_waves_per_eu(3)))
#else
#define HMX_WPE_REDUCE_MFMA16S_KERNEL
#endif
#endif
// The stored-triangle kernels (round 6): two waves per SIMD for 8-byte coefficients (<= 256 registers of the unified file, accumulation registers
// included), three for 4-byte ones (<= 168).  Their dependent chains -- LDS round trip of the transposed tile, sixteen MFMAs into one
// accumulator -- are hidden by the OTHER waves of the SIMD, not by loads in flight: the same instructions at 264 registers (one wave per
// SIMD; the loop over a group's ranges had cost 24 registers) took 1.93 instead of 1.50 ms, SQ_WAVE_CYCLES halved (profiles/r6_pmc_ab_loop.log).
#ifndef HMX_WPE_EXPAND_SYM_MFMA16_KERNEL
#if HMX_INST == 0
#define HMX_WPE_EXPAND_SYM_MFMA16_KERNEL __attribute__((amdgpu_waves_per_eu(2)))
#else
#define HMX_WPE_EXPAND_SYM_MFMA16_KERNEL __attribute__((amdgpu_waves_per_eu(3)))
#endif
#endif
#ifndef HMX_WPE_ROWSYM_MFMA16_KERNEL
// (fp32 too: at three waves -- 168 registers -- the four instantiations of the interval loop spill 32 bytes into the hot loop)
#define HMX_WPE_ROWSYM_MFMA16_KERNEL __attribute__((amdgpu_waves_per_eu(2)))
#endif
// single-vector symmetric sweeps (empty: the compiler's choice; A/B by tools/variant.sh)
#ifndef HMX_WPE_EXPAND_SYM_KERNEL
#define HMX_WPE_EXPAND_SYM_KERNEL
#endif
#ifndef HMX_WPE_ROWSYM_KERNEL
#define HMX_WPE_ROWSYM_KERNEL
#endif
__device__ __forceinline__ hmx_d4 mfma16(double a, double b, hmx_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ hmx_f4 mfma16(float a, float b, hmx_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int mfma16_row(double, int lane, int reg) { return (lane >> 4) + 4 * reg; }
__device__ __forceinline__ int mfma16_row(float, int lane, int reg) { return 4 * (lane >> 4) + reg; }

} // namespace hmx
