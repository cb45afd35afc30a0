// kernels_body.hpp -- the kernels, written against `scalar` / `scalar2` (coefficient type) and `real` (its underlying
// real type) and included four times by engine.hip: namespaces hmx::f64 / hmx::f32 (scalar = real = double / float;
// htool's HMatrix<float,double>: fp32 coefficients, fp64 coordinates) and hmx::z64 / hmx::c32 (scalar = cplx<real>,
// HMX_COMPLEX = 1: htool's HMatrix<std::complex<...>>).  No include guard on purpose.
// Layout and design notes: kernels_common.hpp.  Compiled with -ffp-contract=off, FMAs are explicit (hmx_fma).

__device__ __forceinline__ scalar wave_sum(scalar v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += hmx_shfl_xor(v, o);
    return v;
}

template <typename V>
__device__ __forceinline__ V wave_sum_any(V v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += hmx_shfl_xor(v, o);
    return v;
}

// generator entry in coefficient precision.  KS_INV_DIST: real 1/den; complex (cre + i cim sgn)/den, component-wise division as
// std::complex<double> / double does.  KS_HELMHOLTZ: exp(i k r) / (p0 + p1 r) (real types: its real part).  KS_LAPLACE_SL:
// (cre [+ i cim]) / (4 pi (p0 + r)).
__device__ __forceinline__ scalar eval_scalar(const KernelSpec &ks, double tx, double ty, double tz, double sx, double sy, double sz) {
    if (ks.kind == KS_INV_DIST) {
#if HMX_COMPLEX
        const double den = eval_kernel_den(ks, tx, ty, tz, sx, sy, sz);
        const double u   = tx - sx;
        const double sgn = ks.herm ? (u > 0 ? 1.0 : (u < 0 ? -1.0 : 0.0)) : 1.0;
        return scalar((real)(ks.cre / den), (real)((ks.cim * sgn) / den));
#else
        return (scalar)eval_kernel(ks, tx, ty, tz, sx, sy, sz);
#endif
    }
    const double r = sqrt(eval_dist2(ks, tx, ty, tz, sx, sy, sz));
    if (ks.kind == KS_HELMHOLTZ) {
        const double den = ks.p0 + ks.p1 * r;
        double sn, cs;
        hmx_sincos(ks.wavenumber * r, sn, cs);
#if HMX_COMPLEX
        return scalar((real)(cs / den), (real)(sn / den));
#else
        return (scalar)(cs / den);
#endif
    }
    const double den = HMX_FOUR_PI * (ks.p0 + r); // KS_LAPLACE_SL
#if HMX_COMPLEX
    return scalar((real)(ks.cre / den), (real)(ks.cim / den));
#else
    return (scalar)(ks.cre / den);
#endif
}

// The kernels by stage of the path, in the order a build and a product use them:
#include "kernels_compress.hpp"  // a4-a7, a25: partialACA / sympartialACA (one workgroup per block, workgroup teams, host-generator form), fullACA, SVD, recompression
#include "kernels_pack.hpp"      // a3, a8-a10: the compressed blocks laid out as E- / R-streams
#include "kernels_matvec.hpp"    // a11-a15: reduce, combine, expand -- the headline product
#include "kernels_multi_rhs.hpp" // a18-a19: several right-hand sides per sweep (VALU, matrix cores)
#include "kernels_symmetric.hpp" // symmetric / Hermitian storage and transposed products on the stored data (one vector, several right-hand sides)
#include "kernels_util.hpp"      // a16-a17: permutations; conjugation, bulk download
