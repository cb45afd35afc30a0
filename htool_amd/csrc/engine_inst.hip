// engine_inst.hip -- ONE coefficient type of the engine (kernels_body.hpp + engine_body.hpp) per translation unit:
//   -DHMX_INST=0  hmx::f64  htool's HMatrix<double>            -DHMX_INST=2  hmx::z64  HMatrix<std::complex<double>>
//   -DHMX_INST=1  hmx::f32  HMatrix<float,double>              -DHMX_INST=3  hmx::c32  HMatrix<std::complex<float>>
// The four objects are compiled in parallel (make -j); engine.hip calls into them through engine_api.hpp.
// There is deliberately NO CPU fallback here: every entry point that computes needs a HIP device.
#include "engine_common.hpp"
#include "kernels_common.hpp"

namespace hmx {
#if HMX_INST == 0
#define HMX_COMPLEX 0
#define HMX_SPLIT_COLS 0
namespace f64 {
using real    = double;
using scalar  = double;
using scalar2 = double2;
#elif HMX_INST == 1
#define HMX_COMPLEX 0
#define HMX_SPLIT_COLS 0
namespace f32 {
using real    = float;
using scalar  = float;
using scalar2 = float2;
#elif HMX_INST == 2
#define HMX_COMPLEX 1
#define HMX_SPLIT_COLS 1
namespace z64 { // htool's HMatrix<std::complex<double>, double>
using real    = double;
using scalar  = cplx<double>;
using scalar2 = cplx2<double>;
#elif HMX_INST == 3
#define HMX_COMPLEX 1
#define HMX_SPLIT_COLS 0
namespace c32 { // HMatrix<std::complex<float>, double>
using real    = float;
using scalar  = cplx<float>;
using scalar2 = cplx2<float>;
#else
#error "HMX_INST must be 0 (f64), 1 (f32), 2 (z64) or 3 (c32)"
#endif
#include "engine_api.hpp"
#include "kernels_body.hpp"
#include "engine_body.hpp"
} // namespace f64 / f32 / z64 / c32
} // namespace hmx
