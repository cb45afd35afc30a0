// engine_body.hpp -- host orchestration of the device engine, written against `scalar` (coefficient type; `real` is its
// underlying real type) and included four times by engine.hip: namespaces hmx::f64, hmx::f32 (scalar = double / float) and
// hmx::z64, hmx::c32 (scalar = cplx<double> / cplx<float>, HMX_COMPLEX = 1).  No include guard on purpose.

// The host code by stage, in dependency order:
#include "engine_state.hpp"    // struct HMat: what lives in HBM; launch-order helper
#include "engine_layout.hpp"   // a8-a10: compressed blocks -> E- / R-streams, index tables of the mirrored products
#include "engine_products.hpp" // a11-a19: the sweeps of a product on the device
#include "engine_build.hpp"    // a3-a9, a25: create, options, compression (device / host generator), recompression, finalize
#include "engine_access.hpp"   // a8: blocks back to the host, save / load, statistics
#include "engine_entry.hpp"    // a16, a19, a21: the product entry points behind the C ABI
