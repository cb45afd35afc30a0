// geometry.cpp -- seeded synthetic point clouds used by htool's tests and examples
// (testing/geometry.hpp:11-61: std::mt19937(0) + uniform_real_distribution<double>(0,1), libstdc++).
// bench.py and the tests use these so that the engine and the CPU reference see the same inputs.
#include <cmath>
#include <random>

#include "hmx_host.hpp"

namespace hmx {

namespace {
struct Uniform01 {
    std::mt19937 engine{0};
    std::uniform_real_distribution<double> dist{0, 1};
    double operator()() { return dist(engine); }
};
} // namespace

void make_geometry(const std::string &name, int n, double z, double *xyz) {
    Uniform01 u;
    if (name == "ball") { // create_sphere: uniform in the unit ball
        for (int j = 0; j < n; j++) {
            const double rho   = u();
            const double theta = 2 * M_PI * u();
            const double phi   = std::acos(2 * u() - 1);
            const double r     = std::cbrt(rho);
            xyz[3 * j + 0]     = 0. + r * std::sin(phi) * std::cos(theta);
            xyz[3 * j + 1]     = 0. + r * std::sin(phi) * std::sin(theta);
            xyz[3 * j + 2]     = 0. + r * std::cos(phi);
        }
        return;
    }
    // create_rotated_ellipse(dim, a, b, alpha=0, z, n): planar ellipse embedded in 3-D; disk = (1,1); "disk2d" is
    // the 2-D point cloud (no z column)
    const int dim  = name == "disk2d" ? 2 : 3;
    const double a = name == "ellipse" ? 4. : 1., b = 1., alpha = 0.;
    const double ca = std::cos(alpha), sa = std::sin(alpha);
    for (int j = 0; j < n; j++) {
        const double rho   = u();
        const double theta = u();
        const double r     = std::sqrt(rho);
        const double phi   = 2 * static_cast<double>(M_PI) * theta;
        const double xp    = a * r * std::cos(phi);
        const double yp    = b * r * std::sin(phi);
        xyz[dim * j + 0]   = ca * xp - sa * yp;
        xyz[dim * j + 1]   = sa * xp + ca * yp;
        if (dim == 3)
            xyz[dim * j + 2] = z;
    }
}

} // namespace hmx
