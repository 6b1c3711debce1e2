// TEST INFRASTRUCTURE (tools/voro_sliver_scan.py): per atom the face count under the area rule of rounds 1-5 and under the width rule
#include "../mdapy_amd/csrc/voro_core.hpp"
namespace voroc { constexpr double AREA_TOL = 1e-14; }
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <vector>
using namespace voroc;
// per atom: faces under the area rule (A), faces under the width rule (B), and the list of small faces (area/d^2, width)
extern "C" int dbg_voro(const double *x, const double *y, const double *z, int64_t N, const double *L3, const int *pbc3, const double *origin,
                        double rc, int *nfA, int *nfB, double *small, int small_cap, int *nsmall)
{
    *nsmall = 0;
    for (int64_t i = 0; i < N; ++i) {
        struct C { double d2, v[3]; };
        std::vector<C> cs;
        for (int64_t j = 0; j < N; ++j) {
            if (j == i) continue;
            double d[3] = {x[j] - x[i], y[j] - y[i], z[j] - z[i]};
            for (int a = 0; a < 3; ++a) if (pbc3[a]) d[a] -= L3[a] * std::floor(d[a] / L3[a] + 0.5);
            const double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            if (d2 <= rc * rc) cs.push_back({d2, {d[0], d[1], d[2]}});
        }
        std::sort(cs.begin(), cs.end(), [](const C &a, const C &b) { return a.d2 < b.d2; });
        std::vector<std::array<double, 3>> nrm; std::vector<double> off, dist;
        const double big = 4 * rc;
        const double pi[3] = {x[i] - origin[0], y[i] - origin[1], z[i] - origin[2]};
        for (int a = 0; a < 3; ++a) {
            std::array<double, 3> n{0, 0, 0};
            n[a] = 1; nrm.push_back(n); off.push_back(pbc3[a] ? big : L3[a] - pi[a]); dist.push_back(off.back());
            n[a] = -1; nrm.push_back(n); off.push_back(pbc3[a] ? big : pi[a]); dist.push_back(off.back());
        }
        for (auto &c : cs) { nrm.push_back({c.v[0], c.v[1], c.v[2]}); off.push_back(0.5 * c.d2); dist.push_back(0.5 * std::sqrt(c.d2)); }
        const int nc = (int)nrm.size();
        int a_ = 0, b_ = 0;
        for (int f = 0; f < nc; ++f) {
            if (f < 6 && pbc3[f / 2]) continue;
            ptmc::PolyLocal poly;
            FaceResult2 r = voronoi_face_2d(poly, f, nc, (const double(*)[3])nrm.data(), off.data(), dist.data(), 6, big);
            if (r.overflow) return -2;
            if (r.area > AREA_TOL * dist[f] * dist[f]) ++a_;
            if (r.area > 0 && r.nv >= 3) {
                const double rel = r.area / (dist[f] * dist[f]);
                double w = -1;
                if (rel < 1e-8) { w = ptmc::poly_width2(poly, r.nv); if (*nsmall < small_cap) { small[3 * *nsmall] = (double)i; small[3 * *nsmall + 1] = rel; small[3 * *nsmall + 2] = w; ++*nsmall; } }
                if (face_exists(r, poly, dist[f])) ++b_; // (the product's rule)
            }
        }
        nfA[i] = a_; nfB[i] = b_;
    }
    return 0;
}
