// voro_host.cpp — TEST INFRASTRUCTURE: the product's Voronoi face routine (mdapy_amd/csrc/voro_core.hpp) compiled for the host,
// driven over a brute-force neighbourhood (all minimum images within rc), so that the CPU suite can compare the clipping
// construction with oracle/_ref (the reference's voro++) without a GPU.  Orthogonal boxes.  Never loaded by the product.
#include "../../mdapy_amd/csrc/voro_core.hpp"
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <vector>
using namespace voroc;
extern "C" int voroh_run(const double *x, const double *y, const double *z, int64_t N, const double *L3, const int *pbc3, const double *origin,
                         double rc, double *volume, int *nfaces, double *radius, int *incomplete)
{
    *incomplete = 0;
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
        std::vector<std::array<double, 3>> nrm;
        std::vector<double> off, dist;
        const double big = 4 * rc;
        const double pi[3] = {x[i] - origin[0], y[i] - origin[1], z[i] - origin[2]};
        for (int a = 0; a < 3; ++a) { // walls of open axes, else the bounding cube
            std::array<double, 3> n{0, 0, 0};
            n[a] = 1; nrm.push_back(n); off.push_back(pbc3[a] ? big : L3[a] - pi[a]); dist.push_back(off.back());
            n[a] = -1; nrm.push_back(n); off.push_back(pbc3[a] ? big : pi[a]); dist.push_back(off.back());
        }
        const int first_sorted = 6;
        for (auto &c : cs) { nrm.push_back({c.v[0], c.v[1], c.v[2]}); off.push_back(0.5 * c.d2); dist.push_back(0.5 * std::sqrt(c.d2)); }
        const int nc = (int)nrm.size();
        double vol = 0, mr2 = 0;
        int nf = 0;
        for (int f = 0; f < nc; ++f) {
            if (f < 6 && pbc3[f / 2]) continue; // the cube is not a face
            ptmc::PolyLocal poly;
            FaceResult2 r = voronoi_face_2d(poly, f, nc, (const double(*)[3])nrm.data(), off.data(), dist.data(), first_sorted, big); // the form the device runs
            if (r.overflow) return -2;
            if (face_exists(r, poly, dist[f])) { vol += r.area * dist[f] / 3.0; ++nf; mr2 = std::max(mr2, r.maxr2); }
        }
        volume[i] = vol; nfaces[i] = nf; radius[i] = std::sqrt(mr2);
        if (2 * radius[i] > rc) ++*incomplete;
    }
    return 0;
}
