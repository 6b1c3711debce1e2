// repeat.hip — replicate a cell nx*ny*nz times (cell-major, original atoms innermost).
// Replaces src/repeat_cell.cpp:19-61.
#include "common.hpp"

namespace mdh {

struct Rows9 { double a[9]; };

__global__ __launch_bounds__(256) void k_repeat(double *__restrict__ newp, Rows9 bx, const double *__restrict__ oldp,
                                                int64_t n_old, int ny, int nz, int64_t total)
{
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // output atom
    if (o >= total)
        return;
    const int64_t cell = o / n_old, i = o % n_old;
    const int ix = (int)(cell / ((int64_t)ny * nz));
    const int64_t t = cell % ((int64_t)ny * nz);
    const int iy = (int)(t / nz), iz = (int)(t % nz);
    const double sx = ix * bx.a[0] + iy * bx.a[3] + iz * bx.a[6]; // :48-50
    const double sy = ix * bx.a[1] + iy * bx.a[4] + iz * bx.a[7];
    const double sz = ix * bx.a[2] + iy * bx.a[5] + iz * bx.a[8];
    newp[o * 3 + 0] = oldp[i * 3 + 0] + sx;
    newp[o * 3 + 1] = oldp[i * 3 + 1] + sy;
    newp[o * 3 + 2] = oldp[i * 3 + 2] + sz;
}

} // namespace mdh

using namespace mdh;

extern "C" int mdh_repeat_cell(double *new_pos, const double *old_box9_host, const double *old_pos, int64_t n_old,
                               int nx, int ny, int nz, int space, void *stream)
{
    if (n_old < 0 || nx <= 0 || ny <= 0 || nz <= 0) { set_error("mdh_repeat_cell: invalid argument"); return MDH_ERR_ARG; }
    const int64_t total = n_old * nx * ny * nz;
    if (total == 0)
        return MDH_OK;
    Scope sc(stream);
    const double *dold = sc.stage_in(old_pos, (size_t)n_old * 3, space);
    double *dnew = sc.stage(new_pos, (size_t)total * 3, space, false, true);
    if (sc.failed())
        return sc.error();
    Rows9 bx;
    for (int k = 0; k < 9; ++k) bx.a[k] = old_box9_host[k];
    hipLaunchKernelGGL(k_repeat, dim3(grid_for(total, 256)), dim3(256), 0, sc.stream(), dnew, bx, dold, n_old, ny, nz, total);
    return sc.finish(space);
}

MDH_WARM_UNIT(repeat)
