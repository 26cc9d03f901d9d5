// grid_dev.h — cell addressing shared by the grid build and the search kernels.
#ifndef MVS_GRID_DEV_H_
#define MVS_GRID_DEV_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GridGeom {
    float minx, miny, minz, h, inv_h;
    int32_t nx, ny, nz;
};

// cell-space coordinate of x (float32, the same expression at build and query time)
__device__ inline float grid_cellf(float x, float mn, float inv_h) { return (x - mn) * inv_h; }

__device__ inline int grid_axis(float x, float mn, float inv_h, int n) {
    float f = floorf(grid_cellf(x, mn, inv_h));
    f = fminf(fmaxf(f, 0.0f), (float)(n - 1));     // NaN -> 0
    return (int)f;
}

__device__ inline int grid_cell(const GridGeom& g, float x, float y, float z) {
    const int cx = grid_axis(x, g.minx, g.inv_h, g.nx);
    const int cy = grid_axis(y, g.miny, g.inv_h, g.ny);
    const int cz = grid_axis(z, g.minz, g.inv_h, g.nz);
    return (cz * g.ny + cy) * g.nx + cx;
}
#endif
