// grid_dev.h — cell addressing shared by the grid build and the search kernels.
//
// Cells are stored in a TILED order: the grid is cut into coarse cells of 8x8x8
// fine cells; linear index = coarse * 512 + ((z&7)*8 + (y&7))*8 + (x&7) with
// coarse = ((z>>3)*NY + (y>>3))*NX + (x>>3).  In the cell-sorted point arrays
//   * a run of x-adjacent fine cells inside one coarse cell is one contiguous range
//     (what a near query scans), and
//   * a whole coarse cell is one contiguous range (what a far query scans),
// so both ends of the search cost one (start,end) lookup per range.
#ifndef MVS_GRID_DEV_H_
#define MVS_GRID_DEV_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GridGeom {
    float minx, miny, minz, h, inv_h;
    int32_t nx, ny, nz;      // fine cells that can hold points
    int32_t NX, NY, NZ;      // coarse cells = ceil(n / 8)
};

// cell-space coordinate of x (float32, the same expression at build and query time)
__host__ __device__ inline float grid_cellf(float x, float mn, float inv_h) { return (x - mn) * inv_h; }

__device__ inline int grid_axis(float x, float mn, float inv_h, int n) {
    float f = floorf(grid_cellf(x, mn, inv_h));
    f = fminf(fmaxf(f, 0.0f), (float)(n - 1));     // NaN -> 0
    return (int)f;
}

__host__ __device__ inline int64_t grid_coarse(int NX, int NY, int X, int Y, int Z) { return ((int64_t)Z * NY + Y) * NX + X; }
// linear index of fine cell (x,y,z)
__host__ __device__ inline int64_t grid_index(int NX, int NY, int x, int y, int z) {
    return grid_coarse(NX, NY, x >> 3, y >> 3, z >> 3) * 512 + (((z & 7) * 8 + (y & 7)) * 8 + (x & 7));
}
// linear index of the first cell (x & 7 == 0) of row (y,z) inside coarse column X
__host__ __device__ inline int64_t grid_rowbase(int NX, int NY, int X, int y, int z) {
    return grid_coarse(NX, NY, X, y >> 3, z >> 3) * 512 + ((z & 7) * 8 + (y & 7)) * 8;
}

__device__ inline int64_t grid_cell(const GridGeom& g, float x, float y, float z) {
    const int cx = grid_axis(x, g.minx, g.inv_h, g.nx);
    const int cy = grid_axis(y, g.miny, g.inv_h, g.ny);
    const int cz = grid_axis(z, g.minz, g.inv_h, g.nz);
    return grid_index(g.NX, g.NY, cx, cy, cz);
}
#endif
