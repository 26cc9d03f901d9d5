// camera_dev.h — Camera's four maps as device inlines (R/Camera/Camera.cpp:40-72), shared by geom.hip, srt.hip, consist.hip.
// Operation order is the reference's (and the oracle's); the library is built with -ffp-contract=off.
#ifndef MVS_CAMERA_DEV_H_
#define MVS_CAMERA_DEV_H_
#include "dev_common.h"
#include "geom.h"

__device__ inline int32_t cvt_i32(double x) {            // (int)double with x86 cvttsd2si's out-of-range value
    return (x > -2147483649.0 && x < 2147483648.0) ? (int32_t)x : (int32_t)0x80000000;
}
// GetCamCoordFromImg (:40-44) then GetWorldCoordFromCam (:61-67)
__device__ inline d3 world_from_img(const CamDev& c, int u, int v, double d) {
    const d3 pc = mk3((u - c.cx) * d / c.fx, (v - c.cy) * d / c.fy, d);
    const d3 tmp = mk3(pc.x - c.t[0], pc.y - c.t[1], pc.z - c.t[2]);
    return mulMtv(c.R, tmp);
}
// GetCamCoordFromWorld (:68-72) then GetImgCoordFromCam (:45-48): C truncation toward zero, no z > 0 test
__device__ inline void img_from_world(const CamDev& c, d3 pw, int32_t* u, int32_t* v) {
    const d3 p = mk3(((c.R[0] * pw.x + c.R[1] * pw.y) + c.R[2] * pw.z) + c.t[0],
                     ((c.R[3] * pw.x + c.R[4] * pw.y) + c.R[5] * pw.z) + c.t[1],
                     ((c.R[6] * pw.x + c.R[7] * pw.y) + c.R[8] * pw.z) + c.t[2]);
    *u = cvt_i32(c.fx * p.x / p.z + c.cx + 0.5);
    *v = cvt_i32(c.fy * p.y / p.z + c.cy + 0.5);
}
#endif
