/* mvs_test.h — test hooks of libmvs_hip.so.  NOT part of the drop-in ABI (include/mvs.h): a host of the reference never
 * calls these.  They exist so that tests/ can force paths that timing alone rarely takes and look at tables the library
 * builds on the device.  Every piece of state they set lives in the handle they are given — nothing is process-wide. */
#ifndef MVS_TEST_H_
#define MVS_TEST_H_
#include "mvs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The handle's solver control block (csrc/engine.h, MVS_CTL_*): [0..7] escalation flag, worst rel^2, misses, solves, passes
 * finalized, pending prediction, prediction safety factor; then the ring of the last 32 passes x 8 solves: rel^2 of each solve's
 * result (-1: did not run), and the sweeps it ran (negative: no spare launch was left; a fractional quarter: it stopped on a
 * prediction).  n <= 8 + 2 * 256 + 16 doubles. */
int mvs_test_ctl(mvs_deform_t h, double* out, int n);

/* Cold start: mvs_set_device (or the first entry that needs a device) starts a helper thread that loads the library's code
 * objects and creates the first stream; this waits for it (measurements of the warm / cold first call). */
int mvs_test_preload_wait(void);

/* Tail loop of the patch solver's last launch (csrc/schwarz.hip), for THIS handle:
 *   maxspin  : polls a workgroup waits at the device-wide barrier before it abandons the solve (<= 0: default, 65536);
 *   plan_cap : at most this many launches per global solve, its remaining sweeps run inside the last one (0: no cap);
 *   skip_wg  : the workgroup of every tail launch that never arrives at the barrier, so that the bounded wait of every
 *              other workgroup expires and the solve is abandoned — deterministically (-1: none). */
int mvs_test_tail(mvs_deform_t h, int maxspin, int plan_cap, int skip_wg);

/* Geometry of the handle's target grid: out[0..2] = origin, [3] = cell edge, [4..6] = fine cells per axis, [7] = target points. */
int mvs_test_grid(mvs_deform_t h, double* out /*8*/);

/* mvs_deform_group_iterate re-checks between its batches (32 outer iterations) whether every handle still qualifies for group
 * launches and otherwise finishes the call handle by handle: this makes THIS handle stop qualifying once it has been harvested
 * `after_batches` times inside group calls (0: never). */
int mvs_test_group_leave(mvs_deform_t h, int after_batches);

/* Chebyshev steps every patch ran in the launch of sweep slot `slot` of the handle's last pass (slots are numbered through the
 * pass, solve 0's launches first) -> out[patches].  Equal numbers in every patch = every patch stopped at the same sweep. */
int mvs_test_sweep_steps(mvs_deform_t h, int slot, int32_t* out);

/* The heavy list of the handle's last association: entries, and how many had their coarse nearest-distance walk deferred. */
int mvs_test_heavy_count(mvs_deform_t h, int* n, int* flagged);

/* Tables the device mesh build left (csrc/meshbuild.hip), copied to the host.  what = 0: dims as int64[8] {NP, LS, W,
 * nslices, ne, single_pass, has_patches, total local rows}; 1 slice_off, 2 col, 3 opp0, 4 opp1, 5 vf_ptr, 6 vf, 7 pnloc,
 * 8 pown, 9 pnh, 10 l2g, 11 hl2g, 12 lcol (int16), 13 gent, 14 gcol.  out == NULL: only *bytes. */
int mvs_test_mesh_table(mvs_deform_t h, int what, void* out, int64_t* bytes);

#ifdef __cplusplus
}
#endif
#endif
