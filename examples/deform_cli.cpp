// deform_cli.cpp — Processor::Deform (R/Processor/Processor.cpp:1111-1138) as a stand-alone program on top of the C-ABI:
// what a C++ host of the reference links against once `Deformation`, `Alignment` and the OBJ reader are replaced by
// libmvs_hip.so.  No Python, no torch: only include/mvs.h + include/mvs_io.h.
//
//   deform_cli Model.obj meanbody.obj parts out.obj  dist_thres  r00 r01 r02 r10 r11 r12 r20 r21 r22
//
// (the nine numbers are the rotation of cameras[0][0]; its third row is the view ray handed to Alignment::Align)
#include <cstdio>
#include <cstdlib>

#include "mvs_io.h"

int main(int argc, char** argv) {
    if (argc != 15) {
        std::fprintf(stderr, "usage: %s Model.obj meanbody.obj parts out.obj dist_thres r00 r01 r02 r10 r11 r12 r20 r21 r22\n", argv[0]);
        return 2;
    }
    double R[9];
    for (int k = 0; k < 9; ++k) R[k] = std::atof(argv[6 + k]);
    if (mvs_device_count() == 0) { std::fprintf(stderr, "no HIP device: %s\n", mvs_last_error()); return 3; }
    mvs_deform_params prm;
    mvs_deform_default_params(&prm);
    mvs_deform_stats st;
    const int rc = mvs_processor_deform(argv[1], argv[2], argv[3], R, std::atof(argv[5]), &prm, argv[4], &st);
    if (rc < 0) { std::fprintf(stderr, "mvs_processor_deform failed (%d): %s\n", rc, mvs_last_error()); return 1; }
    if (rc > 0) std::fprintf(stderr, "warning (%d): %s\n", rc, mvs_last_error());
    std::printf("deformed: %d ARAP iterations, %d valid nodes, energy %.6g, solver residual %.2e -> %s\n", st.arap_iters_run, st.n_valid,
                st.energy[st.arap_iters_run > 0 ? st.arap_iters_run - 1 : 0], st.cg_rel_residual, argv[4]);
    return 0;
}
