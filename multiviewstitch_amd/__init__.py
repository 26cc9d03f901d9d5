"""multiviewstitch_amd — MI355X-native SRT + node-driven deformation engine.

Host-side mirror of the reference's solver classes (SRTSolver, Camera,
Deformation — R/Solver, R/Camera, R/Deformation) over the C-ABI in
include/mvs.h.  The compute path is the HIP library libmvs_hip.so
(multiviewstitch_amd/csrc); importing the solver modules fails loudly when it
is missing.  There is no CPU fallback.
"""
__version__ = "0.1.0"
