"""Mesh -> inverse-depth raster (SURVEY §8(f) row 3, R/Model2Depth/Model2Depth.cpp:58-156): known answers for the oracle
(a triangulated depth map rendered back through its own camera), bit-exact parity of the HIP kernels on the GPU."""
import numpy as np
import pytest

from multiviewstitch_amd import scene as S
from tests.util import scene_and_target


def view_mesh(oracle, config=1, k=0):
    sc, _, _, _ = scene_and_target(config)
    pts, nrm, tex, faces = oracle.depth_to_model(sc.depth[k], sc.cams[k], S.MIN_DSP, S.MAX_DSP, S.SMOOTH)
    return sc, pts, faces, tex


def test_oracle_render_back_reproduces_the_depth_map(oracle):
    """The triangulated raster rendered through its own camera: a pixel centre (i+.5, j+.5) lies on the diagonal
    (j,i)-(j+1,i+1) of the quad Depth2Model splits (Depth2Model.cpp:45-77), and inverse depth is affine in window
    space, so the rendered value is the mean of the two diagonal samples."""
    sc, pts, faces, tex = view_mesh(oracle)
    cam, d = sc.cams[0], sc.depth[0].astype(np.float64)
    r = oracle.render_depth(pts, faces, cam).astype(np.float64)
    valid = (d >= S.MIN_DSP) & (d <= S.MAX_DSP)
    # quads that Depth2Model triangulated completely (both triangles passed its smoothness test)
    quad_of_face = tex[faces].min(1)                                              # raster index of the quad's top-left pixel
    full = (np.bincount(quad_of_face, minlength=d.size) == 2).reshape(d.shape)[:-1, :-1]
    want = 0.5 * (d[:-1, :-1] + d[1:, 1:])
    got = r[:-1, :-1]
    assert full.sum() > 10000 and (got[full] > 0).mean() > 0.999
    hit = full & (got > 0)
    assert np.abs(got[hit] / want[hit] - 1).max() < 2e-4                          # float32 vertex stage + float32 depth buffer
    assert not r[~np.pad(valid[:-1, :-1] | valid[1:, 1:] | valid[1:, :-1] | valid[:-1, 1:], ((0, 1), (0, 1)))].any()   # background stays 0


def test_oracle_render_occlusion_and_clipping(oracle):
    cam = S.Camera(120.0, 120.0, 49.5, 39.5, np.eye(3), np.zeros(3), 100, 80)
    quad = lambda z, s: np.array([[-s, -s, z], [s, -s, z], [s, s, z], [-s, s, z]], float)
    pts = np.concatenate([quad(4.0, 1.0), quad(2.0, 0.3), quad(-1.0, 5.0)])       # far wall, near plate, a plate behind the eye
    faces = np.array([[0, 1, 2], [0, 2, 3], [4, 6, 5], [4, 7, 6], [8, 9, 10], [8, 10, 11]], np.int32)   # second plate clockwise
    r = oracle.render_depth(pts, faces, cam)
    assert abs(r[40, 50] - 0.5) < 1e-4 and abs(r[40, 30] - 0.25) < 1e-4 and r[2, 2] == 0     # nearest wins, either winding, behind = dropped
    assert set(np.round(np.unique(r), 3)) == {0.0, 0.25, 0.5}


@pytest.mark.gpu
def test_gpu_render_is_bit_exact(oracle):
    import torch
    from multiviewstitch_amd import processor
    sc, pts, faces, _ = view_mesh(oracle)
    for cam in (sc.cams[0], sc.cams[1]):                     # its own camera, and a view the mesh was not made from
        assert np.array_equal(processor.RenderDepth(pts, faces, cam), oracle.render_depth(pts, faces, cam))
    # closed template mesh, device-resident path
    dev = torch.device("cuda", 0)
    tp, tf = torch.from_numpy(sc.verts).to(dev), torch.from_numpy(sc.faces.astype(np.int32)).to(dev)
    cam = S.Camera(300.0, 300.0, 159.5, 119.5, *S._look_at(np.array([2.5, 1.0, 0.7])), 320, 240)
    out = torch.empty((240, 320), dtype=torch.float32, device=dev)
    processor.RenderDepth((tp.data_ptr(), len(sc.verts)), (tf.data_ptr(), len(sc.faces)), cam, out_dev=out.data_ptr())
    want = oracle.render_depth(sc.verts, sc.faces, cam)
    assert np.array_equal(out.cpu().numpy(), want) and (want > 0).mean() > 0.1
