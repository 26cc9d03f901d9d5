"""Key-frame pair selection around RemoveOutliers (Processor::AlignmentSeq, R/Processor/Processor.cpp:746-765): the last
piece of the match-filter cascade of SURVEY §8(f)4.  The oracle walks the frame pairs one after the other as the reference
does; the engine runs each RANSAC round of ALL pairs as one launch set — selection, masks, residuals and the random
stream's final position must agree bit for bit, also when a pair stops early and shifts the stream for those after it."""
import numpy as np
import pytest

from multiviewstitch_amd import scene as S


def frame_pairs(seed, n1=3, n2=3, counts=None, bad_pair=None):
    """n1 x n2 frame pairs of two 'sequences' (views 0 and 1 of scene 0 shifted by a few degrees of yaw per frame)."""
    sc = S.make_scene(0)
    rng = np.random.default_rng(seed)
    s0, R0, t0 = sc.srt[0]
    s1, R1, t1 = sc.srt[1]
    s01, R01, t01 = s0 / s1, R1.T @ R0, (R1.T @ (t0 - t1)) / s1
    cams1 = [sc.cams[0]] * n1
    cams2 = [sc.cams[1]] * n2
    m = [[None] * n2 for _ in range(n1)]
    for i in range(n1):
        for j in range(n2):
            n = 40 if counts is None else counts[i][j]
            noise = 0.3 + 0.4 * ((3 * i + j) % 4)                  # pairs differ in quality: the selection has something to pick
            mm = S.make_matches(rng, cams1[i], cams2[j], s01, R01, t01, n=max(n, 1), outlier_frac=0.2, noise_px=noise)[:n]
            if bad_pair == (i, j):
                mm[:, 3:] += np.random.default_rng(1).normal(scale=30.0, size=mm[:, 3:].shape)   # garbage: nothing survives the pixel test
            m[i][j] = mm
    return cams1, cams2, m


def test_oracle_selection_follows_the_reference_rules(oracle):
    cams1, cams2, m = frame_pairs(5, counts=[[40, 3, 40], [40, 40, 6], [0, 40, 40]])
    r = oracle.select_keyframe_pair(cams1, cams2, m, min_match_count=7, iters=50, state=3)
    assert r["rc"] == 0 and r["pair_err"][0, 1] == np.inf and r["pair_err"][1, 2] == np.inf and r["pair_err"][2, 0] == np.inf    # skipped: < 7 matches
    assert r["n_keep"][0, 1] == 3 and all(r["keep"][0][1])                                 # their lists stay as they are
    ok = (r["n_keep"] >= 7) & np.isfinite(r["pair_err"])
    best = np.min(r["pair_err"][ok])
    i, j = np.argwhere(ok & (r["pair_err"] == best))[0]                                    # strict '<': the first of equal residuals
    assert (r["frm_idx1"], r["frm_idx2"]) == (i, j) and r["err"] == best
    # nothing qualifies -> the reference would exit(-1): rc -9
    r = oracle.select_keyframe_pair(cams1, cams2, m, min_match_count=50, iters=50, state=3)
    assert r["rc"] == -9 and r["frm_idx1"] == -1


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["plain", "skips", "early_stop"])
def test_gpu_selection_is_bit_equal_to_the_oracle(oracle, case):
    from multiviewstitch_amd import srt, _lib
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box")
    counts = {"plain": None, "skips": [[40, 3, 40], [40, 40, 6], [0, 40, 40]], "early_stop": None}[case]
    cams1, cams2, m = frame_pairs(11, counts=counts, bad_pair=(1, 0) if case == "early_stop" else None)
    o = oracle.select_keyframe_pair(cams1, cams2, m, min_match_count=7, iters=200, state=9)
    g = srt.select_keyframe_pair(cams1, cams2, m, min_match_count=7, iters=200, state=9)
    assert o["rc"] == 0
    assert (g["frm_idx1"], g["frm_idx2"]) == (o["frm_idx1"], o["frm_idx2"]) and g["state"] == o["state"]
    assert np.array_equal(g["n_keep"], o["n_keep"])
    for i in range(3):
        for j in range(3):
            assert np.array_equal(g["keep"][i][j], o["keep"][i][j]), (i, j)
    fin = np.isfinite(o["pair_err"])
    assert np.array_equal(np.isfinite(g["pair_err"]), fin)
    assert np.abs(g["pair_err"][fin] - o["pair_err"][fin]).max() <= 1e-9 * max(1.0, np.abs(o["pair_err"][fin]).max())
    assert abs(g["err"] - o["err"]) <= 1e-9 * max(1.0, abs(o["err"]))
    if case == "early_stop":
        assert o["n_keep"][1, 0] < 3                                   # that pair ran fewer than three rounds: the stream shifted
    # and the degenerate outcome raises like the reference exits
    with pytest.raises(_lib.MvsError) as e:
        srt.select_keyframe_pair(cams1, cams2, m, min_match_count=60, iters=50, state=9)
    assert e.value.code == -9
