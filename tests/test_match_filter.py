"""Match-filter cascade (SURVEY §8(f) row 4, R/Processor/Processor.cpp:644-735): the oracle against a numpy restatement of
the three stages on a synthetic image pair, the HIP path against the oracle."""
import numpy as np
import pytest

W, H, VIEWS, WIN = 96, 72, 3, 3


def scene(seed=0, n=900):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    base = (96 + 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + 30 * np.sin((xx + 2 * yy) / 11.0)).astype(np.float64)
    img1 = np.clip(np.stack([base, base * 0.8 + 20, 255 - base], -1) + rng.normal(scale=2, size=(H, W, 3)), 0, 255).astype(np.uint8)
    shift = 4                                                        # image 2 = image 1 moved 4 px to the right, a little noise
    img2 = np.roll(img1, shift, axis=1).astype(np.int16) + rng.integers(-3, 4, size=(H, W, 3))
    img2 = np.clip(img2, 0, 255).astype(np.uint8)
    # texture-index tables of the generated views: view 0 = identity, views 1, 2 = shifted by one pixel, with holes
    idx = (yy * W + xx).astype(np.int32)
    tex = np.stack([idx, np.roll(idx, 1, axis=1), np.roll(idx, -1, axis=0)]).reshape(VIEWS, -1)
    tex1, tex2 = tex.copy(), tex.copy()
    tex1[:, rng.random(W * H) < 0.05] = -1
    tex2[:, rng.random(W * H) < 0.05] = -1
    valid1 = (rng.random(W * H) > 0.05).astype(np.uint8)
    valid2 = (rng.random(W * H) > 0.05).astype(np.uint8)
    # raw matches: mostly true correspondences (u2 = u1 + shift) seen in random generated views, some gross errors,
    # duplicates, and out-of-range pixels
    u1 = rng.integers(-2, W + 2, n)
    v1 = rng.integers(-2, H + 2, n)
    u2 = u1 + shift + np.where(rng.random(n) < 0.2, rng.integers(-15, 16, n), 0)
    v2 = v1 + np.where(rng.random(n) < 0.2, rng.integers(-10, 11, n), 0)
    raw = np.stack([rng.integers(0, VIEWS, n), u1, v1, rng.integers(0, VIEWS, n), u2, v2], 1).astype(np.int32)
    raw = np.concatenate([raw, raw[:100]])                           # exact duplicates
    return raw, tex1, valid1, tex2, valid2, img1, img2


def np_cascade(raw, tex1, valid1, tex2, valid2, img1, img2, win, ssd_err, interval):
    grey = lambda im: (im.astype(np.int64) @ np.array([4899, 9617, 1868]) + 8192) >> 14
    g1, g2 = grey(img1), grey(img2)
    uniq = set()
    for a1, u1, v1, a2, u2, v2 in raw:
        if not (0 <= u1 < W and 0 <= v1 < H and 0 <= u2 < W and 0 <= v2 < H):
            continue
        i1, i2 = tex1[a1, v1 * W + u1], tex2[a2, v2 * W + u2]
        if i1 != -1 and i2 != -1 and valid1[v1 * W + u1] and valid2[v2 * W + u2]:
            uniq.add((i1 % W, i1 // W, i2 % W, i2 // W))
    m = sorted(uniq)
    n1 = len(m)
    m2 = []
    for u1, v1, u2, v2 in m:
        if min(u1, v1, u2, v2) >= win and u1 < W - win and u2 < W - win and v1 < H - win and v2 < H - win:
            d = g1[v1 - win:v1 + win + 1, u1 - win:u1 + win + 1] - g2[v2 - win:v2 + win + 1, u2 - win:u2 + win + 1]
            if np.sqrt((d.astype(np.float64) ** 2).sum() / (2 * win + 1) ** 2) <= ssd_err:
                m2.append((u1, v1, u2, v2))
    kept = []
    for q in m2:
        if all((k[0] - q[0]) ** 2 + (k[1] - q[1]) ** 2 > interval ** 2 and (k[2] - q[2]) ** 2 + (k[3] - q[3]) ** 2 > interval ** 2 for k in kept):
            kept.append(q)
    return np.array(kept, np.int32).reshape(-1, 4), (n1, len(m2), len(kept))


def test_oracle_cascade_matches_numpy_restatement(oracle):
    args = scene()
    for ssd_err, interval in ((6.0, 5), (12.0, 0), (3.0, 9)):
        got, cnt = oracle.match_filter(*args, WIN, ssd_err, interval)
        want, wcnt = np_cascade(*args, WIN, ssd_err, interval)
        assert tuple(cnt) == wcnt and np.array_equal(got, want)
    got, cnt = oracle.match_filter(*args, WIN, 6.0, 5)
    assert cnt[0] > cnt[1] > cnt[2] > 10
    # what survives is the true 4-pixel shift (+- the one-pixel offsets of the generated views)
    assert (np.abs(got[:, 2] - got[:, 0] - 4) <= 1).mean() > 0.9 and (np.abs(got[:, 3] - got[:, 1]) <= 1).mean() > 0.9


@pytest.mark.gpu
def test_gpu_cascade_matches_oracle(oracle):
    from multiviewstitch_amd import processor
    for seed in (0, 1):
        args = scene(seed)
        for ssd_err, interval in ((6.0, 5), (12.0, 0)):
            g, gc = processor.MatchFilter(*args, WIN, ssd_err, interval)
            o, oc = oracle.match_filter(*args, WIN, ssd_err, interval)
            assert np.array_equal(gc, oc) and np.array_equal(g, o)
    g, gc = processor.MatchFilter(np.empty((0, 6), np.int32), *scene()[1:], WIN, 6.0, 5)
    assert len(g) == 0 and not gc.any()
