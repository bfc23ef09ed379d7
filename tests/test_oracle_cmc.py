"""N4 on the CPU: the ECC restatement recovers known camera motion, the polynomial sin/cos is accurate where it is used,
the warp moves track boxes as upstream's camera_update does, and the tracker with compensation keeps ids through a pan."""
import numpy as np

from oracle import cexact
from oracle.strongsort_np import OracleStrongSort
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.synth import make_stream


def _texture(h, w, seed=0):
    rng = np.random.default_rng(seed)
    base = rng.normal(0, 1, (h // 4 + 8, w // 4 + 8))
    img = np.kron(base, np.ones((4, 4)))[: h + 24, : w + 24]
    k = np.ones(5) / 5
    img = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, img)
    img = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, img)
    img = (img - img.min()) / (img.max() - img.min()) * 255
    return img


def test_sincos_polynomial():
    for t in (0.0, 1e-9, 0.01, -0.3, 0.7, -1.2):
        s, c = cexact.sincos(t)
        assert abs(s - np.sin(t)) < 1e-13 and abs(c - np.cos(t)) < 1e-13


def test_ecc_recovers_translation_and_rotation():
    big = _texture(72, 128)
    T = big[12:84, 12:140].astype(np.uint8)
    I = big[10:82, 15:143].astype(np.uint8)                       # the scene moved by (-3, +2) in the image: x' = x - 3, y' = y + 2
    warp, it = cexact.ecc(T, I)
    assert 1 <= it <= 100
    assert abs(warp[0, 2] + 3) < 0.05 and abs(warp[1, 2] - 2) < 0.05 and abs(warp[0, 0] - 1) < 1e-3 and abs(warp[1, 0]) < 1e-3
    # identical frames: identity after the first check
    warp, it = cexact.ecc(T, T)
    assert np.allclose(warp, [[1, 0, 0], [0, 1, 0]], atol=1e-6) and it >= 1
    # flat frames carry no gradient: no alignment
    assert cexact.ecc(np.full((72, 128), 7, np.uint8), np.full((72, 128), 7, np.uint8))[1] == -1


def test_gray_small_and_camera_update():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    g = cexact.gray_small(img, 72, 128)
    ref = (img[..., 0].astype(np.int64) * 1868 + img[..., 1].astype(np.int64) * 9617 + img[..., 2].astype(np.int64) * 4899 + 8192) >> 14
    assert g.shape == (72, 128) and abs(float(g.mean()) - float(ref.mean())) < 1.0
    mean = np.array([100.0, 200.0, 0.5, 80.0, 1.0, -1.0, 0.0, 0.0])
    out = cexact.camera_update(mean, np.array([[1, 0, 7.0], [0, 1, -3.0]]))
    assert np.allclose(out, [107.0, 197.0, 0.5, 80.0, 1.0, -1.0, 0.0, 0.0])
    th = 0.1
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0]])
    a, b = cexact.camera_update(mean, R), OracleStrongSort(StrongSortConfig(), "numpy").nx.camera_update(mean, R)
    assert np.allclose(a, b, rtol=0, atol=1e-9)


def test_compensation_keeps_ids_through_a_camera_jump():
    """A 60-pixel camera jump between two frames: without compensation the Mahalanobis gate rejects every match and the
    IoU stage cannot recover all of them; with the true warp the ids survive."""
    cfg = StrongSortConfig()
    def run(comp):
        st, orc = make_stream(3, 1280, 720, 12), OracleStrongSort(cfg, "c")
        ids = []
        for k in range(30):
            f = st.next_frame()
            d = f.dets.copy()
            shift = 60.0 if k >= 15 else 0.0
            d[:, [0, 2]] += shift
            warp = np.array([[1, 0, 60.0], [0, 1, 0.0]]) if (comp and k == 15) else None
            rows = orc.update(d, f.feats, (720, 1400), warp)
            ids.append(set(rows[:, 4].astype(int).tolist()))
        return ids
    with_c, without = run(True), run(False)
    assert with_c[20] == with_c[14] and len(with_c[14]) >= 10
    assert max(with_c[29]) < max(without[29])                      # the uncompensated run had to spawn new identities
