"""N4 on the GPU: down-scaled grey frames and ECC warps equal the oracle bit for bit; the tracker with compensation equals
the oracle tracker fed the oracle's warps, frame by frame and in frame groups."""
import numpy as np
import pytest
import torch

from oracle import cexact
from oracle.strongsort_np import OracleStrongSort
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.synth import make_stream
from tests.gpu_util import engine
from tests.test_oracle_cmc import _texture

pytestmark = pytest.mark.gpu


def _scene(H, W, seed):
    """A big BGR canvas to cut shifted views from, with structure at the scale the 0.1x grey images keep."""
    small = np.stack([_texture((H + 200) // 10 + 1, (W + 200) // 10 + 1, seed + c) for c in range(3)], axis=2)
    from scipy.ndimage import zoom
    t = zoom(small, (10, 10, 1), order=1)[: H + 200, : W + 200]
    t = t + np.random.default_rng(seed).normal(0, 3, t.shape)
    return np.clip(t, 0, 255).astype(np.uint8)


def _oracle_warp(prev, cur):
    H, W = cur.shape[:2]
    hs, ws = int(H * 0.1), int(W * 0.1)
    w, it = cexact.ecc(cexact.gray_small(prev, hs, ws), cexact.gray_small(cur, hs, ws))
    out = np.zeros(8)
    if it < 0:
        out[:6], out[6] = [1, 0, 0, 0, 1, 0], -1
    else:
        w = w.copy(); w[0, 2] *= W / ws; w[1, 2] *= H / hs
        out[:6], out[6] = w.reshape(6), it
    return out


@pytest.mark.parametrize("wh,S,F", [((1280, 720), 1, 4), ((640, 480), 2, 3), ((1920, 1080), 1, 2)])
def test_ecc_warps_equal_oracle(wh, S, F):
    W, H = wh
    eng = engine(n_streams=S, debug=False)
    rng = np.random.default_rng(W + S)
    canv = [_scene(H, W, 10 * s) for s in range(S)]
    prev = [None] * S
    for call in range(3):
        frames = np.zeros((F, S, H, W, 3), np.uint8)
        ref = np.zeros((F, S, 8))
        for f in range(F):
            for s in range(S):
                if call == 1 and f == 1 and s == 0:
                    cur = np.full((H, W, 3), 90, np.uint8)                       # a flat frame: no alignment possible
                else:
                    ox, oy = int(rng.integers(60, 140)), int(rng.integers(60, 140))
                    cur = canv[s][oy:oy + H, ox:ox + W]
                    if call == 2:                                                # drift of a few pixels between frames
                        ox0, oy0 = 100 + 7 * f, 100 - 5 * f
                        cur = canv[s][oy0:oy0 + H, ox0:ox0 + W]
                frames[f, s] = cur
                if prev[s] is None:
                    ref[f, s, :6], ref[f, s, 6] = [1, 0, 0, 0, 1, 0], -1
                else:
                    ref[f, s] = _oracle_warp(prev[s], cur)
                prev[s] = cur
        d = torch.from_numpy(frames.reshape(F * S, H, W, 3)).to(eng.device)
        got = eng.cmc_estimate(d, F).cpu().numpy()
        assert np.array_equal(got[..., :7], ref[..., :7]), f"call {call}: max |diff| {np.abs(got[..., :7] - ref[..., :7]).max()}"
        if call == 2:
            assert (ref[1:, :, 6] >= 1).all() and np.abs(ref[1, 0, 2] + 7) < 1.5 and np.abs(ref[1, 0, 5] - 5) < 1.5    # the drift is recovered
    eng.close()


@pytest.mark.parametrize("F", [1, 4])
def test_tracker_with_compensation_equals_oracle(F):
    """camera pans by a few pixels per frame (the detections move with it); warps estimated on the device, applied inside
    k_frame before the prediction; oracle: same warps from its own ECC"""
    cfg = StrongSortConfig()
    W, H = 1280, 720
    eng, orc = engine(cfg, debug=True), OracleStrongSort(cfg, "c")
    canvas = _scene(H, W, 3)
    st = make_stream(9, W, H, 10)
    dev = eng.device
    hw = torch.tensor([[H, W]], dtype=torch.int32, device=dev)
    out, nout = torch.zeros(F, 1, 256, 8, device=dev), torch.zeros(F, 1, dtype=torch.int32, device=dev)
    prev, applied = None, 0
    for k0 in range(0, 24, F):
        hd, hf, hn = np.zeros((F, 1, 128, 6), np.float32), np.zeros((F, 1, 128, 512), np.float32), np.zeros((F, 1), np.int32)
        frames = np.zeros((F, 1, H, W, 3), np.uint8)
        ref = []
        for f in range(F):
            k = k0 + f
            pan = 3 * k if k >= 8 else 0                                            # the camera starts panning at frame 8
            cur = canvas[100:100 + H, 100 + pan:100 + pan + W]
            fr = st.next_frame()
            d = fr.dets.copy(); d[:, [0, 2]] -= pan                                # scene moves left in the image
            d[:, [0, 2]] = np.clip(d[:, [0, 2]], 0, W - 1)
            keep = d[:, 2] - d[:, 0] > 4
            d, ft = d[keep], fr.feats[keep]
            n = len(d)
            hd[f, 0, :n], hf[f, 0, :n], hn[f, 0] = d, ft, n
            frames[f, 0] = cur
            w8 = _oracle_warp(prev, cur) if prev is not None else None
            warp = w8[:6].reshape(2, 3) if (w8 is not None and w8[6] >= 1) else None
            applied += warp is not None
            ref.append(orc.update(d, ft, (H, W), warp))
            prev = cur
        warps = eng.cmc_estimate(torch.from_numpy(frames.reshape(F, H, W, 3)).to(dev), F)
        eng.set_cmc(warps)
        eng.update_group(F, torch.from_numpy(hd).to(dev), torch.from_numpy(hn).to(dev), torch.from_numpy(hf).to(dev), hw, out, nout)
        eng.check_errors()
        ho, hno = out.cpu().numpy(), nout.cpu().numpy()
        for f in range(F):
            got = ho[f, 0, :hno[f, 0]]
            assert got.shape == ref[f].shape and got.tobytes() == ref[f].tobytes(), f"frame {k0 + f}"
    t, o = eng.tracks(0), orc.snapshot()
    assert np.array_equal(t["track_id"], o["track_id"]) and t["mean"].tobytes() == o["mean"].tobytes()
    assert applied >= 20
    eng.close()
