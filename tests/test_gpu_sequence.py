"""Sequence parity: device-resident tracker vs the oracle on seeded synthetic streams.
ids / det indices / classes exact; boxes exact (int-truncated Kalman state); track table and every
stage intermediate (cosine, Mahalanobis, cost matrices, assignment lists) bit-exact."""
import numpy as np
import pytest

from oracle.strongsort_np import OracleStrongSort
from strongsort_yolo_amd.config import StrongSortConfig
from strongsort_yolo_amd.synth import make_stream
from tests.gpu_util import engine, bits_equal, unit

pytestmark = pytest.mark.gpu


def _compare_frame(eng, orc, rows_g, rows_o, k, s=0, f=0, last=None):
    """rows + every stage intermediate of frame k (position f of its group) of stream s"""
    dbg, last = eng.debug(s, f), (orc.last if last is None else last)
    assert rows_g.shape == rows_o.shape, f"frame {k}: {rows_g.shape} vs {rows_o.shape}"
    assert dbg["n_conf"] == len(last["confirmed"]), f"frame {k}: confirmed tracks"
    assert bits_equal(dbg["cos"], last["cos"]), f"frame {k}: cosine (max abs diff {np.abs(dbg['cos'] - last['cos']).max() if dbg['cos'].shape == last['cos'].shape and dbg['cos'].size else 'shape'})"
    assert bits_equal(dbg["maha"], last["maha"]), f"frame {k}: maha"
    assert np.array_equal(dbg["gated"], last["gated"])
    assert bits_equal(dbg["cost_a"], last["cost_a"]), f"frame {k}: cost_a"
    pa = np.full(len(last["confirmed"]), -1, np.int32)
    for r, c in last["pairs_a"]:
        pa[r] = c
    assert np.array_equal(dbg["pairs_a"], pa)
    assert np.array_equal(dbg["cand"], np.asarray(last["cand"], np.int32))
    assert np.array_equal(dbg["cols_b"], np.asarray(last["cols_b"], np.int32))
    if dbg["n_cand"] and dbg["n_cols"]:
        assert bits_equal(dbg["cost_b"], last["cost_b"]), f"frame {k}: cost_b"
    assert bits_equal(rows_g, rows_o), f"frame {k}: output rows differ"


def _compare_table(eng, orc, s=0):
    t, o = eng.tracks(s), orc.snapshot()
    for key in ("track_id", "state", "hits", "age", "tsu", "gal_count"):
        assert np.array_equal(t[key], o[key]), key
    assert t["next_id"] == o["next_id"]
    assert bits_equal(t["mean"], o["mean"]) and bits_equal(t["cov"], o["cov"])
    assert bits_equal(t["smooth"], o["smooth"])


@pytest.mark.parametrize("n_ids,wh,frames,seed", [(30, (1280, 720), 150, 0), (100, (1920, 1080), 112, 1), (8, (640, 480), 60, 2)])
def test_stream_parity(n_ids, wh, frames, seed):
    W, H = wh
    cfg = StrongSortConfig()
    eng = engine(cfg)
    orc = OracleStrongSort(cfg, "c")
    sg, so = make_stream(seed, W, H, n_ids), make_stream(seed, W, H, n_ids)
    for k in range(frames):
        fg, fo = sg.next_frame(), so.next_frame()
        rows_g = eng.update_host(fg.dets, fg.feats, (H, W))
        rows_o = orc.update(fo.dets, fo.feats, (H, W))
        _compare_frame(eng, orc, rows_g, rows_o, k)
        if k % 10 == 9:
            _compare_table(eng, orc)
    _compare_table(eng, orc)
    # gallery content as a set of rows (ring order is irrelevant to the min)
    for i, t in enumerate(orc.tracks[:5]):
        g = eng.gallery(0, i)
        assert len(g) == len(t.gallery)
        a = {r.tobytes() for r in g}
        assert a == {r.tobytes() for r in t.gallery}
    eng.close()


def test_births_deaths_and_empty_frames():
    cfg = StrongSortConfig(max_age=5)
    eng, orc = engine(cfg), OracleStrongSort(cfg, "c")
    sg, so = make_stream(7, 640, 480, 10, p_vanish=0.15, vanish_max=12), make_stream(7, 640, 480, 10, p_vanish=0.15, vanish_max=12)
    empty = (np.zeros((0, 6), np.float32), np.zeros((0, 512), np.float32))
    for k in range(120):
        fg, fo = sg.next_frame(), so.next_frame()
        if k % 17 == 16:                                # drop every detection now and then
            rg, ro = eng.update_host(*empty, (480, 640)), orc.update(*empty, (480, 640))
        else:
            rg, ro = eng.update_host(fg.dets, fg.feats, (480, 640)), orc.update(fo.dets, fo.feats, (480, 640))
        _compare_frame(eng, orc, rg, ro, k)
    _compare_table(eng, orc)
    assert orc.next_id > 11                             # re-births happened
    eng.close()


def _run_streams(S, ids_of, frames, F=1, mute=(), late=(), cfg=None, wh=(1280, 720), stream_kw=None, empty_every=0, opts=None):
    """S streams in one context, fed in groups of F frames (one ss_track_update_group call per group: the
    association kernel sees the detections of all F frames at once), against S single-stream oracles that run frame by
    frame; rows and every stage intermediate of every frame and stream are compared (rows a6-a10).
    `mute` streams never see a detection, `late` streams see their first detection two frames before the end
    (tentative tracks only), `empty_every` drops all detections of every n-th frame."""
    import torch
    cfg = cfg or StrongSortConfig()
    W, H = wh
    eng = engine(cfg, n_streams=S, debug=True)
    for name, value in (opts or {}).items():
        eng.set_option(name, value)
    orcs = [OracleStrongSort(cfg, "c") for _ in range(S)]
    streams = [make_stream(10 + s, W, H, ids_of(s), **(stream_kw or {})) for s in range(S)]
    dev = eng.device
    hw = torch.tensor([[H, W]] * S, dtype=torch.int32, device=dev)
    out = torch.zeros(F, S, 256, 8, device=dev)
    nout = torch.zeros(F, S, dtype=torch.int32, device=dev)
    seen_conf = [0] * S
    for k0 in range(0, frames, F):
        nf = min(F, frames - k0)
        hd, hf, hn = np.zeros((F, S, 128, 6), np.float32), np.zeros((F, S, 128, 512), np.float32), np.zeros((F, S), np.int32)
        ref, lasts = [], []
        for f in range(nf):
            k = k0 + f
            for s in range(S):
                fr = streams[s].next_frame()
                n = len(fr.dets)
                if s in mute or (s in late and k < frames - 2) or (empty_every and k % empty_every == empty_every - 1):
                    n = 0
                hd[f, s, :n], hf[f, s, :n], hn[f, s] = fr.dets[:n], fr.feats[:n], n
                ref.append(orcs[s].update(fr.dets[:n], fr.feats[:n], (H, W)))
                lasts.append(orcs[s].last)
                seen_conf[s] = max(seen_conf[s], len(orcs[s].last["confirmed"]))
        eng.update_group(nf, torch.from_numpy(hd).to(dev), torch.from_numpy(hn).to(dev), torch.from_numpy(hf).to(dev), hw, out, nout)
        eng.check_errors()
        ho, hno = out.cpu().numpy(), nout.cpu().numpy()
        for f in range(nf):
            for s in range(S):
                _compare_frame(eng, orcs[s], ho[f, s, :hno[f, s]], ref[f * S + s], k0 + f, s, f, lasts[f * S + s])
    for s in range(S):
        _compare_table(eng, orcs[s], s)
    eng.close()
    return seen_conf, orcs


def test_multi_stream_batch_equals_single_streams():
    _run_streams(3, lambda s: 20 + 5 * s, 40)


@pytest.mark.parametrize("S,frames", [(4, 112), (8, 30), (32, 22)])
def test_many_streams_per_launch(S, frames):
    """Detections per stream <= 16, 17..32 and > 32 (one, two and three column-tile pairs), galleries of 1, 15, 16,
    17 ... rows while they fill (100 rows at S = 4), a stream that never sees a detection and one without a confirmed
    track — the launch shapes behind bench.py's roofline_batched."""
    ids = [10, 24, 40, 30]
    conf, _ = _run_streams(S, lambda s: ids[s % 4], frames, mute=(S - 1,), late=(S - 2,) if S > 4 else ())
    assert conf[0] > 0 and conf[1] > 16 and conf[2] > 32 and conf[S - 1] == 0
    if S > 4:
        assert conf[S - 2] == 0


@pytest.mark.parametrize("S,F,frames,ids", [(1, 8, 124, 30), (1, 16, 120, 30), (2, 5, 63, 24), (4, 8, 112, 0), (1, 3, 30, 100), (1, 32, 168, 30), (2, 21, 130, 24),
                                            (1, 32, 150, 100)])
def test_frame_groups_equal_frame_by_frame(S, F, frames, ids):
    """The group call (galleries read once for F frames: rows that leave the ring during the group are masked per
    frame in k_assoc, the rows appended during the group are added by k_newrow) == the oracle's frame-by-frame
    result, every frame, every intermediate.  Covers ring wrap-around inside a group (gallery full from frame 103),
    a partial last group, 100 detections per frame (4 column-tile pairs) and mixed streams."""
    mixed = [10, 24, 40, 30]
    wh = (1920, 1080) if ids == 100 else (1280, 720)
    conf, orcs = _run_streams(S, (lambda s: mixed[s % 4]) if ids == 0 else (lambda s: ids + 3 * s), frames, F=F, wh=wh)
    assert max(conf) >= (90 if ids == 100 else 20)
    if frames > 110:
        assert max(len(t.gallery) for t in orcs[0].tracks) == 100           # the ring wrapped


@pytest.mark.parametrize("stage", [0, 1, 2, 4, 5, 104])
@pytest.mark.parametrize("S,F,frames,ids", [(1, 32, 136, 30), (32, 8, 24, 0), (2, 5, 25, 5), (1, 32, 64, 100)])
def test_assoc_operand_staging_variants(stage, S, F, frames, ids):
    """Every staging form of k_assoc's detection operand (`assoc_stage`: registers, or LDS-DMA in 1 / 2 / 4 pieces awaited
    step by step) gives the same bits: full galleries with split tiles and composite tiles (1 x 32 frames), workgroups that
    process several records (32 streams), records of fewer than 8 tiles (waves without a run that only take part in the
    barriers) and four column-tile pairs per frame."""
    mixed = [10, 24, 40, 30]
    wh = (1920, 1080) if ids == 100 else (1280, 720)
    _run_streams(S, (lambda s: mixed[s % 4]) if ids == 0 else (lambda s: ids + s), frames, F=F, wh=wh, opts={"assoc_stage": stage % 100, "assoc_xcd_map": stage // 100})


def test_frame_caps_beyond_the_lds_are_rejected():
    """128 x 128 cost entries + the per-track areas would need 173 KB of LDS: the option refuses it (112 is the largest that fits)."""
    from strongsort_yolo_amd.lib import SSError
    eng = engine(StrongSortConfig())
    with pytest.raises(SSError):
        eng.set_option("frame_caps", 128)
    eng.set_option("frame_caps", 112)
    eng.close()


@pytest.mark.parametrize("caps,ids,wh", [(32, 30, (1280, 720)), (16, 30, (1280, 720)), (64, 100, (1920, 1080)), (0, 100, (1920, 1080)), (32, 12, (640, 480)),
                                         (112, 100, (1920, 1080))])
def test_frame_kernel_work_areas_in_lds_or_global_scratch(caps, ids, wh):
    """`frame_caps`: k_frame keeps its f64 work arrays (Cholesky factors, predicted boxes, detection vectors, cost matrix) in LDS up to
    that many tracks / detections and takes them from the stream's global scratch beyond — 30 and 100 identities against caps of 16,
    32, 64 and the maxima: tracks over, detections over, cost matrix spilled, everything inside; every frame and intermediate equal."""
    _run_streams(2, lambda s: ids + 2 * s, 45, F=5, wh=wh, opts={"frame_caps": caps})


@pytest.mark.parametrize("pack", [0, 1])
@pytest.mark.parametrize("S,F,frames,ids,wh", [(1, 32, 150, 30, (1280, 720)), (4, 16, 64, 0, (1280, 720)), (2, 8, 48, 100, (1920, 1080)), (3, 5, 40, 3, (640, 480))])
def test_association_on_packed_or_per_frame_columns(pack, S, F, frames, ids, wh):
    """assoc_pack: k_assoc's column-tile pairs over the group's detections packed across frames (a pair's 32 columns may belong to
    several frames: per-column ring validity and M address) against pairs of one frame's tiles: every intermediate of every frame
    equals the oracle either way; mixed identity counts, frames without detections, (100, 100) shapes, tiny streams (a group's
    detections fit one pair)."""
    mixed = (30, 8, 100, 3)
    _run_streams(S, (lambda s: mixed[s % 4]) if ids == 0 else (lambda s: ids + s), frames, F=F, wh=wh, opts={"assoc_pack": pack},
                 stream_kw=dict(p_vanish=0.05, vanish_max=6), empty_every=7 if ids == 3 else 0)


@pytest.mark.parametrize("ahead", [1, 0])
@pytest.mark.parametrize("S,F,frames,ids", [(1, 32, 150, 30), (3, 8, 64, 12), (2, 5, 45, 100)])
def test_prediction_one_frame_ahead_or_inside_the_frame_kernel(ahead, S, F, frames, ids):
    """pred_ahead: post_track leaves every live track's state predicted for the next frame (one wave per track, the per-entry operations
    of ss_kf_predict) and k_frame reads only what its gate needs, against k_frame loading, predicting and storing the whole state:
    every intermediate and the track tables (mean / covariance bit for bit) equal the oracle either way; births, deaths, unmatched tracks."""
    wh = (1920, 1080) if ids >= 100 else (1280, 720)
    _run_streams(S, lambda s: ids + 3 * s, frames, F=F, wh=wh, opts={"pred_ahead": ahead}, stream_kw=dict(p_vanish=0.05, vanish_max=6))


@pytest.mark.parametrize("merge", [1, 0])
@pytest.mark.parametrize("S,F,frames,ids", [(1, 32, 150, 30), (3, 8, 64, 12), (2, 5, 45, 100)])
def test_post_and_newrow_as_one_launch_or_two(merge, S, F, frames, ids):
    """k_postnew (the update of a frame and the distances of its new gallery rows in ONE launch, the rows recomputed by the
    new-row units from the double-buffered EMA feature) against the two-launch form: both equal the oracle in every intermediate;
    incl. (100, 100) shapes and groups where tracks are confirmed / die inside the group."""
    wh = (1920, 1080) if ids >= 100 else (1280, 720)
    _run_streams(S, lambda s: ids + 3 * s, frames, F=F, wh=wh, opts={"chain_merge": merge}, stream_kw=dict(p_vanish=0.05, vanish_max=6))


@pytest.mark.parametrize("graph", [1, 0])
def test_chain_graph_replay_equals_oracle(graph):
    """The group's per-frame chain replayed as a captured HIP graph (`track_graph`, fixed caller buffers: captured at the second
    call, replayed from the third) == plain launches == the oracle, every frame; a partial last group (another frame count)
    takes its own graph / plain launches."""
    import torch
    cfg, F, frames = StrongSortConfig(), 8, 93
    eng, orc = engine(cfg, debug=False), OracleStrongSort(cfg, "c")
    eng.set_option("track_graph", graph)
    st = make_stream(91, 1280, 720, 20)
    dev = eng.device
    bd, bf, bn = torch.zeros(F, 1, 128, 6, device=dev), torch.zeros(F, 1, 128, 512, device=dev), torch.zeros(F, 1, dtype=torch.int32, device=dev)
    hw = torch.tensor([[720, 1280]], dtype=torch.int32, device=dev)
    out, nout = torch.zeros(F, 1, 256, 8, device=dev), torch.zeros(F, 1, dtype=torch.int32, device=dev)
    for k0 in range(0, frames, F):
        nf = min(F, frames - k0)
        hd, hf, hn = np.zeros((F, 1, 128, 6), np.float32), np.zeros((F, 1, 128, 512), np.float32), np.zeros((F, 1), np.int32)
        ref = []
        for f in range(nf):
            fr = st.next_frame()
            n = len(fr.dets)
            hd[f, 0, :n], hf[f, 0, :n], hn[f, 0] = fr.dets, fr.feats, n
            ref.append(orc.update(fr.dets, fr.feats, (720, 1280)))
        bd.copy_(torch.from_numpy(hd)); bf.copy_(torch.from_numpy(hf)); bn.copy_(torch.from_numpy(hn))
        eng.update_group(nf, bd, bn, bf, hw, out, nout)
        eng.check_errors()
        ho, hno = out.cpu().numpy(), nout.cpu().numpy()
        for f in range(nf):
            got = ho[f, 0, :hno[f, 0]]
            assert got.shape == ref[f].shape and got.tobytes() == ref[f].tobytes(), f"frame {k0 + f}"
    _compare_table(eng, orc)
    eng.close()


def test_frame_groups_with_births_deaths_and_empty_frames():
    """Tracks die, slots are reused and new tracks get confirmed in the middle of a group; whole frames without
    detections."""
    cfg = StrongSortConfig(max_age=5)
    conf, orcs = _run_streams(2, lambda s: 10 + 2 * s, 120, F=6, cfg=cfg, wh=(640, 480),
                              stream_kw=dict(p_vanish=0.15, vanish_max=12), empty_every=17)
    assert orcs[0].next_id > 11 and orcs[1].next_id > 13                     # re-births happened


@pytest.mark.parametrize("F", [1, 4])
def test_ambiguous_costs_take_the_lsap_path(F):
    """k_frame reads the assignment off the thresholded matrix when the optimum is unique (no row / column with two
    entries under the threshold) and runs the one-wave LSAP otherwise.  Twin identities (co-located boxes, nearly equal
    embeddings) make both stages ambiguous: both paths must occur and both must reproduce the oracle exactly."""
    import torch
    cfg = StrongSortConfig()
    W, H = 1280, 720
    eng, orc = engine(cfg, debug=True), OracleStrongSort(cfg, "c")
    st = make_stream(3, W, H, 12)
    rng = np.random.default_rng(11)
    dev = eng.device
    hw = torch.tensor([[H, W]], dtype=torch.int32, device=dev)
    out, nout = torch.zeros(F, 1, 256, 8, device=dev), torch.zeros(F, 1, dtype=torch.int32, device=dev)
    paths_a, paths_b = set(), set()
    for k0 in range(0, 40, F):
        hd, hf, hn = np.zeros((F, 1, 128, 6), np.float32), np.zeros((F, 1, 128, 512), np.float32), np.zeros((F, 1), np.int32)
        ref, lasts = [], []
        for f in range(F):
            fr = st.next_frame()
            twin = fr.dets.copy()
            twin[:, :4] += np.array([3, 2, 3, 2], np.float32) * (1 if (k0 + f) % 9 else 0)     # some frames: exact ties
            twin[:, 4] -= 0.01
            tf = fr.feats + 0.005 * rng.standard_normal(fr.feats.shape).astype(np.float32)
            tf = (tf / np.linalg.norm(tf, axis=1, keepdims=True)).astype(np.float32)
            dets, feats = np.concatenate([fr.dets, twin]), np.concatenate([fr.feats, tf])
            n = len(dets)
            hd[f, 0, :n], hf[f, 0, :n], hn[f, 0] = dets, feats, n
            ref.append(orc.update(dets, feats, (H, W)))
            lasts.append(orc.last)
        eng.update_group(F, torch.from_numpy(hd).to(dev), torch.from_numpy(hn).to(dev), torch.from_numpy(hf).to(dev), hw, out, nout)
        eng.check_errors()
        ho, hno = out.cpu().numpy(), nout.cpu().numpy()
        for f in range(F):
            _compare_frame(eng, orc, ho[f, 0, :hno[f, 0]], ref[f], k0 + f, 0, f, lasts[f])
            d = eng.debug(0, f)
            paths_a.add(d["path_a"]); paths_b.add(d["path_b"])
    _compare_table(eng, orc)
    assert 2 in paths_a and 2 in paths_b, (paths_a, paths_b)                   # LSAP ran in both stages
    eng.close()
    # ... and the plain streams of the other tests take the shortcut
    eng, st = engine(cfg, debug=True), make_stream(0, W, H, 30)
    for k in range(8):
        fr = st.next_frame()
        eng.update_host(fr.dets, fr.feats, (H, W))
    assert eng.debug(0)["path_a"] == 1
    eng.close()


def test_capacity_error_is_loud():
    from strongsort_yolo_amd.lib import SSError
    eng = engine()
    with pytest.raises(SSError):
        eng.update_host(np.zeros((129, 6), np.float32), np.zeros((129, 512), np.float32), (480, 640))
    eng.close()


def test_golden_tracker_vector_through_the_c_abi():
    """tests/golden/tracker_6ids_24frames.npz (inputs + oracle outputs) replayed on the GPU."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tracker_6ids_24frames.npz"))
    eng = engine(StrongSortConfig())
    H, W = z["hw"]
    for k in range(len(z["counts"])):
        n = int(z["counts"][k])
        got = eng.update_host(z["dets"][k, :n], z["feats"][k, :n], (H, W))
        ref = z["rows"][k, : int(z["nrows"][k])]
        assert got.shape == ref.shape and got.tobytes() == ref.tobytes(), f"frame {k}"
        dbg = eng.debug(0)
        assert bits_equal(dbg["cost_a"], z["cost_a"][k][: dbg["n_conf"], : dbg["n_dets"]])
    t = eng.tracks(0)
    assert np.array_equal(t["track_id"], z["final_ids"]) and t["next_id"] == int(z["next_id"])
    assert bits_equal(t["mean"], z["final_mean"]) and bits_equal(t["cov"], z["final_cov"])
    eng.close()


def test_near_capacity_stream():
    """110 identities at 1920x1080: 110 x 110 = 12,100 cost entries (LDS cap 14,336), 7 detection column tiles."""
    cfg = StrongSortConfig()
    eng, orc = engine(cfg, debug=False), OracleStrongSort(cfg, "c")
    sg, so = make_stream(44, 1920, 1080, 110), make_stream(44, 1920, 1080, 110)
    for k in range(8):
        fg, fo = sg.next_frame(), so.next_frame()
        got, ref = eng.update_host(fg.dets, fg.feats, (1080, 1920)), orc.update(fo.dets, fo.feats, (1080, 1920))
        assert got.shape == ref.shape and got.tobytes() == ref.tobytes(), f"frame {k}"
    assert len(orc.tracks) >= 100
    eng.close()


def test_track_capacity_overflow_keeps_the_tables_consistent():
    """More births than free slots (256 per stream): the detections that do not fit start no track, SS_ERR_CAPACITY is raised at
    the next check and stays until ss_reset; the stream's tables stay consistent (unique ids, counts within capacity) —
    the defined behaviour that include/strongsort_hip.h states."""
    from strongsort_yolo_amd import lib
    cfg = StrongSortConfig()
    eng = engine(cfg, debug=False)
    rng = np.random.default_rng(9)
    feats = [unit(rng, 128) for _ in range(3)]
    for k in range(7):                                   # three sets of 128 detections at disjoint places, three frames each: a set's
        g = k // 3                                       # tracks are confirmed and then coast (max_age 30) while the next set is born
        dets = np.zeros((128, 6), np.float32)
        gx, gy = np.meshgrid(np.arange(16), np.arange(8))
        dets[:, 0] = 10 + gx.ravel() * 110 + 36 * g; dets[:, 1] = 10 + gy.ravel() * 125 + 40 * g
        dets[:, 2] = dets[:, 0] + 30; dets[:, 3] = dets[:, 1] + 60; dets[:, 4] = 0.9
        try:
            eng.update_host(dets, feats[g], (1080, 1920))
            raised = False
        except lib.SSError as e:
            raised = e.code == lib.SS_ERR_CAPACITY
        assert raised == (k == 6), k
    t = eng.tracks(0)
    assert len(t["track_id"]) == 256 and len(set(t["track_id"].tolist())) == 256 and t["next_id"] == 257
    with pytest.raises(lib.SSError):
        eng.check_errors()                                # sticky until reset
    eng.reset(-1)
    eng.check_errors()
    assert len(eng.tracks(0)["track_id"]) == 0
    eng.close()
