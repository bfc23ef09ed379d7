"""The fp32 ReID kernels (csrc/ss_ops32.hip, the accuracy mode behind model.track's ReID forward, yolo_multi_model.py:41) against
the plain torch fp32 / fp64 modules they replace.

Bounds: the kernels sum in fp32 in their own fixed order, so a single operator agrees with an fp64 evaluation of the same
operator to a few fp32 ulps of the output scale (5e-6 x scale asserted), the whole 30-layer network to 1e-4 of the embedding
scale (measured 2.3e-5; the CPU side is fp32 too); on the 150-frame stream the appearance distances stay within north_star's 1e-4 and the ids are identical."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _close(got, ref64, rel=5e-6):
    ref = ref64.to(torch.float64).cpu()
    got = got.to(torch.float64).cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, (err, scale, err / scale)


@pytest.mark.parametrize("K,N", [(16, 16), (16, 64), (64, 16), (64, 24), (64, 64), (64, 96), (96, 24), (96, 32), (96, 96), (96, 128),
                                 (128, 32), (128, 128)])
def test_pointwise_matches_fp64(K, N):
    from strongsort_yolo_amd import fused32, nets
    g = torch.Generator().manual_seed(K * 1000 + N)
    for (n, h, w), relu, use_res in (((3, 5, 7), True, False), ((2, 16, 8), False, True), ((1, 64, 32), True, True)):
        m = nets.ConvBR(K, N, 1, relu=relu)
        with torch.no_grad():
            m.conv.weight.copy_(torch.randn(m.conv.weight.shape, generator=g))
            m.conv.bias.copy_(torch.randn(N, generator=g))
        x = torch.randn(n, K, h, w, generator=g)
        res = torch.randn(n, N, h, w, generator=g) if use_res else None
        ref = F.conv2d(x.double(), m.conv.weight.double(), m.conv.bias.double())
        if use_res:
            ref = ref + res.double()
        if relu:
            ref = F.relu(ref)
        m = m.to(DEV)
        got = fused32.pointwise(_cl(x.to(DEV)), m, m.conv, relu=relu, res=_cl(res.to(DEV)) if use_res else None)
        assert got.is_contiguous(memory_format=torch.channels_last)
        _close(got, ref)


def test_stem_matches_fp64():
    from strongsort_yolo_amd import fused32, nets
    g = torch.Generator().manual_seed(7)
    m = nets.ConvBR(3, 16, 7, 2, 3)
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn(m.conv.weight.shape, generator=g) * 0.2)
        m.conv.bias.copy_(torch.randn(16, generator=g))
    x = torch.randn(5, 3, 256, 128, generator=g)
    ref = F.max_pool2d(F.relu(F.conv2d(x.double(), m.conv.weight.double(), m.conv.bias.double(), 2, 3)), 3, 2, 1)
    m = m.to(DEV)
    got = fused32.stem(_cl(x.to(DEV)), m)
    assert got.shape == (5, 16, 64, 32)
    _close(got, ref)


@pytest.mark.parametrize("walk", [1, 2, 4, 8, 16])
def test_stem_walking_form_equals_band_form(walk):
    """k32_stemW (a workgroup walks `walk` bands down an image: input-row ring, shared convolution row kept, next rows requested ahead) keeps
    k32_stem's summation order: bit-equal outputs, including a valid-image count below the batch."""
    from strongsort_yolo_amd import fused32, nets
    g = torch.Generator().manual_seed(11)
    m = nets.ConvBR(3, 16, 7, 2, 3)
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn(m.conv.weight.shape, generator=g) * 0.2)
        m.conv.bias.copy_(torch.randn(16, generator=g))
    x = _cl((torch.randn(7, 3, 256, 128, generator=g) * 3).to(DEV))
    m = m.to(DEV)
    try:
        fused32.set_option("stem_walk", -1)
        ref = fused32.stem(x, m).clone()
        fused32.set_option("stem_walk", walk)
        got = fused32.stem(x, m)
        assert got.cpu().numpy().tobytes() == ref.cpu().numpy().tobytes()
        ref64 = F.max_pool2d(F.relu(F.conv2d(x.cpu().double(), m.conv.weight.cpu().double(), m.conv.bias.cpu().double(), 2, 3)), 3, 2, 1)
        _close(got, ref64)
        nv = torch.tensor([4], dtype=torch.int32, device=DEV)
        with fused32.valid_images(nv):
            part = fused32.stem(x, m)
        assert part[:4].cpu().numpy().tobytes() == ref[:4].cpu().numpy().tobytes()
    finally:
        fused32.set_option("stem_walk", 0)


@pytest.mark.parametrize("n", [3, 140, 600])
def test_stem_on_byte_crops_equals_stem_on_float_crops(n):
    """k32_stemW<true> (byte crops, /255 + mean + std applied while the rows are staged) == the float form on fused32.crops_from_u8 of the same
    bytes, bit for bit, at 2 / 4 / 8 bands per workgroup and with a valid-image count below the batch."""
    from strongsort_yolo_amd import fused32, nets
    g = torch.Generator().manual_seed(n)
    m = nets.ConvBR(3, 16, 7, 2, 3)
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn(m.conv.weight.shape, generator=g) * 0.2)
        m.conv.bias.copy_(torch.randn(16, generator=g))
    m = m.to(DEV)
    xb = torch.randint(0, 256, (n, 3, 256, 128), dtype=torch.uint8, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    ref = fused32.stem(fused32.crops_from_u8(xb), m)
    got = fused32.stem(xb, m)
    assert got.cpu().numpy().tobytes() == ref.cpu().numpy().tobytes()
    nv = torch.tensor([max(1, n // 3)], dtype=torch.int32, device=DEV)
    with fused32.valid_images(nv):
        part = fused32.stem(xb, m)
    k = int(nv.item())
    assert part[:k].cpu().numpy().tobytes() == ref[:k].cpu().numpy().tobytes()


@pytest.mark.parametrize("u8", [False, True])
@pytest.mark.parametrize("n", [5, 200])
def test_stem_with_the_first_blocks_conv1_in_the_same_launch(n, u8):
    """ss_op32_stem_conv1: (x0, relu(conv1(x0))) from one launch == the stem launch + k32_pw on its output, bit for bit (byte and float crops)."""
    from strongsort_yolo_amd import fused32, nets
    g = torch.Generator().manual_seed(100 + n)
    m, c1 = nets.ConvBR(3, 16, 7, 2, 3), nets.ConvBR(16, 16, 1)
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn(m.conv.weight.shape, generator=g) * 0.2); m.conv.bias.copy_(torch.randn(16, generator=g))
        c1.conv.weight.copy_(torch.randn(c1.conv.weight.shape, generator=g) * 0.3); c1.conv.bias.copy_(torch.randn(16, generator=g) * 0.5)
    m, c1 = m.to(DEV), c1.to(DEV)
    xb = torch.randint(0, 256, (n, 3, 256, 128), dtype=torch.uint8, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    x = xb if u8 else fused32.crops_from_u8(xb)
    try:
        fused32.STEM_CONV1 = False
        y_ref, y1_ref = fused32.stem(x, m, c1)
        fused32.STEM_CONV1 = True
        y, y1 = fused32.stem(x, m, c1)
    finally:
        fused32.STEM_CONV1 = True
    assert y.cpu().numpy().tobytes() == y_ref.cpu().numpy().tobytes()
    assert y1.cpu().numpy().tobytes() == y1_ref.cpu().numpy().tobytes()
    assert float(y1.abs().max()) > 0


def _block(c1, c2, seed):
    from strongsort_yolo_amd import nets
    g = torch.Generator().manual_seed(seed)
    blk = nets.OSBlock(c1, c2)
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.4 if p.dim() > 1 else 0.2))
    return blk, g


@pytest.mark.parametrize("form", [2, 1, 0])
@pytest.mark.parametrize("c1,c2,n,h,w", [(16, 64, 3, 64, 32), (64, 96, 3, 32, 16), (96, 128, 5, 16, 8), (16, 64, 4, 64, 32), (64, 96, 1, 32, 16)])
def test_chains_match_fp64(c1, c2, n, h, w, form):
    """Every kernel form: 2 = k32_chainsR (register-resident row stream, one chain per wave, two images per workgroup: odd and even
    image counts; 64 x 32 and 32 x 16 maps — the default), 1 = k32_chains3 (LDS phases, conflict-free lane map), 0 = k32_chains
    (always for 16 x 8)."""
    from strongsort_yolo_amd import fused32
    fused32.set_option("chains_form", form)
    fused32.set_option("chains_min_n", 1)              # (batches below 128 images would otherwise take the band forms: the per-frame call's batches)
    try:
        _chains_case(c1, c2, n, h, w)
    finally:
        fused32.set_option("chains_form", 2)
        fused32.set_option("chains_min_n", 128)


def _chains_case(c1, c2, n, h, w):
    from strongsort_yolo_amd import fused32
    blk, g = _block(c1, c2, c1 + h)
    mid = c2 // 4
    x1 = torch.randn(n, mid, h, w, generator=g).abs()                      # conv1's output is a ReLU output
    blk64 = blk.double()
    with torch.no_grad():
        refs = [s(x1.double()) for s in blk64.streams]
    blk = blk.float().to(DEV)
    ys, psum = fused32.chains(_cl(x1.to(DEV)), blk)
    torch.cuda.synchronize()
    for y, r in zip(ys, refs):
        _close(y, r)
    sums = torch.stack([r.sum((2, 3)) for r in refs])                     # [4, n, mid]
    _close(psum.sum(2), sums, rel=2e-5)


@pytest.mark.parametrize("min_n", [1, 128])
@pytest.mark.parametrize("k", [1, 2, 4, 5, 7, 8])
def test_block_parts_match_fp64(k, min_n):
    """chains + tail (gates, conv3, shortcut, the following ConvBR [+ average pool]) of every OSBlock position against the modules in fp64,
    with the row-stream chains (what large batches take: min_n 1 forces them for this 3-image batch) and with the band forms small batches take."""
    from strongsort_yolo_amd import fused32
    fused32.set_option("chains_min_n", min_n)
    try:
        _block_parts_case(k)
    finally:
        fused32.set_option("chains_min_n", 128)


def _block_parts_case(k):
    from strongsort_yolo_amd import nets
    g = torch.Generator().manual_seed(100 + k)
    net = nets.build_reid(1)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.2))
    blk, nxt, pool = net._blocks(k)
    c1 = blk.conv1.conv.in_channels
    h, w = {1: (64, 32), 2: (64, 32), 4: (32, 16), 5: (32, 16), 7: (16, 8), 8: (16, 8)}[k]
    x = torch.randn(3, c1, h, w, generator=g).abs()
    net64 = nets.build_reid(1).double()
    net64.load_state_dict({kk: v.double() for kk, v in net.state_dict().items()})
    b64, n64, _ = net64._blocks(k)
    with torch.no_grad():
        o = b64(x.double())
        o2 = n64(o)
        if pool:
            o2 = F.avg_pool2d(o2, 2, 2)
    net = net.to(DEV)
    with torch.no_grad():
        st = net._part32(k, _cl(x.to(DEV)))
    if k in (1, 4, 7):
        _close(st[0], o, rel=2e-5)
        _close(st[1], o2, rel=2e-5)
    else:
        assert len(st) == 1
        _close(st[0], o2, rel=2e-5)


def test_head_matches_fp64():
    from strongsort_yolo_amd import fused32
    g = torch.Generator().manual_seed(3)
    fc = torch.nn.Linear(128, 512)
    x = torch.randn(7, 128, 16, 8, generator=g)
    ref = F.relu(F.linear(x.double().mean((2, 3)), fc.weight.double(), fc.bias.double()))
    got = fused32.head(_cl(x.to(DEV)), fc.to(DEV))
    _close(got, ref)


def _calibrated(n_crops=48, seed=11):
    import bench
    from strongsort_yolo_amd import nets
    g = torch.Generator().manual_seed(seed)
    crops = torch.randn(n_crops, 3, 256, 128, generator=g)
    return bench.calibrate_reid_(nets.build_reid(1).float(), crops), crops


def test_whole_network_matches_the_cpu_fp32_network():
    """OSNet-x0.25 on the fp32 kernels vs the same module on the CPU (the oracle's network): embeddings within 1e-4 of their scale,
    cosine distances between crops within 2e-5; every cut of the 10-part split gives the unsplit result bit for bit; the launches
    really are the library's (fused32.ENABLED = False goes to the torch modules and differs in the last bits)."""
    from strongsort_yolo_amd import fused32
    net, crops = _calibrated()
    with torch.no_grad():
        ref = net(crops)
        gnet = __import__("copy").deepcopy(net).to(DEV).to(memory_format=torch.channels_last)
        xg = _cl(crops.to(DEV))
        got = gnet(xg)
        for cut in range(0, 11):
            again = gnet.forward_b(gnet.forward_a(xg, cut), cut)
            assert torch.equal(again, got), cut
        fused32.ENABLED = False
        lib = gnet(xg)
        fused32.ENABLED = True
    _close(got, ref, rel=1e-4)                 # two fp32 evaluations of 30 layers in different summation orders (measured 2.3e-5)
    u = lambda e: e.double().cpu() / e.double().cpu().norm(dim=1, keepdim=True)
    dg, dr = 1.0 - u(got) @ u(got).T, 1.0 - u(ref) @ u(ref).T
    assert (dg - dr).abs().max().item() <= 2e-5
    assert dr[~torch.eye(len(dr), dtype=bool)].min().item() > 1e-3          # a network that tells crops apart
    _close(lib, ref, rel=1e-4)


def test_valid_image_count_skips_the_rest_of_the_batch():
    from strongsort_yolo_amd import fused32
    net, crops = _calibrated(16, seed=12)
    gnet = net.to(DEV).to(memory_format=torch.channels_last)
    xg = _cl(crops.to(DEV))
    nv = torch.tensor([5], dtype=torch.int32, device=DEV)
    with torch.no_grad():
        full = gnet(xg)
        with fused32.valid_images(nv):
            part = gnet(xg)
    assert torch.equal(part[:5], full[:5])
    assert part[5:].abs().sum().item() == 0.0


def test_graph_replay_of_the_fp32_network_equals_eager():
    net, crops = _calibrated(32, seed=13)
    gnet = net.to(DEV).to(memory_format=torch.channels_last)
    xg = _cl(crops.to(DEV))
    st = torch.cuda.Stream(DEV)
    with torch.no_grad(), torch.cuda.stream(st):
        eager = gnet(xg).clone()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=st):
            out = gnet(xg)
        out.zero_()
        gph.replay()
        st.synchronize()
    assert torch.equal(out, eager)


def test_accuracy_mode_meets_the_float_bound_on_the_true_reid_path():
    """north_star: float distances within 1e-4 of the CPU reference, ids identical.  Rendered frames -> HIP crops (fp32) -> the
    fp32 HIP OSNet -> HIP tracker, beside C-oracle crops -> the same network in CPU fp32 -> C-oracle tracker (bench.reid_f16_vs_f32,
    the measurement the bench line carries), 150 frames."""
    import bench
    from strongsort_yolo_amd.config import StrongSortConfig, DetectConfig
    r = bench.reid_f16_vs_f32("yolov8n", 1280, 720, 30, StrongSortConfig(), DetectConfig(), device=0, frames=150, reid_half=False)
    assert r["reid_kernels"] == "hip-fp32", r
    assert r["cost_matrix_cosine_max_abs_err"] <= 1e-4 and r["within_bound"], r
    assert r["embedding_unit_max_abs_err"] <= 1e-4, r
    assert r["id_match_rate"] == 1.0 and r["first_divergent_frame"] is None, r
    assert r["cost_matrix_frames_compared"] >= 140 and r["rows_compared"] > 3000, r


def test_f16_mode_error_is_bounded_and_recorded():
    """The throughput default (f16 activations) does NOT meet 1e-4 on this network; its error is pinned here so that a change of the
    f16 kernels cannot make it silently worse: distances within 6e-2, the same tracks up to one renumbering on >= 99 % of the rows."""
    import bench
    from strongsort_yolo_amd.config import StrongSortConfig, DetectConfig
    r = bench.reid_f16_vs_f32("yolov8n", 1280, 720, 30, StrongSortConfig(), DetectConfig(), device=0, frames=150, reid_half=True)
    assert r["cost_matrix_cosine_max_abs_err"] <= 6e-2 and r["embedding_unit_max_abs_err"] <= 6e-2, r
    assert r["id_match_rate_up_to_relabeling"] >= 0.99, r
