"""CPU restatements of the index arithmetic the network-side HIP kernels rely on (csrc/ss_ops.hip), checked against
torch's own operators in fp32.  They pin the host-side weight layouts (`fused.weight_nk`, `weight_n9k`, `stem_weight`,
the stacked LightConv weights of an OSNet block) and the kernels' addressing schemes independent of a GPU; the GPU
tests (`tests/test_gpu_nets.py`) then check the kernels themselves."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from strongsort_yolo_amd import fused, nets


def test_weight_n9k_is_the_implicit_gemm_operand():
    """k_pw<…,CONV3>: out[p][n] = sum_k W[n][k] * X[p][k] with k = tap*Cin + c and tap = 3*dy + dx, input pixel
    (stride*oy - 1 + dy, stride*ox - 1 + dx), zeros outside."""
    torch.manual_seed(0)
    for cin, cout, stride, hw in [(8, 16, 1, (5, 7)), (16, 8, 2, (6, 9)), (24, 40, 2, (7, 7))]:
        conv = torch.nn.Conv2d(cin, cout, 3, stride, 1)
        assert fused.conv3x3_ok(conv)
        w = fused.weight_n9k(torch.nn.Module(), conv)
        assert w.shape == (cout, 9 * cin)
        x = torch.randn(2, cin, *hw)
        H, W = hw
        OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
        xp = F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1)                       # [B, H+2, W+2, Cin], padded index = index + 1
        cols = torch.empty(2, OH, OW, 9 * cin)
        for tap in range(9):
            dy, dx = divmod(tap, 3)
            cols[..., tap * cin:(tap + 1) * cin] = xp[:, dy:dy + stride * (OH - 1) + 1:stride, dx:dx + stride * (OW - 1) + 1:stride]
        got = (cols.reshape(-1, 9 * cin) @ w.t()).reshape(2, OH, OW, cout).permute(0, 3, 1, 2)
        ref = F.conv2d(x, conv.weight, None, stride, 1)
        assert got.shape == ref.shape and torch.allclose(got, ref, atol=1e-5)


def test_weight_nk_and_the_permuted_k_assignment():
    """k_pw feeds the MFMA with lane group q holding k = 8q..8q+7 of a 32-wide chunk (elements 0-3 to the first
    16x16x16 instruction, 4-7 to the second): any k <-> slot assignment used for BOTH operands leaves the dot
    product unchanged.  Restated with an explicit permutation."""
    torch.manual_seed(1)
    conv = torch.nn.Conv2d(32, 16, 1)
    assert fused.pointwise_ok(conv)
    w = fused.weight_nk(torch.nn.Module(), conv)                             # [N, K]
    x = torch.randn(10, 32)
    slot = np.array([[8 * q + j for q in range(4) for j in range(4)], [8 * q + 4 + j for q in range(4) for j in range(4)]])
    assert sorted(slot.reshape(-1).tolist()) == list(range(32))              # the two instructions cover every k once
    acc = sum(x[:, slot[i]] @ w[:, slot[i]].t() for i in range(2))
    assert torch.allclose(acc, F.conv2d(x.view(10, 32, 1, 1), conv.weight).view(10, 16), atol=1e-5)


def test_stem_weight_and_flat_row_windows():
    """k_osnet_stem keeps each NHWC input row as a flat array of halfs with 16 halfs of zero margin in front; the 21 taps
    (kx, ch) of conv column c on row ky are then the window starting at half 6c + 7.  Wave r handles the columns c = 4n + r:
    their windows start (6r + 7) % 8 halfs into the ALIGNED 8-half block 3n + (6r + 7) // 8, and the MFMA's 32-wide K axis is
    that block and the next three, against weights [r][ky][oc][32] shifted by the same amount (zeros elsewhere)."""
    torch.manual_seed(2)
    conv = torch.nn.Conv2d(3, 16, 7, 2, 3)
    wp = fused.stem_weight(torch.nn.Module(), conv)
    assert wp.shape == (4, 7, 16, 32)
    for r in range(4):
        sh = (6 * r + 7) % 8
        assert sh + 21 <= 32 and (wp[r, :, :, :sh] == 0).all() and (wp[r, :, :, sh + 21:] == 0).all()
    H = 16
    x = torch.randn(1, 3, H, 128)
    assert fused.stem_ok(x, conv)
    rows = torch.zeros(H + 6, 416)                                           # 3 zero rows above and below, margins zero
    rows[3:H + 3, 16:16 + 384] = x[0].permute(1, 2, 0).reshape(H, 384)
    out = torch.empty(16, H // 2, 64)
    for rr in range(H // 2):
        for c in range(64):
            r, n = c % 4, c // 4
            blk = 3 * n + (6 * r + 7) // 8
            assert 8 * blk + (6 * r + 7) % 8 == 6 * c + 7 and 8 * blk + 32 <= 416
            win = torch.stack([rows[2 * rr + ky, 8 * blk:8 * blk + 32] for ky in range(7)])      # [7, 32]
            out[:, rr, c] = (wp[r] * win[:, None, :]).sum((0, 2))
    ref = F.conv2d(x, conv.weight, None, 2, 3)[0]
    assert torch.allclose(out, ref, atol=1e-4)
    # ReLU outputs are >= 0, so a 0 in a padded pooling position never changes the maximum (the kernel's pad value)
    a = F.relu(ref + conv.bias.view(16, 1, 1))
    pooled_zero_pad = F.max_pool2d(F.pad(a, (1, 1, 1, 1), value=0.0), 3, 2, 0)
    assert torch.equal(pooled_zero_pad, F.max_pool2d(a, 3, 2, 1))


def test_conv0_weight_and_flat_row_windows():
    """k_conv0 (3x3 / stride 2 / pad 1 on 3 channels): with 8 halfs of margin in front of a 64-column tile's span of the flat
    NHWC row, the 9 taps of conv column c = 4n + r start (6r + 5) % 8 halfs into the aligned block 3n + (6r + 5) // 8 and stay
    inside a 16-wide window; weights [r][ky][oc][16] are shifted by the same amount."""
    torch.manual_seed(4)
    conv = torch.nn.Conv2d(3, 16, 3, 2, 1)
    wp = fused.conv0_weight(torch.nn.Module(), conv)
    assert wp.shape == (4, 3, 16, 16)
    H, W = 6, 256
    x = torch.randn(1, 3, H, W)
    rows = torch.zeros(H + 2, 8 + W * 3 + 8)                                 # one zero row above / below, margins zero
    rows[1:H + 1, 8:8 + W * 3] = x[0].permute(1, 2, 0).reshape(H, W * 3)
    out = torch.empty(16, H // 2, W // 2)
    for oy in range(H // 2):
        for c in range(W // 2):
            c0, cl = (c // 64) * 64, c % 64
            r, n = cl % 4, cl // 4
            blk, sh = 3 * n + (6 * r + 5) // 8, (6 * r + 5) % 8
            assert 8 * blk + sh == 6 * cl + 5 and sh + 9 <= 16 and 8 * blk + 16 <= 400
            base = 6 * c0                                                    # LDS half 0 of the tile = flat 6 c0 - 8 = rows[.., 6 c0]
            win = torch.stack([rows[2 * oy + ky, base + 8 * blk:base + 8 * blk + 16] for ky in range(3)])      # [3, 16]
            out[:, oy, c] = (wp[r] * win[:, None, :]).sum((0, 2))
    assert torch.allclose(out, F.conv2d(x, conv.weight, None, 2, 1)[0], atol=1e-4)


def test_osnet_block_layer_order_matches_the_chain_kernel():
    """k_osnet_streams takes the ten LightConv layers as [chain1: 1 layer][chain2: 2][chain3: 3][chain4: 4]; chain t
    starts at index t(t-1)/2.  The module order `for st in streams for m in st` must be that order."""
    blk = nets.OSBlock(16, 64)
    layers = [m for st in blk.streams for m in st]
    assert len(layers) == 10 and [len(st) for st in blk.streams] == [1, 2, 3, 4]
    for t in range(1, 5):
        base = t * (t - 1) // 2
        assert all(layers[base + l] is blk.streams[t - 1][l] for l in range(t))


def test_chain_halo_rows_shrink_one_per_layer():
    """Row bookkeeping of k_osnet_streams: band of TH rows, chain depth t, local row i <-> image row y0 - t + i; layer
    l computes the pointwise product on local rows [l-1, R-l] and the depthwise output on [l, R-l-1] (R = TH + 2t), so
    the last layer yields exactly the band.  Checked by propagating 'valid row' sets through real 3x3 dependencies."""
    TH = 16
    for t in range(1, 5):
        R = TH + 2 * t
        have = set(range(R))                                                 # input rows present in LDS / global
        for l in range(1, t + 1):
            pw = set(range(l - 1, R - l + 1))
            assert pw <= have                                                # pointwise needs only rows we hold
            dw = set(range(l, R - l))
            assert all({r - 1, r, r + 1} <= pw for r in dw)                  # 3x3 reads inside the pointwise rows
            have = dw
        assert have == set(range(t, TH + t))                                 # = the band


def test_avgpool_and_place_guards():
    conv = torch.nn.Conv2d(12, 16, 1)                                        # Cin % 8 != 0 -> library path
    assert not fused.pointwise_ok(conv)
    assert not fused.conv3x3_ok(torch.nn.Conv2d(16, 16, 3, 1, 0))            # pad 0
    assert not fused.conv3x3_ok(torch.nn.Conv2d(16, 16, 3, 1, 1, groups=16))
    assert fused.place_ok(16, 48) and not fused.place_ok(12, 36)
    x = torch.zeros(1, 16, 64, 32)
    assert fused.lightconv_ok(x) and fused.streams_ok(x)
    assert not fused.streams_ok(torch.zeros(1, 16, 64, 100))                 # W % 8 != 0
    assert not fused.streams_ok(torch.zeros(1, 32, 64, 64))                  # LDS budget


@pytest.mark.parametrize("name,last", [("yolov8n", 22), ("yolov5n", 24), ("yolo11n-pose", 23), ("yolo11n", 23), ("yolov8n-seg", 22), ("yolo11n-seg", 23)])
def test_ultralytics_state_dict_is_folded_and_mapped(name, last, tmp_path):
    """An Ultralytics-style state_dict (layer indices, Conv + BatchNorm pairs, DFL projection, num_batches_tracked) built from
    one of the networks here loads — strictly — into a fresh one and reproduces its forward pass: the layer-index map and the
    BatchNorm folding of nets.convert_ultralytics_state_dict (what `YOLO("yolo11n-pose.pt")` of
    /root/reference/yolo_multi_model.py:17 needs once a real exported file is there)."""
    from strongsort_yolo_amd import nets
    src = nets.build_detector(name, 3).float()
    g = torch.Generator().manual_seed(11)
    fake, convs = {}, {n for n, m in src.named_modules() if isinstance(m, nets.Conv)}
    eps = 1e-3

    def uname(key):                                               # b3.cv1.conv.weight -> model.3.cv1.conv.weight
        head, rest = key.split(".", 1)
        return f"model.{last if head == 'detect' else int(head[1:])}.{rest}"

    sd = src.state_dict()
    for key, v in sd.items():
        mod = key.rsplit(".", 2)[0]
        if mod in convs and key.endswith(".conv.weight"):
            gamma, var = torch.rand(v.shape[0], generator=g) + 0.5, torch.rand(v.shape[0], generator=g) + 0.5
            mean, beta = torch.randn(v.shape[0], generator=g), sd[mod + ".conv.bias"]
            scale = gamma / torch.sqrt(var + eps)
            fake[uname(key)] = v / scale.view(-1, 1, 1, 1)                        # folds back to v
            pre = uname(mod + ".bn.")
            fake[pre + "weight"], fake[pre + "running_var"], fake[pre + "running_mean"] = gamma, var, mean
            fake[pre + "bias"] = beta + mean * scale                             # folds back to beta
            fake[pre + "num_batches_tracked"] = torch.tensor(7)
        elif mod in convs and key.endswith(".conv.bias"):
            continue
        else:
            fake[uname(key)] = v
    fake[f"model.{last}.dfl.conv.weight"] = torch.arange(16.0).view(1, 16, 1, 1)
    path = tmp_path / (name + ".pt")
    torch.save(fake, path)
    dst = nets.build_detector(name, 99).float()
    assert nets.load_weights(dst, str(path), name)
    for k, v in sd.items():
        assert torch.allclose(dst.state_dict()[k], v, rtol=1e-4, atol=1e-5), k
    x = torch.randn(1, 3, 64, 96, generator=g)
    with torch.no_grad():
        a, b = dst(x), src(x)
        if name.endswith("-seg"):                                 # (rows with 32 mask coefficients, prototypes at stride 4)
            assert a[0].shape == (1, 4 + 80 + 32, 126) and a[1].shape == (1, 32, 16, 24)
            assert any(k.startswith("detect.proto.upsample") for k in sd) and f"model.{last}.proto.cv3.bn.weight" in fake
            a, b = torch.cat([t.flatten() for t in a]), torch.cat([t.flatten() for t in b])
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-3)
    # a layer that does not line up is an error, not a silently wrong network
    bad = dict(fake); bad["model.40.conv.weight"] = torch.zeros(1)
    torch.save(bad, path)
    with pytest.raises(ValueError):
        nets.load_weights(nets.build_detector(name, 1).float(), str(path), name)


def test_bottleneck_launch_eligibility():
    """Host-side rule for the one-launch C2f bottleneck (csrc k_bneck): two 3x3 / stride 1 SiLU convolutions of one width in
    {16, 32, 64, 128}; everything else (e = 0.5 bottlenecks of C3k2, 1x1 + 3x3 pairs of C3, other widths) takes the per-layer path."""
    assert fused.bottleneck_ok(nets.Bottleneck(32, 32, True, e=1.0)) and fused.bottleneck_ok(nets.Bottleneck(128, 128, False, e=1.0))
    assert not fused.bottleneck_ok(nets.Bottleneck(32, 32, True, e=0.5))            # hidden width 16 != 32
    assert not fused.bottleneck_ok(nets.Bottleneck(48, 48, True, e=1.0))            # width without a kernel instance
    assert not fused.bottleneck_ok(nets.Bottleneck(32, 32, True, k=(1, 3), e=1.0))  # C3's 1x1 -> 3x3 pair
    m = nets.Bottleneck(64, 64, True, e=1.0)
    m.cv2.act = torch.nn.Identity()
    assert not fused.bottleneck_ok(m)                                               # the kernel's activation is SiLU on both layers
    c2f = nets.C2f(64, 64, n=2, shortcut=True)
    assert all(fused.bottleneck_ok(b) and b.add for b in c2f.m) and c2f.c == 32


def test_bench_stage_cut_defaults_cover_every_preset():
    import bench
    assert set(bench.REID_SPLIT) == set(bench.PRESETS) and all(0 <= v <= nets.OSNet.N_PARTS for v in bench.REID_SPLIT.values())
    assert set(bench.REID_SPLIT_FP32) == set(bench.PRESETS) and all(0 <= v <= nets.OSNet.N_PARTS for v in bench.REID_SPLIT_FP32.values())


@pytest.mark.parametrize("name,published", [
    ("yolov8n", 3157200), ("yolov8s", 11166560), ("yolov8m", 25902640), ("yolov8n-pose", 3295470), ("yolov8n-seg", 3409968),
    ("yolo11n", 2624080), ("yolo11s", 9458752), ("yolov5n", 2654816), ("yolov5s", 9153152)])
def test_parameter_counts_equal_the_published_model_summaries(name, published):
    """Third-party pin on the network GRAPHS ('not even the published parameter counts as a test'): the Ultralytics model
    summaries print these totals for the unfused models (yolov8n: '225 layers, 3157200 parameters'; yolo11n: '319 layers, 2,624,080
    parameters'; the 'u' heads for yolov5).  The modules here are built in fused-inference form — a biased convolution per Conv +
    BatchNorm pair — so the published count is ours + one more vector per such pair (BatchNorm has weight AND bias where the folded
    convolution keeps one bias) + the 16 fixed weights of the DFL projection."""
    m = nets.build_detector(name)
    ours = sum(p.numel() for p in m.parameters())
    bn_pairs = sum(mod.conv.out_channels for mod in m.modules() if isinstance(mod, nets.Conv))
    assert ours + bn_pairs + 16 == published


def test_fused_parameter_count_of_the_reference_default_model():
    """`yolo11n-pose.pt` (/root/reference/yolo_multi_model.py:17): Ultralytics prints 2,866,468 parameters for the FUSED model (BatchNorm
    folded, which is the form built here) — ours + the 16 DFL weights.  OSNet-x0.25 is published as a 0.2 M-parameter network."""
    m = nets.build_detector("yolo11n-pose")
    assert sum(p.numel() for p in m.parameters()) + 16 == 2866468
    r = sum(p.numel() for p in nets.build_reid().parameters())
    assert 190_000 < r < 215_000


def test_padded_head_branch_computes_the_same_channels():
    """fused.padded_branch / padded_last: the pose head's 51-channel branch with its widths zero-padded to 64 / 56 and a one-class
    head's last 1x1 padded to 8 rows (what the convolution kernels take) compute the original channels exactly — SiLU(0) = 0 keeps
    the padding at zero through both 3x3 layers — and zeros in the padded ones (fp32 torch convolutions on the prepared weights)."""
    torch.manual_seed(5)
    det = nets.Detect(1, (64, 128, 256), nk=51).float()
    seq = det.cv4[0]
    (w0, b0), (w1, b1), (w2, b2) = fused.padded_branch(torch.nn.Module(), seq)
    assert w0.shape == (64, 9 * 64) and w1.shape == (64, 9 * 64) and w2.shape == (56, 64) and b2.shape == (56,)
    x = torch.randn(2, 64, 9, 11)
    as_conv = lambda w, k: w.view(w.shape[0], k, k, -1).permute(0, 3, 1, 2).contiguous()         # [N, kh*kw*Cin] tap-major -> [N, Cin, kh, kw]
    t = F.silu(F.conv2d(x, as_conv(w0, 3), b0, padding=1))
    assert (t[:, 51:] == 0).all()
    t = F.silu(F.conv2d(t, as_conv(w1, 3), b1, padding=1))
    assert (t[:, 51:] == 0).all()
    got = F.conv2d(t, w2.view(56, 64, 1, 1), b2)
    ref = seq(x)
    assert (got[:, 51:] == 0).all() and torch.allclose(got[:, :51], ref, rtol=0, atol=1e-5)
    last = det.cv3[0][2]                                           # Conv2d(c3, 1, 1): one class
    w, b = fused.padded_last(torch.nn.Module(), last)
    assert w.shape == (8, last.in_channels) and (w[1:] == 0).all() and (b[1:] == 0).all()
    assert torch.equal(w[:1], last.weight.detach().reshape(1, -1)) and torch.equal(b[:1], last.bias.detach())


def test_merged_pointwise_pairs_keep_the_concat_order():
    """C3k._w12 / nets._w_pair: cv1 and cv2 of a C3k / ELAN block read the same input and their outputs are concatenated [cv1 | cv2] —
    one 1x1 with both sets of output rows in that order."""
    torch.manual_seed(6)
    blk = nets.C3k(32, 32, 2).float()
    w, b = blk._w12()
    x = torch.randn(1, 32, 5, 7)
    got = F.silu(F.conv2d(x, w.view(w.shape[0], -1, 1, 1), b))
    assert torch.allclose(got, torch.cat((blk.cv1(x), blk.cv2(x)), 1), rtol=0, atol=1e-6)
    el = nets.ELAN(32, 16, 64).float()
    w, b = nets._w_pair(el, el.cv1.conv, el.cv2.conv)
    got = F.silu(F.conv2d(x, w.view(w.shape[0], -1, 1, 1), b))
    assert torch.allclose(got, torch.cat((el.cv1(x), el.cv2(x)), 1), rtol=0, atol=1e-6)


def test_loading_weights_drops_the_prepared_copies(tmp_path):
    """nets.load_weights after a forward pass: the kernel-side weight copies cached on the modules (fused.weight_nk, padded_branch, ...)
    are dropped, so the next forward prepares them from the loaded tensors."""
    torch.manual_seed(7)
    det = nets.Detect(1, (64, 128, 256), nk=51)
    c = det.cv2[0][2]
    before = fused.weight_nk(c, c).clone()
    fused.padded_branch(det.cv4[0], det.cv4[0])
    assert "_w_nk" in c.__dict__ and "_padded" in det.cv4[0].__dict__
    other = nets.Detect(1, (64, 128, 256), nk=51)
    f = tmp_path / "w.pt"
    torch.save(other.state_dict(), f)
    assert nets.load_weights(det, str(f), "test head")
    assert "_w_nk" not in c.__dict__ and "_padded" not in det.cv4[0].__dict__
    after = fused.weight_nk(c, c)
    assert not torch.equal(before, after) and torch.equal(after, other.cv2[0][2].weight.detach().reshape(after.shape))


def test_padded_bottleneck_weights_compute_the_same_block():
    """fused.bottleneck_padded's prepared weights (a c -> c/2 -> c Bottleneck as c -> c -> c with zero rows / columns) through plain torch
    convolutions: the block's output, exactly the same sums plus zeros."""
    torch.manual_seed(8)
    m = nets.Bottleneck(32, 32, True, e=0.5).float()
    assert fused.bottleneck_padded_ok(m) and not fused.bottleneck_ok(m)
    c, hid = 32, 16
    a, b = m.cv1.conv, m.cv2.conv
    w1 = torch.zeros(c, 3, 3, c); w1[:hid] = a.weight.detach().permute(0, 2, 3, 1)
    b1 = torch.zeros(c); b1[:hid] = a.bias.detach()
    w2 = torch.zeros(c, 3, 3, c); w2[:, :, :, :hid] = b.weight.detach().permute(0, 2, 3, 1)
    x = torch.randn(2, 32, 7, 9)
    t = F.silu(F.conv2d(x, w1.permute(0, 3, 1, 2), b1, padding=1))
    assert (t[:, hid:] == 0).all()
    got = x + F.silu(F.conv2d(t, w2.permute(0, 3, 1, 2), b.bias, padding=1))
    assert torch.allclose(got, m(x), rtol=0, atol=1e-5)
