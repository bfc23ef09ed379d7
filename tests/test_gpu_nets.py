"""Fused NHWC-half glue operators (csrc/ss_ops.hip) vs the plain torch modules they replace."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,shape", [("yolov8n", (1, 3, 384, 640)), ("osnet", (8, 3, 256, 128)), ("yolov8n-pose", (2, 3, 128, 160))])
def test_fused_ops_match_torch_modules(name, shape):
    from strongsort_yolo_amd import fused, nets
    dev = torch.device("cuda", 0)
    m = (nets.build_reid() if name == "osnet" else nets.build_detector(name)).to(dev, torch.float16).to(memory_format=torch.channels_last)
    x = torch.randn(*shape, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        fused.ENABLED = True
        a = m(x).float()
        fused.ENABLED = False
        b = m(x).float()
        fused.ENABLED = True
    assert a.shape == b.shape and torch.isfinite(a).all()
    scale = b.abs().max().item() + 1e-6
    assert (a - b).abs().max().item() <= 2e-2 * scale, ((a - b).abs().max().item(), scale)


@pytest.mark.parametrize("shape", [(3, 16, 64, 32), (2, 24, 32, 16), (5, 32, 16, 8), (1, 16, 20, 24), (2, 32, 7, 8)])
def test_lightconv_matches_pointwise_plus_depthwise(shape):
    """Fused LightConv3x3 (MFMA pointwise -> LDS -> depthwise+bias+ReLU) vs the two-step form: the pointwise
    product rounded to half (as the GEMM writes it), then the depthwise in fp32.  Covers bands whose halo rows
    fall outside the image (H not a multiple of the 16-row band, H < band)."""
    import torch.nn.functional as F
    from strongsort_yolo_amd import fused
    n, c, h, w = shape
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(c * 100 + h)
    x = torch.randn(n, c, h, w, generator=g).to(dev, torch.float16).contiguous(memory_format=torch.channels_last)
    w1 = (torch.randn(c, c, generator=g) / c ** 0.5).to(dev, torch.float16)
    wd = (torch.randn(c, 9, generator=g) / 3).to(dev, torch.float16)
    b = torch.randn(c, generator=g).to(dev, torch.float16)
    assert fused.lightconv_ok(x)
    got = fused.lightconv(x, w1, wd.t().contiguous(), b).float()
    mid = F.conv2d(x.float(), w1.float().view(c, c, 1, 1)).half().float()
    ref = F.relu(F.conv2d(mid, wd.float().view(c, 1, 3, 3), b.float(), padding=1, groups=c))
    ref = ref.half().float()
    err = (got - ref).abs().max().item()
    # one half-ulp flip of an intermediate can move an output by ~|w| * ulp(mid); outputs themselves are half
    assert err <= 2e-2 * (ref.abs().max().item() + 1e-6), err
    assert (got != ref).float().mean().item() < 0.02       # and nearly every element is bit-identical
