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
