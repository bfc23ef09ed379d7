"""fp32 NHWC operators of the ReID accuracy mode (csrc/ss_ops32.hip) used by nets.OSNet on CUDA float tensors.

The reference's model.track passes no half= (/root/reference/yolo_multi_model.py:41): its ReID arithmetic is fp32, and
north_star bounds the float distances at 1e-4.  `YOLO(..., reid_fp32=True)` / `FramePipeline(reid_half=False)` therefore run
OSNet-x0.25 with fp32 activations and weights on these hand-written kernels (v_mfma_f32_16x16x4_f32); f16 (fused.py) stays the
throughput default.  Inference only; no CPU path (the modules in nets.py take their plain torch form on CPU tensors).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _lib

ENABLED = True          # A/B: False sends fp32 CUDA tensors to PyTorch-ROCm's library convolutions (the round-4 accuracy mode)

_nv = None              # device int32[1]: images of the batch that are real (packed ReID batches), set by valid_images


def _st(x):
    return C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _ck(rc):
    if rc != 0:
        raise _lib.SSError(rc, "fp32 ReID operator failed")


def _cl(x):
    return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)


def set_option(name: str, value: int):
    """Process-wide A/B switch of the fp32 operators (csrc ss_op32_set_option): chains_form 2 (default) / 1 / 0."""
    _ck(_lib.load().ss_op32_set_option(name.encode(), int(value)))


def usable(x) -> bool:
    return ENABLED and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4


def osnet_ok(net, x) -> bool:
    """The kernels are instantiated for OSNet-x0.25 on 256 x 128 crops (channels 16 / 64 / 96 / 128, 128 -> F head)."""
    if not usable(x) or x.shape[1:] != (3, 256, 128):
        return False
    c = (net.conv1.conv.out_channels, net.conv2[0].conv3.conv.out_channels, net.conv3[0].conv3.conv.out_channels,
         net.conv4[0].conv3.conv.out_channels)
    return c == (16, 64, 96, 128) and net.fc.in_features == 128 and net.conv1.conv.kernel_size == (7, 7)


class valid_images:
    """`with valid_images(n_dev):` — the launches made inside compute only the first n_dev[0] (device int32) images of their
    batch; the grids stay fixed (graph replay), the workgroups of the other images leave at once."""

    def __init__(self, n_dev):
        self.n_dev = n_dev

    def __enter__(self):
        global _nv
        self._old, _nv = _nv, self.n_dev
        return self

    def __exit__(self, *a):
        global _nv
        _nv = self._old
        return False


def _cached(mod, name, like, build):
    t = mod.__dict__.get(name)
    first = t[0] if isinstance(t, tuple) else t
    if t is None or first.device != like.device:
        t = build()
        mod.__dict__[name] = t
    return t


def stem(x, cbr):
    """relu(conv7x7/2(x) + b) -> max pool 3x3/2: x [N, 3, 256, 128] channels-last float -> [N, 16, 64, 32]."""
    x = _cl(x)
    n, _, h, w = x.shape
    conv = cbr.conv

    def build():
        wk = torch.zeros(16, 148, dtype=torch.float32, device=conv.weight.device)
        wk[:, :147] = conv.weight.detach().float().permute(0, 2, 3, 1).reshape(16, 147)       # k = (ky * 7 + kx) * 3 + c
        return wk.contiguous()

    wk = _cached(cbr, "_w32_stem", conv.weight, build)
    y = torch.empty((n, 16, h // 4, w // 4), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op32_stem(_st(x), _p(x), _p(wk), _p(conv.bias), _p(y), n, h, w, _p(_nv)))
    return y


def _w_nk(mod, conv):
    return _cached(mod, "_w32_nk", conv.weight, lambda: conv.weight.detach().float().reshape(conv.weight.shape[0], -1).contiguous())


def pointwise(x, mod, conv, relu=True, res=None):
    """[relu](conv1x1(x) + b (+ res)); `mod` carries the cached weight copy."""
    x = _cl(x)
    n, k, h, w = x.shape
    co = conv.out_channels
    out = torch.empty((n, co, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if res is not None:
        res = _cl(res)
    _ck(_lib.load().ss_op32_pointwise(_st(x), _p(x), _p(_w_nk(mod, conv)), _p(conv.bias), _p(res), _p(out), n * h * w, k, co, int(relu),
                                      _p(_nv), h * w))
    return out


def _block_w(blk, like):
    def build():
        layers = [m for st in blk.streams for m in st]
        c = layers[0].pw.weight.shape[0]
        w1 = torch.stack([m.pw.weight.detach().float().reshape(c, c) for m in layers]).contiguous()
        w9 = torch.stack([m.dw.weight.detach().float().reshape(c, 9).t() for m in layers]).contiguous()          # [10][9][C] tap-major
        b = torch.stack([m.dw.bias.detach().float() for m in layers]).contiguous()
        g = blk.gate
        hid = g.fc1.weight.shape[0]
        return (w1, w9, b, g.fc1.weight.detach().float().reshape(hid, c).contiguous(), g.fc1.bias.detach().float().contiguous(),
                g.fc2.weight.detach().float().reshape(c, hid).contiguous(), g.fc2.bias.detach().float().contiguous())
    return _cached(blk, "_sw32", like, build)


def chains(x1, blk):
    """The block's four LightConv chains on x1 [N, mid, H, W] -> ([y_1..y_4], psum [4, N, bands, mid]) in one launch."""
    x1 = _cl(x1)
    n, c, h, w = x1.shape
    L = _lib.load()
    bands = L.ss_op32_chains_bands(h, w, c)
    if bands < 1:
        raise _lib.SSError(bands, f"fp32 LightConv chains: unsupported map {c} x {h} x {w}")
    w1, w9, b = _block_w(blk, x1)[:3]
    ys = [torch.empty_like(x1, memory_format=torch.channels_last) for _ in range(4)]
    psum = torch.empty(4, n, bands, c, dtype=torch.float32, device=x1.device)
    arr = (C.c_void_p * 4)(*[y.data_ptr() for y in ys])
    _ck(L.ss_op32_chains(_st(x1), _p(x1), _p(w1), _p(w9), _p(b), arr, _p(psum), n, h, w, c, _p(_nv)))
    return ys, psum


def tail(ys, psum, blk, x, nxt, pool, want_out):
    """-> (o or None, o2): o = relu(conv3(sum_t gate_t * ys[t]) + shortcut(x)), o2 = relu(nxt(o)) (2x2-averaged when pool);
    shortcut = x or blk.down(x).  Two launches (gates per image, then the persistent tail)."""
    x = _cl(x)
    n, mid, h, w = ys[0].shape
    c3, c4 = blk.conv3.conv, nxt.conv
    c2, n2 = c3.out_channels, c4.out_channels
    gw1, gb1, gw2, gb2 = _block_w(blk, x)[3:]
    c1, wd, bd = 0, None, None
    if blk.down is not None:
        c1, wd, bd = x.shape[1], _w_nk(blk.down, blk.down.conv), blk.down.conv.bias
    out = torch.empty((n, c2, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last) if want_out else None
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    out2 = torch.empty((n, n2, oh, ow), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    arr = (C.c_void_p * 4)(*[y.data_ptr() for y in ys])
    gates = torch.empty(4, n, mid, dtype=torch.float32, device=x.device)            # workspace of the gate launch
    _ck(_lib.load().ss_op32_tail(_st(x), arr, _p(psum), psum.shape[2], _p(gw1), _p(gb1), _p(gw2), _p(gb2), gw1.shape[0], _p(gates),
                                 _p(_w_nk(blk.conv3, c3)), _p(c3.bias), _p(x), c1, _p(wd), _p(bd), _p(out), _p(_w_nk(nxt, c4)), _p(c4.bias),
                                 _p(out2), int(pool), n, h, w, mid, c2, n2, _p(_nv)))
    return out, out2


def head(x, fc):
    """relu(fc(mean_hw(x))): x [N, 128, H, W] channels-last float -> [N, F] float.  Rows of images past the valid count are zero."""
    x = _cl(x)
    n, c, h, w = x.shape
    out = torch.zeros((n, fc.out_features), dtype=torch.float32, device=x.device) if _nv is not None else \
        torch.empty((n, fc.out_features), dtype=torch.float32, device=x.device)
    _ck(_lib.load().ss_op32_head(_st(x), _p(x), _p(fc.weight), _p(fc.bias), _p(out), n, h * w, c, fc.out_features, _p(_nv)))
    return out
