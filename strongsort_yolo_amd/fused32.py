"""fp32 NHWC operators of the ReID accuracy mode (csrc/ss_ops32.hip) used by nets.OSNet on CUDA float tensors.

The reference's model.track passes no half= (/root/reference/yolo_multi_model.py:41): its ReID arithmetic is fp32, and
north_star bounds the float distances at 1e-4.  `YOLO(..., reid_fp32=True)` / `FramePipeline(reid_half=False)` therefore run
OSNet-x0.25 with fp32 activations and weights on these hand-written kernels (v_mfma_f32_16x16x4_f32); f16 (fused.py) stays the
throughput default.  Inference only; no CPU path (the modules in nets.py take their plain torch form on CPU tensors).
"""
from __future__ import annotations

import ctypes as C
import threading

import torch

from . import lib as _lib
from . import fused as _fused

ENABLED = True          # A/B: False sends fp32 CUDA tensors to PyTorch-ROCm's library convolutions (the round-4 accuracy mode)

_tls = threading.local()  # .nv: device int32[1] = images of the batch that are real (packed ReID batches), set by valid_images for THIS thread


def _nvp(x):
    """The valid-image count of this thread's `with valid_images(...)` block as a launch argument; it must live on x's device."""
    nv = getattr(_tls, "nv", None)
    if nv is not None and nv.device != x.device:
        raise _lib.SSError(-1, f"valid_images count on {nv.device}, batch on {x.device}")
    return _p(nv)



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
        self._old, _tls.nv = getattr(_tls, "nv", None), self.n_dev
        return self

    def __exit__(self, *a):
        _tls.nv = self._old
        return False


def _cached(mod, name, like, build):
    t = mod.__dict__.get(name)
    first = t[0] if isinstance(t, tuple) else t
    if t is None or first.device != like.device:
        t = build()
        mod.__dict__[name] = t
    return t


def usable_u8(x) -> bool:
    """Byte crops (engine.crop_norm_* into a uint8 tensor: the rounded bilinear values, NHWC) the stem normalises itself."""
    return (ENABLED and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.uint8 and x.dim() == 4 and tuple(x.shape[1:]) == (3, 256, 128)
            and x.is_contiguous(memory_format=torch.channels_last))


_MEAN, _STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


_LUT = {}


def crops_from_u8(x):
    """Byte crops -> the float crops a4 would have written, channels-last: ((q / 255) - mean) / sd per value with IEEE float32 operations.
    The 3 x 256 table is computed on the HOST (numpy): PyTorch-ROCm's device division is not correctly rounded (322 of the 768 entries
    come out one ulp off), a4's and k32_stemW's are (csrc/Makefile: -fhip-fp32-correctly-rounded-divide-sqrt)."""
    import numpy as np
    lut = _LUT.get(x.device)
    if lut is None:
        q = np.arange(256, dtype=np.float32) / np.float32(255.0)
        tab = np.stack([(q - np.float32(m)) / np.float32(sd) for m, sd in zip(_MEAN, _STD)]).astype(np.float32)      # [3, 256]
        lut = _LUT[x.device] = torch.from_numpy(tab).to(x.device)
    idx = x.long()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device).contiguous(memory_format=torch.channels_last)
    for c in range(3):
        out[:, c] = lut[c][idx[:, c]]
    return out


STEM_CONV1 = True       # A/B: the first OSBlock's conv1 from the stem's launch (False: a k32_pw launch that reads the stem's output back)


def stem(x, cbr, conv1=None):
    """relu(conv7x7/2(x) + b) -> max pool 3x3/2: x [N, 3, 256, 128] channels-last float (or uint8 byte crops) -> [N, 16, 64, 32].
    conv1 (a ConvBR 16 -> 16, 1x1, ReLU: the first OSBlock's): -> (y, relu(conv1(y))) from ONE launch."""
    if x.dtype != torch.uint8:
        x = _cl(x)
    n, _, h, w = x.shape
    conv = cbr.conv

    def build():
        wk = torch.zeros(16, 148, dtype=torch.float32, device=conv.weight.device)
        wk[:, :147] = conv.weight.detach().float().permute(0, 2, 3, 1).reshape(16, 147)       # k = (ky * 7 + kx) * 3 + c
        return wk.contiguous()

    wk = _cached(cbr, "_w32_stem", conv.weight, build)
    y = torch.empty((n, 16, h // 4, w // 4), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if conv1 is not None:
        c1 = conv1.conv
        if STEM_CONV1 and c1.in_channels == 16 and c1.out_channels == 16 and c1.kernel_size == (1, 1) and conv1.relu and c1.bias is not None:
            y1 = torch.empty_like(y)
            _ck(_lib.load().ss_op32_stem_conv1(_st(x), _p(x), int(x.dtype == torch.uint8), _p(wk), _p(conv.bias), _p(y), _p(_w_nk(conv1, c1)), _p(c1.bias), _p(y1),
                                               n, h, w, _nvp(x)))
            return y, y1
    fn = _lib.load().ss_op32_stem_u8 if x.dtype == torch.uint8 else _lib.load().ss_op32_stem
    _ck(fn(_st(x), _p(x), _p(wk), _p(conv.bias), _p(y), n, h, w, _nvp(x)))
    return y if conv1 is None else (y, pointwise(y, conv1, conv1.conv, relu=True))


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
                                      _nvp(x), h * w))
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
    bands = L.ss_op32_chains_bands(n, h, w, c)
    if bands < 1:
        raise _lib.SSError(bands, f"fp32 LightConv chains: unsupported map {c} x {h} x {w}")
    w1, w9, b = _block_w(blk, x1)[:3]
    ys = [torch.empty_like(x1, memory_format=torch.channels_last) for _ in range(4)]
    psum = torch.empty(4, n, bands, c, dtype=torch.float32, device=x1.device)
    arr = (C.c_void_p * 4)(*[y.data_ptr() for y in ys])
    _ck(L.ss_op32_chains(_st(x1), _p(x1), _p(w1), _p(w9), _p(b), arr, _p(psum), n, h, w, c, _nvp(x1)))
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
                                 _p(out2), int(pool), n, h, w, mid, c2, n2, _nvp(x)))
    return out, out2


def head(x, fc):
    """relu(fc(mean_hw(x))): x [N, 128, H, W] channels-last float -> [N, F] float.  Rows of images past the valid count are zero."""
    x = _cl(x)
    n, c, h, w = x.shape
    out = torch.zeros((n, fc.out_features), dtype=torch.float32, device=x.device) if getattr(_tls, "nv", None) is not None else \
        torch.empty((n, fc.out_features), dtype=torch.float32, device=x.device)
    _ck(_lib.load().ss_op32_head(_st(x), _p(x), _p(fc.weight), _p(fc.bias), _p(out), n, h * w, c, fc.out_features, _nvp(x)))
    return out


# ---- the detector's convolutions in fp32 (csrc k32_conv / k32_conv0) ---------------------------------------------------------
DET = True              # A/B: False sends the fp32 detector to PyTorch-ROCm's library convolutions (bias / SiLU / concat as separate passes)


def _nhwc_view(t):
    """-> pixel stride (floats) of a channels-last tensor or of a channel slice of one, else None."""
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if sc != 1 or sw < c or sw % 4 or sh != w * sw or (n > 1 and sn != h * sh) or t.data_ptr() % 16:
        return None
    return sw


def conv_ok(x, conv) -> bool:
    """k32_conv covers this convolution on this input: fp32 CUDA NHWC (or a channel slice of one), 1x1 / 3x3, stride 1 / 2 (1x1: 1),
    pad k // 2, no groups / dilation, channel counts multiples of 16, a bias."""
    if not (DET and usable(x)) or conv.groups != 1 or conv.dilation != (1, 1) or conv.bias is None:
        return False
    k, s = conv.kernel_size, conv.stride
    if k[0] != k[1] or s[0] != s[1] or k[0] not in (1, 3) or s[0] not in (1, 2) or (k[0] == 1 and s[0] != 1) or conv.padding != (k[0] // 2, k[0] // 2):
        return False
    return conv.in_channels % 16 == 0 and conv.out_channels % 16 == 0 and x.shape[1] == conv.in_channels and _nhwc_view(x) is not None


def conv0_ok(x, conv) -> bool:
    return (DET and usable(x) and conv.in_channels == 3 and conv.out_channels == 16 and conv.kernel_size == (3, 3) and conv.stride == (2, 2)
            and conv.padding == (1, 1) and conv.groups == 1 and conv.bias is not None and x.is_contiguous(memory_format=torch.channels_last))


def _w_khwc(mod, conv):
    """[Cout][kh][kw][Cin] copy of the weight (the K order of k32_conv), cached on `mod`."""
    return _cached(mod, "_w32_khwc", conv.weight, lambda: conv.weight.detach().float().permute(0, 2, 3, 1).contiguous())


def conv(x, mod, conv, act="silu", out=None, res=None):
    """act(conv(x) + b) (+ res, added after the activation) on k32_conv.  x / out / res: channels-last tensors or channel slices of
    wider channels-last tensors; out (when given) is written in place and returned."""
    n, ci, h, w = x.shape
    k, s = conv.kernel_size[0], conv.stride[0]
    oh, ow = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    co = conv.out_channels
    if out is None:
        out = torch.empty((n, co, oh, ow), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    xs, os_ = _nhwc_view(x), _nhwc_view(out)
    rs = _nhwc_view(res) if res is not None else 0
    if xs is None or os_ is None or rs is None or tuple(out.shape) != (n, co, oh, ow) or (res is not None and tuple(res.shape) != tuple(out.shape)):
        raise _lib.SSError(-1, "fp32 convolution: operand is not a channels-last tensor / channel slice of the expected shape")
    _ck(_lib.load().ss_op32_conv(_st(x), _p(x), xs, _p(_w_khwc(mod, conv)), _p(conv.bias), _p(res), rs, _p(out), os_, n, h, w, ci, co, k, s,
                                 1 if act == "silu" else 0))
    return out


def conv0(x, mod, conv, act="silu", out=None):
    """The first convolution (3 -> 16, 3x3, stride 2) on k32_conv0."""
    n, _, h, w = x.shape
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if out is None:
        out = torch.empty((n, 16, oh, ow), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op32_conv0(_st(x), _p(x), _p(_w_khwc(mod, conv)), _p(conv.bias), _p(out), _nhwc_view(out), n, h, w, 1 if act == "silu" else 0))
    return out


def upcat_ok(lo, hi) -> bool:
    if not (DET and usable(lo) and usable(hi)) or _nhwc_view(lo) is None or _nhwc_view(hi) is None:
        return False
    n, cl, h, w = lo.shape
    return tuple(hi.shape[2:]) == (2 * h, 2 * w) and hi.shape[0] == n and cl % 4 == 0 and hi.shape[1] % 4 == 0


def upcat(lo, hi, lo_first=True):
    """cat(upsample2x_nearest(lo), hi) (or cat(hi, up(lo))) along channels, one launch (k32_upcat)."""
    n, cl, h, w = lo.shape
    ch = hi.shape[1]
    out = torch.empty((n, cl + ch, 2 * h, 2 * w), dtype=torch.float32, device=lo.device, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op32_upcat(_st(lo), _p(lo), _nhwc_view(lo), cl, _p(hi), _nhwc_view(hi), ch, _p(out), n, 2 * h, 2 * w, int(lo_first)))
    return out


def v8_decode(boxes, clss, strides, nc, ext=None, n_ext=0, ext_mode=0):
    """DFL + dist2bbox + sigmoid + level concat of the anchor-free head in one launch (k32_v8_decode) -> [B, 4 + nc (+ n_ext), A] float32.
    boxes / clss / ext: the branches' outputs per level (bias included), made dense channels-last here if they are not."""
    boxes, clss = [_cl(t) for t in boxes], [_cl(t) for t in clss]
    ext = [_cl(t) for t in ext] if n_ext else None
    B = boxes[0].shape[0]
    A = sum(t.shape[2] * t.shape[3] for t in boxes)
    pred = _fused.decode_into.target((B, 4 + nc + n_ext, A), boxes[0].device)
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    ints = lambda v: (C.c_int * 3)(*v)
    _ck(_lib.load().ss_op32_v8_decode(_st(pred), arr(boxes), arr(clss), arr(ext) if n_ext else None, n_ext, ext[0].shape[1] if n_ext else 0, ext_mode,
                                      ints([t.shape[2] for t in boxes]), ints([t.shape[3] for t in boxes]), ints(strides), B, nc, clss[0].shape[1], _p(pred)))
    return pred


def sppf_pools(x):
    """cat(x, pool5(x), pool5(pool5(x)), pool5^3(x)) along channels, one launch (k32_sppf)."""
    n, c, h, w = x.shape
    out = torch.empty((n, 4 * c, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op32_sppf_pools(_st(x), _p(x), _nhwc_view(x), _p(out), n, h, w, c))
    return out
