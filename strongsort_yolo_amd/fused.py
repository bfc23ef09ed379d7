"""Fused NHWC-half glue operators (csrc/ss_ops.hip) used by nets.py on the GPU.

Inference only.  Enabled for CUDA half tensors; on CPU (oracle-side baseline, CPU tests) the modules
in nets.py take their plain torch path instead.
"""
from __future__ import annotations

import ctypes as C
import threading

import torch

from . import lib as _lib

import os as _os

ENABLED = True


def _flag(name: str) -> bool:
    """The fused kernel families are module attributes (POINTWISE, CONV3X3, BNECK, GROUP, HEAD, ...), all on.  They are plain
    process state: flip them with set_flags(NAME=False) (tests, A/B measurements) — nothing here reads the environment at
    import; measurement tools that take their switches from the command line / environment call flags_from_env() themselves."""
    return True


FLAG_NAMES = ("POINTWISE", "CONV3X3", "BNECK", "GROUP", "HEAD", "HEAD_TILE16", "LIGHTCONV", "CONV0", "STEM", "STEM_CONV1", "STREAMS", "TAIL", "GLUE", "DW3X3", "HEAD_EXT", "C3K2", "PSA", "OSHEAD")


def set_flags(**kw):
    """A/B switches: set_flags(HEAD=False) takes the grouped head launches instead of k_head, ...; BNECK_C=(16, 32) restricts the
    widths that take k_bneck.  Returns the previous values (to restore)."""
    g, old = globals(), {}
    for k, v in kw.items():
        if k not in FLAG_NAMES and k != "BNECK_C":
            raise ValueError(f"unknown fused switch {k!r}")
        old[k] = g[k]
        g[k] = tuple(int(c) for c in v) if k == "BNECK_C" else bool(v)
    return old


def flags_from_env(env=None):
    """For measurement tools: SS_FUSED_<NAME>=0|1 and SS_BNECK_C=16,32,.. from the (given) environment -> set_flags."""
    env = _os.environ if env is None else env
    kw = {n: env["SS_FUSED_" + n] != "0" for n in FLAG_NAMES if "SS_FUSED_" + n in env}
    if "SS_BNECK_C" in env:
        kw["BNECK_C"] = tuple(int(c) for c in env["SS_BNECK_C"].split(","))
    return set_flags(**kw)

ACT = {"none": 0, "relu": 1, "silu": 2, "sigmoid": 3}


def usable(x: torch.Tensor) -> bool:
    return ENABLED and x.is_cuda and x.dtype == torch.float16 and x.dim() == 4


def _cl(x):
    return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)


def _st(x):
    return C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _ck(rc):
    if rc != 0:
        raise _lib.SSError(rc, "fused op failed")


class valid_images:
    """`with valid_images(n_dev, batch):` — the OSNet-side launches of `batch` images made inside, on the current torch stream,
    compute only the first n_dev[0] (device int32) of them (packed ReID batches; csrc ss_op_set_valid_images: the setting
    belongs to this thread and this stream)."""

    def __init__(self, n_dev, batch):
        self.n_dev, self.batch = n_dev, int(batch)
        self._stream = None

    _tls = threading.local()   # .active: nesting depth of THIS thread's `with valid_images(...)` blocks that carry a count (its launches skip images)

    def __enter__(self):
        if self.n_dev is not None:
            self._stream = C.c_void_p(torch.cuda.current_stream(self.n_dev.device).cuda_stream)
            _ck(_lib.load().ss_op_set_valid_images(self._stream, _p(self.n_dev), self.batch))
            valid_images._tls.active = getattr(valid_images._tls, "active", 0) + 1
        return self

    def __exit__(self, *a):
        if self.n_dev is not None:
            valid_images._tls.active -= 1
            _ck(_lib.load().ss_op_set_valid_images(self._stream, None, 0))
        return False


def _nv_active() -> bool:
    return getattr(valid_images._tls, "active", 0) > 0


def set_option(name: str, value: int):
    """Process-wide A/B switch of the operators: pw_epilogue, pw_splitk, osnet_chains (csrc ss_op_set_option)."""
    _ck(_lib.load().ss_op_set_option(name.encode(), int(value)))


def bias_act_(x, bias, act="none", res=None):
    """x <- act(x + bias[c] (+ res)) in place; x NCHW-shaped, channels-last memory."""
    x = _cl(x)
    n, c, h, w = x.shape
    if res is not None:
        res = _cl(res)
    _ck(_lib.load().ss_op_bias_act_f16(_st(x), _p(x), _p(bias), _p(res), n * h * w, c, ACT[act]))
    return x


def bias_act_place(y, bias, act, out, c_off, res=None, res_after=False, out2=None, c0=0):
    """out[:, c_off:c_off+C] = act(y + bias) (+ res after the activation), `out` a wider channels-last tensor (a
    concat buffer); channels [c0, c0+out2.C) are mirrored into the dense tensor `out2`.  One launch."""
    y = _cl(y)
    n, c, h, w = y.shape
    if res is not None:
        res = _cl(res)
    ct = out.shape[1]
    dst = C.c_void_p(out.data_ptr() + 2 * c_off)
    _ck(_lib.load().ss_op_bias_act_place_f16(_st(y), _p(y), _p(bias), _p(res), n * h * w, c, ACT[act], int(res_after), dst, ct,
                                             _p(out2), c0, 0 if out2 is None else out2.shape[1]))


POINTWISE = _flag("POINTWISE")            # own MFMA kernel for 1x1 convolutions (off: MIOpen / hipBLASLt + separate epilogue)


def pointwise_ok(conv) -> bool:
    return POINTWISE and is_pointwise(conv) and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0 and conv.bias is not None


def weight_nk(mod, conv):
    """[Cout, Cin] view of a 1x1 conv weight (already the layout k_pw stages), cached on the module."""
    w = getattr(mod, "_w_nk", None)
    if w is None or w.device != conv.weight.device or w.dtype != conv.weight.dtype:
        w = conv.weight.detach().reshape(conv.weight.shape[0], -1).contiguous()
        mod._w_nk = w
    return w


def pointwise(x, w_nk, bias, act="none", res=None, res_after=False, out=None, c_off=0, out2=None, c0=0):
    """act(conv1x1(x) + bias) (+ res, before or after the activation) in ONE launch; optionally written into the
    channel slice [c_off, c_off+N) of a wider channels-last tensor `out`, with channels [c0, c0+out2.C) mirrored
    into the dense tensor `out2` (C2f blocks)."""
    x = _cl(x)
    b, k, h, w = x.shape
    n = w_nk.shape[0]
    if res is not None:
        res = _cl(res)
    ret = out
    if out is None:
        ret = out = torch.empty((b, n, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    dst = C.c_void_p(out.data_ptr() + 2 * c_off)
    _ck(_lib.load().ss_op_pointwise_f16(_st(x), _p(x), _p(w_nk), _p(bias), _p(res), b * h * w, k, n, ACT[act], int(res_after),
                                        dst, out.shape[1], _p(out2), c0, 0 if out2 is None else out2.shape[1]))
    return ret


CONV3X3 = _flag("CONV3X3")              # own implicit-GEMM kernel for 3x3 convolutions (off: MIOpen + separate epilogue)


def conv3x3_ok(conv) -> bool:
    return (CONV3X3 and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.stride in ((1, 1), (2, 2)) and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0
            and conv.bias is not None)


def weight_n9k(mod, conv):
    """[Cout, 3*3*Cin] (tap-major, channel-minor) copy of a 3x3 conv weight, cached on the module."""
    w = getattr(mod, "_w_n9k", None)
    if w is None or w.device != conv.weight.device or w.dtype != conv.weight.dtype:
        w = conv.weight.detach().permute(0, 2, 3, 1).reshape(conv.weight.shape[0], -1).contiguous()
        mod._w_n9k = w
    return w


def conv3x3(x, w_n9k, bias, stride=1, act="none", res=None, res_after=False, out=None, c_off=0, out2=None, c0=0):
    """act(conv3x3(x, pad 1, stride 1|2) + bias) (+ res) in one launch; placement arguments as `pointwise`."""
    x = _cl(x)
    b, k, h, w = x.shape
    n = w_n9k.shape[0]
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    if res is not None:
        res = _cl(res)
    ret = out
    if out is None:
        ret = out = torch.empty((b, n, oh, ow), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    dst = C.c_void_p(out.data_ptr() + 2 * c_off)
    _ck(_lib.load().ss_op_conv3x3_f16(_st(x), _p(x), _p(w_n9k), _p(bias), _p(res), b, h, w, k, n, stride, ACT[act], int(res_after),
                                      dst, out.shape[1], _p(out2), c0, 0 if out2 is None else out2.shape[1]))
    return ret


BNECK_C = (16, 32, 64, 128)             # channel counts that take the fused bottleneck kernel (A/B: set_flags(BNECK_C=...))
BNECK = _flag("BNECK")                  # a C2f bottleneck (3x3 + 3x3 + shortcut) in one launch, the intermediate in LDS


def bottleneck_ok(m) -> bool:
    a, b = m.cv1.conv, m.cv2.conv
    silu = all(type(cv.act).__name__ == "SiLU" for cv in (m.cv1, m.cv2))
    return (BNECK and silu and conv3x3_ok(a) and conv3x3_ok(b) and a.stride == (1, 1) and b.stride == (1, 1)
            and a.in_channels == a.out_channels == b.out_channels and a.in_channels in BNECK_C)


def bottleneck(x, m, out, c_off, out2=None):
    """out[:, c_off:c_off+C] (= out2) = [x +] silu(conv3x3(silu(conv3x3(x)))) of nets.Bottleneck `m`; x dense channels-last."""
    x = _cl(x)
    b, c, h, w = x.shape
    a, bb = m.cv1.conv, m.cv2.conv
    _ck(_lib.load().ss_op_bottleneck_f16(_st(x), _p(x), _p(weight_n9k(m.cv1, a)), _p(a.bias), _p(weight_n9k(m.cv2, bb)), _p(bb.bias),
                                         b, h, w, c, int(m.add), C.c_void_p(out.data_ptr() + 2 * c_off), out.shape[1], _p(out2)))
    return out


def bottleneck_padded_ok(m) -> bool:
    """A Bottleneck c -> c/2 -> c (v11's e = 0.5) can run on k_bneck with its hidden width zero-padded to c."""
    a, b = m.cv1.conv, m.cv2.conv
    silu = all(type(cv.act).__name__ == "SiLU" for cv in (m.cv1, m.cv2))
    return (BNECK and silu and conv3x3_ok(a) and conv3x3_ok(b) and a.stride == (1, 1) and b.stride == (1, 1) and a.in_channels == b.out_channels
            and a.out_channels == b.in_channels and a.out_channels < a.in_channels and a.in_channels in BNECK_C)


def bottleneck_padded(x, m, out, c_off, out2=None):
    """fused.bottleneck for a Bottleneck whose hidden width is smaller than its in / out width c: the first convolution's missing
    output channels get zero weights and biases (SiLU(0) = 0), the second one's matching input columns zero weights — the equal-width
    kernel computes the same sums plus exact zeros.  Prepared weights cached on `m`."""
    x = _cl(x)
    b, c, h, w = x.shape
    a, bb = m.cv1.conv, m.cv2.conv
    p = getattr(m, "_padded", None)
    if p is None or p[0].device != a.weight.device or p[0].dtype != a.weight.dtype:
        hid = a.out_channels
        w1 = torch.zeros(c, 3, 3, c, device=a.weight.device, dtype=a.weight.dtype)
        w1[:hid] = a.weight.detach().permute(0, 2, 3, 1)
        b1 = torch.zeros(c, device=a.weight.device, dtype=a.weight.dtype)
        b1[:hid] = a.bias.detach()
        w2 = torch.zeros(c, 3, 3, c, device=a.weight.device, dtype=a.weight.dtype)
        w2[:, :, :, :hid] = bb.weight.detach().permute(0, 2, 3, 1)
        p = m._padded = (w1.reshape(c, -1).contiguous(), b1, w2.reshape(c, -1).contiguous())
    _ck(_lib.load().ss_op_bottleneck_f16(_st(x), _p(x), _p(p[0]), _p(p[1]), _p(p[2]), _p(bb.bias), b, h, w, c, int(m.add),
                                         C.c_void_p(out.data_ptr() + 2 * c_off), out.shape[1], _p(out2)))
    return out


GROUP = _flag("GROUP")                  # independent convolutions of the detect head in one launch per depth


def conv_group(items):
    """items: [(x, w_prepared, bias, ksize, stride, act)], all 3x3 (pad 1; weight_n9k) or all 1x1 (weight_nk), N <= 80, at most
    12 entries.  Returns the dense channels-last outputs; ONE launch (csrc k_pw_group)."""
    n = len(items)
    descs = (_lib.ss_conv_desc * n)()
    outs, keep = [], []
    for d, (x, w, bias, ksize, stride, act) in zip(descs, items):
        x = _cl(x)
        b, cin, h, wd = x.shape
        N = w.shape[0]
        oh, ow = (h - 1) // stride + 1, (wd - 1) // stride + 1
        y = torch.empty((b, N, oh, ow), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        d.x, d.w, d.bias, d.out = x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr()
        d.B, d.H, d.W, d.Cin, d.N, d.ksize, d.stride, d.act = b, h, wd, cin, N, ksize, stride, ACT[act]
        outs.append(y); keep.append(x)
    _ck(_lib.load().ss_op_conv_group_f16(_st(outs[0]), n, descs))
    return outs


# a level of the v8 detect head (both branches, 3 layers each) in one launch, intermediates in LDS (csrc k_head): bit-identical to the
# grouped launches and measured NEUTRAL (r04, same box: yolov8n on 32 frames 0.964-0.989 ms with it, 0.968-0.980 without; inside the two-stream
# pipeline 10.5-10.9 k vs 10.95 k frames/s) — its 60-117 KB of LDS per workgroup leave one or two workgroups per CU.  Off by default;
# set_flags(HEAD=True) / bench.py --fused HEAD=1 turns it on.
HEAD = False
HEAD_TILE16 = False                     # A/B: 8 x 16 tiles at the stride-8 level (measured neutral, r04)


def head_level_ok(x, box_seq, cls_seq) -> bool:
    """nets.Detect branch pairs that k_head covers: Conv3x3+SiLU, Conv3x3+SiLU, Conv2d 1x1 with 64 (box) / 80 (class) mid channels."""
    if not (HEAD and usable(x) and x.shape[1] in (64, 128, 256)):
        return False
    for seq, cm in ((box_seq, 64), (cls_seq, 80)):
        a, b, c = seq[0], seq[1], seq[2]
        if not (hasattr(a, "conv") and hasattr(b, "conv") and conv3x3_ok(a.conv) and conv3x3_ok(b.conv) and a.conv.stride == (1, 1)
                and b.conv.stride == (1, 1) and type(a.act).__name__ == "SiLU" and type(b.act).__name__ == "SiLU"
                and a.conv.in_channels == x.shape[1] and a.conv.out_channels == cm and b.conv.in_channels == cm and b.conv.out_channels == cm
                and pointwise_ok(c) and c.in_channels == cm and c.out_channels <= cm):
            return False
    return True


def head_level(x, box_seq, cls_seq, tile16=None):
    """-> (box branch output [B, 64, H, W], class branch output [B, nc, H, W]) of one level, one launch (csrc k_head)."""
    x = _cl(x)
    b, cin, h, w = x.shape
    outs, keep = [], []
    arr = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
    w1, b1, w2, b2, w3, b3 = [], [], [], [], [], []
    for seq in (box_seq, cls_seq):
        w1.append(weight_n9k(seq[0], seq[0].conv)); b1.append(seq[0].conv.bias)
        w2.append(weight_n9k(seq[1], seq[1].conv)); b2.append(seq[1].conv.bias)
        w3.append(weight_nk(seq[2], seq[2])); b3.append(seq[2].bias)
        outs.append(torch.empty((b, seq[2].out_channels, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last))
    nout = (C.c_int * 2)(*[o.shape[1] for o in outs])
    t16 = HEAD_TILE16 if tile16 is None else bool(tile16)
    _ck(_lib.load().ss_op_head_f16(_st(x), _p(x), arr(w1), arr(b1), arr(w2), arr(b2), arr(w3), arr(b3), arr(outs), nout, b, h, w, cin, int(t16)))
    return outs[0], outs[1]


def place_ok(c: int, ctot: int) -> bool:
    return c % 8 == 0 and ctot % 8 == 0


class decode_into:
    """`with decode_into(buf):` — the head decode launched inside (this thread; v8_decode here and in fused32) writes its row tensor
    [B, 4 + nc (+ n_ext), A] float32 straight into `buf` when shape, dtype and device match, instead of a fresh tensor the caller then
    copies (the frame pipeline's static NMS input: 54 MB per 32 frames copied once more)."""
    _tls = threading.local()

    def __init__(self, buf):
        self.buf = buf

    def __enter__(self):
        self.prev = getattr(decode_into._tls, "buf", None)
        decode_into._tls.buf = self.buf
        return self

    def __exit__(self, *exc):
        decode_into._tls.buf = self.prev
        return False

    @staticmethod
    def target(shape, device):
        b = getattr(decode_into._tls, "buf", None)
        if b is not None and tuple(b.shape) == tuple(shape) and b.dtype == torch.float32 and b.device == device and b.is_contiguous():
            return b
        return torch.empty(*shape, dtype=torch.float32, device=device)


def v8_decode(boxes, clss, box_bias, cls_bias, strides, nc, ext=None, n_ext=0, ext_mode=0):
    """DFL + dist2bbox + sigmoid + level concat of the anchor-free head in one launch -> [B, 4+nc(+n_ext), A] float32.
    ext: the third branch's three outputs [B, >= n_ext, H, W] (bias added), decoded as keypoint triplets (ext_mode 1) or copied raw
    (ext_mode 0: mask coefficients) into rows 4+nc.. of the same tensor."""
    boxes, clss = [_cl(t) for t in boxes], [_cl(t) for t in clss]
    B = boxes[0].shape[0]
    H = (C.c_int * 3)(*[t.shape[2] for t in boxes]); W = (C.c_int * 3)(*[t.shape[3] for t in boxes])
    A = sum(t.shape[2] * t.shape[3] for t in boxes)
    pred = decode_into.target((B, 4 + nc + n_ext, A), boxes[0].device)
    arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    if n_ext or clss[0].shape[1] != nc:                  # a third branch, or class tensors padded past nc channels
        ext = [_cl(t) for t in ext] if n_ext else clss
        _ck(_lib.load().ss_op_v8_decode_ext_f16(_st(pred), arr(boxes), arr(clss), arr(box_bias), arr(cls_bias), arr(ext), n_ext, ext[0].shape[1],
                                                ext_mode, H, W, (C.c_int * 3)(*strides), B, nc, clss[0].shape[1], _p(pred)))
    else:
        _ck(_lib.load().ss_op_v8_decode_f16(_st(pred), arr(boxes), arr(clss), arr(box_bias), arr(cls_bias), H, W,
                                            (C.c_int * 3)(*strides), B, nc, _p(pred)))
    return pred


HEAD_EXT = _flag("HEAD_EXT")            # pose / segmentation heads: the third branch on the convolution kernels (zero-padded widths) and its rows written by the decode launch
C3K2 = _flag("C3K2")                    # v11's C3k2 / C3k blocks with every producer writing its slice of the concat buffer (no chunk / cat / add launches)
DW3X3 = _flag("DW3X3")                  # depthwise 3x3 / stride 1 convolutions (the v11 head's DWConv) on k_dw3x3 instead of MIOpen


def dw3x3_ok(conv) -> bool:
    return (DW3X3 and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.stride == (1, 1)
            and conv.groups == conv.in_channels == conv.out_channels and conv.in_channels % 8 == 0 and conv.bias is not None)


def weight_dw9(mod, conv):
    """[9, C] (tap-major) copy of a depthwise 3x3 weight, cached on the module."""
    w = getattr(mod, "_w_dw9", None)
    if w is None or w.device != conv.weight.device or w.dtype != conv.weight.dtype:
        w = conv.weight.detach().reshape(conv.weight.shape[0], 9).t().contiguous()
        mod._w_dw9 = w
    return w


PREPARED = ("_w_nk", "_w_n9k", "_w_dw9", "_w_t", "_w_c0", "_w_stem", "_w1", "_w9", "_padded", "_padded_last", "_w12_", "_w_pair_", "_sw", "_gw",
            "_w32_stem", "_w32_nk", "_sw32", "_w32_khwc")       # ... and the fp32 accuracy mode's copies (fused32.py)


def clear_prepared(module):
    """Drop every kernel-side copy of the weights cached on the sub-modules of `module` (weight_nk / weight_n9k / padded_branch / the
    OSBlock stacks and operand tables ...): the next forward rebuilds them from the tensors the modules hold NOW.  nets.load_weights
    calls it, so a checkpoint loaded after a forward pass cannot run on stale copies."""
    for m in module.modules():
        for a in PREPARED:
            if a in m.__dict__:
                del m.__dict__[a]


def padded_last(mod, conv):
    """A 1x1 Conv2d whose output count is not a multiple of 8 (the class branch of a one-class head) as (w_nk, bias) zero-padded to 8
    rows, cached on `mod`; conv.in_channels % 8 == 0."""
    p = getattr(mod, "_padded_last", None)
    if p is None or p[0].device != conv.weight.device or p[0].dtype != conv.weight.dtype:
        n, np_ = conv.out_channels, (conv.out_channels + 7) // 8 * 8
        w = torch.zeros(np_, conv.in_channels, device=conv.weight.device, dtype=conv.weight.dtype)
        w[:n] = conv.weight.detach().reshape(n, -1)
        b = torch.zeros(np_, device=conv.weight.device, dtype=conv.weight.dtype)
        b[:n] = conv.bias.detach()
        p = mod._padded_last = (w, b)
    return p


def padded_branch(mod, seq, mult=16):
    """A head branch Conv 3x3 -> Conv 3x3 -> Conv2d 1x1 whose hidden width is not a multiple of 8 (the pose branch: 51 channels) as
    zero-padded weights the convolution kernels take: [(w_n9k, bias)] x 2 + (w_nk, bias) with the hidden width rounded up to `mult`
    and the output rows to 8; the padded channels carry exact zeros (SiLU(0) = 0).  Cached on `mod`."""
    p = getattr(mod, "_padded", None)
    c0 = seq[0].conv
    if p is None or p[0][0].device != c0.weight.device or p[0][0].dtype != c0.weight.dtype:
        up = lambda v, m: (v + m - 1) // m * m
        h, hp = c0.out_channels, up(c0.out_channels, mult)
        n, np_ = seq[2].out_channels, up(seq[2].out_channels, 8)
        dev, dt = c0.weight.device, c0.weight.dtype
        def pad(w, rows, cols_in):                       # [N, Cin, kh, kw] -> [rows, kh*kw*cols_in], tap-major
            N, Cin, kh, kw = w.shape
            o = torch.zeros(rows, kh, kw, cols_in, device=dev, dtype=dt)
            o[:N, :, :, :Cin] = w.detach().permute(0, 2, 3, 1)
            return o.reshape(rows, -1).contiguous()
        def padb(b, rows):
            o = torch.zeros(rows, device=dev, dtype=dt)
            o[:b.shape[0]] = b.detach()
            return o
        p = mod._padded = ((pad(c0.weight, hp, c0.in_channels), padb(c0.bias, hp)),
                           (pad(seq[1].conv.weight, hp, hp), padb(seq[1].conv.bias, hp)),
                           (pad(seq[2].weight, np_, hp), padb(seq[2].bias, np_)))
    return p


def dwconv3x3(x, w9, bias, act="relu"):
    x = _cl(x)
    n, c, h, w = x.shape
    y = torch.empty_like(x, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op_dwconv3x3_f16(_st(x), _p(x), _p(w9), _p(bias), _p(y), n, h, w, c, ACT[act]))
    return y


LIGHTCONV = _flag("LIGHTCONV")            # fused pointwise + depthwise path of OSNet's LightConv3x3 (off: GEMM + dwconv3x3)


def lightconv_ok(x) -> bool:
    n, c, h, w = x.shape
    return LIGHTCONV and c in (16, 24, 32) and w % 8 == 0 and 18 * (w + 2) * ((c + 15) // 16 * 16) * 2 <= 65536


def lightconv(x, w1, w9, bias):
    """relu(dw3x3(pw1x1(x)) + bias) in one launch.  w1 [C,C] (out, in) of the pointwise conv, w9 [9,C] tap-major."""
    x = _cl(x)
    n, c, h, w = x.shape
    y = torch.empty_like(x, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op_lightconv_f16(_st(x), _p(x), _p(w1), _p(w9), _p(bias), _p(y), n, h, w, c))
    return y


CONV0 = _flag("CONV0")                  # the detector's first convolution (3 input channels) on its own MFMA kernel


def conv0_ok(x, conv) -> bool:
    n, c, h, w = x.shape
    return (CONV0 and usable(x) and c == 3 and w % 128 == 0 and h >= 2 and conv.kernel_size == (3, 3) and conv.stride == (2, 2)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.out_channels in (16, 32, 48)
            and conv.bias is not None)


def conv0_weight(mod, conv):
    """[4][3][Cout][16]: for conv columns c = 4n + r, per (ky, out channel) the 9 (kx, ch) taps in the order they lie in an
    NHWC input row, placed (6r + 5) % 8 halfs into a 16-wide K window of two aligned 8-half blocks (k_conv0)."""
    w = getattr(mod, "_w_c0", None)
    if w is None or w.device != conv.weight.device or w.dtype != conv.weight.dtype:
        co = conv.out_channels
        taps = conv.weight.detach().permute(2, 0, 3, 1).reshape(3, co, 9)           # [ky][oc][3*kx + ch]
        w = torch.zeros(4, 3, co, 16, dtype=conv.weight.dtype, device=conv.weight.device)
        for r in range(4):
            sh = (6 * r + 5) % 8
            w[r, :, :, sh:sh + 9] = taps
        mod._w_c0 = w
    return w


def conv0(x, w_prep, bias, act="silu"):
    x = _cl(x)
    n, c, h, w = x.shape
    co = w_prep.shape[2]
    y = torch.empty((n, co, (h - 1) // 2 + 1, w // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op_conv0_f16(_st(x), _p(x), _p(w_prep), _p(bias), _p(y), n, h, w, co, ACT[act]))
    return y


STEM = _flag("STEM")                    # OSNet conv1 7x7/2 + bias + ReLU + max pool 3x3/2 in one launch


def stem_ok(x, conv) -> bool:
    n, c, h, w = x.shape
    return (STEM and c == 3 and w == 128 and h % 16 == 0 and conv.kernel_size == (7, 7) and conv.stride == (2, 2)
            and conv.padding == (3, 3) and conv.out_channels == 16 and conv.groups == 1 and conv.bias is not None)


def stem_weight(mod, conv):
    """[4][7][16][32]: for conv columns c = 4n + r (r = 0..3), per (ky, out channel) the 21 (kx, ch) taps in the order they lie
    in an NHWC input row, placed (6r + 7) % 8 halfs into a 32-wide K window of aligned 8-half blocks (k_osnet_stem)."""
    w = getattr(mod, "_w_stem", None)
    if w is None or w.device != conv.weight.device or w.dtype != conv.weight.dtype:
        taps = conv.weight.detach().permute(2, 0, 3, 1).reshape(7, 16, 21)          # [ky][oc][3*kx + ch]
        w = torch.zeros(4, 7, 16, 32, dtype=conv.weight.dtype, device=conv.weight.device)
        for r in range(4):
            sh = (6 * r + 7) % 8
            w[r, :, :, sh:sh + 21] = taps
        mod._w_stem = w
    return w


STEM_CONV1 = _flag("STEM_CONV1")        # the first OSBlock's conv1 (16 -> 16) on the stem's pooled tile


def osnet_stem(x, w_prep, bias, conv1=None):
    """conv1 = (w [16,16], b [16]): also returns relu(conv1x1(y, w) + b) (the first OSBlock's conv1) from the same launch."""
    x = _cl(x)
    n, c, h, w = x.shape
    y = torch.empty((n, 16, h // 4, w // 4), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    y1 = torch.empty_like(y, memory_format=torch.channels_last) if conv1 is not None else None
    w1, b1 = conv1 if conv1 is not None else (None, None)
    _ck(_lib.load().ss_op_osnet_stem_f16(_st(x), _p(x), _p(w_prep), _p(bias), _p(y), n, h, w, _p(w1), _p(b1), _p(y1)))
    return y if conv1 is None else (y, y1)


STREAMS = _flag("STREAMS")              # all four LightConv chains of an OSNet block in one launch (off: one launch per layer)


def streams_ok_dims(c, w) -> bool:
    return STREAMS and LIGHTCONV and c in (16, 24, 32) and w % 8 == 0 and 24 * (2 * w + 2) * ((c + 15) // 16 * 16) * 2 <= 65536


def streams_ok(x) -> bool:
    return streams_ok_dims(x.shape[1], x.shape[3])


def dwtab(w9, bias):
    """Depthwise taps w9 [L,9,C] (tap-major) + biases [L,C] of LightConv layers -> the operand table of the matrix-core depthwise
    (uint8 tensor; build once per set of weights)."""
    layers, _, c = w9.shape
    L = _lib.load()
    out = torch.empty(int(L.ss_op_dwtab_bytes(layers, c)), dtype=torch.uint8, device=w9.device)
    _ck(L.ss_op_dwtab_f16(_st(w9), _p(w9.contiguous()), _p(bias.contiguous()), layers, c, _p(out)))
    return out


def osnet_streams(x, w1, w9, bias, tab=None):
    """x [N,C,H,W] channels-last half; w1 [10,C,C], w9 [10,9,C], bias [10,C]: the layers of the 1-, 2-, 3- and 4-deep
    chains in that order; tab = dwtab(w9, bias) (built here when not given).  Returns the four chain outputs and the per-band
    channel sums psum [4,N,bands,C] (float; the library picks the band height)."""
    x = _cl(x)
    n, c, h, w = x.shape
    if tab is None:
        tab = dwtab(w9, bias)
    bands = _lib.load().ss_op_osnet_streams_bands(n, h, w, c)
    ys = [torch.empty_like(x, memory_format=torch.channels_last) for _ in range(4)]
    psum = torch.empty(4, n, bands, c, dtype=torch.float32, device=x.device)
    arr = (C.c_void_p * 4)(*[y.data_ptr() for y in ys])
    _ck(_lib.load().ss_op_osnet_streams_f16(_st(x), _p(x), _p(w1), _p(tab), arr, _p(psum), n, h, w, c))
    return ys, psum


def gate_apply(xs, psum, w1, b1, w2, b2):
    """sum_t x_t * sigmoid(fc2(relu(fc1(mean_hw(x_t))))) with the means taken from partial sums psum [T,N,parts,C]."""
    n, c, h, w = xs[0].shape
    out = torch.empty_like(xs[0], memory_format=torch.channels_last)
    arr = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    _ck(_lib.load().ss_op_gate_apply_f16(_st(out), arr, len(xs), _p(w1), _p(b1), _p(w2), _p(b2), _p(psum), psum.shape[2],
                                         1.0 / (h * w), _p(out), n, h * w, c, w1.shape[0]))
    return out


TAIL = _flag("TAIL")                    # gate + conv3 + shortcut + ReLU + the following 1x1 ConvBR (+ 2x2 average) in one launch


def tail_ok(mid, c2, n2, h, w, pool) -> bool:
    return (TAIL and (mid, c2, n2) in ((16, 64, 16), (16, 64, 64), (24, 96, 24), (24, 96, 96), (32, 128, 32), (32, 128, 128))
            and (h * w) % 128 == 0 and 128 % w == 0 and (not pool or ((128 // w) % 2 == 0 and w % 2 == 0)))


def tail_down_ok(c1, mid, c2, n2) -> bool:
    return TAIL and (c1, mid, c2, n2) in ((16, 16, 64, 16), (64, 24, 96, 24), (96, 32, 128, 32))


def osnet_tail(ys, psum, gate_w, w3, b3, idn, want_out, w4, b4, pool, down=None):
    """-> (o or None, o2): o = relu(conv3(sum_t ys[t]*gate_t) + b3 + idn), o2 = relu(conv4(o) + b4), 2x2-averaged when
    `pool`; w3 [C2, MID], w4 [N2, C2] (out, in).  down = (wd [C2, C1], bd): `idn` is the block input and the shortcut is its 1x1
    convolution, computed inside (tail_down_ok).  Bit-identical to (pointwise +) gate_apply + pointwise + pointwise (+ avgpool2)."""
    n, mid, h, w = ys[0].shape
    idn = _cl(idn)
    c2, n2 = w3.shape[0], w4.shape[0]
    c1, wd, bd = (idn.shape[1], down[0], down[1]) if down is not None else (0, None, None)       # down: shortcut = conv1x1(idn, wd) + bd
    out = torch.empty((n, c2, h, w), dtype=idn.dtype, device=idn.device, memory_format=torch.channels_last) if want_out else None
    oh, ow = (h // 2, w // 2) if pool else (h, w)
    out2 = torch.empty((n, n2, oh, ow), dtype=idn.dtype, device=idn.device, memory_format=torch.channels_last)
    arr = (C.c_void_p * 4)(*[y.data_ptr() for y in ys])
    gw1, gb1, gw2, gb2 = gate_w
    gates = torch.empty((n, 4, 32), dtype=torch.float32, device=idn.device)
    _ck(_lib.load().ss_op_osnet_tail_f16(_st(idn), arr, _p(psum), psum.shape[2], 1.0 / (h * w), _p(gw1), _p(gb1), _p(gw2), _p(gb2),
                                         gw1.shape[0], _p(gates), _p(w3), _p(b3), _p(idn), c1, _p(wd), _p(bd), _p(out), _p(w4), _p(b4),
                                         _p(out2), int(pool), n, h, w, mid, c2, n2))
    return out, out2


def gate_sum(xs, w1, b1, w2, b2):
    xs = [_cl(x) for x in xs]
    n, c, h, w = xs[0].shape
    out = torch.empty_like(xs[0], memory_format=torch.channels_last)
    means = torch.empty(len(xs) * n * c, dtype=torch.float32, device=out.device)
    arr = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    _ck(_lib.load().ss_op_gate_sum_f16(_st(out), arr, len(xs), _p(w1), _p(b1), _p(w2), _p(b2), _p(means), _p(out),
                                       n, h * w, c, w1.shape[0]))
    return out


def maxpool(x, k, stride, pad):
    x = _cl(x)
    n, c, h, w = x.shape
    if c % 8:
        return torch.nn.functional.max_pool2d(x, k, stride, pad)
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op_maxpool_f16(_st(x), _p(x), _p(y), n, h, w, c, k, stride, pad))
    return y


GLUE = _flag("GLUE")                    # upsample + concat and the SPPF pooling pyramid as single launches


def upcat_ok(lo, hi) -> bool:
    return (GLUE and usable(lo) and usable(hi) and lo.shape[1] % 8 == 0 and hi.shape[1] % 8 == 0 and hi.shape[0] == lo.shape[0]
            and hi.shape[2] == 2 * lo.shape[2] and hi.shape[3] == 2 * lo.shape[3])


def upcat(lo, hi, lo_first=True):
    """torch.cat((F.interpolate(lo, scale_factor=2, mode="nearest"), hi), 1) (or hi first) in one launch."""
    lo, hi = _cl(lo), _cl(hi)
    b, c1, h, w = lo.shape
    c2 = hi.shape[1]
    out = torch.empty((b, c1 + c2, 2 * h, 2 * w), dtype=lo.dtype, device=lo.device, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op_upcat_f16(_st(lo), _p(lo), _p(hi), _p(out), b, h, w, c1, c2, int(lo_first)))
    return out


def sppf_pools_ok(x) -> bool:
    return GLUE and usable(x) and x.shape[1] % 8 == 0 and x.shape[2] * x.shape[3] <= 1024


def sppf_pools(x):
    """cat(x, m(x), m(m(x)), m(m(m(x)))) with m = max_pool2d(5, 1, 2), one launch."""
    x = _cl(x)
    b, c, h, w = x.shape
    out = torch.empty((b, 4 * c, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op_sppf_pools_f16(_st(x), _p(x), _p(out), b, h, w, c))
    return out


OSHEAD = _flag("OSHEAD")                # OSNet's average pool + fully connected layer + ReLU in one launch


def osnet_head_ok(x, fc) -> bool:
    return OSHEAD and usable(x) and x.shape[1] == 128 and fc.in_features == 128 and fc.bias is not None and x.is_contiguous(memory_format=torch.channels_last)


def osnet_head(x, fc):
    """relu(fc(mean_hw(x))): x [N, 128, H, W] channels-last half -> [N, F] half."""
    n, c, h, w = x.shape
    # rows of images past the valid count of a packed batch are not computed: zeros, not uninitialised memory, for whoever reads the full tensor
    out = torch.zeros((n, fc.out_features), dtype=x.dtype, device=x.device) if _nv_active() else torch.empty((n, fc.out_features), dtype=x.dtype, device=x.device)
    _ck(_lib.load().ss_op_osnet_head_f16(_st(x), _p(x), _p(fc.weight), _p(fc.bias), _p(out), n, h * w, c, fc.out_features))
    return out


PSA = _flag("PSA")                      # v11's C2PSA attention (QK^T, softmax, PV, + positional term) in one launch


def psa_ok(x, heads, key_dim, head_dim) -> bool:
    return PSA and usable(x) and key_dim == 32 and head_dim == 64 and x.shape[2] * x.shape[3] <= 256 and heads <= 64


def psa_attention(qkv, pe, heads, scale):
    """qkv [B, heads*128, H, W] channels-last half (per head q 32 | k 32 | v 64), pe [B, heads*64, H, W] or None ->
    softmax(scale q^T k) applied to v, + pe: [B, heads*64, H, W] channels-last."""
    qkv = _cl(qkv)
    b, _, h, w = qkv.shape
    out = torch.empty((b, heads * 64, h, w), dtype=qkv.dtype, device=qkv.device, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op_psa_attention_f16(_st(qkv), _p(qkv), _p(_cl(pe) if pe is not None else None), _p(out), b, h * w, heads, float(scale)))
    return out


def avgpool2(x):
    """2x2 / stride 2 average pooling on a channels-last half tensor (even H, W; C % 8 == 0), else torch's."""
    x = _cl(x)
    n, c, h, w = x.shape
    if c % 8 or h % 2 or w % 2:
        return torch.nn.functional.avg_pool2d(x, 2, 2)
    y = torch.empty((n, c, h // 2, w // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _ck(_lib.load().ss_op_avgpool2_f16(_st(x), _p(x), _p(y), n, h, w, c))
    return y


def conv1x1(x, w_t, bias=None):
    """Pointwise convolution as one GEMM on the NHWC view: [N*H*W, Cin] @ [Cin, Cout] (+ bias in the epilogue).
    MIOpen's implicit-GEMM kernels for these shapes need a separate zero-fill launch (split-K); a plain GEMM does not."""
    x = _cl(x)
    n, c, h, w = x.shape
    x2 = x.permute(0, 2, 3, 1).reshape(-1, c)
    y2 = torch.addmm(bias, x2, w_t) if bias is not None else torch.mm(x2, w_t)
    return y2.view(n, h, w, -1).permute(0, 3, 1, 2)


def is_pointwise(conv) -> bool:
    return conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1 and conv.padding == (0, 0)


def weight_t(mod, conv):
    """[Cin, Cout] contiguous copy of a 1x1 conv weight, cached on the module (inference: weights are static)."""
    wt = getattr(mod, "_w_t", None)
    if wt is None or wt.device != conv.weight.device or wt.dtype != conv.weight.dtype:
        wt = conv.weight.detach().reshape(conv.weight.shape[0], -1).t().contiguous()
        mod._w_t = wt
    return wt
