"""Detector and ReID backbones (rows a2 / a5 of SURVEY.md §8a) as plain PyTorch modules.

These are the dense-conv part of the path (north_star: "PyTorch-ROCm for the YOLOv5/7/8 detector and
OSNet-x0.25 ReID backbones").  The modules are plain PyTorch; on half channels-last CUDA tensors their 1x1 / 3x3
convolutions, the OSNet stem and LightConv chains and the v8 head decode dispatch to the hand-written gfx950
kernels of csrc/ss_ops.hip (through fused.py), everything else (and the CPU / fp32 path used as the oracle-side
baseline) to PyTorch-ROCm's libraries.  No weights exist offline, so they are seeded random-init
networks of the published architectures; their job is to load the GPU with the real layer shapes.  Conv+BN
pairs are built in their fused inference form (conv with bias).

They stand behind `YOLO(weights)` (/root/reference/yolo_multi_model.py:14-17) and the forward pass
inside model.track / model.predict (:41, :173).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused, fused32


# --------------------------------------------------------------------------------------------------
# YOLO building blocks
# --------------------------------------------------------------------------------------------------
class Conv(nn.Module):
    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2 if p is None else p, groups=g, bias=True)
        self.act = nn.SiLU(inplace=True) if act else nn.Identity()

    def forward(self, x, out=None, res=None):
        """out / res (fp32 kernels only): write into this channels-last tensor / channel slice; add `res` after the activation."""
        act = "silu" if isinstance(self.act, nn.SiLU) else ("none" if isinstance(self.act, nn.Identity) else None)
        if act is not None and fused32.conv_ok(x, self.conv):       # fp32: one implicit-GEMM launch, bias + SiLU (+ shortcut) in its epilogue
            return fused32.conv(x, self, self.conv, act, out=out, res=res)
        if act is not None and fused32.conv0_ok(x, self.conv):
            return fused32.conv0(x, self, self.conv, act, out=out)
        if out is not None or res is not None:
            y = self.act(self.conv(x))
            y = y if res is None else y + res
            if out is None:
                return y
            out.copy_(y)
            return out
        if fused.usable(x):
            c = self.conv
            act = "silu" if isinstance(self.act, nn.SiLU) else "none"
            if fused.pointwise_ok(c):        # 1x1: own MFMA kernel, bias + SiLU in its epilogue (one launch)
                return fused.pointwise(x, fused.weight_nk(self, c), c.bias, act)
            if fused.conv3x3_ok(c):          # 3x3: the same kernel as an implicit GEMM
                return fused.conv3x3(x, fused.weight_n9k(self, c), c.bias, c.stride[0], act)
            if fused.conv0_ok(x, c):         # the first convolution (3 input channels, stride 2)
                return fused.conv0(x, fused.conv0_weight(self, c), c.bias, act)
            if fused.dw3x3_ok(c):            # depthwise 3x3 (v11 head): MIOpen has only its naive kernel for these (55 us a launch)
                return fused.dwconv3x3(x, fused.weight_dw9(self, c), c.bias, act)
            # k x k: conv without bias (MIOpen) + one fused bias+SiLU pass
            y = F.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
            return fused.bias_act_(y, c.bias, act)
        return self.act(self.conv(x))


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1, self.cv2 = Conv(c1, c_, k[0]), Conv(c_, c2, k[1])
        self.add = shortcut and c1 == c2

    def forward(self, x, out=None):
        if out is not None or (self.add and fused32.conv_ok(x, self.cv1.conv)):      # fp32 kernels: the shortcut in the second convolution's epilogue
            return self.cv2(self.cv1(x), out=out, res=x if self.add else None)
        return x + self.cv2(self.cv1(x)) if self.add else self.cv2(self.cv1(x))


class C2f(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=False):
        super().__init__()
        self.c = c2 // 2
        self.cv1 = Conv(c1, 2 * self.c, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, e=1.0) for _ in range(n))

    def placed_ok(self, x) -> bool:
        return fused.usable(x) and fused.place_ok(self.c, (2 + len(self.m)) * self.c) and all(type(m) is Bottleneck for m in self.m)

    def forward(self, x, also=None, c_off=0):
        """also / c_off: the block's output is ALSO written into channels [c_off, c_off + c2) of the channels-last tensor `also` (a
        later concat's buffer) by the last 1x1's epilogue — only with the placed path (placed_ok) and a pointwise-capable cv2."""
        if self.placed_ok(x):
            return self._forward_placed(x, also, c_off)
        if fused32.conv_ok(x, self.cv1.conv) and all(type(m) is Bottleneck for m in self.m):
            # fp32 kernels: every producer writes its channel slice of the concat buffer, every consumer reads a slice — no chunk / add / cat pass
            c, n = self.c, len(self.m)
            B, _, H, W = x.shape
            cat = torch.empty((B, (2 + n) * c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            self.cv1(x, out=cat[:, :2 * c])
            for i, m in enumerate(self.m):
                m(cat[:, (1 + i) * c:(2 + i) * c], out=cat[:, (2 + i) * c:(3 + i) * c])
            if also is not None:         # the last 1x1 writes its slice of a later concat's buffer; the block's output is that slice
                return self.cv2(cat, out=also[:, c_off:c_off + self.cv2.conv.out_channels])
            return self.cv2(cat)
        y = list(self.cv1(x).chunk(2, 1))
        for m in self.m:
            y.append(m(y[-1]))
        y = self.cv2(torch.cat(y, 1))
        if also is not None:
            also[:, c_off:c_off + y.shape[1]] = y
        return y

    def _forward_placed(self, x, also=None, c_off=0):
        """Same arithmetic, no chunk / add / cat launches: every producer's bias+SiLU epilogue writes straight into
        its channel slice of the concat buffer (and a dense copy of the half the next 3x3 conv reads)."""
        c, n = self.c, len(self.m)
        B, _, H, W = x.shape
        cat = torch.empty((B, (2 + n) * c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        dense = lambda: torch.empty((B, c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        cv = self.cv1.conv
        cur = dense()
        if fused.pointwise_ok(cv):
            fused.pointwise(x, fused.weight_nk(self.cv1, cv), cv.bias, "silu", out=cat, c_off=0, out2=cur, c0=c)
        else:
            fused.bias_act_place(F.conv2d(x, cv.weight, None, cv.stride, cv.padding), cv.bias, "silu", cat, 0, out2=cur, c0=c)
        for i, m in enumerate(self.m):
            cv = m.cv2.conv
            nxt = dense() if i + 1 < n else None
            if fused.bottleneck_ok(m):       # both convolutions + shortcut + placement in one launch (csrc k_bneck)
                fused.bottleneck(cur, m, cat, (2 + i) * c, out2=nxt)
            elif fused.conv3x3_ok(cv):       # conv + bias + SiLU + shortcut + placement in one launch
                fused.conv3x3(m.cv1(cur), fused.weight_n9k(m.cv2, cv), cv.bias, cv.stride[0], "silu", res=cur if m.add else None,
                              res_after=True, out=cat, c_off=(2 + i) * c, out2=nxt, c0=0)
            else:
                t = F.conv2d(m.cv1(cur), cv.weight, None, cv.stride, cv.padding)
                fused.bias_act_place(t, cv.bias, "silu", cat, (2 + i) * c, res=cur if m.add else None, res_after=True,
                                     out2=nxt, c0=0)
            cur = nxt
        cv = self.cv2.conv
        if also is not None and fused.pointwise_ok(cv) and isinstance(self.cv2.act, nn.SiLU):
            y = torch.empty((B, cv.out_channels, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            fused.pointwise(cat, fused.weight_nk(self.cv2, cv), cv.bias, "silu", out=also, c_off=c_off, out2=y, c0=0)
            return y
        y = self.cv2(cat)
        if also is not None:
            also[:, c_off:c_off + y.shape[1]] = y
        return y


def _w_pair(mod, a, b):
    """Two 1x1 convolutions on the same input as one: (weights [Ca + Cb, Cin], bias), cached on `mod`."""
    w = getattr(mod, "_w_pair_", None)
    if w is None or w[0].device != a.weight.device or w[0].dtype != a.weight.dtype:
        w = mod._w_pair_ = (torch.cat((a.weight.detach().reshape(a.out_channels, -1), b.weight.detach().reshape(b.out_channels, -1)), 0).contiguous(),
                            torch.cat((a.bias.detach(), b.bias.detach())).contiguous())
    return w


class C3(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=True, k=(1, 3)):
        super().__init__()
        c_ = c2 // 2
        self.cv1, self.cv2, self.cv3 = Conv(c1, c_, 1), Conv(c1, c_, 1), Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, k=k, e=1.0) for _ in range(n)))

    def forward(self, x):
        if fused.C3K2 and fused.usable(x) and self.placed_ok():
            B, _, H, W = x.shape
            out = torch.empty((B, self.cv3.conv.out_channels, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            self.forward_placed(x, out, 0)
            return out
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))

    def placed_ok(self) -> bool:
        a, b, c3 = self.cv1.conv, self.cv2.conv, self.cv3.conv
        silu = all(isinstance(cv.act, nn.SiLU) for cv in (self.cv1, self.cv2, self.cv3))
        inner = all(type(m) is Bottleneck and isinstance(m.cv1.act, nn.SiLU) and isinstance(m.cv2.act, nn.SiLU) and
                    (fused.bottleneck_ok(m) or ((fused.conv3x3_ok(m.cv1.conv) or fused.pointwise_ok(m.cv1.conv)) and fused.conv3x3_ok(m.cv2.conv)
                                                and m.cv2.conv.stride == (1, 1)))
                    for m in self.m)
        return (silu and inner and len(self.m) >= 1 and fused.pointwise_ok(a) and fused.pointwise_ok(b) and fused.pointwise_ok(c3)
                and a.in_channels == b.in_channels and fused.place_ok(a.out_channels, 2 * a.out_channels))

    def _w12(self):
        """cv1 and cv2 read the same input: one 1x1 with both sets of output rows ([cv1 | cv2], the concat order of forward)."""
        return _w_pair(self, self.cv1.conv, self.cv2.conv)

    def forward_placed(self, x, out, c_off, out2=None):
        """The same arithmetic with every producer writing where its consumer reads: [cv1 | cv2] in one pointwise launch into the
        inner concat buffer (cv1's half mirrored densely for the first bottleneck), the bottlenecks' results (shortcut in the second
        convolution's epilogue) into cv1's slot, cv3's output into channels [c_off, c_off + c2) of `out` (+ the dense copy `out2`)."""
        c_ = self.cv1.conv.out_channels
        B, _, H, W = x.shape
        inner = torch.empty((B, 2 * c_, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        dense = lambda: torch.empty((B, c_, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        cur = dense()
        w12, b12 = self._w12()
        fused.pointwise(x, w12, b12, "silu", out=inner, c_off=0, out2=cur, c0=0)
        for j, m in enumerate(self.m):
            nxt = dense() if j + 1 < len(self.m) else None
            if fused.bottleneck_ok(m):
                fused.bottleneck(cur, m, inner, 0, out2=nxt)
            else:
                cv = m.cv2.conv
                fused.conv3x3(m.cv1(cur), fused.weight_n9k(m.cv2, cv), cv.bias, 1, "silu", res=cur if m.add else None, res_after=True,
                              out=inner, c_off=0, out2=nxt, c0=0)
            cur = nxt
        c3 = self.cv3.conv
        fused.pointwise(inner, fused.weight_nk(self.cv3, c3), c3.bias, "silu", out=out, c_off=c_off, out2=out2, c0=0)


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1, self.cv2 = Conv(c1, c_, 1), Conv(c_ * 4, c2, 1)
        self.m = nn.MaxPool2d(k, 1, k // 2)

    def forward(self, x):
        y = [self.cv1(x)]
        if fused.sppf_pools_ok(y[0]):                        # the three pools and the concat in one launch
            return self.cv2(fused.sppf_pools(y[0]))
        if fused32.DET and fused32.usable(y[0]) and self.m.kernel_size == 5 and y[0].shape[1] % 4 == 0 and fused32._nhwc_view(y[0]) is not None:
            return self.cv2(fused32.sppf_pools(y[0]))
        pool = (lambda t: fused.maxpool(t, 5, 1, 2)) if fused.usable(x) else self.m
        y.extend(pool(y[-1]) for _ in range(3))
        return self.cv2(torch.cat(y, 1))


class Proto(nn.Module):
    """Mask prototypes of the segmentation head (Ultralytics `Proto`): 3x3 -> 2x transposed convolution -> 3x3 -> 1x1 on the
    stride-8 map, [B, nm, H/4, W/4]."""

    def __init__(self, c1, c_=256, c2=32):
        super().__init__()
        self.cv1, self.upsample = Conv(c1, c_, 3), nn.ConvTranspose2d(c_, c_, 2, 2, 0, bias=True)
        self.cv2, self.cv3 = Conv(c_, c_, 3), Conv(c_, c2, 1)

    def forward(self, x):
        return self.cv3(self.cv2(self.upsample(self.cv1(x))))


class Detect(nn.Module):
    """Anchor-free v8 head with DFL decode.  Output [B, 4+nc(+nk | +nm), A] (xywh in input pixels); with nm > 0 (the
    segmentation head, Ultralytics `Segment`: /root/reference/yolo_multi_model.py:14 names 'yolov8n-seg.pt') the rows end in
    the nm raw mask coefficients and forward returns (rows, prototypes [B, nm, H/4, W/4])."""

    def __init__(self, nc, ch, nk=0, nm=0, npr=256):
        super().__init__()
        if nk and nm:
            raise ValueError("a head carries keypoints or mask coefficients, not both")
        self.nc, self.nk, self.nm, self.reg_max = nc, nk, nm, 16
        c2, c3 = max(16, ch[0] // 4, 64), max(ch[0], min(nc, 100))
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 64, 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, nc, 1)) for x in ch)
        if nk or nm:
            c4 = max(ch[0] // 4, nk or nm)
            self.cv4 = nn.ModuleList(nn.Sequential(Conv(x, c4, 3), Conv(c4, c4, 3), nn.Conv2d(c4, nk or nm, 1)) for x in ch)
        if nm:
            self.proto = Proto(ch[0], npr, nm)
        self.strides = (8, 16, 32)
        self.register_buffer("proj", torch.arange(16, dtype=torch.float32).view(1, 1, 16, 1), persistent=False)
        self._anchors = None

    def _make_anchors(self, feats):
        pts, strd = [], []
        for f, s in zip(feats, self.strides):
            h, w = f.shape[2:]
            sx = torch.arange(w, device=f.device, dtype=f.dtype) + 0.5
            sy = torch.arange(h, device=f.device, dtype=f.dtype) + 0.5
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            pts.append(torch.stack((xx, yy), 0).view(2, -1))
            strd.append(torch.full((1, h * w), float(s), device=f.device, dtype=f.dtype))
        return torch.cat(pts, 1).unsqueeze(0), torch.cat(strd, 1).unsqueeze(0)

    def _zero_bias(self, like):
        z = getattr(self, "_zeros", None)
        if z is None or z.device != like.device:
            z = self._zeros = torch.zeros(max(64, self.nc), dtype=like.dtype, device=like.device)
        return [z] * 3

    @staticmethod
    def _last_ok(conv):
        return fused.POINTWISE and fused.is_pointwise(conv) and conv.in_channels % 8 == 0 and conv.bias is not None

    @staticmethod
    def _last_wb(conv):
        """(w_nk, bias) of a branch's last 1x1, its rows zero-padded to a multiple of 8 when needed (a one-class head)."""
        return (fused.weight_nk(conv, conv), conv.bias) if conv.out_channels % 8 == 0 else fused.padded_last(conv, conv)

    def _ext_ok(self):
        """The third branch (keypoints / mask coefficients) can run on the convolution kernels with zero-padded widths."""
        return fused.HEAD_EXT and all(s[0].conv.in_channels % 8 == 0 and isinstance(s[0].act, nn.SiLU) and s[0].conv.kernel_size == (3, 3)
                                      and s[0].conv.out_channels <= 80 for s in self.cv4)

    def _decode(self, feats, box, cls, bb, cb, ext=None):
        """box / cls / ext: the branches' last-layer outputs per level -> the prediction tensor (+ prototypes for a mask head)."""
        n_ext = (self.nk or self.nm) if ext is not None else 0
        pred = fused.v8_decode(box, cls, bb, cb, self.strides, self.nc, ext, n_ext, 1 if self.nk else 0)
        return (pred, self.proto(feats[0])) if self.nm else pred

    def forward(self, feats):
        B = feats[0].shape[0]
        ext_head = bool(self.nk or self.nm)
        if fused.usable(feats[0]) and (not ext_head or self._ext_ok()):    # branch tensors -> [B, 4+nc(+nk|nm), A] float in one launch
            seqs = list(self.cv2) + list(self.cv3)
            pads = [fused.padded_branch(self.cv4[i], self.cv4[i]) for i in range(len(feats))] if ext_head else []
            if not ext_head and len(feats) == 3 and all(fused.head_level_ok(f, self.cv2[i], self.cv3[i]) for i, f in enumerate(feats)):
                # a level's two branches, three layers each, in ONE launch with the intermediates in LDS (csrc k_head): 3 launches
                # instead of the 3 grouped ones per depth, without the round trips of the 64- / 80-channel intermediates
                t = [fused.head_level(f, self.cv2[i], self.cv3[i]) for i, f in enumerate(feats)]
                z = self._zero_bias(feats[0])
                return self._decode(feats, [a for a, _ in t], [b for _, b in t], z, z)
            if (fused.GROUP and len(feats) == 3 and all(self._last_ok(s[2]) and fused.conv3x3_ok(s[0].conv) and
                                                         fused.conv3x3_ok(s[1].conv) and isinstance(s[0].act, nn.SiLU) and
                                                         s[2].out_channels <= 80 and s[0].conv.out_channels <= 80 for s in seqs)):
                # the branches are independent: one grouped launch per depth instead of 18 (27 with a third branch) small ones
                xs = list(feats) * (3 if ext_head else 2)
                w = [[(fused.weight_n9k(s[d], s[d].conv), s[d].conv.bias) for s in seqs] + [p[d] for p in pads] for d in (0, 1)]
                w.append([self._last_wb(s[2]) for s in seqs] + [p[2] for p in pads])
                t = fused.conv_group([(x, wd, b, 3, 1, "silu") for (wd, b), x in zip(w[0], xs)])
                t = fused.conv_group([(x, wd, b, 3, 1, "silu") for (wd, b), x in zip(w[1], t)])
                t = fused.conv_group([(x, wd, b, 1, 1, "none") for (wd, b), x in zip(w[2], t)])
                z = self._zero_bias(feats[0])
                return self._decode(feats, t[:3], t[3:6], z, z, t[6:] if ext_head else None)
            ext = None
            if ext_head:                     # the third branch layer by layer on the convolution kernels (padded widths)
                ext = []
                for p, f in zip(pads, feats):
                    u = fused.conv3x3(f, p[0][0], p[0][1], 1, "silu")
                    u = fused.conv3x3(u, p[1][0], p[1][1], 1, "silu")
                    ext.append(fused.pointwise(u, p[2][0], p[2][1]))
            if all(self._last_ok(s[2]) for s in list(self.cv2) + list(self.cv3)):
                # final 1x1 of every branch on the pointwise kernel (bias in its epilogue; the decode adds zeros)
                last = lambda seq, f: fused.pointwise(seq[1](seq[0](f)), *self._last_wb(seq[2]))
                bb = cb = self._zero_bias(feats[0])
            else:
                last = lambda seq, f: F.conv2d(seq[1](seq[0](f)), seq[2].weight)
                bb, cb = [s[2].bias for s in self.cv2], [s[2].bias for s in self.cv3]
            return self._decode(feats, [last(self.cv2[i], f) for i, f in enumerate(feats)],
                                [last(self.cv3[i], f) for i, f in enumerate(feats)], bb, cb, ext)
        return self._forward_torch(feats)

    @staticmethod
    def _branch(seq, f):
        """A head branch (3x3, 3x3, plain 1x1); the last layer on the fp32 kernel too when its widths allow."""
        t = seq[1](seq[0](f))
        return fused32.conv(t, seq[2], seq[2], "none") if fused32.conv_ok(t, seq[2]) else seq[2](t)

    def _forward_torch(self, feats):
        B = feats[0].shape[0]
        if fused32.DET and fused32.usable(feats[0]) and len(feats) == 3 and all(fused32.conv_ok(f, self.cv2[i][0].conv) for i, f in enumerate(feats)):
            # fp32 kernels: the branches on k32_conv, the decode (DFL, dist2bbox, sigmoid, level concat, keypoints / coefficients) in one launch
            box = [self._branch(self.cv2[i], f) for i, f in enumerate(feats)]
            cls = [self._branch(self.cv3[i], f) for i, f in enumerate(feats)]
            ext = [self._branch(self.cv4[i], f) for i, f in enumerate(feats)] if (self.nk or self.nm) else None
            pred = fused32.v8_decode(box, cls, self.strides, self.nc, ext, self.nk or self.nm, 1 if self.nk else 0)
            return (pred, self.proto(feats[0])) if self.nm else pred
        box = torch.cat([self._branch(self.cv2[i], f).reshape(B, 64, -1) for i, f in enumerate(feats)], 2)
        cls = torch.cat([self._branch(self.cv3[i], f).reshape(B, self.nc, -1) for i, f in enumerate(feats)], 2)
        if self._anchors is None or self._anchors[0].shape[-1] != box.shape[-1] or self._anchors[0].dtype != box.dtype or self._anchors[0].device != box.device:
            self._anchors = self._make_anchors(feats)
        anchors, strides = self._anchors
        d = box.view(B, 4, 16, -1).softmax(2)
        d = (d * self.proj.to(d.dtype)).sum(2)                       # DFL expectation, [B,4,A]
        lt, rb = d[:, :2], d[:, 2:]
        x1y1, x2y2 = anchors - lt, anchors + rb
        out = [torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides, cls.sigmoid()]
        if self.nk:
            # keypoint decode (Ultralytics Pose.kpts_decode): xy = (2 k + anchor - 0.5) * stride, visibility = sigmoid
            k = torch.cat([self.cv4[i](f).reshape(B, self.nk, -1) for i, f in enumerate(feats)], 2).view(B, self.nk // 3, 3, -1)
            xy = (k[:, :, :2] * 2.0 + (anchors - 0.5).unsqueeze(1)) * strides.unsqueeze(1)
            out.append(torch.cat((xy, k[:, :, 2:].sigmoid()), 2).view(B, self.nk, -1))
        if self.nm:
            # mask coefficients stay raw (Ultralytics Segment: `torch.cat([x, mc], 1), p`); the masks are assembled from them
            # and the prototypes after NMS (yolo.Masks)
            out.append(torch.cat([self.cv4[i](f).reshape(B, self.nm, -1) for i, f in enumerate(feats)], 2))
            return torch.cat(out, 1), self.proto(feats[0])
        return torch.cat(out, 1)


_V8_SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768)}


class YOLOv8(nn.Module):
    def __init__(self, scale="n", nc=80, nk=0, nm=0):
        super().__init__()
        d, w, mc = _V8_SCALES[scale]
        c = lambda x: int(math.ceil(min(x, mc) * w / 8) * 8)
        n = lambda x: max(round(x * d), 1)
        self.b0, self.b1 = Conv(3, c(64), 3, 2), Conv(c(64), c(128), 3, 2)
        self.b2 = C2f(c(128), c(128), n(3), True)
        self.b3, self.b4 = Conv(c(128), c(256), 3, 2), C2f(c(256), c(256), n(6), True)
        self.b5, self.b6 = Conv(c(256), c(512), 3, 2), C2f(c(512), c(512), n(6), True)
        self.b7, self.b8 = Conv(c(512), c(1024), 3, 2), C2f(c(1024), c(1024), n(3), True)
        self.b9 = SPPF(c(1024), c(1024))
        self.h12 = C2f(c(1024) + c(512), c(512), n(3))
        self.h15 = C2f(c(512) + c(256), c(256), n(3))
        self.h16, self.h18 = Conv(c(256), c(256), 3, 2), C2f(c(256) + c(512), c(512), n(3))
        self.h19, self.h21 = Conv(c(512), c(512), 3, 2), C2f(c(512) + c(1024), c(1024), n(3))
        self.detect = Detect(nc, (c(256), c(512), c(1024)), nk, nm, c(256))
        self.nc, self.nk, self.nm = nc, nk, nm

    def forward_backbone(self, x):
        p3 = self.b4(self.b3(self.b2(self.b1(self.b0(x)))))
        p4 = self.b6(self.b5(p3))
        p5 = self.b9(self.b8(self.b7(p4)))
        return p3, p4, p5

    def forward_head(self, p3, p4, p5):
        c16, c19 = self.h16.conv, self.h19.conv
        if (fused.C3K2 and fused.usable(p3) and fused.conv3x3_ok(c16) and fused.conv3x3_ok(c19) and isinstance(self.h16.act, nn.SiLU)
                and isinstance(self.h19.act, nn.SiLU) and c16.out_channels % 8 == 0 and c19.out_channels % 8 == 0):
            # the two down-path concats as placement: h12 lands in h18's input from its own last 1x1, the stride-2 convolutions write
            # their slices; only p5 (made in the backbone half, which may be another captured graph) is copied
            B = p3.shape[0]
            x18 = torch.empty((B, c16.out_channels + self.h12.cv2.conv.out_channels, p4.shape[2], p4.shape[3]), dtype=p3.dtype, device=p3.device,
                              memory_format=torch.channels_last)
            h12 = self.h12(_upcat(p5, p4), also=x18, c_off=c16.out_channels)
            h15 = self.h15(_upcat(h12, p3))
            fused.conv3x3(h15, fused.weight_n9k(self.h16, c16), c16.bias, 2, "silu", out=x18, c_off=0)
            h18 = self.h18(x18)
            x21 = torch.empty((B, c19.out_channels + p5.shape[1], p5.shape[2], p5.shape[3]), dtype=p3.dtype, device=p3.device,
                              memory_format=torch.channels_last)
            fused.conv3x3(h18, fused.weight_n9k(self.h19, c19), c19.bias, 2, "silu", out=x21, c_off=0)
            x21[:, c19.out_channels:] = p5
            return self.detect([h15, h18, self.h21(x21)])
        if fused32.conv_ok(p3, c16) and fused32.conv_ok(p4, c19):
            # fp32 kernels: the two down-path concats as placement — h12 lands in h18's input from its own last 1x1, the stride-2
            # convolutions write their slices; only p5 is copied
            B = p3.shape[0]
            new = lambda c, like: torch.empty((B, c, like.shape[2], like.shape[3]), dtype=p3.dtype, device=p3.device, memory_format=torch.channels_last)
            x18 = new(c16.out_channels + self.h12.cv2.conv.out_channels, p4)
            h12 = self.h12(_upcat(p5, p4), also=x18, c_off=c16.out_channels)
            h15 = self.h15(_upcat(h12, p3))
            self.h16(h15, out=x18[:, :c16.out_channels])
            h18 = self.h18(x18)
            x21 = new(c19.out_channels + p5.shape[1], p5)
            self.h19(h18, out=x21[:, :c19.out_channels])
            x21[:, c19.out_channels:] = p5
            return self.detect([h15, h18, self.h21(x21)])
        h12 = self.h12(_upcat(p5, p4))
        h15 = self.h15(_upcat(h12, p3))
        h18 = self.h18(torch.cat((self.h16(h15), h12), 1))
        h21 = self.h21(torch.cat((self.h19(h18), p5), 1))
        return self.detect([h15, h18, h21])

    def forward(self, x):
        self._own32 = bool(fused32.DET and fused32.usable(x))       # fp32 CUDA input: the convolutions run on csrc k32_conv / k32_conv0
        return self.forward_head(*self.forward_backbone(x))


# --------------------------------------------------------------------------------------------------
# YOLO11 (the reference's default weights file: `YOLO("yolo11n-pose.pt")`, /root/reference/yolo_multi_model.py:17)
# --------------------------------------------------------------------------------------------------
# Block structure restated from the published Ultralytics yolo11.yaml / modules (C3k2, C3k, C2PSA, PSABlock, Attention, the
# depthwise-separable class branch of the v11 Detect head); nothing of it is in the reference snapshot, and no weights exist
# offline, so the only check available here is structural (strict state_dict key / shape match through
# `convert_ultralytics_state_dict`).  Module attribute names follow the Ultralytics layer indices (b0..b10, h13.., detect = 23).
class C3k(C3):
    """C3 with n Bottlenecks of two k x k convolutions (e = 1.0 inside)."""

    def __init__(self, c1, c2, n=2, shortcut=True, k=3):
        super().__init__(c1, c2, n, shortcut, k=(k, k))


class C3k2(nn.Module):
    """C2f whose inner blocks are C3k (c3k=True) or plain Bottlenecks with e = 0.5; hidden width c = int(c2 * e)."""

    def __init__(self, c1, c2, n=1, c3k=False, e=0.5, shortcut=True):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(C3k(self.c, self.c, 2, shortcut) if c3k else Bottleneck(self.c, self.c, shortcut) for _ in range(n))

    def _placed_ok(self, x) -> bool:
        def inner_ok(m):
            if type(m) is C3k:
                return m.placed_ok()
            return (type(m) is Bottleneck and isinstance(m.cv1.act, nn.SiLU) and isinstance(m.cv2.act, nn.SiLU) and
                    fused.conv3x3_ok(m.cv1.conv) and fused.conv3x3_ok(m.cv2.conv) and m.cv2.conv.stride == (1, 1))
        return (fused.C3K2 and fused.usable(x) and fused.place_ok(self.c, (2 + len(self.m)) * self.c) and fused.pointwise_ok(self.cv1.conv)
                and isinstance(self.cv1.act, nn.SiLU) and all(inner_ok(m) for m in self.m))

    def forward(self, x):
        if self._placed_ok(x):
            return self._forward_placed(x)
        y = list(self.cv1(x).chunk(2, 1))
        for m in self.m:
            y.append(m(y[-1]))
        return self.cv2(torch.cat(y, 1))

    def _forward_placed(self, x):
        """C2f._forward_placed for v11's inner blocks: no chunk / cat / add launches; a Bottleneck's second convolution adds the
        shortcut and writes its slice of the concat buffer, a C3k block places its output itself."""
        c, n = self.c, len(self.m)
        B, _, H, W = x.shape
        cat = torch.empty((B, (2 + n) * c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        dense = lambda: torch.empty((B, c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        cv = self.cv1.conv
        cur = dense()
        fused.pointwise(x, fused.weight_nk(self.cv1, cv), cv.bias, "silu", out=cat, c_off=0, out2=cur, c0=c)
        for i, m in enumerate(self.m):
            nxt = dense() if i + 1 < n else None
            if type(m) is C3k:
                m.forward_placed(cur, cat, (2 + i) * c, nxt)
            elif fused.bottleneck_padded_ok(m):      # c -> c/2 -> c: one launch of the equal-width kernel on zero-padded weights
                fused.bottleneck_padded(cur, m, cat, (2 + i) * c, out2=nxt)
            else:
                cv = m.cv2.conv
                fused.conv3x3(m.cv1(cur), fused.weight_n9k(m.cv2, cv), cv.bias, 1, "silu", res=cur if m.add else None, res_after=True,
                              out=cat, c_off=(2 + i) * c, out2=nxt, c0=0)
            cur = nxt
        return self.cv2(cat)


class Attention(nn.Module):
    """Multi-head self-attention over the H*W positions of a feature map (key_dim = head_dim * attn_ratio) + a depthwise 3x3
    positional term on v."""

    def __init__(self, dim, num_heads=8, attn_ratio=0.5):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.key_dim = int(self.head_dim * attn_ratio)
        self.scale = self.key_dim ** -0.5
        h = dim + self.key_dim * num_heads * 2
        self.qkv, self.proj = Conv(dim, h, 1, act=False), Conv(dim, dim, 1, act=False)
        self.pe = Conv(dim, dim, 3, 1, g=dim, act=False)

    def forward(self, x, res=None):
        """res: added to the output (PSABlock's shortcut)."""
        B, C, H, W = x.shape
        N = H * W
        if fused.psa_ok(x, self.num_heads, self.key_dim, self.head_dim) and x.is_contiguous(memory_format=torch.channels_last):
            # one launch for QK^T, softmax, PV and the sum with the positional term; v's dense copy feeds the depthwise convolution
            qkv = self.qkv(x)                                                           # [B, heads*128, H, W], NHWC in memory
            per = 2 * self.key_dim + self.head_dim
            v = qkv.permute(0, 2, 3, 1).reshape(B, H, W, self.num_heads, per)[..., 2 * self.key_dim:].reshape(B, H, W, C).permute(0, 3, 1, 2)
            y = fused.psa_attention(qkv, self.pe(v), self.num_heads, self.scale)
            pc = self.proj.conv
            if res is not None and fused.pointwise_ok(pc):                               # PSABlock's x + attn(x) in the projection's epilogue
                # res_after: the projection is rounded to half BEFORE the shortcut is added, as `res + self.proj(y)` rounds it
                return fused.pointwise(y, fused.weight_nk(self.proj, pc), pc.bias, "none", res=res, res_after=True)
            return self.proj(y) if res is None else res + self.proj(y)
        qkv = self.qkv(x).contiguous().view(B, self.num_heads, self.key_dim * 2 + self.head_dim, N)
        q, k, v = qkv.split([self.key_dim, self.key_dim, self.head_dim], dim=2)
        attn = ((q.transpose(-2, -1) @ k) * self.scale).softmax(dim=-1)
        cl = x.is_contiguous(memory_format=torch.channels_last)
        vv = v.reshape(B, C, H, W)
        y = (v @ attn.transpose(-2, -1)).reshape(B, C, H, W) + self.pe(vv.contiguous(memory_format=torch.channels_last) if cl else vv)
        if cl:
            y = y.contiguous(memory_format=torch.channels_last)
        return self.proj(y) if res is None else res + self.proj(y)


class PSABlock(nn.Module):
    def __init__(self, c, attn_ratio=0.5, num_heads=4):
        super().__init__()
        self.attn = Attention(c, num_heads, attn_ratio)
        self.ffn = nn.Sequential(Conv(c, c * 2, 1), Conv(c * 2, c, 1, act=False))

    def forward(self, x, out=None, c_off=0):
        """out / c_off: the result also goes to channels [c_off, c_off + c) of `out` (C2PSA's concat buffer) when the fused path runs."""
        x = self.attn(x, res=x)
        f1 = self.ffn[1].conv
        if fused.usable(x) and fused.pointwise_ok(f1):                                   # x + ffn(x) in the second 1x1's epilogue
            if out is not None:
                fused.pointwise(self.ffn[0](x), fused.weight_nk(self.ffn[1], f1), f1.bias, "none", res=x, res_after=True, out=out, c_off=c_off)
                return None
            return fused.pointwise(self.ffn[0](x), fused.weight_nk(self.ffn[1], f1), f1.bias, "none", res=x, res_after=True)   # (two roundings, as x + ffn(x))
        y = x + self.ffn(x)
        if out is not None:
            out[:, c_off:c_off + y.shape[1]] = y
            return None
        return y


class C2PSA(nn.Module):
    def __init__(self, c1, n=1, e=0.5):
        super().__init__()
        self.c = int(c1 * e)
        self.cv1, self.cv2 = Conv(c1, 2 * self.c, 1), Conv(2 * self.c, c1, 1)
        self.m = nn.Sequential(*(PSABlock(self.c, 0.5, max(self.c // 64, 1)) for _ in range(n)))

    def forward(self, x):
        cv = self.cv1.conv
        if (fused.C3K2 and fused.usable(x) and fused.pointwise_ok(cv) and isinstance(self.cv1.act, nn.SiLU) and fused.place_ok(self.c, 2 * self.c)
                and len(self.m) >= 1 and all(fused.pointwise_ok(blk.ffn[1].conv) for blk in self.m)):
            # cv1 writes [a | b] into cv2's input and mirrors b densely for the attention blocks; the last block's shortcut epilogue
            # writes its result over b's slot: no split / contiguous / cat launches
            B, _, H, W = x.shape
            cat = torch.empty((B, 2 * self.c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            b = torch.empty((B, self.c, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            fused.pointwise(x, fused.weight_nk(self.cv1, cv), cv.bias, "silu", out=cat, c_off=0, out2=b, c0=self.c)
            for i, blk in enumerate(self.m):
                b = blk(b) if i + 1 < len(self.m) else blk(b, out=cat, c_off=self.c)
            return self.cv2(cat)
        a, b = self.cv1(x).split((self.c, self.c), 1)
        if x.is_contiguous(memory_format=torch.channels_last):
            b = b.contiguous(memory_format=torch.channels_last)
        return self.cv2(torch.cat((a, self.m(b)), 1))


class Detect11(Detect):
    """v11 head: the class branch is depthwise-separable (DWConv 3x3 -> Conv 1x1, twice) before the final 1x1."""

    def __init__(self, nc, ch, nk=0, nm=0, npr=256):
        super().__init__(nc, ch, nk, nm, npr)
        c3 = max(ch[0], min(nc, 100))
        self.cv3 = nn.ModuleList(nn.Sequential(nn.Sequential(Conv(x, x, 3, g=x), Conv(x, c3, 1)),
                                               nn.Sequential(Conv(c3, c3, 3, g=c3), Conv(c3, c3, 1)), nn.Conv2d(c3, nc, 1)) for x in ch)

    def forward(self, feats):
        ext_head = bool(self.nk or self.nm)
        if (fused.usable(feats[0]) and (not ext_head or self._ext_ok()) and len(feats) == 3
                and all(self._last_ok(s[2]) for s in list(self.cv2) + list(self.cv3))):
            z = self._zero_bias(feats[0])
            # class branch: DWConv 3x3 -> 1x1, twice, -> 1x1 (each layer one launch of its own kernel)
            cls = [fused.pointwise(s[1](s[0](f)), *self._last_wb(s[2])) for s, f in zip(self.cv3, feats)]
            grp = list(self.cv2)
            pads = [fused.padded_branch(self.cv4[i], self.cv4[i]) for i in range(3)] if ext_head else []
            if fused.GROUP and all(fused.conv3x3_ok(s[0].conv) and fused.conv3x3_ok(s[1].conv) and isinstance(s[0].act, nn.SiLU)
                                   and s[0].conv.out_channels <= 80 for s in grp):
                # the box branches (and the keypoint / coefficient branches) are independent 3x3 -> 3x3 -> 1x1 chains: one grouped
                # launch per depth
                xs = list(feats) * (2 if ext_head else 1)
                w = [[(fused.weight_n9k(s[d], s[d].conv), s[d].conv.bias) for s in grp] + [p[d] for p in pads] for d in (0, 1)]
                w.append([self._last_wb(s[2]) for s in grp] + [p[2] for p in pads])
                t = fused.conv_group([(x, wd, b, 3, 1, "silu") for (wd, b), x in zip(w[0], xs)])
                t = fused.conv_group([(x, wd, b, 3, 1, "silu") for (wd, b), x in zip(w[1], t)])
                t = fused.conv_group([(x, wd, b, 1, 1, "none") for (wd, b), x in zip(w[2], t)])
                return self._decode(feats, t[:3], cls, z, z, t[3:] if ext_head else None)
            box = [fused.pointwise(s[1](s[0](f)), *self._last_wb(s[2])) for s, f in zip(self.cv2, feats)]
            ext = None
            if ext_head:
                ext = []
                for p, f in zip(pads, feats):
                    u = fused.conv3x3(f, p[0][0], p[0][1], 1, "silu")
                    u = fused.conv3x3(u, p[1][0], p[1][1], 1, "silu")
                    ext.append(fused.pointwise(u, p[2][0], p[2][1]))
            return self._decode(feats, box, cls, z, z, ext)
        return self._forward_torch(feats)


_V11_SCALES = {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024), "m": (0.50, 1.00, 512)}


class YOLO11(nn.Module):
    def __init__(self, scale="n", nc=80, nk=0, nm=0):
        super().__init__()
        d, w, mc = _V11_SCALES[scale]
        c = lambda x: int(math.ceil(min(x, mc) * w / 8) * 8)
        n = lambda x: max(round(x * d), 1)
        k3 = scale in "mlx"                                       # the larger scales use C3k inner blocks everywhere
        self.b0, self.b1 = Conv(3, c(64), 3, 2), Conv(c(64), c(128), 3, 2)
        self.b2 = C3k2(c(128), c(256), n(2), k3, 0.25)
        self.b3, self.b4 = Conv(c(256), c(256), 3, 2), C3k2(c(256), c(512), n(2), k3, 0.25)
        self.b5, self.b6 = Conv(c(512), c(512), 3, 2), C3k2(c(512), c(512), n(2), True)
        self.b7, self.b8 = Conv(c(512), c(1024), 3, 2), C3k2(c(1024), c(1024), n(2), True)
        self.b9, self.b10 = SPPF(c(1024), c(1024)), C2PSA(c(1024), n(2))
        self.h13 = C3k2(c(1024) + c(512), c(512), n(2), k3)
        self.h16 = C3k2(c(512) + c(512), c(256), n(2), k3)
        self.h17, self.h19 = Conv(c(256), c(256), 3, 2), C3k2(c(256) + c(512), c(512), n(2), k3)
        self.h20, self.h22 = Conv(c(512), c(512), 3, 2), C3k2(c(512) + c(1024), c(1024), n(2), True)
        self.detect = Detect11(nc, (c(256), c(512), c(1024)), nk, nm, c(256))
        self.nc, self.nk, self.nm = nc, nk, nm

    def forward(self, x):
        p3 = self.b4(self.b3(self.b2(self.b1(self.b0(x)))))
        p4 = self.b6(self.b5(p3))
        p5 = self.b10(self.b9(self.b8(self.b7(p4))))
        h13 = self.h13(_upcat(p5, p4))
        h16 = self.h16(_upcat(h13, p3))
        h19 = self.h19(torch.cat((self.h17(h16), h13), 1))
        h22 = self.h22(torch.cat((self.h20(h19), p5), 1))
        return self.detect([h16, h19, h22])


class YOLOv5u(nn.Module):
    """YOLOv5 backbone/neck (C3 blocks) with the anchor-free head — what `YOLO('yolov5n.pt')`
    (yolo_multi_model.py:15) resolves to in current Ultralytics."""

    def __init__(self, scale="n", nc=80):
        super().__init__()
        d, w, mc = _V8_SCALES[scale]
        c = lambda x: int(math.ceil(min(x, mc) * w / 8) * 8)
        n = lambda x: max(round(x * d), 1)
        self.b0, self.b1, self.b2 = Conv(3, c(64), 6, 2, 2), Conv(c(64), c(128), 3, 2), C3(c(128), c(128), n(3))
        self.b3, self.b4 = Conv(c(128), c(256), 3, 2), C3(c(256), c(256), n(6))
        self.b5, self.b6 = Conv(c(256), c(512), 3, 2), C3(c(512), c(512), n(9))
        self.b7, self.b8, self.b9 = Conv(c(512), c(1024), 3, 2), C3(c(1024), c(1024), n(3)), SPPF(c(1024), c(1024))
        self.h10, self.h13 = Conv(c(1024), c(512), 1), C3(c(1024), c(512), n(3), False)
        self.h14, self.h17 = Conv(c(512), c(256), 1), C3(c(512), c(256), n(3), False)
        self.h18, self.h20 = Conv(c(256), c(256), 3, 2), C3(c(512), c(512), n(3), False)
        self.h21, self.h23 = Conv(c(512), c(512), 3, 2), C3(c(1024), c(1024), n(3), False)
        self.detect = Detect(nc, (c(256), c(512), c(1024)))
        self.nc, self.nk = nc, 0

    def forward(self, x):
        p3 = self.b4(self.b3(self.b2(self.b1(self.b0(x)))))
        p4 = self.b6(self.b5(p3))
        p5 = self.h10(self.b9(self.b8(self.b7(p4))))
        h13 = self.h14(self.h13(_upcat(p5, p4)))
        h17 = self.h17(_upcat(h13, p3))
        h20 = self.h20(torch.cat((self.h18(h17), h13), 1))
        h23 = self.h23(torch.cat((self.h21(h20), p5), 1))
        return self.detect([h17, h20, h23])


# --------------------------------------------------------------------------------------------------
# YOLOv7-style detector (ELAN backbone + SPPCSPC neck), anchor-free head for a uniform NMS layout
# --------------------------------------------------------------------------------------------------
class ELAN(nn.Module):
    def __init__(self, c1, c_, c2, depth=4):
        super().__init__()
        self.cv1, self.cv2 = Conv(c1, c_, 1), Conv(c1, c_, 1)
        self.m = nn.ModuleList(Conv(c_, c_, 3) for _ in range(depth))
        self.out = Conv(c_ * (2 + depth // 2), c2, 1)

    def _placed_ok(self, x) -> bool:
        a, b = self.cv1.conv, self.cv2.conv
        return (fused.C3K2 and fused.usable(x) and fused.pointwise_ok(a) and fused.pointwise_ok(b) and fused.place_ok(a.out_channels, self.out.conv.in_channels)
                and all(isinstance(c.act, nn.SiLU) for c in [self.cv1, self.cv2] + list(self.m))
                and all(fused.conv3x3_ok(m.conv) and m.conv.stride == (1, 1) for m in self.m))

    def forward(self, x):
        if self._placed_ok(x):
            return self._forward_placed(x)
        y = [self.cv1(x), self.cv2(x)]
        t = y[-1]
        for i, m in enumerate(self.m):
            t = m(t)
            if i % 2 == 1:
                y.append(t)
        return self.out(torch.cat(y, 1))

    def _forward_placed(self, x):
        """The same arithmetic without the concat copy: [cv1 | cv2] as one 1x1 launch into the concat buffer (cv2's half mirrored
        densely for the 3x3 chain), every second 3x3 writes its slice (and a dense copy for the next one)."""
        c_ = self.cv1.conv.out_channels
        B, _, H, W = x.shape
        cat = torch.empty((B, self.out.conv.in_channels, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        dense = lambda: torch.empty((B, c_, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        cur = dense()
        w12, b12 = _w_pair(self, self.cv1.conv, self.cv2.conv)
        fused.pointwise(x, w12, b12, "silu", out=cat, c_off=0, out2=cur, c0=c_)
        n = len(self.m)
        for i, m in enumerate(self.m):
            cv = m.conv
            if i % 2 == 1:
                nxt = dense() if i + 1 < n else None
                fused.conv3x3(cur, fused.weight_n9k(m, cv), cv.bias, 1, "silu", out=cat, c_off=(2 + i // 2) * c_, out2=nxt, c0=0)
                cur = nxt
            else:
                cur = fused.conv3x3(cur, fused.weight_n9k(m, cv), cv.bias, 1, "silu")
        return self.out(cat)


class MP(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.cv1, self.cv2, self.cv3 = Conv(c1, c2 // 2, 1), Conv(c1, c2 // 2, 1), Conv(c2 // 2, c2 // 2, 3, 2)

    def forward(self, x):
        a, c3 = self.cv1.conv, self.cv3.conv
        if (fused.C3K2 and fused.usable(x) and fused.pointwise_ok(a) and fused.conv3x3_ok(c3) and isinstance(self.cv1.act, nn.SiLU)
                and isinstance(self.cv3.act, nn.SiLU) and fused.place_ok(a.out_channels, 2 * a.out_channels) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0):
            h = a.out_channels                           # both branches write their half of the output
            B, _, H, W = x.shape
            out = torch.empty((B, 2 * h, H // 2, W // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            fused.conv3x3(self.cv2(x), fused.weight_n9k(self.cv3, c3), c3.bias, 2, "silu", out=out, c_off=0)
            fused.pointwise(F.max_pool2d(x, 2, 2), fused.weight_nk(self.cv1, a), a.bias, "silu", out=out, c_off=h)
            return out
        return torch.cat((self.cv3(self.cv2(x)), self.cv1(F.max_pool2d(x, 2, 2))), 1)


class SPPCSPC(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        c_ = c2
        self.cv1, self.cv2 = Conv(c1, c_, 1), Conv(c1, c_, 1)
        self.cv3, self.cv4 = Conv(c_, c_, 3), Conv(c_, c_, 1)
        self.cv5, self.cv6, self.cv7 = Conv(4 * c_, c_, 1), Conv(c_, c_, 3), Conv(2 * c_, c2, 1)

    def forward(self, x):
        x1 = self.cv4(self.cv3(self.cv1(x)))
        c2, c6 = self.cv2.conv, self.cv6.conv
        if (fused.C3K2 and fused.sppf_pools_ok(x1) and fused.pointwise_ok(c2) and fused.conv3x3_ok(c6) and isinstance(self.cv2.act, nn.SiLU)
                and isinstance(self.cv6.act, nn.SiLU) and fused.place_ok(c6.out_channels, 2 * c6.out_channels)):
            # max pools of 5, 9, 13 = the cascade of three 5-pools (max over nested windows): the SPPF launch gives cat(x1, p5, p9, p13);
            # cv6 and cv2 write their halves of cv7's input
            t = self.cv5(fused.sppf_pools(x1))
            c_ = c6.out_channels
            B, _, H, W = x.shape
            cat = torch.empty((B, 2 * c_, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            fused.conv3x3(t, fused.weight_n9k(self.cv6, c6), c6.bias, 1, "silu", out=cat, c_off=0)
            fused.pointwise(x, fused.weight_nk(self.cv2, c2), c2.bias, "silu", out=cat, c_off=c_)
            return self.cv7(cat)
        y1 = self.cv6(self.cv5(torch.cat([x1] + [F.max_pool2d(x1, k, 1, k // 2) for k in (5, 9, 13)], 1)))
        return self.cv7(torch.cat((y1, self.cv2(x)), 1))


class YOLOv7(nn.Module):
    def __init__(self, nc=80):
        super().__init__()
        self.stem = nn.Sequential(Conv(3, 32, 3, 1), Conv(32, 64, 3, 2), Conv(64, 64, 3, 1), Conv(64, 128, 3, 2))
        self.e1 = ELAN(128, 64, 256)
        self.mp1, self.e2 = MP(256, 256), ELAN(256, 128, 512)
        self.mp2, self.e3 = MP(512, 512), ELAN(512, 256, 1024)
        self.mp3, self.e4 = MP(1024, 1024), ELAN(1024, 256, 1024)
        self.spp = SPPCSPC(1024, 512)
        self.l1, self.r1, self.n1 = Conv(512, 256, 1), Conv(1024, 256, 1), ELAN(512, 128, 256)
        self.l2, self.r2, self.n2 = Conv(256, 128, 1), Conv(512, 128, 1), ELAN(256, 64, 128)
        self.d1, self.n3 = MP(128, 256), ELAN(512, 128, 256)
        self.d2, self.n4 = MP(256, 512), ELAN(1024, 256, 512)
        self.detect = Detect(nc, (128, 256, 512))
        self.nc, self.nk = nc, 0

    def forward(self, x):
        c3 = self.e2(self.mp1(self.e1(self.stem(x))))
        c4 = self.e3(self.mp2(c3))
        c5 = self.spp(self.e4(self.mp3(c4)))
        p4 = self.n1(_upcat(self.l1(c5), self.r1(c4), lo_first=False))
        p3 = self.n2(_upcat(self.l2(p4), self.r2(c3), lo_first=False))
        n4 = self.n3(torch.cat((self.d1(p3), p4), 1))
        n5 = self.n4(torch.cat((self.d2(n4), c5), 1))
        return self.detect([p3, n4, n5])


# --------------------------------------------------------------------------------------------------
# OSNet (x0.25: channels 16/64/96/128, 512-d embedding)
# --------------------------------------------------------------------------------------------------
class ConvBR(nn.Module):
    def __init__(self, c1, c2, k, s=1, p=0, g=1, relu=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, p, groups=g, bias=True)
        self.relu = relu

    def forward(self, x, res=None):
        if fused.usable(x):
            c = self.conv
            if fused.pointwise_ok(c):        # one MFMA launch: bias (+ shortcut) + ReLU in the epilogue
                return fused.pointwise(x, fused.weight_nk(self, c), c.bias, "relu" if self.relu else "none", res=res)
            if fused.is_pointwise(c):
                if not self.relu and res is None:
                    return fused.conv1x1(x, fused.weight_t(self, c), c.bias)           # bias in the GEMM epilogue
                y = fused.conv1x1(x, fused.weight_t(self, c))
            else:
                y = F.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
            return fused.bias_act_(y, c.bias, "relu" if self.relu else "none", res)
        x = self.conv(x)
        if res is not None:
            x = x + res
        return F.relu(x, inplace=True) if self.relu else x


class LightConv3x3(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.pw = nn.Conv2d(c1, c2, 1, bias=False)
        self.dw = nn.Conv2d(c2, c2, 3, 1, 1, groups=c2, bias=True)

    def forward(self, x):
        if fused.usable(x):          # 1x1 conv (MIOpen) + fused depthwise 3x3 + bias + ReLU
            if getattr(self, "_w9", None) is None or self._w9.device != x.device:
                self._w9 = self.dw.weight.detach().reshape(self.dw.weight.shape[0], 9).t().contiguous()
                self._w1 = self.pw.weight.detach().reshape(self.pw.weight.shape[0], -1).contiguous()
            if self._w1.shape[0] == self._w1.shape[1] and fused.lightconv_ok(x):      # one pass: MFMA pointwise -> LDS -> depthwise
                return fused.lightconv(x, self._w1, self._w9, self.dw.bias)
            return fused.dwconv3x3(fused.conv1x1(x, fused.weight_t(self, self.pw)), self._w9, self.dw.bias, "relu")
        return F.relu(self.dw(self.pw(x)), inplace=True)


class ChannelGate(nn.Module):
    def __init__(self, c, reduction=16):
        super().__init__()
        self.fc1 = nn.Conv2d(c, max(c // reduction, 1), 1)
        self.fc2 = nn.Conv2d(max(c // reduction, 1), c, 1)

    def forward(self, x):
        g = x.mean((2, 3), keepdim=True)
        return x * torch.sigmoid(self.fc2(F.relu(self.fc1(g), inplace=True)))


class OSBlock(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        mid = c2 // 4
        self.conv1 = ConvBR(c1, mid, 1)
        self.streams = nn.ModuleList(nn.Sequential(*(LightConv3x3(mid, mid) for _ in range(t))) for t in (1, 2, 3, 4))
        self.gate = ChannelGate(mid)
        self.conv3 = ConvBR(mid, c2, 1, relu=False)
        self.down = ConvBR(c1, c2, 1, relu=False) if c1 != c2 else None

    def _gate_w(self):
        g = self.gate
        cr, c = g.fc1.weight.shape[0], g.fc1.weight.shape[1]
        return (g.fc1.weight.reshape(cr, c), g.fc1.bias, g.fc2.weight.reshape(c, cr), g.fc2.bias)

    def _stream_w(self, x1):
        sw = getattr(self, "_sw", None)
        if sw is None or sw[0].device != x1.device:
            c = x1.shape[1]
            layers = [m for st in self.streams for m in st]
            w9 = torch.stack([m.dw.weight.detach().reshape(c, 9).t() for m in layers]).contiguous()
            b = torch.stack([m.dw.bias.detach() for m in layers]).contiguous()
            sw = self._sw = (torch.stack([m.pw.weight.detach().reshape(c, c) for m in layers]).contiguous(), w9, b, fused.dwtab(w9, b))
        return sw

    def tail_ok(self, x, nxt, pool) -> bool:
        """The fused tail (gate + conv3 + shortcut + ReLU + the following 1x1 ConvBR [+ 2x2 average]) covers this block."""
        n, _, h, w = x.shape
        mid, c2 = self.conv1.conv.out_channels, self.conv3.conv.out_channels
        return (fused.usable(x) and isinstance(nxt, ConvBR) and nxt.relu and fused.pointwise_ok(nxt.conv) and nxt.conv.in_channels == c2
                and fused.pointwise_ok(self.conv3.conv) and fused.streams_ok_dims(mid, w)
                and fused.tail_ok(mid, c2, nxt.conv.out_channels, h, w, pool))

    def forward_tail(self, x, x1, nxt, pool, want_out):
        """-> (block output or None, relu(nxt(block output)) [2x2-averaged when pool]); x1 = this block's conv1(x) if the previous
        block's tail already produced it."""
        c3, c4 = self.conv3.conv, nxt.conv
        down = None
        if self.down is not None and fused.pointwise_ok(self.down.conv) and not self.down.relu and not pool and fused.tail_down_ok(
                x.shape[1], c3.in_channels, c3.out_channels, c4.out_channels):
            down = (fused.weight_nk(self.down, self.down.conv), self.down.conv.bias)      # the tail computes the shortcut from x itself
        idn = x if (self.down is None or down is not None) else self.down(x)
        if x1 is None:
            x1 = self.conv1(x)
        ys, psum = fused.osnet_streams(x1, *self._stream_w(x1))
        return fused.osnet_tail(ys, psum, self._gate_w(), fused.weight_nk(self.conv3, c3), c3.bias, idn, want_out,
                                fused.weight_nk(nxt, c4), c4.bias, pool, down)

    def forward(self, x, x1=None):
        idn = x if self.down is None else self.down(x)
        if x1 is None:
            x1 = self.conv1(x)
        if fused.usable(x):          # all four gates + their sum in two launches; bias + residual + ReLU in one
            gw = self._gate_w()
            if fused.streams_ok(x1):     # the ten LightConv layers of the four chains in one launch + gate from its sums
                ys, psum = fused.osnet_streams(x1, *self._stream_w(x1))
                x2 = fused.gate_apply(ys, psum, *gw)
            else:
                x2 = fused.gate_sum([s(x1) for s in self.streams], *gw)
            c3 = self.conv3.conv
            if fused.pointwise_ok(c3):
                return fused.pointwise(x2, fused.weight_nk(self.conv3, c3), c3.bias, "relu", res=idn)
            y = fused.conv1x1(x2, fused.weight_t(self.conv3, c3))
            return fused.bias_act_(y, c3.bias, "relu", idn)
        x2 = sum(self.gate(s(x1)) for s in self.streams)
        return F.relu(self.conv3(x2) + idn, inplace=True)


class OSNet(nn.Module):
    def __init__(self, channels=(16, 64, 96, 128), feature_dim=512):
        super().__init__()
        c = channels
        self.conv1 = ConvBR(3, c[0], 7, 2, 3)
        self.conv2 = nn.Sequential(OSBlock(c[0], c[1]), OSBlock(c[1], c[1]), ConvBR(c[1], c[1], 1), nn.AvgPool2d(2, 2))
        self.conv3 = nn.Sequential(OSBlock(c[1], c[2]), OSBlock(c[2], c[2]), ConvBR(c[2], c[2], 1), nn.AvgPool2d(2, 2))
        self.conv4 = nn.Sequential(OSBlock(c[2], c[3]), OSBlock(c[3], c[3]))
        self.conv5 = ConvBR(c[3], c[3], 1)
        self.fc = nn.Linear(c[3], feature_dim)

    N_PARTS = 10

    def _block_part(self, k, s):
        """Parts 1, 2, 4, 5, 7, 8: an OSBlock.  On the GPU the block's tail also runs the 1x1 convolution that follows it —
        the next block's conv1 (state becomes (block output, that conv1's output)) or the stage's ConvBR (+ average pool;
        state becomes a 1-tuple: the following part is already applied)."""
        blk, nxt, pool = self._blocks(k)
        x, x1 = s if isinstance(s, tuple) else (s, None)
        if blk.tail_ok(x, nxt, pool):
            want_out = k in (1, 4, 7)
            out, o2 = blk.forward_tail(x, x1, nxt, pool, want_out)
            return (out, o2) if want_out else (o2,)
        return blk(x, x1)

    def _blocks(self, k):
        return {1: (self.conv2[0], self.conv2[1].conv1, False), 2: (self.conv2[1], self.conv2[2], True),
                4: (self.conv3[0], self.conv3[1].conv1, False), 5: (self.conv3[1], self.conv3[2], True),
                7: (self.conv4[0], self.conv4[1].conv1, False), 8: (self.conv4[1], self.conv5, False)}[k]

    def _part32(self, k, s):
        """Part k with fp32 activations on the hand-written fp32 kernels (fused32.py, csrc/ss_ops32.hip; the accuracy mode).
        Same states as the half path: part 0 -> (x0, conv1 of the first block); a first block of a stage -> (block output, next
        block's conv1); a second block -> (the ConvBR after it [+ 2x2 average],); parts 3, 6, 9 only unwrap."""
        f = fused32
        if k == 0:
            return f.stem(s, self.conv1, self.conv2[0].conv1)      # (x0, the first block's conv1 of x0) from one launch
        if k in (1, 2, 4, 5, 7, 8):
            blk, nxt, pool = self._blocks(k)
            x, x1 = s if (isinstance(s, tuple) and len(s) == 2) else (s[0] if isinstance(s, tuple) else s, None)
            if x1 is None:
                x1 = f.pointwise(x, blk.conv1, blk.conv1.conv, relu=True)
            ys, psum = f.chains(x1, blk)
            want_out = k in (1, 4, 7)
            out, o2 = f.tail(ys, psum, blk, x, nxt, pool, want_out)
            return (out, o2) if want_out else (o2,)
        assert isinstance(s, tuple) and len(s) == 1           # the previous block's tail already ran this part
        return s[0]

    def _fp32_kernels(self, x) -> bool:
        first = x[0] if isinstance(x, tuple) else x
        if not fused32.usable(first):
            return False
        if getattr(self, "_ok32", None) is None:
            self._ok32 = fused32.osnet_ok(self, torch.empty(1, 3, 256, 128, device=first.device))
        return self._ok32

    def _part(self, k, x):
        """The backbone as 10 consecutive parts, so a frame pipeline can cut it anywhere to balance its stages.  The state
        between parts is a tensor or a tuple of tensors (see _block_part)."""
        if k == 0 and isinstance(x, torch.Tensor) and x.dtype == torch.uint8:      # byte crops (a4 with SS_DST_U8): the fp32 stem normalises them itself
            if fused32.usable_u8(x):
                if getattr(self, "_ok32", None) is None:
                    self._ok32 = fused32.osnet_ok(self, torch.empty(1, 3, 256, 128, device=x.device))
                if self._ok32:
                    return self._part32(0, x)
            x = fused32.crops_from_u8(x).to(self.conv1.conv.weight.dtype)
        if self._fp32_kernels(x):
            first = x[0] if isinstance(x, tuple) else x
            want = {0: (256, 128), 1: (64, 32), 2: (64, 32), 4: (32, 16), 5: (32, 16), 7: (16, 8), 8: (16, 8)}.get(k)
            if (want is None and isinstance(x, tuple) and len(x) == 1) or (want is not None and tuple(first.shape[2:]) == want):
                return self._part32(k, x)
        if k == 0:
            if fused.usable(x) and fused.stem_ok(x, self.conv1.conv) and self.conv1.relu:      # conv + bias + ReLU + pool, one launch
                c1 = self.conv2[0].conv1
                if (fused.STEM_CONV1 and fused.TAIL and c1.relu and fused.pointwise_ok(c1.conv) and c1.conv.in_channels == 16
                        and c1.conv.out_channels == 16):        # the first block's conv1 from the same launch: state (x0, x1)
                    return fused.osnet_stem(x, fused.stem_weight(self.conv1, self.conv1.conv), self.conv1.conv.bias,
                                            (fused.weight_nk(c1, c1.conv), c1.conv.bias))
                return fused.osnet_stem(x, fused.stem_weight(self.conv1, self.conv1.conv), self.conv1.conv.bias)
            x = self.conv1(x)
            return fused.maxpool(x, 3, 2, 1) if fused.usable(x) else F.max_pool2d(x, 3, 2, 1)
        if k in (1, 2, 4, 5, 7, 8):
            return self._block_part(k, x)
        if isinstance(x, tuple):                            # the previous block's tail already ran this part
            assert len(x) == 1
            return x[0]
        if k == 3:
            t = self.conv2[2](x)
            return fused.avgpool2(t) if fused.usable(t) else self.conv2[3](t)
        if k == 6:
            t = self.conv3[2](x)
            return fused.avgpool2(t) if fused.usable(t) else self.conv3[3](t)
        return self.conv5(x)

    def forward_a(self, x, upto: int = 5):
        """Parts [0, upto): the default runs through the first block of conv3 (~half the launches)."""
        for k in range(upto):
            x = self._part(k, x)
        return x

    def forward_b(self, x, start: int = 5):
        """`x`: what forward_a(., start) returned (a tensor or a tuple of tensors)."""
        for k in range(start, self.N_PARTS):
            x = self._part(k, x)
        if fused.osnet_head_ok(x, self.fc):                  # average pool + fc + ReLU in one launch
            return fused.osnet_head(x, self.fc)
        if self._fp32_kernels(x) and x.shape[1] == 128:
            return fused32.head(x, self.fc)
        return F.relu(self.fc(x.mean((2, 3))))

    def forward(self, x):
        return self.forward_b(self.forward_a(x, 0), 0)


def _upcat(lo, hi, lo_first=True):
    """cat(upsample2x(lo), hi) (or cat(hi, upsample2x(lo))) along channels."""
    if fused.upcat_ok(lo, hi):
        return fused.upcat(lo, hi, lo_first)
    if fused32.upcat_ok(lo, hi):
        return fused32.upcat(lo, hi, lo_first)
    up = F.interpolate(lo, scale_factor=2.0, mode="nearest")
    return torch.cat((up, hi) if lo_first else (hi, up), 1)


def osnet_x0_25():
    return OSNet((16, 64, 96, 128), 512)


DETECTORS = {
    "yolov8n": lambda: YOLOv8("n"), "yolov8s": lambda: YOLOv8("s"), "yolov8m": lambda: YOLOv8("m"),
    "yolov8n-pose": lambda: YOLOv8("n", nc=1, nk=51), "yolo11n-pose": lambda: YOLO11("n", nc=1, nk=51),
    "yolov8n-seg": lambda: YOLOv8("n", nm=32), "yolov8s-seg": lambda: YOLOv8("s", nm=32), "yolo11n-seg": lambda: YOLO11("n", nm=32),
    "yolo11n": lambda: YOLO11("n"), "yolo11s": lambda: YOLO11("s"), "yolo11s-pose": lambda: YOLO11("s", nc=1, nk=51),
    "yolov5n": lambda: YOLOv5u("n"), "yolov5s": lambda: YOLOv5u("s"),
    "yolov7": lambda: YOLOv7(),
}


def build_detector(name: str, seed: int = 0) -> nn.Module:
    """Seeded random-init detector named like the reference's weight files
    (yolo_multi_model.py:14-17: 'yolov8n-seg.pt', 'yolov5n.pt', 'yolo11n.pt', 'yolo11n-pose.pt')."""
    key = name.lower().replace(".pt", "")
    if key not in DETECTORS:
        raise ValueError(f"unknown detector '{name}'; known: {sorted(DETECTORS)}")
    g = torch.random.fork_rng()
    with g:
        torch.manual_seed(seed)
        m = DETECTORS[key]()
    return m.eval()


def load_weights(module: nn.Module, path, what: str, random_init_ok: bool = False) -> bool:
    """Load a plain state_dict (or {'state_dict': ...} / {'model': state_dict}) into `module`.  A missing file is an
    error unless random-init weights were asked for explicitly (`random_init_ok`, or SS_RANDOM_INIT=1 in the
    environment — the only mode available offline): a tracker that silently runs on random weights
    returns plausible-looking garbage.  Pickled module checkpoints (Ultralytics `{'model': nn.Module}`) are refused:
    `torch.load(weights_only=True)` never executes checkpoint code."""
    import os
    import warnings
    if not path or not os.path.isfile(path):
        if random_init_ok or os.environ.get("SS_RANDOM_INIT") == "1":
            warnings.warn(f"{what}: no weights file ({path!r}); running on SEEDED RANDOM-INIT weights — outputs are "
                          f"synthetic-load only", RuntimeWarning, stacklevel=3)
            return False
        raise FileNotFoundError(f"{what}: weights file {path!r} not found (pass random_init_ok=True / --random-init or set "
                                f"SS_RANDOM_INIT=1 to run the seeded random-init network of the same architecture)")
    ck = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("state_dict", "model"):
        if isinstance(ck, dict) and isinstance(ck.get(key), dict):
            ck = ck[key]
    if not isinstance(ck, dict) or not all(isinstance(v, torch.Tensor) for v in ck.values()):
        raise ValueError(f"{what}: {path} is not a plain state_dict; export one with torch.save(model.state_dict(), ...)")
    if any(k.startswith("model.") for k in ck):                 # an Ultralytics DetectionModel state_dict: layer indices + Conv/BN pairs
        ck = convert_ultralytics_state_dict(module, ck)
    module.load_state_dict(ck, strict=True)
    fused.clear_prepared(module)                                   # kernel-side copies built from the previous tensors, if any
    return True


def convert_ultralytics_state_dict(module: nn.Module, sd: dict, bn_eps: float = 1e-3) -> dict:
    """Keys of an Ultralytics model's `model.model.state_dict()` (exported where `ultralytics` is installed:
    `torch.save(YOLO("yolo11n-pose.pt").model.state_dict(), "yolo11n-pose.pt")`) -> this module's keys:
      * `model.<i>.` -> the attribute that stands for layer i here (`b<i>` / `h<i>`, the last indexed layer = `detect`);
      * every Conv + BatchNorm pair (`X.conv.weight`, `X.bn.{weight,bias,running_mean,running_var}`; Ultralytics uses
        eps = 1e-3) folded into the biased convolution the networks here are built with;
      * the fixed DFL projection (`dfl.conv.weight` = arange(16)) and `num_batches_tracked` dropped.
    The result is loaded strictly, so a layer that does not line up raises instead of running on wrong weights."""
    idx = {}
    for name, _ in module.named_children():
        if name[:1] in "bh" and name[1:].isdigit():
            idx[int(name[1:])] = name
    layers = sorted({int(k.split(".")[1]) for k in sd if k.startswith("model.") and k.split(".")[1].isdigit()})
    if hasattr(module, "detect") and layers:
        idx.setdefault(layers[-1], "detect")
    out, bn = {}, {}
    for k, v in sd.items():
        parts = k.split(".")
        if parts[0] != "model" or not parts[1].isdigit():
            raise ValueError(f"unexpected key {k!r} in an Ultralytics state_dict")
        i = int(parts[1])
        if i not in idx:
            raise ValueError(f"layer {i} of the checkpoint has no counterpart in {type(module).__name__} ({k!r})")
        rest = parts[2:]
        if rest[-1] == "num_batches_tracked" or rest[:2] == ["dfl", "conv"]:
            continue
        name = ".".join([idx[i]] + rest)
        if len(rest) >= 2 and rest[-2] == "bn":
            bn.setdefault(".".join([idx[i]] + rest[:-2]), {})[rest[-1]] = v.float()
        else:
            out[name] = v
    for prefix, p in bn.items():
        wk = prefix + ".conv.weight"
        if wk not in out or not {"weight", "bias", "running_mean", "running_var"} <= set(p):
            raise ValueError(f"incomplete Conv/BatchNorm pair at {prefix!r}")
        w = out[wk].float()
        scale = p["weight"] / torch.sqrt(p["running_var"] + bn_eps)
        out[wk] = (w * scale.view(-1, 1, 1, 1)).to(out[wk].dtype)
        b0 = out.get(prefix + ".conv.bias")
        out[prefix + ".conv.bias"] = (p["bias"] - p["running_mean"] * scale + (b0.float() * scale if b0 is not None else 0)).to(out[wk].dtype)
    return out


def build_reid(seed: int = 1) -> nn.Module:
    with torch.random.fork_rng():
        torch.manual_seed(seed)
        m = osnet_x0_25()
    return m.eval()
