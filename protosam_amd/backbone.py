"""`TVDeeplabRes101Encoder` (models/backbone/torchvision_backbones.py:12-52) on the HIP path: the ResNet-101 trunk of
torchvision's `deeplabv3_resnet101` (output stride 8: layer3 / layer4 dilated instead of strided) followed by the 1x1
`localconv` 2048 -> 256. This is the reference's `which_model = 'dlfcn_res101'` encoder (BASELINE config 1).

torchvision is absent here and from /root/reference => the architecture is restated from torchvision 0.15's published
`resnet101(replace_stride_with_dilation=[False, True, True])` (Bottleneck v1.5: stride on the 3x3 conv; the first block of
a dilated layer keeps the previous dilation) and PARITY IS UNPINNED; parameter names follow torchvision's module tree so
the reference's checkpoints load (`backbone.conv1`, `backbone.bn1`, `backbone.layer{1..4}.{i}.conv{1,2,3}` / `bn{1,2,3}` /
`downsample.{0,1}`, `localconv`, plus the unused `asppconv` / `aspp_out.*` the reference keeps in its state dict).

Execution: activations are token-major (NHWC) fp16; every convolution is `psam_gemm_f16` on an im2col view (1x1 convs need
none) with eval-mode BatchNorm folded into weights and bias and ReLU / identity-add fused in the epilogue (epilogue 3).
Channel counts below 128 (stem, layer1 width 64) are zero-padded to 128 for the GEMM's N granularity.
"""

import torch
import torch.nn as nn

from . import ops

_LAYERS = (3, 4, 23, 3)
_WIDTHS = (64, 128, 256, 512)


def _bn(c):
    return nn.BatchNorm2d(c)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, dilation, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = _bn(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), _bn(planes * 4))
        self.stride, self.dilation = stride, dilation


class ResNet101Trunk(nn.Module):
    def __init__(self, layers=_LAYERS):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = _bn(64)
        inplanes, dilation = 64, 1
        for li, (n, w) in enumerate(zip(layers, _WIDTHS)):
            stride = 1 if li == 0 else 2
            prev = dilation
            if li >= 2:            # replace_stride_with_dilation = [False, True, True]
                dilation *= stride
                stride = 1
            blocks = [Bottleneck(inplanes, w, stride, prev, stride != 1 or inplanes != w * 4)]
            inplanes = w * 4
            blocks += [Bottleneck(inplanes, w, 1, dilation, False) for _ in range(1, n)]
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))


class _ASPPPlaceholder(nn.Module):
    """Parameters of torchvision's ASPP head the reference keeps (unused with use_aspp=False) for strict loading."""

    def __init__(self):
        super().__init__()
        def cbr(cin, k, d=1):
            return nn.Sequential(nn.Conv2d(cin, 256, k, padding=0 if k == 1 else d, dilation=d, bias=False), _bn(256), nn.ReLU())
        pool = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(2048, 256, 1, bias=False), _bn(256), nn.ReLU())
        self.convs = nn.ModuleList([cbr(2048, 1), cbr(2048, 3, 12), cbr(2048, 3, 24), cbr(2048, 3, 36), pool])
        self.project = nn.Sequential(nn.Conv2d(5 * 256, 256, 1, bias=False), _bn(256), nn.ReLU(), nn.Dropout(0.5))


def _fold(conv, bn, cin_pad=None, cout_pad=None, order="khwc"):
    """conv weight [Cout, Cin, kh, kw] + eval BatchNorm -> (half [Cout_p, K] with column (ky*kw + kx)*Cin_p + c, fp32 bias)."""
    w = conv.weight.detach().float()
    cout, cin, kh, kw = w.shape
    if bn is not None:
        s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        b = bn.bias.detach().float() - bn.running_mean.detach().float() * s
        w = w * s[:, None, None, None]
    else:
        b = torch.zeros(cout, device=w.device)
    cin_p, cout_p = cin_pad or cin, cout_pad or cout
    if order == "khwc":
        wp = torch.zeros((cout_p, kh, kw, cin_p), device=w.device)
        wp[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
    else:                  # stem: the weight's own (c, ky, kx) order, K padded to a multiple of 64
        wp = torch.zeros((cout_p, cin_p), device=w.device)
        wp[:cout, :cin * kh * kw] = w.reshape(cout, -1)
    bp = torch.zeros(cout_p, device=w.device)
    bp[:cout] = b
    return wp.reshape(cout_p, -1).to(torch.float16).contiguous(), bp.contiguous()


class TVDeeplabRes101Encoder(nn.Module):
    def __init__(self, use_coco_init=False, aux_dim_keep=64, use_aspp=False, layers=_LAYERS):
        super().__init__()
        if use_coco_init:
            raise NotImplementedError("ms-coco initialisation downloads torchvision weights; load a checkpoint instead")
        if use_aspp:
            raise NotImplementedError("use_aspp=True is never enabled by the reference (torchvision_backbones.py:16)")
        self.aux_dim_keep = aux_dim_keep
        self.backbone = ResNet101Trunk(layers)
        self.localconv = nn.Conv2d(2048, 256, kernel_size=1, stride=1, bias=False)
        self.asppconv = nn.Conv2d(256, 256, kernel_size=1, bias=False)
        self.aspp_out = nn.Sequential(_ASPPPlaceholder(), nn.Conv2d(256, 256, 3, padding=1, bias=False))
        self.use_aspp = use_aspp
        self.embed_dim = 256
        self._cache = None

    def _apply(self, fn, *a, **k):
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1
        self._cache = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):   # also reached by a parent's recursive load
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1
        self._cache = None
        return super()._load_from_state_dict(*a, **k)

    def _packed(self):
        if self._cache is not None:
            return self._cache
        bb = self.backbone
        pk = dict(stem=_fold(bb.conv1, bb.bn1, cin_pad=192, cout_pad=128, order="stem"), blocks=[])
        for li in range(4):
            for blk in getattr(bb, f"layer{li + 1}"):
                cin = blk.conv1.in_channels
                width = blk.conv1.out_channels
                cin_p, w_p = max(cin, 128), max(width, 128)
                d = dict(c1=_fold(blk.conv1, blk.bn1, cin_p, w_p), c2=_fold(blk.conv2, blk.bn2, w_p, w_p),
                         c3=_fold(blk.conv3, blk.bn3, w_p, None), stride=blk.stride, dil=blk.dilation, cin=cin_p, width=w_p,
                         cout=width * 4, ds=None)
                if blk.downsample is not None:
                    d["ds"] = _fold(blk.downsample[0], blk.downsample[1], cin_p, None)
                pk["blocks"].append(d)
        pk["local"] = _fold(self.localconv, None)
        self._cache = pk
        return pk

    @torch.no_grad()
    def forward_tokens(self, x_in):
        """[B,3,H,W] fp32 -> token-major high-level features fp32 [B, (H/8)*(W/8), 256]."""
        if self.training:
            raise NotImplementedError("BatchNorm is folded for inference; call .eval()")
        pk = self._packed()
        B = x_in.shape[0]
        cols, H, W = ops.im2col_stem(x_in.float().contiguous(), 192)
        w, b = pk["stem"]
        x = ops.gemm(cols, w, b, epilogue=ops.EPI_RELU_F16)                       # [B*H*W, 128] (64 real channels)
        x, H, W = ops.maxpool3x3s2(x, B, H, W, 128)
        for d in pk["blocks"]:
            w1, b1 = d["c1"]
            y = ops.gemm(x, w1, b1, epilogue=ops.EPI_RELU_F16)                    # 1x1
            cols, Ho, Wo = ops.im2col(y, B, H, W, d["width"], 3, 3, d["stride"], d["dil"], d["dil"])
            w2, b2 = d["c2"]
            y = ops.gemm(cols, w2, b2, epilogue=ops.EPI_RELU_F16)                 # 3x3 (stride / dilation)
            if d["ds"] is not None:
                idn = x
                if d["stride"] != 1:
                    idn, _, _ = ops.im2col(x, B, H, W, d["cin"], 1, 1, d["stride"], 1, 0)
                wd, bd = d["ds"]
                idn = ops.gemm(idn, wd, bd, epilogue=ops.EPI_F16)                 # downsample conv + BN (no ReLU)
            else:
                idn = x
            w3, b3 = d["c3"]
            x = ops.gemm(y, w3, b3, epilogue=ops.EPI_RELU_F16, resid=idn)         # 1x1 + identity + ReLU
            H, W = Ho, Wo
        wl, bl = pk["local"]
        out = ops.gemm(x, wl, None, epilogue=ops.EPI_F32)                         # localconv, fp32 token-major
        return out.view(B, H * W, 256), H, W

    def forward(self, x_in, low_level):
        if low_level:
            raise NotImplementedError("low-level (aux) features are unused by the inference path (grid_proto_fewshot.py:100)")
        tok, H, W = self.forward_tokens(x_in)
        return tok.view(x_in.shape[0], H, W, 256).permute(0, 3, 1, 2)
