"""Oracle: torchvision's deeplabv3_resnet101 trunk + `localconv` (models/backbone/torchvision_backbones.py:12-52) as
functional fp32 torch over a state dict with torchvision's key names. Test infrastructure only (see oracle/__init__.py).

torchvision==0.15.2 (requirements.txt:65) is absent here and from /root/reference, so this is restated from its published
`resnet101(replace_stride_with_dilation=[False, True, True])`: 7x7/2 stem + BN + ReLU + MaxPool(3,2,1); four
layers of [3, 4, 23, 3] Bottlenecks (1x1 -> 3x3 (stride, dilation, padding = dilation) -> 1x1 x4, BN after each, ReLU after
the first two and after the identity add; a 1x1 strided conv + BN on the identity when shape changes); layer2 strides by
2, layer3 / layer4 trade their stride for dilation 2 / 4, the first block of each keeping the previous dilation.

PINNED against independent code (tests/test_oracle_resnet_cpu.py), in two steps, since no third-party dilated ResNet is installed:
  1. with `dilate=(False, False, False)` this function IS the plain ResNet-101 (v1.5: the stride sits in the 3x3 convolution), and
     equals `transformers.ResNetModel` (HuggingFace's implementation, installed here) on the same state dict, key names mapped;
  2. trading a layer's stride for dilation is exact: the dilated network's map sampled at every 2nd (layer3) / 4th (layer4) position
     equals the strided network's map (the atrous identity), which ties the shipped `dilate=(False, True, True)` form to step 1.
torchvision itself (its weights' key names, which `TVDeeplabRes101Encoder` exposes) stays un-run: what is pinned is the architecture.
"""
import torch.nn.functional as F

LAYERS = (3, 4, 23, 3)


def _bn(sd, pre, x, eps=1e-5):
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"], False,
                        0.0, eps)


def encoder(x, sd, pre="", layers=LAYERS, dilate=(False, True, True), head=True):
    """[B,3,H,W] -> [B,256,H/8,W/8] (TVDeeplabRes101Encoder.forward with low_level=False, use_aspp=False).
    dilate: torchvision's `replace_stride_with_dilation` for layer2 / layer3 / layer4; head=False: the trunk's 2048-channel map."""
    b = pre + "backbone."
    x = F.relu(_bn(sd, b + "bn1.", F.conv2d(x, sd[b + "conv1.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    dilation = 1
    for li, n in enumerate(layers):
        stride = 1 if li == 0 else 2
        prev = dilation
        if li >= 1 and dilate[li - 1]:
            dilation *= stride
            stride = 1
        for i in range(n):
            p = f"{b}layer{li + 1}.{i}."
            s, d = (stride, prev) if i == 0 else (1, dilation)
            idn = x
            y = F.relu(_bn(sd, p + "bn1.", F.conv2d(x, sd[p + "conv1.weight"])))
            y = F.relu(_bn(sd, p + "bn2.", F.conv2d(y, sd[p + "conv2.weight"], stride=s, padding=d, dilation=d)))
            y = _bn(sd, p + "bn3.", F.conv2d(y, sd[p + "conv3.weight"]))
            if p + "downsample.0.weight" in sd:
                idn = _bn(sd, p + "downsample.1.", F.conv2d(x, sd[p + "downsample.0.weight"], stride=s))
            x = F.relu(y + idn)
    return F.conv2d(x, sd[pre + "localconv.weight"]) if head else x
