"""CPU: pins oracle/resnet.py (the dilated ResNet-101 trunk of BASELINE config 1's coarse model; torchvision is absent) against
independent code, in the two steps its header states: (1) the un-dilated form equals HuggingFace's `transformers.ResNetModel` on the
same weights, (2) the dilated form sampled at the strided positions equals the un-dilated form (trading stride for dilation is exact)."""
import pytest
import torch

LAYERS = (3, 4, 23, 3)


def _synth_trunk_sd(seed=1234):
    """torchvision-named state dict of the ResNet-101 trunk + localconv, from the product's own module (same keys as
    deeplabv3_resnet101().backbone; tests/test_alpnet_gpu.py loads it with strict=True)."""
    from protosam_amd.backbone import TVDeeplabRes101Encoder
    from protosam_amd.synth import synth_state_dict
    return synth_state_dict(TVDeeplabRes101Encoder(False), seed)


def _to_hf(sd):
    """torchvision's key names -> transformers.ResNetModel's."""
    out = {}
    for k, v in sd.items():
        if not k.startswith("backbone."):
            continue
        k = k[len("backbone."):]
        parts = k.split(".")
        if parts[0] == "conv1":
            nk = "embedder.embedder.convolution." + parts[1]
        elif parts[0] == "bn1":
            nk = "embedder.embedder.normalization." + parts[1]
        else:
            stage, blk = int(parts[0][5:]) - 1, int(parts[1])
            base = f"encoder.stages.{stage}.layers.{blk}."
            if parts[2] == "downsample":
                nk = base + "shortcut." + ("convolution." if parts[3] == "0" else "normalization.") + parts[4]
            else:
                j = int(parts[2][-1]) - 1
                nk = base + f"layer.{j}." + ("convolution." if parts[2].startswith("conv") else "normalization.") + parts[3]
        out[nk] = v
    return out


def test_undilated_oracle_equals_huggingface_resnet101():
    transformers = pytest.importorskip("transformers")
    from oracle import resnet as ores
    sd = _synth_trunk_sd()
    cfg = transformers.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=list(LAYERS),
                                    layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False,
                                    downsample_in_bottleneck=False)
    hf = transformers.ResNetModel(cfg).eval()
    missing, unexpected = hf.load_state_dict(_to_hf(sd), strict=False)
    assert not unexpected and all("num_batches_tracked" in m for m in missing), (missing[:4], unexpected[:4])
    x = torch.randn((2, 3, 96, 128), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = hf(x).last_hidden_state                                            # [2, 2048, 3, 4]: output stride 32
        got = ores.encoder(x, sd, dilate=(False, False, False), head=False)
    assert got.shape == ref.shape == (2, 2048, 3, 4)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"oracle ResNet-101 (no dilation) vs transformers.ResNetModel: max err / max |ref| {err:.2e}")
    assert err < 1e-5


def test_dilation_is_the_strided_network_sampled_densely():
    """The shipped form (layer3 / layer4 dilated: output stride 8) at every 4th position == the plain ResNet-101 (output stride 32);
    and each conversion separately. Exact up to fp32 summation order."""
    from oracle import resnet as ores
    sd = _synth_trunk_sd()
    x = torch.randn((1, 3, 128, 96), generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        plain = ores.encoder(x, sd, dilate=(False, False, False), head=False)        # [1,2048,4,3]
        d4 = ores.encoder(x, sd, dilate=(False, False, True), head=False)            # layer4 dilated: stride 16
        d34 = ores.encoder(x, sd, dilate=(False, True, True), head=False)            # the shipped form: stride 8
        shipped = ores.encoder(x, sd)                                                # + localconv
    assert d4.shape == (1, 2048, 8, 6) and d34.shape == (1, 2048, 16, 12) and shipped.shape == (1, 256, 16, 12)
    scale = plain.abs().max().item()
    e4 = (d4[..., ::2, ::2] - plain).abs().max().item() / scale
    e34 = (d34[..., ::4, ::4] - plain).abs().max().item() / scale
    print(f"dilated layer4 sampled /2 vs plain: {e4:.2e}; dilated layer3+4 sampled /4 vs plain: {e34:.2e}")
    assert e4 < 1e-5 and e34 < 1e-5
    head = torch.nn.functional.conv2d(d34, sd["localconv.weight"])
    assert torch.equal(head, shipped)
