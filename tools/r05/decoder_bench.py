"""Round 5: the prompt encoder + mask decoder stage alone, as a 16-slice step runs it: 27 prompt sets (nine tokens each) on 16 image
embeddings. us per call of MaskDecoder.predict_masks_tokens; PSAM_T2I_ALL / PSAM_UPSCALE_MFMA = 0 select the round-1 kernels (A/B:
run the script twice; PSAM_T2I_SPLIT=0: token-to-image attention unsplit).   [N_IMG=1 P_SETS=2] python tools/r05/decoder_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protosam_amd.segment_anything import sam_model_registry
from protosam_amd.synth import synth_state_dict
dev = torch.device("cuda:0")
sam = sam_model_registry["vit_b"](encoder_depth=0)
sam.load_state_dict(synth_state_dict(sam, 1234), strict=True)
sam = sam.to(dev).eval()
dec, pe = sam.mask_decoder, sam.prompt_encoder
g = torch.Generator().manual_seed(3)
n_img, P, T = int(os.environ.get("N_IMG", 16)), int(os.environ.get("P_SETS", 27)), 9
feat = torch.randn((n_img, 4096, 256), generator=g).to(dev)
tokens = torch.randn((P, T, 256), generator=g).to(dev)
iop = (torch.arange(P) * n_img // P).to(torch.int32).to(dev)
pk = pe._packed()
with torch.no_grad():
    for _ in range(3):
        masks, iou, _ = dec.predict_masks_tokens(feat, pk["pe_tok"], tokens, pk["no_mask"], img_of_prompt=iop)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        masks, iou, _ = dec.predict_masks_tokens(feat, pk["pe_tok"], tokens, pk["no_mask"], img_of_prompt=iop)
    e1.record()
    torch.cuda.synchronize()
print(f"predict_masks_tokens, {P} prompt sets x {T} tokens on {n_img} images: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us per call "
      f"(PSAM_T2I_SPLIT={os.environ.get('PSAM_T2I_SPLIT', '1')} PSAM_T2I_ALL={os.environ.get('PSAM_T2I_ALL', '1')} PSAM_UPSCALE_MFMA={os.environ.get('PSAM_UPSCALE_MFMA', '1')}); "
      f"checksum masks {masks.double().sum().item():.6e} iou {iou.double().sum().item():.6e}")
