"""Where does the heavy-tailed SAM fixture (synth.heavy_tail_sam_) lose parity? HIP image encoder vs the CPU oracle's on one image,
per truncated depth and per ingredient of the fixture.   python tools/heavy_probe.py [vit_b|vit_h] [depths, e.g. 1,2,4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import glue, sam_image_encoder as oenc
from protosam_amd.segment_anything import sam_model_registry
from protosam_amd.synth import heavy_tail_sam_, synth_state_dict, synth_pair
dev = torch.device("cuda:0")
mt = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
depths = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]
torch.set_num_threads(min(os.cpu_count() or 1, 32))
_, _, q_img, _ = synth_pair(512, seed=0)
img = (q_img[0].permute(1, 2, 0).numpy() * 40 + 128).clip(0, 255).astype("uint8")
import numpy as np
img = np.ascontiguousarray(np.kron(img, np.ones((2, 2, 1), dtype=np.uint8)))          # 1024 x 1024 x 3
x = glue.sam_preprocess(img)
for parts in ((), ("scale",), ("massive",), ("student",), ("scale", "massive", "student")):
    for depth in depths:
        sam = sam_model_registry[mt](encoder_depth=depth)
        sd = synth_state_dict(sam, 1234)
        heavy_tail_sam_(sd, 1234, parts=parts)
        sam.load_state_dict(sd)
        sam = sam.to(dev).eval()
        with torch.no_grad():
            ref = oenc.image_encoder(x, {k: v.float() for k, v in sd.items()}, model_type=mt, depth=depth)     # [1,256,64,64]
            got = sam.image_encoder(x.to(dev))
            xres = sam.image_encoder._ws[1]["x"].float().cpu()
        e = (got.cpu() - ref).abs()
        print(f"{mt} parts={'+'.join(parts) or 'none':22s} depth {depth}: embedding max err {e.max():.3e} mean {e.mean():.3e} (|ref| max {ref.abs().max():.2f}), "
              f"residual stream |x| max {xres.abs().max():.1f} finite {bool(torch.isfinite(xres).all())}", flush=True)
        del sam
