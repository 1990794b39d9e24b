"""CPU oracle for the ProtoSAM per-query-slice hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / the timed CPU baseline. The
product (``protosam_amd``) never imports it and has no CPU fallback.

It is a plain fp32 PyTorch-CPU / numpy restatement of the reference algorithm, written functionally over
state dicts that use the reference's parameter names; every function cites the reference file:line it
follows (paths relative to /root/reference).

Pin status (see DESIGN.md "Oracle"):
  * ALP module, FewShotSeg.forward glue, SAM image encoder / prompt encoder / mask decoder /
    postprocess, ResizeLongestSide: PINNED against the reference's own modules imported in the build
    container (``oracle/validate_against_reference.py``), golden vectors committed under ``tests/golden``.
  * The ORCHESTRATION (round 2): the reference's ``ProtoSAM.forward`` (eight flag sets, an empty coarse mask,
    ``coarse_pred_only``), ``SamPredictor.set_image`` / ``predict``, ``ProtoMedSAM.forward``, ``cca`` /
    ``get_connected_components``, ``get_dice_iou_precision_recall`` and ``SamAutomaticMaskGenerator`` (layer-0 grid, crop
    layers, small-region removal) are executed end to end on CPU by the same script with the absent dependencies' primitives
    (cv2, torchvision.ops) injected from the restatements; ``oracle/glue.py`` / ``oracle/amg.py`` reproduce them bit for bit
    (masks exact, scores <= 1e-5). Records in ``tests/golden/reference_outputs.npz``; full-depth oracle records of
    configs 3 / 4 / 5 in ``tests/golden/fullsize_cfg*.npz`` (``oracle/make_fullsize_goldens.py``).
  * DINOv2 ViT (``facebookresearch/dinov2`` via torch.hub, absent from /root/reference, no network):
    PARITY UNPINNED against the hub code; restated from the public architecture and cross-checked
    against the independent ``transformers.Dinov2Model`` implementation at 518^2 (no interpolation) and at the sizes the
    configurations use, 504^2 and 1022^2 (bicubic position-embedding interpolation: the residual against the
    ``size=`` convention of transformers is stated by the validation script).
  * cv2.connectedComponentsWithStats (opencv-python 4.10.0.84, absent): PARITY UNPINNED against cv2;
    cross-checked against scipy.ndimage.label / center_of_mass. Label numbering is by raster order of each
    component's first pixel; downstream results are order-invariant (union of masks).
"""
