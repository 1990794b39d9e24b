"""protosam_amd: MI355X-native (gfx950) implementation of ProtoSAM's per-query-slice inference hot path.

Python mirrors of the reference's class API live here (ProtoSAM, ProtoMedSAM, SamWrapper, FewShotSeg,
MultiProtoAsConv, sam_model_registry, SamPredictor, ...); every arithmetic stage runs in hand-written HIP
kernels behind the C ABI declared in include/protosam_hip.h (libprotosam_hip.so). There is no CPU fallback.
"""
__version__ = "0.1.0"
