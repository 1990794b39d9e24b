"""ctypes binding of ``libprotosam_hip.so`` (the C-ABI declared in ``include/protosam_hip.h``).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing, or a kernel
launcher returns a non-zero status, this module raises. ``torch`` is imported first so that the HIP
runtime bundled with PyTorch-ROCm (SONAME ``libamdhip64.so.7``) is the one the kernels run on; streams
and device pointers handed across the boundary then belong to a single runtime.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below: loads PyTorch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSAM_LIB_PATH") or os.path.join(_HERE, "libprotosam_hip.so")   # (override: A/B of library builds)

c_void_p, c_int, c_float, c_longlong, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong, ctypes.c_size_t

# name -> argtypes; every entry point returns int (0 = ok). Kept in the same order as the header.
SIGNATURES = {
    "psam_gemm_f16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p],
    "psam_gemm_f16_heads": [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p],
    "psam_gemm_f16_ln": [c_void_p] * 6 + [c_int] * 12 + [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "psam_ln_finalize": [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p],
    "psam_gemm_splitk_ranges": [c_int, c_int, c_int],
    "psam_gemm_f16_splitk_ln": [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int,
                                c_void_p],
    "psam_gemm_set_tile": [c_int],
    "psam_gemm_asm_variant": [c_int],
    "psam_gemm_set_option": [ctypes.c_char_p, c_int],
    "psam_gemm_set_workspace": [c_void_p, c_size_t],
    "psam_layernorm": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                       c_int, c_int, c_void_p],
    "psam_attention_f16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                           c_float, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "psam_attention_set_variant": [c_int],
    "psam_attention_fused_relpos": [c_int] * 6,
    "psam_relpos": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_float, c_int, c_void_p],
    "psam_alp_bank": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float,
                      c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "psam_alp_sim": [c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_float, c_float,
                     c_void_p, c_void_p, c_int, c_void_p],
    "psam_patchify_bilinear": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "psam_bilinear_nchw": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "psam_resize2d": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "psam_bilinear_tokens": [c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "psam_prob_argmax": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "psam_broadcast_rows": [c_void_p, c_int, c_void_p, c_int, c_longlong, c_longlong, c_void_p],
    "psam_minmax": [c_void_p, c_int, c_longlong, c_void_p, c_void_p],
    "psam_im2col3x3": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "psam_cast_f16": [c_void_p, c_void_p, c_longlong, c_void_p],
    "psam_cast_f32": [c_void_p, c_void_p, c_longlong, c_void_p],
    "psam_gelu_f32": [c_void_p, c_longlong, c_void_p],
    "psam_split_f16": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p],
    "psam_small_linear": [c_void_p] * 6 + [c_int] * 4 + [c_longlong] * 4 + [c_int] * 3 + [c_void_p],
    "psam_small_attention": [c_void_p] * 4 + [c_int] * 10 + [c_void_p],
    "psam_t2i_attention": [c_void_p] * 4 + [c_int] * 5 + [c_void_p],
    "psam_t2i_attention_split": [c_void_p] * 4 + [c_int] * 6 + [c_void_p] * 2,
    "psam_gemm_f32": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p],
    "psam_gemm_f32_heads": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p],
    "psam_gemm_f32x3": [c_void_p, c_void_p, c_int] + [c_void_p] * 5 + [c_int] * 8 + [c_float, c_void_p],
    "psam_small_linear_splitk": [c_void_p] * 6 + [c_int] * 6 + [c_void_p],
    "psam_ln_pe": [c_void_p] * 8 + [c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p],
    "psam_dense_pe": [c_void_p, c_int, c_int, c_void_p, c_void_p],
    "psam_prompt_tokens": [c_void_p] * 5 + [c_int, c_int, c_float, c_void_p, c_void_p],
    "psam_upscale_tail": [c_void_p] * 7 + [c_int, c_int, c_void_p],
    "psam_mask_upsample": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "psam_mask_union": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p],
    "psam_mask_stats": [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                        c_void_p, c_void_p],
    "psam_im2col": [c_void_p] + [c_int] * 10 + [c_void_p, c_void_p],
    "psam_im2col_stem": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "psam_maxpool3x3s2": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "psam_rotate_nearest": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                            c_int, c_void_p],
    "psam_resize_aa": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "psam_volume_stats": [c_void_p, c_int, ctypes.c_longlong, c_float, c_float, c_void_p, c_void_p],
    "psam_volume_slices": [c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, c_int, c_int,
                           c_void_p, c_void_p],
    "psam_neg_points": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p],
    "psam_mask_downscale": [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p],
    "psam_plane_stats": [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p],
    "psam_mask_binarize": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                           c_void_p, c_void_p],
    "psam_normalize_chw": [c_void_p, c_int, c_int, c_longlong, ctypes.POINTER(c_float), ctypes.POINTER(c_float),
                           c_void_p, c_void_p],
    "psam_ccl": [c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 9 + [c_void_p],
    "psam_ccl_batch": [c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int] + [c_void_p] * 9 + [c_void_p],
    "psam_sam_patchify": [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float),
                          c_int, c_void_p, c_void_p, c_void_p],
}

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def _preload_torch_hip_runtime():
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)


def lib():
    """Return the loaded library, loading it on first use. Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: build it with `make -C protosam_amd/csrc` (or __graft_entry__.build()). "
            "protosam_amd has no CPU fallback.")
    _preload_torch_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is missing -> loud
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = L
    return L


def check(status, name):
    if status != 0:
        raise RuntimeError(f"{name} failed with status {status} (1 = bad argument, 2 = launch error)")
