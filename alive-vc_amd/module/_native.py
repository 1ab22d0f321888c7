"""ctypes binding of libalive_vc.so (C ABI: include/alive_vc.h).

The product path has no CPU fallback: if the shared library is missing or a
tensor is not on a HIP device, these helpers raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALIVE_VC_LIB", os.path.join(os.path.dirname(_HERE), "libalive_vc.so"))   # override: kernel A/B builds

_lib = None


class AliveConv(C.Structure):
    _fields_ = [
        ("W", C.c_void_p), ("bias", C.c_void_p), ("X", C.c_void_p),
        ("N", C.c_int), ("Ci", C.c_int), ("Tin", C.c_int), ("Co", C.c_int), ("K_pad", C.c_int),
        ("KW", C.c_int), ("stride", C.c_int), ("dil", C.c_int), ("pad_left", C.c_int), ("pad_mode", C.c_int),
        ("Tout", C.c_int), ("up", C.c_int), ("act", C.c_int),
        ("post_add", C.c_void_p), ("ch_scale", C.c_void_p), ("residual", C.c_void_p), ("skip", C.c_void_p),
        ("Y", C.c_void_p), ("Z", C.c_void_p), ("film", C.c_void_p),
        ("film_rows", C.c_int), ("Lf", C.c_int), ("film_scale_row", C.c_int), ("film_shift_row", C.c_int),
        ("precision", C.c_int), ("Ci_pad", C.c_int),
        ("film_t0", C.c_int), ("film_f0", C.c_int), ("film_ld", C.c_int),
        ("Xp", C.c_void_p), ("Zp", C.c_void_p), ("Yp", C.c_void_p), ("yp_planes", C.c_int),
    ]


class AliveGemm(C.Structure):
    _fields_ = [
        ("W", C.c_void_p), ("bias", C.c_void_p), ("P", C.c_void_p),
        ("N", C.c_int), ("T", C.c_int), ("Ci", C.c_int), ("Co", C.c_int), ("planes", C.c_int), ("act", C.c_int),
        ("post_add", C.c_void_p), ("ch_scale", C.c_void_p), ("residual", C.c_void_p),
        ("Y", C.c_void_p), ("Pout", C.c_void_p),
        ("b_plane", C.c_int64), ("b_win", C.c_int64), ("b_row", C.c_int), ("b_cblk", C.c_int), ("b_blk", C.c_int64),
        ("arg_val", C.c_void_p), ("arg_idx", C.c_void_p),
        ("Y2", C.c_void_p), ("y_split", C.c_int),
        ("f16s", C.c_int), ("wscale", C.c_void_p), ("in_unscale", C.c_float), ("pout_scale", C.c_float),
    ]


_VP, _I, _I64, _F, _D, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/alive_vc.h one to one
PROTOTYPES = {
    "alive_last_error": (C.c_char_p, []),
    "alive_version": (_I, []),
    "alive_library_padded_rows": (_I64, [_I64]),
    "alive_library_pack": (_I, [_VP, _I64, _I, _VP, _VP, _VP, _VP]),
    "alive_knn_workspace_bytes": (_SZ, [_I64, _I64]),
    "alive_knn_workspace_bytes_strict": (_SZ, [_I64, _I64]),
    "alive_knn_workspace_bytes_fast": (_SZ, [_I64, _I64]),
    "alive_knn_search": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I64, _I64, _I, _VP, _VP, _VP, _VP]),
    "alive_library_fp8_bytes": (_SZ, [_I64]),
    "alive_library_rounding_bound": (_I, [_VP, _VP, _VP, _I64, _VP, _VP]),
    "alive_library_pack_lo": (_I, [_VP, _VP, _VP, _I64, _VP, _VP]),
    "alive_knn_search_strict": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _I64, _I64, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "alive_library_pack_fp8": (_I, [_VP, _I64, _VP, _VP]),
    "alive_knn_search_fp8": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _I64, _I64, _I, _VP, _VP, _VP, _VP]),
    "alive_knn_search_stats": (_VP, [_I, _I, _I64, _VP]),
    "alive_knn_search_timed": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I64, _I64, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "alive_knn_search_fp8_timed": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _I64, _I64, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "alive_library_pack_fp6": (_I, [_VP, _I64, _VP, _VP]),
    "alive_knn_rot_coordinates": (_I, []),
    "alive_knn_rot_leading": (_I, []),
    "alive_knn_rot_mixed": (_I, []),
    "alive_library_pack_fp8_rot": (_I, [_VP, _I64, _I64, _I64, _F, _VP, _VP]),
    "alive_knn_search_fp8_rot_timed": (_I, [_VP, _VP, _F, _I, _I, _VP, _VP, _VP, _VP, _I64, _I64, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "alive_knn_search_fp6": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _I64, _I64, _I, _VP, _VP, _VP, _VP]),
    "alive_knn_search_fp6_timed": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _I64, _I64, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "alive_dedup_pass": (_I, [_VP, _VP, _I64, _I, _D, _VP, _VP, _VP]),
    "alive_knn_merge_gather": (_I, [_VP, _VP, _I, _I, _D, _VP, _VP, _I, _I, _VP, _VP, _VP]),
    "alive_conv1d": (_I, [C.POINTER(AliveConv), _VP]),
    "alive_decoder_precision": (_I, [_I]),
    "alive_encoder_precision": (_I, [_I]),
    "alive_f16_saturations": (_I, [_I]),
    "alive_f16_saturations_clear": (_I, [_VP]),
    "alive_gelu_film": (_I, [_VP, _I, _I, _I, _VP, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "alive_planes_bytes": (_SZ, [_I64, _I, _I]),
    "alive_to_planes": (_I, [_VP, _I, _I, _I, _I, _VP, _VP]),
    "alive_gemm_planes": (_I, [C.POINTER(AliveGemm), _VP]),
    "alive_argmax_merge": (_I, [_VP, _VP, _I, _I64, _VP, _VP]),
    "alive_filter_block_small_weights": (_I, [_I]),
    "alive_filter_block_small": (_I, [_VP, _I, _I, _I, _VP, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "alive_filter_block_small_range": (_I, [_VP, _I, _I, _I, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "alive_filter_block64_weights": (_I64, []),
    "alive_filter_block64": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "alive_filter_block64_range": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "alive_filter_block64_range_fp16": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "alive_filter_block256_workspace_bytes": (_I64, [_I, _I]),
    "alive_filter_block256_fp16": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _I64, _VP]),
    "alive_filter_block64s_workspace_bytes": (_I64, [_I, _I]),
    "alive_filter_block64s_fp16": (_I, [_VP, _I, _I, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _I64, _VP]),
    "alive_filter_source_in": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "alive_filter_source_out": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP]),
    "alive_dwconv_norm": (_I, [_VP, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _VP, _I, _I, _I, _F, _VP, _VP]),
    "alive_dwconv_norm_planes": (_I, [_VP, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _VP, _I, _I, _I, _F, _I, _VP, _VP]),
    "alive_dwconv_norm_planes_f16s": (_I, [_VP, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _VP, _I, _I, _I, _F, _F, _VP, _VP]),
    "alive_channel_norm": (_I, [_VP, _I, _I, _I, _VP, _VP, _F, _VP, _VP]),
    "alive_argmax_channels": (_I, [_VP, _I, _I, _I, _VP, _VP]),
    "alive_oscillator_workspace_bytes": (_SZ, [_I, _I, _I]),
    "alive_oscillator": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _F, _I, _I, _VP, _VP, _VP, _VP]),
    "alive_oscillator_range": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _F, _I, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "alive_dft_basis_bytes": (_SZ, []),
    "alive_dft_basis": (_I, [_VP, _VP]),
    "alive_spectrogram_workspace_bytes": (_SZ, [_I, _I]),
    "alive_spectrogram": (_I, [_VP, _VP, _I, _I, _VP, _VP, _VP]),
    "alive_resample_taps": (_I, [_I, _I]),
    "alive_resample_length": (_I64, [_I64, _I, _I]),
    "alive_resample_filter": (_I, [_I, _I, _VP, _VP]),
    "alive_resample": (_I, [_VP, _I, _I, _I, _I, _VP, _F, _F, _VP, _I, _VP]),
    "alive_pcm16_to_float": (_I, [_VP, _I64, _VP, _VP]),
    "alive_float_to_pcm16": (_I, [_VP, _I64, _VP, _VP]),
    "alive_weight_count": (_I, [_I]),
    "alive_weight_name": (C.c_char_p, [_I, _I]),
    "alive_front_end_workspace_bytes": (_SZ, [_I, _I]),
    "alive_front_end": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP]),
    "alive_content_encoder_workspace_bytes": (_SZ, [_I, _I]),
    "alive_content_encoder": (_I, [_VP, _VP, _I, _I, _VP, _VP, _VP]),
    "alive_f0_estimate_workspace_bytes": (_SZ, [_I, _I]),
    "alive_f0_estimate": (_I, [_VP, _VP, _I, _I, _VP, _VP, _VP]),
    "alive_decoder_workspace_bytes": (_SZ, [_I, _I]),
    "alive_decoder_forward": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "alive_decoder_forward_range": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP]),
    "alive_pitch_transform": (_I, [_VP, _I, _I, _I, _F, _F, _F, _VP]),
}


def lib():
    """Load libalive_vc.so (built by `make -C alive-vc_amd/csrc` / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().alive_last_error().decode()
        raise ValueError(f"{what}: {msg}" if what else msg)


def ptr(t):
    """device pointer of a contiguous fp32/int32/bf16 HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libalive_vc operates on HIP device tensors only (got a CPU tensor)")
    if not t.is_contiguous():
        raise RuntimeError("libalive_vc needs contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


class Workspace:
    """Grow-only device scratch, one per (purpose, device, stream); never allocates inside a call once it has reached its
    steady-state size (hipGraph friendly).  Keyed by the current stream: window batches that run on side streams
    (module/pipeline.py) get scratch of their own."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes, device):
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf


def weight_names(model):
    L = lib()
    return [L.alive_weight_name(model, i).decode() for i in range(L.alive_weight_count(model))]


class WeightTable:
    """Array of device pointers in the order the .so expects, keeping the tensors alive."""

    def __init__(self, model, packed: dict):
        names = weight_names(model)
        missing = [n for n in names if n not in packed]
        if missing:
            raise KeyError(f"packed weights missing {missing[:4]}...")
        self.tensors = [packed[n] for n in names]
        for n, t in zip(names, self.tensors):
            if t.dtype not in (torch.float32, torch.bfloat16) or not t.is_contiguous() or not t.is_cuda:
                raise RuntimeError(f"packed weight {n} must be a contiguous fp32 / bf16 HIP tensor")
        self.array = (C.c_void_p * len(names))(*[t.data_ptr() for t in self.tensors])
        self.names = names

    def tensor(self, name):
        return self.tensors[self.names.index(name)]
