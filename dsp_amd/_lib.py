"""Loader of libdsp_stft_hip.so (ctypes).  There is no fallback: a missing or unloadable library raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSP_AMD_LIB selects another build of the same library (kernel A/B runs)
LIB_PATH = os.environ.get("DSP_AMD_LIB") or os.path.join(_HERE, "libdsp_stft_hip.so")

_sz, _int, _vp = C.c_size_t, C.c_int, C.c_void_p
_f, _d = C.c_float, C.c_double

# every symbol include/dsp_stft.h declares: name -> (restype, argtypes)
_STFT = [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _int, _int, _int, _int]
_FRAME = [_vp, _vp, _sz, _sz, _sz, _int]
_AW = [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _int]
_AWI = [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _int]
_R2HC = [_vp, _vp, _sz, _sz, _sz, _sz, _int, _sz, _sz, _int]
_BDEV = [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _int, _int, _int, _int, _vp]
_BHOST = _BDEV[:-1]

SIGNATURES = {
    "stft_s": (_int, _STFT), "stft_d": (_int, _STFT),
    "stft_flt_s": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _f, _f, _int, _int, _int]),
    "stft_flt_d": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _d, _d, _int, _int, _int]),
    "fft_power_c": (_int, [_vp, _vp, _sz]), "fft_power_z": (_int, [_vp, _vp, _sz]),
    "frame_univar_s": (_int, _FRAME), "frame_univar_d": (_int, _FRAME),
    "frame_univar_c": (_int, _FRAME), "frame_univar_z": (_int, _FRAME),
    "frame_univar_flt_s": (_int, [_vp, _vp, _sz, _sz, _sz, _f, _f]),
    "frame_univar_flt_d": (_int, [_vp, _vp, _sz, _sz, _sz, _d, _d]),
    "frame_univar_flt_c": (_int, [_vp, _vp, _sz, _sz, _sz, _f, _f]),
    "frame_univar_flt_z": (_int, [_vp, _vp, _sz, _sz, _sz, _d, _d]),
    "apply_win_s": (_int, _AW), "apply_win_d": (_int, _AW), "apply_win_c": (_int, _AW), "apply_win_z": (_int, _AW),
    "apply_win_inplace_s": (_int, _AWI), "apply_win_inplace_d": (_int, _AWI),
    "apply_win_inplace_c": (_int, _AWI), "apply_win_inplace_z": (_int, _AWI),
    "fft_fftw_r2hc_s": (_int, _R2HC), "fft_fftw_r2hc_d": (_int, _R2HC),
    "fft_fftw_s": (_int, _R2HC), "fft_fftw_d": (_int, _R2HC), "fft_fftw_c": (_int, _R2HC), "fft_fftw_z": (_int, _R2HC),
    "dsp_stft_num_frames": (_sz, [_sz, _sz, _sz, _int]),
    "dsp_set_device": (_int, [_int]),
    "dsp_stft_release_buffers": (_int, []),
    "dsp_stft_flt_set_mode": (_int, [_int]),
    "dsp_stft_batch_dev_s": (_int, _BDEV), "dsp_stft_batch_dev_d": (_int, _BDEV),
    "dsp_stft_batch_dev_noalloc_s": (_int, _BDEV),
    "dsp_stft_cplx_batch_dev_s": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _int, _int, _vp]),
    "dsp_stft_cplx_batch_dev_d": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _int, _int, _vp]),
    "dsp_shard_range": (None, [_sz, _sz, _sz, _vp, _vp]),
    "dsp_stft_multi_s": (_int, _BHOST + [_vp, _int]), "dsp_stft_multi_d": (_int, _BHOST + [_vp, _int]),
    "dsp_stft_multi_dev_s": (_int, [_vp, _vp, _vp, _vp] + _BDEV[3:-1] + [_vp, _int]),
    "dsp_stft_multi_dev_d": (_int, [_vp, _vp, _vp, _vp] + _BDEV[3:-1] + [_vp, _int]),
    "dsp_stft_batch_dev_strided_s": (_int, _BDEV[:6] + [_sz] + _BDEV[6:]), "dsp_stft_batch_dev_strided_d": (_int, _BDEV[:6] + [_sz] + _BDEV[6:]),
    "dsp_stft_flt_batch_dev_strided_s": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _f, _f, _int, _int, _int, _vp]),
    "dsp_stft_flt_batch_dev_strided_d": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _d, _d, _int, _int, _int, _vp]),
    "dsp_stft_flt_batch_dev_s": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _f, _f, _int, _int, _int, _vp]),
    "dsp_stft_flt_batch_dev_d": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _d, _d, _int, _int, _int, _vp]),
    "dsp_stft_batch_strided_s": (_int, _BHOST[:6] + [_sz] + _BHOST[6:]), "dsp_stft_batch_strided_d": (_int, _BHOST[:6] + [_sz] + _BHOST[6:]),
    "dsp_stft_flt_batch_strided_s": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _f, _f, _int, _int, _int]),
    "dsp_stft_flt_batch_strided_d": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _d, _d, _int, _int, _int]),
    "dsp_stft_batch_s": (_int, _BHOST), "dsp_stft_batch_d": (_int, _BHOST),
    "dsp_stft_flt_multi_s": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _f, _f, _int, _int, _int, _vp, _int]),
    "dsp_stft_flt_multi_d": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _d, _d, _int, _int, _int, _vp, _int]),
    "dsp_stft_flt_batch_s": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _f, _f, _int, _int, _int]),
    "dsp_stft_flt_batch_d": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _d, _d, _int, _int, _int]),
    "dsp_fft_power_dev_c": (_int, [_vp, _vp, _sz, _vp]), "dsp_fft_power_dev_z": (_int, [_vp, _vp, _sz, _vp]),
    "window_univar_s": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _int]), "window_univar_d": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _int]),
    "window_univar_flt_s": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _f, _f]), "window_univar_flt_d": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _d, _d]),
    "window_univar_c": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _int]), "window_univar_z": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _int]),
    "window_univar_flt_c": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _f, _f]), "window_univar_flt_z": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _d, _d]),
    "hann_s": (_int, [_vp, _sz, _sz]), "hann_d": (_int, [_vp, _sz, _sz]),
    "hamming_s": (_int, [_vp, _sz, _sz]), "hamming_d": (_int, [_vp, _sz, _sz]),
    "povey_s": (_int, [_vp, _sz, _sz]), "povey_d": (_int, [_vp, _sz, _sz]),
    "blackmanharris_s": (_int, [_vp, _sz, _sz]), "blackmanharris_d": (_int, [_vp, _sz, _sz]),
    "dsp_window_dev_s": (_int, [_vp, _int, _sz, _sz, _vp]), "dsp_window_dev_d": (_int, [_vp, _int, _sz, _sz, _vp]),
    "dsp_window_par_dev_s": (_int, [_vp, _int, _sz, _f, _int, _sz, _vp]), "dsp_window_par_dev_d": (_int, [_vp, _int, _sz, _d, _int, _sz, _vp]),
    "blackman_s": (_int, [_vp, _sz, _int, _sz]), "blackman_d": (_int, [_vp, _sz, _int, _sz]),
    "bartlett_s": (_int, [_vp, _sz, _sz]), "bartlett_d": (_int, [_vp, _sz, _sz]),
    "triangular_s": (_int, [_vp, _sz, _sz]), "triangular_d": (_int, [_vp, _sz, _sz]),
    "rectangular_s": (_int, [_vp, _sz, _sz]), "rectangular_d": (_int, [_vp, _sz, _sz]),
    "flattop_s": (_int, [_vp, _sz, _sz]), "flattop_d": (_int, [_vp, _sz, _sz]),
    "gauss_s": (_int, [_vp, _sz, _f, _sz]), "gauss_d": (_int, [_vp, _sz, _d, _sz]),
    "tukey_s": (_int, [_vp, _sz, _f, _sz]), "tukey_d": (_int, [_vp, _sz, _d, _sz]),
    "planck_s": (_int, [_vp, _sz, _f, _sz]), "planck_d": (_int, [_vp, _sz, _d, _sz]),
    "hann_c": (_int, [_vp, _sz, _sz]), "hann_z": (_int, [_vp, _sz, _sz]),
    "hamming_c": (_int, [_vp, _sz, _sz]), "hamming_z": (_int, [_vp, _sz, _sz]),
    "povey_c": (_int, [_vp, _sz, _sz]), "povey_z": (_int, [_vp, _sz, _sz]),
    "blackmanharris_c": (_int, [_vp, _sz, _sz]), "blackmanharris_z": (_int, [_vp, _sz, _sz]),
    "bartlett_c": (_int, [_vp, _sz, _sz]), "bartlett_z": (_int, [_vp, _sz, _sz]),
    "triangular_c": (_int, [_vp, _sz, _sz]), "triangular_z": (_int, [_vp, _sz, _sz]),
    "rectangular_c": (_int, [_vp, _sz, _sz]), "rectangular_z": (_int, [_vp, _sz, _sz]),
    "flattop_c": (_int, [_vp, _sz, _sz]), "flattop_z": (_int, [_vp, _sz, _sz]),
    "blackman_c": (_int, [_vp, _sz, _int, _sz]), "blackman_z": (_int, [_vp, _sz, _int, _sz]),
    "gauss_c": (_int, [_vp, _sz, _f, _sz]), "gauss_z": (_int, [_vp, _sz, _d, _sz]),
    "tukey_c": (_int, [_vp, _sz, _f, _sz]), "tukey_z": (_int, [_vp, _sz, _d, _sz]),
    "planck_c": (_int, [_vp, _sz, _f, _sz]), "planck_z": (_int, [_vp, _sz, _d, _sz]),
    "convert_freqs_s": (_int, [_vp, _sz, C.c_char_p, C.c_char_p]), "convert_freqs_d": (_int, [_vp, _sz, C.c_char_p, C.c_char_p]),
    "pow_compress_s": (_int, [_vp, _vp, _sz, _f, _f]), "pow_compress_d": (_int, [_vp, _vp, _sz, _d, _d]),
    "pow_compress_inplace_s": (_int, [_vp, _sz, _f, _f]), "pow_compress_inplace_d": (_int, [_vp, _sz, _d, _d]),
    "get_stft_freqs_s": (_int, [_vp, _sz, _sz, _f]), "get_stft_freqs_d": (_int, [_vp, _sz, _sz, _d]),
    "dsp_pow_compress_dev_s": (_int, [_vp, _vp, _sz, _f, _f, _vp]), "dsp_pow_compress_dev_d": (_int, [_vp, _vp, _sz, _d, _d, _vp]),
    "dsp_stft_pc_batch_dev_s": (_int, _BDEV[:-1] + [_f, _f, _vp]), "dsp_stft_pc_batch_dev_d": (_int, _BDEV[:-1] + [_d, _d, _vp]),
    "dsp_stft_pc_batch_s": (_int, _BHOST + [_f, _f]), "dsp_stft_pc_batch_d": (_int, _BHOST + [_d, _d]),
    "dsp_stft_kernel_family": (_int, [_sz, _sz]),
    "dsp_stft_force_generic": (_int, [_int]),
    "dsp_stft_version": (C.c_char_p, []),
}

_lib = None


def lib():
    """The loaded library with argtypes set.  Raises if it was not built or does not load."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C dsp_amd/csrc` "
                               "(or __graft_entry__.build()); dsp_amd has no CPU fallback")
        if not os.environ.get("DSP_AMD_NO_TORCH"):
            # share one HIP runtime with PyTorch when both live in a process: importing torch first
            # makes the loader resolve libamdhip64.so.7 to the copy torch already mapped
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib
