"""dsp_amd: MI355X-native STFT power-spectrogram path (framing, windowing, real FFT, power) behind the
reference's C entry points.  The compute lives in libdsp_stft_hip.so (hand-written HIP, gfx950);
this package is the ctypes front end plus the signal-sharding helper for multi-GPU runs."""
from .api import (DspError, apply_win, fft, fft_power, fft_power_dev, fft_r2hc, force_generic, frame_univar,  # noqa: F401
                  frame_univar_flt, kernel_family, nextpow2, num_frames, set_flt_mode, stft, stft_batch, stft_cplx_dev, stft_dev,
                  stft_flt, stft_flt_batch, stft_flt_dev, stft_flt_multi, stft_multi, stft_multi_dev, shard_range, pow_compress, pow_compress_dev, get_stft_freqs, stft_pc_batch, stft_pc_dev, window, window_dev, window_univar, window_univar_flt, convert_freqs)
from .windows import WINDOWS, blackmanharris, hamming, hann, povey  # noqa: F401

__all__ = ["DspError", "apply_win", "fft", "fft_power", "fft_power_dev", "fft_r2hc", "force_generic", "frame_univar",
           "frame_univar_flt", "kernel_family", "nextpow2", "num_frames", "set_flt_mode", "stft", "stft_batch",
           "stft_cplx_dev", "stft_dev", "stft_flt", "stft_flt_batch", "stft_flt_dev", "stft_flt_multi", "stft_multi", "stft_multi_dev", "shard_range", "pow_compress", "pow_compress_dev", "get_stft_freqs", "stft_pc_batch", "stft_pc_dev", "window", "window_dev", "window_univar", "window_univar_flt", "convert_freqs", "WINDOWS", "blackmanharris", "hamming", "hann", "povey"]
