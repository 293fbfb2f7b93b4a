"""radio-observer_amd -- MI355X-native STFT / waterfall / bolid-scan hot path.

Layout:
  csrc/      hand-written gfx950 kernels + the C ABI (include/ro_stft.h)
  capi.py    ctypes binding of that C ABI (plumbing for tests / bench)
  build.py   in-tree hipcc build of libro_stft.so

The directory name follows the reference repo (`radio-observer`), so import it
with importlib:  ro = importlib.import_module("radio-observer_amd").
There is no CPU fallback: without libro_stft.so / a gfx950 device the compute
entry points raise.
"""
from . import build as _build  # noqa: F401
from . import capi  # noqa: F401
from .capi import (  # noqa: F401
    Stft, StftError, Bands, ScanRecord, PinnedArray, library, library_path, hip_runtimes, require_one_hip_runtime,
    clamp_overlap, fft_sample_rate, frequency_to_bin, bin_to_frequency, time_to_fft_samples,
    row_count, window_table, bins_supported, device_count, shard_rows, shard_samples, shard_max_rows,
    stitch_rows, ln_levels,
    RO_WINDOW_NUTTALL, RO_WINDOW_HANN, RO_WINDOW_CUSTOM, RO_IQ_F32, RO_IQ_I16, RO_IQ_F64,
    RO_PRECISION_F32, RO_PRECISION_F64,
)

def sharding():
    """time-chunk sharding helpers (radio-observer_amd/timeshard.py; imports torch lazily)"""
    import importlib
    return importlib.import_module(__name__ + ".timeshard")


__all__ = ["capi", "Stft", "StftError", "Bands", "ScanRecord", "library"]
