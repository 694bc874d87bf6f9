"""aidadsp-lv2_amd — MI355X-native hot path of AIDA-X's rt-neural-generic plugin.

The product is the C-ABI shared library ``lib/libaidax_hip.so`` (declared in
``include/aidax.h``) and the LV2 shell ``lv2/rt-neural-generic.so`` that calls it.
This Python package is only a thin ctypes binding over that C ABI for tests and
``bench.py``; it adds no compute of its own and has no CPU fallback.

The directory name carries a hyphen, so import it with
``importlib.import_module("aidadsp-lv2_amd")``.
"""
from .binding import (  # noqa: F401
    AidaxError, Controls, StreamDsp, Hub, Model, ModelInfo, Pool, lib, lib_path, default_controls,
    biquad_design, db_to_coeff, lpf_fc, declared_symbols, device_count, pick_device, pick_hub, many_streams_form,
    ALL_STREAMS, START_WARMUP, START_RESET,
)
from . import workloads  # noqa: F401,E402
