"""Quantisation method registry for the hot path (python/sglang/srt/layers/quantization/__init__.py:73-107)."""
from .awq import AWQConfig
from .base_config import QuantizationConfig
from .fp8 import Fp8Config
from .w8a8_fp8 import W8A8Fp8Config

QUANTIZATION_METHODS = {"fp8": Fp8Config, "w8a8_fp8": W8A8Fp8Config, "awq": AWQConfig}


def get_quantization_config(quantization: str):
    if quantization not in QUANTIZATION_METHODS:
        raise ValueError(f"Invalid quantization method: {quantization}. Available methods: {list(QUANTIZATION_METHODS)}")
    return QUANTIZATION_METHODS[quantization]
