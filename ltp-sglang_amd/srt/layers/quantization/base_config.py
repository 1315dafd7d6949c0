"""Quantisation plugin ABCs (python/sglang/srt/layers/quantization/base_config.py:15-80)."""
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional

import torch


class QuantizeMethodBase(ABC):
    @abstractmethod
    def create_weights(self, layer: torch.nn.Module, *weight_args, **extra_weight_attrs):
        raise NotImplementedError()

    @abstractmethod
    def apply(self, layer: torch.nn.Module, *args, **kwargs) -> torch.Tensor:
        raise NotImplementedError()

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        return


class LinearMethodBase(QuantizeMethodBase):
    @abstractmethod
    def create_weights(self, layer, input_size_per_partition: int, output_partition_sizes: List[int], input_size: int,
                       output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        raise NotImplementedError()

    @abstractmethod
    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError()


class QuantizationConfig(ABC):
    def __init__(self):
        self.packed_modules_mapping: Dict[str, List[str]] = dict()

    @abstractmethod
    def get_name(self) -> str:
        raise NotImplementedError()

    @abstractmethod
    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        raise NotImplementedError()

    @classmethod
    @abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        raise NotImplementedError()

    @abstractmethod
    def get_quant_method(self, layer: torch.nn.Module, prefix: str) -> Optional[QuantizeMethodBase]:
        raise NotImplementedError()

    @classmethod
    def get_min_capability(cls) -> int:
        """CUDA compute capability gate of the reference (base_config.py:126-135); meaningless on gfx950: always satisfied."""
        return 0

    @staticmethod
    def get_config_filenames() -> List[str]:
        return []

    @classmethod
    def override_quantization_method(cls, hf_quant_cfg, user_quant) -> Optional[str]:
        return None

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        for key in keys:
            if key in config:
                return config[key]
        raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")

    @staticmethod
    def get_from_keys_or(config: Dict[str, Any], keys: List[str], default: Any) -> Any:
        try:
            return QuantizationConfig.get_from_keys(config, keys)
        except ValueError:
            return default

    def get_scaled_act_names(self) -> List[str]:
        return []
