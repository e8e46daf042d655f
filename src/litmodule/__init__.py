"""src.litmodule (reference: src/litmodule/__init__.py:3-8)."""
from phantom_vlb_amd.litmodule import VLBLitModule, VLBLitModuleConfig

__all__ = ["VLBLitModuleConfig", "VLBLitModule"]
