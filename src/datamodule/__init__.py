"""src.datamodule (reference: src/datamodule/__init__.py:3-10)."""
from phantom_vlb_amd.datamodule import VLB_Dataset, VLBDataModule, VLBDataModuleConfig, VLBDatasets

__all__ = ["VLBDataModule", "VLBDataModuleConfig", "VLB_Dataset", "VLBDatasets"]
