"""Drop-in import surface of the reference's ``src`` package (src/__init__.py:3-13), backed by
phantom_vlb_amd (HIP kernels).  ``from src import HRFConvolveLayer, RidgeRegressionLayer, ...`` works."""
from phantom_vlb_amd.utils import HRFConvolveLayer, LogValAccuracyCallback, RidgeRegressionLayer, get_hrf_weight

__all__ = ["HRFConvolveLayer", "RidgeRegressionLayer", "get_hrf_weight", "LogValAccuracyCallback"]
