"""phantom_vlb_amd: the phantom_vlb fine-tuning hot path on MI355X (gfx950) HIP kernels.

Importing the package loads ``libvlb.so`` (built by ``make -C phantom_vlb_amd/csrc`` or
``__graft_entry__.build()``); there is no CPU fallback - a missing library is an ImportError.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is absent)

__all__ = ["_lib"]
