"""ycnr_als -- Python loader and host-side mirror for the MI355X ALS path.

The product is libycnr_als.so (HIP, ../../csrc) behind the C ABI of include/ycnr_als.h; this
package only binds it (ctypes) and mirrors the reference's trainer interface for the bench
and the parity tests.  The NodeJS host lives in ../../lib/emf and ../../addon.
"""
from . import _lib
from ._lib import YcnrError
from .trainer import (AlsDevice, als_calc_portion, pin_fixed_factors, unpin_fixed_factors, release_portion_state, rmse_portion, split_to_sets,
                      rating_stats, recommend_items)

__all__ = ["AlsDevice", "als_calc_portion", "pin_fixed_factors", "unpin_fixed_factors", "release_portion_state", "rmse_portion", "split_to_sets", "rating_stats", "recommend_items", "YcnrError", "_lib"]
