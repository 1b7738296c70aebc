"""centrifuge_amd — MI355X-native Centrifuge classification path.

The product is libcentrifuge_amd.so (HIP kernels + C ABI, include/centrifuge_amd.h)
and the drop-in `centrifuge-class` front end (centrifuge_amd/bin), both built
from centrifuge_amd/csrc.  This package only binds the C ABI for tests and bench.
"""
from . import capi  # noqa: F401
