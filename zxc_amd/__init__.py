"""zxc_amd — MI355X-native ZXC block decode and LZ77 block encode.

The product is ``libzxc_mi355x.so`` (HIP kernels + C host API, built by
``__graft_entry__.build()`` from ``zxc_amd/csrc``); this package is only the thin
ctypes view of its C-ABI used by tests and bench.py. There is no Python or CPU
decoder behind it: if the library or a GPU is missing, calls raise.
"""
from .api import (ZxcError, Seekable, compress, decompress, get_decompressed_size, decode_blocks_device,  # noqa: F401
                  lib, lib_path, JOB_DTYPE, error_name)
from . import api  # noqa: F401  (stream_* helpers live there)
