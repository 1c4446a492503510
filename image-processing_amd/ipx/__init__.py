"""ipx -- MI355X-native backend for the reference's /api/process image hot path.

`ipx.ops`   tensor-level operators over libipx.so (HIP, gfx950)
`ipx.luts`  host-side 256-entry tables built with the reference's own numpy expressions
`ipx.app`   Flask app keeping server.py's request/response contract
`ipx.sharder` one-image-per-GPU batch sharder and the row-partitioned giant-image path
"""
from . import _lib  # noqa: F401
from ._lib import IpxError  # noqa: F401

__all__ = ["_lib", "IpxError"]
