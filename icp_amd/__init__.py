"""icp_amd: MI355X (gfx950) backend for the registration path of alex-van-vliet/icp.

The product is the C-ABI shared library ``icp_amd/lib/libicp_hip.so`` (sources in
``icp_amd/csrc``, interface in ``include/icp_hip.h``).  This package is the Python host-side
mirror of the reference's operator interface for that path:

    icp(m, p, iterations, threshold, vp_threshold) -> (4x4 transform, moved p)

(reference: libgpu/libgpu/icp.hh:19-21, libcpu/libcpu/icp.hh:25-27).
"""
from .capi import Context, FlatTree, IcpHipError, center_host, load, mean_host  # noqa: F401
from .icp import closest, icp  # noqa: F401
