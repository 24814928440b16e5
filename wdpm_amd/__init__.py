"""wdpm_amd — MI355X-native WDPM water-redistribution path."""
from .capi import (ADD, SUBTRACT, DRAIN, MODULES, KERNEL_AUTO, KERNEL_PASS, KERNEL_FUSED, OPT_SIGNED_ZERO_SAFE, OPT_DEM32,  # noqa: F401
                   HALO_AUTO, HALO_RCCL, HALO_PEER, HALO_HOST, HALO_NAMES,
                   Context, Lib, WdpmError, load, load_hip, HIP_LIB_PATH)
