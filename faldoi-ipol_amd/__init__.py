"""faldoi-ipol_amd -- MI355X (gfx950) implementation of FALDOI's variational hot path.

csrc/       hand-written HIP kernels + the C ABI (include/faldoi_hip.h) -> libfaldoi_hip.so
capi.py     ctypes binding of that ABI (what a host-language caller would write)
flo.py      .flo codec (the file format between the reference's executables)
build.py    in-tree hipcc build

The directory name follows the upstream project and is not a Python identifier; import it as
`faldoi_ipol_amd` (the loader shim faldoi_ipol_amd.py at the repository root does that).
"""
from . import batch, build, capi, flo, shard  # noqa: F401
from .capi import Context, FaldoiError, LocalTvl1, Tvl1Batch, default_params  # noqa: F401
