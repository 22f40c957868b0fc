"""metagraph_amd: host-side mirror of the DBGAligner interface over libmgx.so (HIP, gfx950)."""
# Load order: the torch wheel bundles its own copy of the HIP runtime.  A process that initialises /opt/rocm's copy first
# (through libmgx.so) and torch's second ends up with two runtimes, and the second one sees no GPUs.  Importing torch first
# makes libmgx.so bind to the runtime torch loaded, so both share one.  (torch here is plumbing for device tensors and
# torch.distributed; the C++ host tools never load it.)
# MGX_NO_TORCH=1 skips it for processes that will never touch torch (short hardware checks: the first `import torch` on a
# fresh box takes a minute or two).
import os as _os

if not _os.environ.get("MGX_NO_TORCH"):
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass
