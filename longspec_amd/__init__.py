"""longspec_amd -- MI355X-native draft-then-verify decode path of LongSpec.

Hand-written HIP kernels (gfx950) behind a C ABI (include/longspec_hip.h), driven
from PyTorch-ROCm through the reference's own Python API (`LlamaGlide`,
`tree_spec_generate`, ...).  See DESIGN.md.
"""
__version__ = "0.1.0"
