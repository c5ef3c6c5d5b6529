"""Host-side data pipeline with the reference's config surface (u2pl/dataset/*):
list formats, n_sup resampling, tensor-space transforms, DistributedSampler loaders.
Plain PIL / torch-CPU code: it feeds the GPU step and is not on the kernel path."""
from .builder import get_loader  # noqa: F401
