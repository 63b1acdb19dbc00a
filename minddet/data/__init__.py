"""minddet.data: input shapes / synthetic batches for the hot path (reference stub
minddet/data/__init__.py:1-3).  The reference's dataset pipelines are out of scope."""
from minddet_amd.data import synthetic_images  # noqa: F401
