"""minddet.models: registry-style model build surface (reference stub minddet/models/__init__.py:1-3;
pattern from minddet/models/centerpoint/det3d_ms/models/{builder,registry}.py)."""
from minddet_amd import graphs  # noqa: F401  (registers the modules)
from minddet_amd.config import Config  # noqa: F401
from minddet_amd.registry import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, READERS, ROI_HEAD, SECOND_STAGE,  # noqa: F401
                                  Registry, build_backbone, build_detector, build_from_cfg, build_head, build_neck,
                                  build_roi_head)
