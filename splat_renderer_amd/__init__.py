"""splat_renderer_amd — MI355X-native tile-raster hot path of ath92/splat-renderer.

HIP kernels (csrc/) behind the C ABI in include/splat.h, plus host classes that keep the
reference's Renderer/Camera/PointManager/SplatPropertyManager + stage-class surface (host.py).
Importing the package does not need a GPU; creating a Device does (there is no CPU fallback).
"""
from . import scene  # noqa: F401
from ._lib import (MODE_FRONT_TO_BACK, MODE_REFERENCE_LITERAL, STAGE_BIN, STAGE_COMPOSITE, STAGE_NAMES,  # noqa: F401
                   STAGE_PROJECT, STAGE_SORT, CompositeCfg, SplatError)
from .camera import Camera  # noqa: F401
from .host import (Buffer, CommandEncoder, ComputeShaderRenderer, DepthKeyExtractor, Device, GPUTileBinner,  # noqa: F401
                   PerTileSorter, PipelinedRenderer, PointManager, PrefixSumScanner, PropertyPlanes, RadixSorter, Renderer, SequentialRenderer,
                   SplatProjector, SplatPropertyManager, TileRenderer)
from .frameloop import FrameLoop, MouseEvent, OrbitCameraController, SdfSplatSource, read_png, write_png  # noqa: F401
from . import sdf  # noqa: F401
from .sdf import CurvatureSampler, GradientSampler, PositionUpdater, SDFScene  # noqa: F401
