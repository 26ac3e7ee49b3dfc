"""planarslam_amd — MI355X-native (gfx950) hot path of PlanarSLAM behind a C ABI.

Python here is thin plumbing (ctypes) over libplanar_hip.so; see include/planar_abi.h."""
from ._lib import KP_DTYPE, Context, PlanarError, lib  # noqa: F401
from .orb import ORBextractor  # noqa: F401
from .optimizer import Optimizer  # noqa: F401
from .matcher import LSDmatcher, ORBmatcher, distinctive_descriptors, hamming_knn  # noqa: F401
from .planes import PlaneClouds, PlaneDetection, SurfaceNormals, flag_matched_plane_points  # noqa: F401
from .ba import Communicator, HostedCommunicator, local_bundle_adjustment, shard_problem  # noqa: F401
