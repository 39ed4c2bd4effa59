"""Import alias: the package directory is ``slam-duckietown_amd/`` (not a valid Python identifier).

``import slam_duckietown_amd`` resolves submodules from that directory.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 "slam-duckietown_amd"))

from .ekf_bindings import (  # noqa: E402,F401
    EKF_pose_estimation, EkfConfig, EkfError, EkfSlam, build_library, device_count, library_path, load_library,
    predict, update,
)
from .frontend import associate, delta_phi, displacement  # noqa: E402,F401
from .replay import GpuBackend, ReplayResult  # noqa: E402,F401
from .node_adapter import EkfNodeAdapter  # noqa: E402,F401
from .replay import replay as replay_events  # noqa: E402,F401  (the submodule keeps the name `replay`)
