"""nyx_amd — MI355X-native batched propagation path behind the interface of nyx-space/nyx's
Propagator / MonteCarlo.  The numeric path is the HIP library
``nyx_amd/libnyx_hip.so`` (C-ABI in include/nyx_hip.h); importing this package does not load it,
any compute call does and fails loudly if it is missing."""
from . import _abi  # noqa: F401
from .propagator import *  # noqa: F401,F403
from ._abi import SCHED_CALIBRATED, SCHED_EXPLICIT, SCHED_MODEL, Tuning  # noqa: F401,E402
from .mc import DispersedState, MonteCarlo, MvnSpacecraft, PropResult, Results, Run, StateDispersion, shard_bounds  # noqa: F401,E402
from .params import StateError, StateParameter, state_value  # noqa: F401,E402
from .rng import Pcg64Mcg  # noqa: F401,E402
from .od import KalmanODProcess, Predicted, ProcessNoise3D, predict_until  # noqa: F401,E402
