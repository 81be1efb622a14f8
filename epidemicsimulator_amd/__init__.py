"""epidemicsimulator_amd -- MI355X (gfx950) implementation of the per-timestep Citizen update
loop of NoSuchThingAsRandom/EpidemicSimulator's `sim` crate, behind the C ABI of include/esim.h.

Only what that path needs lives here: `csrc/` (HIP kernels + C ABI + synthetic populations),
and the host-side mirror of the reference's `Simulator` interface.
"""
from . import _lib
from ._lib import EsimError, default_params
from .population import Population, PRESETS
from .simulator import Simulator, StatisticsRecorder, RECORD_DTYPE

__all__ = ["Simulator", "StatisticsRecorder", "Population", "PRESETS", "EsimError",
           "default_params", "RECORD_DTYPE", "_lib"]
