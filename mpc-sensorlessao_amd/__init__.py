"""MI355X-native fastMPC inner solver behind the `Fast_MPC2` interface of
jinsungkim96/MPC-SensorlessAO (Fast_MPC/VAR_2/Fast_MPC2.m).  The compute path is the HIP library
behind include/fastmpc.h; nothing here solves on the CPU."""
from ._lib import (FastMPCError, FMPC_OK, FMPC_W_LINESEARCH, FMPC_E_NULL, FMPC_E_DIM,
                   FMPC_E_UNSUPPORTED, FMPC_E_NOT_PD_PHI, FMPC_E_NOT_PD_SCHUR, FMPC_E_HIP,
                   FMPC_E_ALLOC, FMPC_E_NO_DEVICE, FMPC_PATH_GENERIC, FMPC_PATH_WAVE, FMPC_PATH_SHARED,
                   FMPC_PATH_PANEL, FMPC_PATH_RAMP, LIB_PATH, load)
from .handle import FastMPCHandle
from .fast_mpc2 import Fast_MPC2, Fast_MPC2_VAR1, deinterleave
from . import synthetic
from .sharded import ShardedFastMPC, shard_range
from .closed_loop import ClosedLoop, AOLoop
from .lanes import SolveLanes
from .recorded import RecordedSolves
from .var_identify import identify_var2_device
from .estimator import PhaseDiversityEstimator
from . import _lib

__all__ = ["FastMPCHandle", "Fast_MPC2", "Fast_MPC2_VAR1", "deinterleave", "FastMPCError",
           "ShardedFastMPC", "shard_range", "ClosedLoop", "AOLoop", "SolveLanes", "RecordedSolves", "PhaseDiversityEstimator", "synthetic", "load", "LIB_PATH",
           "identify_var2_device"]
