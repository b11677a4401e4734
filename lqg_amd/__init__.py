"""lqg_amd — MI355X-native batched LQG / inverse-optimal-control solve path with the call surface of
RothkopfLab/lqg (`import lqg_amd as lqg`).  Compute happens only in lqg_amd/csrc/liblqg_hip.so
(hand-written HIP kernels for gfx950 behind the C ABI of include/lqg_hip.h)."""
__version__ = "0.1.0"

from lqg_amd.spec import LQGSpec
from lqg_amd.system import LQG, Actor, Dynamics, System
from lqg_amd.tracking.basic import (
    BoundedActor,
    OptimalActor,
    RelativeObservationBoundedActor,
)
from lqg_amd.tracking.subjective import SubjectiveActor
from lqg_amd.tracking.point_mass import PointMassBoundedActor

__all__ = [
    "LQG", "Actor", "Dynamics", "System", "LQGSpec",
    "BoundedActor",
    "OptimalActor",
    "RelativeObservationBoundedActor",
    "SubjectiveActor",
    "PointMassBoundedActor",
]
