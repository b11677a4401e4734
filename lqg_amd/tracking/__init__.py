from lqg_amd.tracking.basic import (
    BoundedActor,
    OptimalActor,
    RelativeObservationBoundedActor,
)
from lqg_amd.tracking.subjective import SubjectiveActor
from lqg_amd.tracking.point_mass import PointMassBoundedActor

__all__ = [
    "BoundedActor",
    "OptimalActor",
    "RelativeObservationBoundedActor",
    "SubjectiveActor",
    "PointMassBoundedActor",
]
