"""eigentrajectory_amd -- MI355X-native (gfx950) EigenTrajectory SVD-descriptor path.

Same public surface as the reference package (EigenTrajectory/__init__.py:1-2
exports ``EigenTrajectory`` and ``TrajNorm``); the descriptor / anchor / k-means
modules are importable for the callers that use them directly.  All compute goes
through the C ABI of ``libetamd.so`` (include/eigentraj.h).
"""
from .model import EigenTrajectory
from .normalizer import TrajNorm
from .descriptor import ETDescriptor
from .anchor import ETAnchor
from .kmeans import BatchKMeans

__all__ = ["EigenTrajectory", "TrajNorm", "ETDescriptor", "ETAnchor", "BatchKMeans"]
