"""covest_amd -- MI355X (gfx950) implementation of CovEst's likelihood grid-search
hot path behind the reference's own model API.  See DESIGN.md."""
from .models import BasicModel, RepeatsModel, models, select_model  # noqa: F401
from .estimator import CoverageEstimator  # noqa: F401
from .grid import DenseGrid, dense_grid_argmin, optimize_grid, initial_grid  # noqa: F401

__version__ = "0.1.0"
