"""covest_amd -- MI355X (gfx950) implementation of CovEst's likelihood grid-search
hot path behind the reference's own model API, and of the steps either side of it
(k-mer histogram, histogram down-sampling, result record).  See DESIGN.md."""
__version__ = "0.1.0"

from .models import BasicModel, RepeatsModel, models, select_model  # noqa: F401,E402
from .estimator import CoverageEstimator  # noqa: F401,E402
from .grid import DenseGrid, dense_grid_argmin, optimize_grid, initial_grid  # noqa: F401,E402
from .hist_steps import (load_histogram, save_histogram, process_histogram, sample_histogram,  # noqa: F401,E402
                         compute_coverage_apx)
from .report import print_output  # noqa: F401,E402
