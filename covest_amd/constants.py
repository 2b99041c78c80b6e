"""Constants of the grid-search path (values of covest/constants.py:2-23 that the
path reads; nothing else of that module is mirrored)."""
from multiprocessing import cpu_count

GRID_DEPTH = 3              # covest/constants.py:2
INF = float('inf')          # :3
STEP = 1.1                  # :7
OPTIMIZATION_METHOD = 'L-BFGS-B'  # :8
INITIAL_GRID_COUNT = 20     # :10
INITIAL_GRID_STEP = 3       # :11
DEFAULT_ERR_SCALE = 1       # :12
DEFAULT_K = 21              # :13
DEFAULT_READ_LENGTH = 100   # :14
DEFAULT_MIN_SINGLECOPY_RATIO = 0.3  # :16
MAX_ERRORS = 8              # :20

try:
    DEFAULT_THREAD_COUNT = cpu_count()  # :23-28 (accepted and ignored by the GPU path)
except NotImplementedError:
    DEFAULT_THREAD_COUNT = 2
