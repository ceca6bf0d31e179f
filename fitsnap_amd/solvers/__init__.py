"""HIP-backed linear solvers behind the reference's Solver plugin API
(fitsnap3lib/solvers/)."""
from .solver import Solver  # noqa: F401
from .svd import SVD  # noqa: F401
from .ridge import RIDGE  # noqa: F401
from .ard import ARD  # noqa: F401
from .anl import ANL  # noqa: F401
from .lasso import LASSO  # noqa: F401
from .solver_factory import solver, search  # noqa: F401
