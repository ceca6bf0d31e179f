from .snap import Snap, to_coeff_string, to_param_string  # noqa: F401
