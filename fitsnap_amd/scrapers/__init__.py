"""Origin of the per-row weights (fitsnap3lib/scrapers/scrape.py:323-353)."""
from .weighting import apply_weighting  # noqa: F401
