"""Drop-in entry point name of the reference (``python -m fitsnap3``, fitsnap3/__main__.py)."""
