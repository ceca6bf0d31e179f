"""Plugin discovery shared by the solver and calculator factories.

FitSNAP finds its plugins by walking ``__subclasses__()`` of a base class and comparing class names
case-insensitively with the name given in the input file (reference: fitsnap3lib/solvers/solver_factory.py:18-34 for
direct subclasses, fitsnap3lib/calculators/calculator_factory.py:11-38 for grandchildren).  Being imported is what
registers a plugin.  The object is allocated without running ``__init__`` first, exactly like the reference does, so
that ``__init__`` receives the name under which the plugin was requested."""


def _descendants(base, generation):
    level = [base]
    for _ in range(generation):
        level = [child for parent in level for child in parent.__subclasses__()]
    return level


def find_plugin(base, wanted, generation, family):
    """Uninitialised instance of the plugin class named ``wanted`` (case-insensitive), ``generation`` levels below
    ``base`` (1 = child, 2 = grandchild).  When several classes carry the name the last one registered wins, as in
    the reference's loop.  Unknown name -> IndexError with the reference's message."""
    key = str(wanted).lower()
    match = [cls for cls in _descendants(base, generation) if cls.__name__.lower() == key]
    if not match:
        raise IndexError("{} was not found in fitsnap {}".format(wanted, family))
    return base.__new__(match[-1])
