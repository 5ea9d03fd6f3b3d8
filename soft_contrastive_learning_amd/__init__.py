"""MI355X-native hot path of soft-contrastive visual localisation.

``model.nets`` / ``model.losses`` / ``pointnetvlad_cls`` mirror the reference's Python
interface for the path; everything under them runs in ``libscl_hip.so`` (hand-written
gfx950 kernels behind the C-ABI of ``include/scl_hip.h``).  Importing the package does
not load the library; the first op does, and raises if it was never built.
"""
__version__ = "0.1.0"


def install_as_learnlarge():
    """Make the reference's own import lines resolve to this backend: its scripts import
    ``learnlarge.model.nets``, ``learnlarge.model.losses``, ``learnlarge.util.cv`` / ``.io`` / ``.meta``
    (train/train.py:15-25, evaluation/inference.py:11-16, evaluation/top-n.py:8-10) and
    ``pointnetvlad.pointnetvlad_cls``.  Registers
    those names in ``sys.modules`` (a ``learnlarge`` package that is really installed is left
    alone and the call raises) and returns the alias package."""
    import importlib
    import sys
    import types
    if 'learnlarge' in sys.modules and not getattr(sys.modules['learnlarge'], '_scl_alias', False):
        raise RuntimeError("a real 'learnlarge' package is already imported")
    names = {'model': 'model', 'model.nets': 'model.nets', 'model.losses': 'model.losses',
             'util': 'util', 'util.cv': 'util.cv', 'util.io': 'util.io', 'util.meta': 'util.meta'}
    pkg = types.ModuleType('learnlarge')
    pkg._scl_alias = True
    pkg.__path__ = []                                   # a package: `import learnlarge.model.nets`
    sys.modules['learnlarge'] = pkg
    for alias, real in names.items():
        mod = importlib.import_module(__name__ + '.' + real)
        sys.modules['learnlarge.' + alias] = mod
        parent, _, leaf = alias.rpartition('.')
        setattr(sys.modules['learnlarge' + ('.' + parent if parent else '')], leaf, mod)
    # `from pointnetvlad.pointnetvlad_cls import triplet_loss, ...` (train/train.py:25); also top level
    cls = importlib.import_module(__name__ + '.pointnetvlad_cls')
    pn = types.ModuleType('pointnetvlad')
    pn._scl_alias = True
    pn.__path__ = []
    pn.pointnetvlad_cls = cls
    sys.modules.update({'pointnetvlad': pn, 'pointnetvlad.pointnetvlad_cls': cls, 'pointnetvlad_cls': cls})
    return pkg
