"""
Zero-edit drop-in for the reference's ``src/models`` package.

Put this directory in front of the reference's ``src`` on the module search path (and the repo root
behind it, so that ``textocvp_amd`` resolves):

    PYTHONPATH=/path/to/repo/dropin:/path/to/repo  python src/05_evaluate_predictor.py ...

``from models.SAVi import SAVi``, ``from models.Predictors.predictor_wrapper import PredictorWrapper`` and
every other ``models.*`` import of the reference's ``lib/setup_model.py:43-127``,
``base/basePredictorTrainer.py:20`` and ``data/*.py`` then resolves to the MODULE OBJECTS of
``textocvp_amd.models.*`` (one copy, same classes whichever name imported them), so ``lib/setup_model.py``
needs no edit.  Nothing is computed here: a ``sys.meta_path`` finder maps the names.
"""

import importlib
import importlib.abc
import importlib.machinery
import sys

_TARGET = "textocvp_amd.models"


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)        # the real module object, not a copy

    def exec_module(self, module):                          # already executed under its real name
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != __name__ and not fullname.startswith(__name__ + "."):
            return None
        real = _TARGET + fullname[len(__name__):]
        try:
            found = importlib.util.find_spec(real)
        except ModuleNotFoundError:
            found = None
        if found is None:
            return None
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(real),
                                              is_package=found.submodule_search_locations is not None)


import importlib.util  # noqa: E402

if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_real = importlib.import_module(_TARGET)
sys.modules[__name__] = _real                               # `import models` IS textocvp_amd.models
