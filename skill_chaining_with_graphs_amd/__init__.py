"""Import shim: `skill-chaining-with-graphs_amd/` (the package directory the layout calls for) is not a
valid Python identifier, so this package of the importable name re-exports it in place."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "skill-chaining-with-graphs_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
