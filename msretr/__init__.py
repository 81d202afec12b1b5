"""Importable alias of the package directory `modern-search-engines-project_amd/` (whose name, fixed by
the project layout, is not a valid Python identifier).  `import msretr` gives that package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "modern-search-engines-project_amd")]
with open(_os.path.join(__path__[0], "__init__.py"), encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _os, _f
