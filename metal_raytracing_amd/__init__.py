"""Import alias: `metal-raytracing_amd/` (hyphenated, named after the reference repository) is not a
valid Python identifier, so this package points its search path at that directory and re-exports it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "metal-raytracing_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
