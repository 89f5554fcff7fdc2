"""Import alias: ``combinatorial-rl-tasks_amd/`` (the package directory the build contract
names) is not a valid Python identifier, so this stub re-roots itself onto that directory.
All code lives there; nothing is implemented here."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "combinatorial-rl-tasks_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
