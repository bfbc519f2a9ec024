"""Import shim: the product package lives in ``lrp-imagecaptioning_amd/`` (a
directory name Python cannot import directly because of the hyphen).  This
shim makes ``import lrp_imagecaptioning_amd`` resolve to that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "lrp-imagecaptioning_amd")
__path__[:] = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
