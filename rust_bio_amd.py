"""Import shim: the package directory is `rust-bio_amd/` (a hyphen is not importable), so this
module becomes the package `rust_bio_amd` by pointing __path__ at it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "rust-bio_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
