"""Drop-in `sige` package: put the directory that contains this package on
PYTHONPATH and `import sige` / `from sige.nn import Gather` resolve to sige_amd."""
import sige_amd.compat as _compat

_compat.install(force=True)  # (this stub IS the `sige` being imported)
