# (no re-exports: import the submodule you need).  The one thing that must happen as early as possible -- before the HIP
# runtime initialises -- is the launch configuration the package is measured in; see _lib.py.
import os as _os

_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
