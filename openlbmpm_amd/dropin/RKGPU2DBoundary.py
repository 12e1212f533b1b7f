"""Drop-in for the reference module RKCG2D/RKGPU2DBoundary.py: its sixteen boundary kernels as pre-built HIP kernels
(include/lbmpm_kernels.h, module tag `rkb`), callable as kernel[grid, block](...).  Twelve have the body of their
namesakes in AcceleratedRKGPU2D.py; four do not, and keep THIS module's semantics:
  ghostPointsConstantVelocityRK          10 arguments (no forceX / forceY)                          RKGPU2DBoundary.py:58
  calConstPressureLowerGPU               row test on the GRID index of the node, not the compact one              :414
  ghostPointsConstPressureLowerRK        likewise                                                                 :452
  constantVelocityZHBoundaryHigherNewRK  the retreating (blue) fluid copies its unknowns from the row above      :535
(The reference module itself has no imports -- `cuda` is an undefined name in it)"""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _runtime import export as _export  # noqa: E402

_export("rkb", globals())
