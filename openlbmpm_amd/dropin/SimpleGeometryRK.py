"""The module RKD2Q9.py:21 imports but the reference does not ship; same function as
ShanChen2D/SimpleGeometry.py:11-27."""
import numpy as np


def defineGeometry(xDomain, yDomain):
    void = np.ones([yDomain, xDomain], dtype=bool)
    solid = np.zeros([yDomain, xDomain], dtype=bool)
    void[10:-10, 0] = 0; void[10:-10, -1] = 0
    solid[10:-10, 0] = 1; solid[10:-10, -1] = 1
    return void, solid
