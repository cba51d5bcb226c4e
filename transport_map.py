"""
Drop-in module: ``from transport_map import *`` (as every example of
MaxRamgraber/Triangular-Transport-Toolbox does, e.g. example_01.py:12) gives the
MI355X-native ``transport_map`` class.
"""
import numpy as np  # noqa: F401  (the reference module exports np as well)

from triangular_transport_toolbox_amd.transport_map import transport_map  # noqa: F401

__all__ = ['transport_map', 'np']
