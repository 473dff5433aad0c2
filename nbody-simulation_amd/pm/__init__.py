"""MI355X-native Particle-Mesh force engine: Python host over the C ABI of libpm.so."""
from ._lib import PMError, load
from .simulation import Simulation

__all__ = ["Simulation", "PMError", "load"]
