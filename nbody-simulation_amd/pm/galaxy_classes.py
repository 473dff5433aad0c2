"""Parameter records of the reference's galaxy classes (galaxy_classes.py:5-66), kept so that code
written against them keeps working.  They hold no arithmetic; unlike the reference, importing this
module builds nothing and prints nothing."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class Galaxy:
    """galaxy_classes.py:5-23: total mass defaults to the particle count; str() is the file stem the
    generators use, "{morphology}_{sampling}_{seed}"."""
    morphology: str
    sampling: str
    N_particles: int
    M_tot: Optional[float] = None
    seed: Optional[int] = None

    def __post_init__(self):
        if self.M_tot is None:
            self.M_tot = self.N_particles

    def __str__(self) -> str:
        return f"{self.morphology}_{self.sampling}_{self.seed}"


@dataclass
class DiskGalaxy(Galaxy):
    """galaxy_classes.py:25-42: Rd = radial scale length, z0 = vertical scale height."""
    Rd: Optional[float] = None
    z0: Optional[float] = None

    def galaxy_type(self):
        return self.morphology


def generate_galaxy(morphology, sampling, N_particles, Rd=None, z0=None, M_tot=None, seed=None):
    """galaxy_classes.py:55-66: only "disk" is known; anything else raises ValueError."""
    if morphology == "disk":
        return DiskGalaxy("disk", sampling, N_particles, M_tot, seed, Rd, z0)
    raise ValueError(f"Unknown morphology: {morphology!r}")
