"""Parameter records of the reference's galaxy classes (galaxy_classes.py:5-66), kept so that code
written against them keeps working: same constructor arguments in the same positions, same attributes,
same str().  They hold no arithmetic; unlike the reference, importing this module builds nothing and
prints nothing (galaxy_classes.py:68-69 does both)."""


class Galaxy:
    """galaxy_classes.py:5-23: Galaxy(morphology, sampling, N_particles, M_tot=None, seed=None).  The total
    mass defaults to the particle count; str() is the file stem the generators use,
    "{morphology}_{sampling}_{seed}"."""

    def __init__(self, morphology, sampling, N_particles, M_tot=None, seed=None):
        if M_tot is None:
            M_tot = N_particles
        self.morphology = morphology
        self.sampling = sampling
        self.seed = seed
        self.N_particles = N_particles
        self.M_tot = M_tot

    def __str__(self) -> str:
        return f"{self.morphology}_{self.sampling}_{self.seed}"

    def __repr__(self) -> str:
        fields = ", ".join(f"{k}={v!r}" for k, v in vars(self).items())
        return f"{type(self).__name__}({fields})"

    def __eq__(self, other):
        return type(other) is type(self) and vars(other) == vars(self)


class DiskGalaxy(Galaxy):
    """galaxy_classes.py:25-42: DiskGalaxy(morphology, sampling, N_particles, Rd, z0, M_tot=None, seed=None) --
    Rd and z0 come fourth and fifth, before the total mass.  Rd = radial scale length, z0 = vertical scale
    height.  (The reference's defaults for Rd and z0 are the type `float` itself, galaxy_classes.py:31-32, an
    evident slip for an annotation; they are kept so that a call that omits them sees what it saw there.)"""

    def __init__(self, morphology, sampling, N_particles, Rd=float, z0=float, M_tot=None, seed=None):
        super().__init__(morphology, sampling, N_particles, M_tot, seed)
        self.Rd = Rd
        self.z0 = z0

    def galaxy_type(self):
        return self.morphology


def generate_galaxy(morphology, sampling, N_particles, Rd=None, z0=None, M_tot=None, seed=None):
    """galaxy_classes.py:55-66: only "disk" is known; anything else raises ValueError."""
    if morphology == "disk":
        return DiskGalaxy("disk", sampling, N_particles, Rd, z0, M_tot, seed)
    raise ValueError(f"Unknown morphology: {morphology!r}")
