"""ctypes binding of libpm.so (include/pm.h).  There is no CPU fallback: a missing
library or a failing call raises."""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PM_LIBRARY names another build of the same library (e.g. a tuning variant)
LIB_PATH = os.environ.get("PM_LIBRARY") or os.path.join(os.path.dirname(_HERE), "lib", "libpm.so")

PM_F32, PM_F64 = 0, 1
MESH_DENSITY, MESH_POTENTIAL, MESH_GX, MESH_GY, MESH_GZ, SLAB_ZSPECTRUM, SLAB_MESH_ZSPECTRUM, SLAB_MESH_POTENTIAL = range(8)
FLAG_TIMING, FLAG_FUSED_GRADIENT, FLAG_ROCFFT_ONLY, FLAG_NGP, FLAG_F64, FLAG_FULL_SORT = 1, 2, 4, 8, 32, 64
IC_KINDS = {"uniform": 0, "two_galaxy": 1, "disk": 2}
STAGES = ("bin", "deposit", "fft_fwd", "green", "fft_inv", "gradient", "gather")
KERNELS = ("sort_pass", "reorder", "deposit_rows", "col_y_fwd", "col_x_green", "col_y_inv", "row_inv", "gather")

_c_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double

# name -> (restype, argtypes); every symbol include/pm.h declares
SIGNATURES = {
    "pm_create": (_c_p, [_int, _dbl, _i64, _int, ctypes.c_uint]),
    "pm_destroy": (None, [_c_p]),
    "pm_set_particles": (_int, [_c_p, _c_p, _int, _c_p, _int, _c_p, _i64]),
    "pm_generate": (_int, [_c_p, _int, _i64, ctypes.c_uint64, _c_p, _int]),
    "pm_generate_galaxy": (_int, [_int, _int, _i64, _c_p, _int, ctypes.c_uint64, _c_p, _c_p, _c_p, _c_p]),
    "pm_deposit": (_int, [_c_p]),
    "pm_potential": (_int, [_c_p]),
    "pm_set_softening": (_int, [_c_p, _dbl]),
    "pm_gradient": (_int, [_c_p]),
    "pm_forces": (_int, [_c_p]),
    "pm_gather_kick_drift": (_int, [_c_p, _dbl]),
    "pm_step": (_int, [_c_p, _dbl, _int]),
    "pm_run": (_int, [_c_p, _dbl, _int, _c_p, _int, _c_p, _c_p]),
    "pm_host_alloc": (_c_p, [ctypes.c_size_t]),
    "pm_host_free": (None, [_c_p]),
    "pm_get_particles": (_int, [_c_p, _c_p, ctypes.POINTER(_int), _c_p, ctypes.POINTER(_int)]),
    "pm_get_forces": (_int, [_c_p, _c_p]),
    "pm_get_mesh": (_int, [_c_p, _int, _c_p]),
    "pm_set_mesh": (_int, [_c_p, _int, _c_p]),
    "pm_get_cell_keys": (_int, [_c_p, _c_p, _c_p]),
    "pm_energies": (_int, [_c_p, ctypes.POINTER(_dbl), ctypes.POINTER(_dbl)]),
    "pm_energies_noalloc": (_int, [_c_p, _c_p]),
    "pm_deposit_ngp": (_int, [_c_p]),
    "pm_forces_ngp": (_int, [_c_p]),
    "pm_get_mesh_f64": (_int, [_c_p, _int, _c_p]),
    "pm_set_mesh_f64": (_int, [_c_p, _int, _c_p]),
    "pm_potential_f64": (_int, [_c_p]),
    "pm_gradient_f64": (_int, [_c_p]),
    "pm_forces_ngp_f64": (_int, [_c_p]),
    "pm_deposit_f64": (_int, [_c_p]),
    "pm_forces_f64": (_int, [_c_p]),
    "pm_get_forces_f64": (_int, [_c_p, _c_p]),
    "pm_create_slab": (_c_p, [_int, _dbl, _i64, _int, ctypes.c_uint, _int, _int]),
    "pm_create_slab_windowed": (_c_p, [_int, _dbl, _i64, _int, ctypes.c_uint, _int, _int, _int]),
    "pm_slab_set_window": (_int, [_c_p, _c_p, _int, _int]),
    "pm_slab_window": (_int, [_c_p, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "pm_slab_rekey": (_int, [_c_p]),
    "pm_slab_copy_planes": (_int, [_c_p, _int, _int, _int, _int, _int]),
    "pm_slab_plane_counts": (_int, [_c_p, _c_p]),
    "pm_generate_plane_counts": (_int, [_int, _int, _dbl, _int, _i64, ctypes.c_uint64, _c_p, _int, _c_p]),
    "pm_slab_set_particles": (_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _i64]),
    "pm_slab_deposit": (_int, [_c_p]),
    "pm_slab_planes": (_int, [_c_p, _int, _int, _int, _c_p, _int]),
    "pm_slab_fft2": (_int, [_c_p, _int]),
    "pm_slab_pack": (_int, [_c_p, _c_p]),
    "pm_slab_unpack": (_int, [_c_p, _c_p]),
    "pm_slab_solve_x": (_int, [_c_p, _c_p]),
    "pm_slab_gather_kick_drift": (_int, [_c_p, _dbl]),
    "pm_slab_outbox": (_int, [_c_p, _c_p, _c_p, _i64]),
    "pm_slab_inbox": (_int, [_c_p, _c_p, _i64]),
    "pm_slab_outbox_counts": (_int, [_c_p, _c_p]),
    "pm_slab_outbox_records": (_int, [_c_p, _c_p, _c_p, _i64]),
    "pm_slab_get_particles": (_int, [_c_p, _c_p, _c_p, _c_p]),
    "pm_slab_pitch": (_int, [_c_p]),
    "pm_slab_forward_planes": (_int, [_c_p, _c_p, _int, _int]),
    "pm_slab_inverse_planes": (_int, [_c_p, _c_p, _int, _int]),
    "pm_stage_times": (_int, [_c_p, _c_p, _c_p]),
    "pm_kernel_times": (_int, [_c_p, _c_p, _c_p]),
    "pm_last_binning": (_int, [_c_p, _c_p]),
    "pm_set_flags": (_int, [_c_p, ctypes.c_uint]),
    "pm_set_stream": (_int, [_c_p, _c_p]),
    "pm_synchronize": (_int, [_c_p]),
    "pm_num_particles": (_i64, [_c_p]),
    "pm_last_error": (ctypes.c_char_p, [_c_p]),
}

_lib = None


class PMError(RuntimeError):
    pass


def load():
    """Load libpm.so once and declare every prototype."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PMError(f"{LIB_PATH} is missing: build it with `make -C nbody-simulation_amd` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
        # PyTorch ships a HIP runtime of its own, and the first one loaded serves the whole process: torch does not
        # find the GPU behind the system's runtime, while this library works on either.  So where torch is installed
        # it is imported first (PM_NO_TORCH=1 skips that for processes that never use torch).
        if "torch" not in sys.modules and not os.environ.get("PM_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


PM_ERANGE = 6


def check(ctx, rc):
    if rc != 0:
        msg = load().pm_last_error(ctx)
        text = f"libpm error {rc}: {msg.decode() if msg else '?'}"
        if rc == PM_ERANGE:
            raise IndexError(text)          # what the reference's NumPy indexing raises
        raise PMError(text)


class _PinnedBlock:
    """Owner of one pm_host_alloc allocation; freed when the last array over it is gone."""

    def __init__(self, nbytes):
        self.lib = load()
        self.address = self.lib.pm_host_alloc(nbytes)
        if not self.address:
            raise MemoryError(f"pm_host_alloc({nbytes}): " + self.lib.pm_last_error(None).decode())

    def __del__(self):
        address, self.address = getattr(self, "address", None), None
        if address:
            try:
                self.lib.pm_host_free(address)
            except Exception:       # interpreter shutdown: the library may be gone already, and so is the process's memory
                pass


def pinned_empty(shape, dtype=np.float32):
    """np.empty in page-locked host memory (pm_host_alloc): copies from the device into it run beside kernels.
    The memory lives as long as the array or any view of it."""
    dtype = np.dtype(dtype)
    nbytes = max(int(np.prod(shape)) * dtype.itemsize, 1)
    block = _PinnedBlock(nbytes)
    buf = (ctypes.c_char * nbytes).from_address(block.address)
    buf._block = block                      # the ctypes object is the base of every array made from it
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def ptr(a):
    return None if a is None else a.ctypes.data_as(_c_p)


def as_particle_array(a, name):
    """C-contiguous (N,3) float32/float64 view or copy; other dtypes are widened to float64."""
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError(f"{name} must have shape (N, 3), got {a.shape}")
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return np.ascontiguousarray(a)
