/*
 * pm.h -- C ABI of libpm.so, the MI355X (gfx950) Particle-Mesh force engine.
 *
 * This is the drop-in boundary for the force/integration path of
 * MappaCrappa/Nbody-Simulation.  Each entry point names the reference
 * interface it replaces (file:line under the reference's root).  All
 * functions are extern "C", take plain pointers and sizes, return 0 on
 * success or a PM_E* code, and leave a message for pm_last_error().
 *
 * Ownership and threading: a context owns every device allocation, its HIP
 * stream and its FFT plans; the caller owns every host buffer.  One context
 * = one GPU = one host thread at a time (calls on one context must not
 * overlap; different contexts are independent).  Host buffers are read or
 * written synchronously: when a pm_set_* / pm_get_* call returns, the copy is
 * done.  The compute calls (pm_deposit ... pm_step) are asynchronous on the
 * context's stream; pm_synchronize() waits for them.
 *
 * Mesh arrays are C-order (G,G,G) float32, index [i][j][k] <-> (x,y,z), k
 * fastest (functions.py:22,61).  Particle arrays on the host are AoS (N,3).
 */
#ifndef PM_H
#define PM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pm_ctx pm_ctx;

/* error codes */
enum { PM_OK = 0, PM_EINVAL = 1, PM_ENOMEM = 2, PM_EHIP = 3, PM_EFFT = 4, PM_ESTATE = 5, PM_ERANGE = 6 };

/* host dtypes of particle arrays */
enum { PM_F32 = 0, PM_F64 = 1 };

/* meshes addressable by pm_get_mesh / pm_set_mesh; the PM_SLAB_* selectors are for pm_slab_planes only */
enum { PM_MESH_DENSITY = 0, PM_MESH_POTENTIAL = 1, PM_MESH_GX = 2, PM_MESH_GY = 3, PM_MESH_GZ = 4, PM_SLAB_ZSPECTRUM = 5,
       PM_SLAB_MESH_ZSPECTRUM = 6, PM_SLAB_MESH_POTENTIAL = 7 };

/* pm_create flags */
enum {
    PM_FLAG_TIMING = 1u,      /* record HIP events around every stage (pm_stage_times) */
    PM_FLAG_FUSED_GRADIENT = 2u, /* pm_step keeps intermediate meshes on chip where it can: the potential is differenced
                                    inside the gather (no gx,gy,gz), and at grid_size 512 or 1024 the deposit emits each density
                                    row already transformed along z (no density mesh) */
    PM_FLAG_NGP = 8u,         /* pm_create only: pm_step runs the nearest-grid-point branch (CIC_interpolation = False,
                                 main.py:9, 40-41), whose meshes and forces are float64 */
    PM_FLAG_ROCFFT_ONLY = 4u, /* pm_create only: 3-D rocFFT plans for every grid size (no hand-written column FFT) */
    PM_FLAG_F64 = 32u,        /* pm_create only: pm_step runs the cloud-in-cell branch with every array in float64 -- what
                                 main.py:50-60 computes when the mesh of functions.py:22, the forces of functions.py:102 and
                                 the positions after main.py:60 are float64.  The reference has no such switch: this is the
                                 build mode that is held to 1e-12 against a float64 restatement.  Particles must be given
                                 as PM_F64 positions and velocities; not tuned */
    PM_FLAG_FULL_SORT = 64u   /* the particles are brought into cell order by sorting all of them every time.  Without it
                                 a step in which few particles changed their cell (the drift counts them) puts the new
                                 order together from the old one -- the movers are sorted and merged in -- which gives
                                 the same order and is faster (for sets of 2^20 particles and more; the environment variable
                                 PM_REBIN_MIN, read by pm_create, sets another bound); the flag is there to compare the two */
};

/* stages reported by pm_stage_times, in this order */
enum { PM_STAGE_BIN = 0, PM_STAGE_DEPOSIT = 1, PM_STAGE_FFT_FWD = 2, PM_STAGE_GREEN = 3, PM_STAGE_FFT_INV = 4,
       PM_STAGE_GRADIENT = 5, PM_STAGE_GATHER = 6, PM_NSTAGES = 7 };

/* single kernel launches reported by pm_kernel_times, in this order: a pass of the radix sort (one per digit place),
 * the reorder into cell order, the deposit's row kernel, the column transforms along y (forward), along x (forward,
 * Green multiply, inverse) and along y (inverse) and the inverse row transform of the hand-written Poisson solve,
 * the gather (with kick, drift and the next cell keys) */
enum { PM_KERNEL_SORT_PASS = 0, PM_KERNEL_REORDER = 1, PM_KERNEL_DEPOSIT_ROWS = 2, PM_KERNEL_COL_Y_FWD = 3, PM_KERNEL_COL_X_GREEN = 4,
       PM_KERNEL_COL_Y_INV = 5, PM_KERNEL_ROW_INV = 6, PM_KERNEL_GATHER = 7, PM_NKERNELS = 8 };

/* Replaces the module constants grid_size / box_size (main.py:5-6) and the
 * per-call allocations of every reference function.  n_capacity is the
 * largest particle count the context will hold.  Returns NULL on failure
 * (pm_last_error(NULL) then describes why). */
pm_ctx *pm_create(int grid_size, double box_size, int64_t n_capacity, int device, unsigned flags);
void pm_destroy(pm_ctx *ctx);

/* Replaces the bare arrays positions / velocities / masses created at
 * main.py:25-33.  pos and vel are host AoS (n,3) of dtype PM_F32 or PM_F64
 * (the shipped default run has float64 velocities throughout and float64
 * positions until the first drift, main.py:33,58-60); vel may be NULL
 * (zeros), mass may be NULL (ones). */
int pm_set_particles(pm_ctx *ctx, const void *pos, int pos_dtype, const void *vel, int vel_dtype,
                     const float *mass, int64_t n);

/* Synthetic particle sets drawn on the device into the context (float32 positions and velocities, unit masses), in
 * place of pm_set_particles: the sets of 10^7 .. 10^9 particles the engine is measured on need not be built and
 * shipped from the host.  The distributions are those of the reference's generators, not their random streams:
 *   PM_IC_UNIFORM     positions uniform in the box.                                    params: sigma_v
 *   PM_IC_TWO_GALAXY  two Plummer spheres rho ~ (1 + r^2/a^2)^-5/2 (galaxy_generation_functions.py:267-288) truncated
 *                     at rmax, n/2 particles about centre1 and the rest about centre2, each moving with half of the
 *                     relative velocity (zero total momentum, functions.py:320-338).
 *                     params: a1, a2, rmax, centre1[3], centre2[3], relative_velocity[3], sigma_v
 *   PM_IC_DISK        exponential disk rho ~ exp(-R/Rd) exp(-|z|/z0): R ~ Gamma(2, Rd) truncated at Rmax, z ~
 *                     Laplace(0, z0) truncated at zmax, uniform azimuth (galaxy_generation_functions.py:97-136).
 *                     params: Rd, z0, Rmax, zmax, centre[3], sigma_v
 * Every set adds an isotropic Gaussian velocity of dispersion sigma_v per axis; positions are wrapped into the box as
 * main.py:60 wraps them.  Particle p draws from Philox4x32-10 with counter (p, purpose, attempt) and key seed, so the
 * set depends on (kind, n, seed, params) only.  nparams must be the count listed for the kind. */
enum { PM_IC_UNIFORM = 0, PM_IC_TWO_GALAXY = 1, PM_IC_DISK = 2 };
int pm_generate(pm_ctx *ctx, int kind, int64_t n, uint64_t seed, const double *params, int nparams);

/* generate_ellipse_importance(N_particles, a, b, c, M_tot, seed) and generate_diffuse_sphere_importance(N_particles, R,
 * M_tot, seed): galaxy_generation_functions.py:181-244 (with virial_ellipsoid, :9-94) and :267-383, computed on the
 * device in float64 -- importance weights normalised to the total mass, enclosed mass over the particles sorted by
 * radius, circular speeds, tangential directions.  No context: the call allocates what it needs on `device` and
 * frees it.  positions (n,3), velocities (n,3), masses (n): host arrays of doubles, written in the particles' order
 * (PM_GALAXY_DIFFUSE_SPHERE: the first int(0.5 n) are the stars, label "1", the rest dark matter, label "2").
 *   PM_GALAXY_ELLIPSE          params: a, b, c, M_tot, disp (the reference's default dispersion fraction is 0.1)
 *   PM_GALAXY_DIFFUSE_SPHERE   params: R, M_tot, stellar_fraction (0.25 in the reference), R_dm (390 in the reference)
 * normals: the standard normal deviates the generator consumes, in the order the reference draws them, or NULL.
 *   ellipse: (n,3) for the positions, then n and n for the in-plane dispersion (5 n doubles);
 *   diffuse sphere: (n,3) for the positions, stars first, then (n,3) for the directions (6 n doubles).
 * With the deviates NumPy's default_rng(seed) hands out, the result is the reference's galaxy for that seed (to the
 * rounding of sums taken in another order).  NULL: the deviates are drawn on the device (Philox4x32-10, counter =
 * particle index, key = seed): the same distribution, another stream. */
enum { PM_GALAXY_ELLIPSE = 0, PM_GALAXY_DIFFUSE_SPHERE = 1 };
int pm_generate_galaxy(int device, int kind, int64_t n, const double *params, int nparams, uint64_t seed, const double *normals,
                       double *positions, double *velocities, double *masses);

/* density_func(positions, grid_size, box_size, masses), the CIC branch:
 * functions.py:19-70, call site main.py:50. */
int pm_deposit(pm_ctx *ctx);
/* compute_potential(density, grid_size): functions.py:73-89, call site main.py:51.
 * Parity with the reference is by tolerance here, by construction: a real-to-complex transform (hand-written for G =
 * 128 .. 1024, rocFFT otherwise) stands in for pocketfft's complex one, 1/k^2 is a float32 reciprocal multiply folded
 * with the 1/G^3 of the inverse transform, and the mean that functions.py:81 subtracts is not subtracted (the k = 0
 * mode is set to zero either way, functions.py:86): the result agrees within 1e-5 of the largest |potential|, not bit
 * for bit.  Returns PM_ESTATE when the density mesh does not hold a deposit (see pm_get_mesh). */
int pm_potential(pm_ctx *ctx);
/* Optional Gaussian softening of the Poisson solve: the Green's function becomes 1/k^2 * exp(-(sigma k_phys)^2 / 2),
 * i.e. the density is smoothed with a Gaussian of standard deviation `length` (in the units of box_size) before the
 * potential is taken; k_phys = 2 pi k G / box_size for the k of functions.py:75.  The reference has no softening
 * (functions.py:85 divides by the bare k^2): length = 0, the default, is the reference's solve bit for bit, and any
 * other value leaves its results.  Applies to every solve of the context from the next call on (pm_potential, pm_step,
 * pm_run, the float64 solve of the NGP branch, pm_slab_solve_x). */
int pm_set_softening(pm_ctx *ctx, double length);
/* np.gradient(potential): call site main.py:54. */
int pm_gradient(pm_ctx *ctx);
/* force_func(potential_gradient, positions, grid_size, box_size), the CIC
 * branch: functions.py:99-180, call site main.py:55.  Result via pm_get_forces. */
int pm_forces(pm_ctx *ctx);
/* force_CIC fused with the in-line kick / drift / wrap of main.py:58-60. */
int pm_gather_kick_drift(pm_ctx *ctx, double dt);
/* nsteps passes of the loop body main.py:50-60 with all state resident. */
int pm_step(pm_ctx *ctx, double dt, int nsteps);
/* The whole main loop main.py:44-67: nsteps passes of the loop body, and after each of them
 *  - trajectory.append(positions.copy()) (main.py:36, 61): if trajectory is not NULL, the positions after step s, in
 *    the caller's original order, go to trajectory[s] of a host array (nsteps, n, 3) float32.  The copy of step s
 *    runs on a stream of its own beside step s + 1 (two snapshot buffers on the device), which hides it when the host
 *    array is page-locked (pm_host_alloc); into pageable memory the copies are correct but the host waits for each;
 *  - every energy_every-th step (step % energy_every == 0, main.py:64; 0 = never): KE and PE as pm_energies computes
 *    them go to ke[step / energy_every], pe[step / energy_every] (arrays of (nsteps - 1) / energy_every + 1 doubles).
 * Returns when everything has arrived. */
int pm_run(pm_ctx *ctx, double dt, int nsteps, float *trajectory, int energy_every, double *ke, double *pe);
/* Page-locked host memory for pm_run's trajectory (hipHostMalloc / hipHostFree).  No counterpart in the reference,
 * whose trajectory is a list of NumPy arrays; NULL on failure (message via pm_last_error(NULL)). */
void *pm_host_alloc(size_t bytes);
void pm_host_free(void *p);

/* Particle state in the caller's original order.  pos is written as float32
 * (n,3) once a drift has happened, in the dtype given to pm_set_particles
 * before; *pos_dtype / *vel_dtype report what was written. */
int pm_get_particles(pm_ctx *ctx, void *pos, int *pos_dtype, void *vel, int *vel_dtype);
int pm_get_forces(pm_ctx *ctx, float *forces /* (n,3) */);
/* A mesh can be read (and the stage that consumes it can run) once the stage that writes it has run or pm_set_mesh
 * has put it there; otherwise PM_ESTATE.  pm_step with PM_FLAG_FUSED_GRADIENT writes no gradient meshes and, on the
 * hand-written transforms, no density mesh: after it only PM_MESH_POTENTIAL can be read.  (Such a step transforms back
 * only the potential rows that its own gather reads; pm_get_mesh, pm_gradient and the energy calls have the other rows
 * transformed before they read the mesh -- one more kernel on the context's stream, no allocation -- so what they see is
 * the whole potential of functions.py:89.) */
int pm_get_mesh(pm_ctx *ctx, int which, float *out /* G*G*G */);
/* Upload a mesh, so that one stage can be run on the caller's input
 * (granular mode: compute_potential(density), np.gradient(potential),
 * force_CIC(gradient, ...) each take arrays the caller made). */
int pm_set_mesh(pm_ctx *ctx, int which, const float *in /* G*G*G */);

/* Cell order of the last binning: keys[r] = (i*G+j)*G+k of the r-th particle
 * in cell order (cell (i,j,k) as functions.py:27-39 computes it), perm[r] =
 * its index in the caller's original order.  Ties keep storage order. */
int pm_get_cell_keys(pm_ctx *ctx, int64_t *keys, int64_t *perm);
/* How the last binning got its order (a diagnostic of this engine; the reference has no cell order).  Returns 1 if it
 * was put together from the order of the step before, with *movers = the particles that had changed their cell; 0 if
 * the whole set was sorted, with *movers = the latest estimate of them that a drift has sent to the host (a sample
 * scaled up; -1 if there is none yet); negative on error.  Synchronises. */
int pm_last_binning(pm_ctx *ctx, int64_t *movers);

/* compute_kinetic_energy(velocities, masses) and compute_potential_energy(positions, masses,
 * potential, grid_size, box_size): functions.py:352-361, call sites main.py:65-66.  Uses the live
 * particle state and the potential mesh the context holds (in pm_step's flow: positions after the
 * drift, potential from before it, as the reference does).  Sums run in float64. */
int pm_energies(pm_ctx *ctx, double *ke, double *pe);

/* pm_energies for callers that cannot allocate or wait: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  It has no counterpart in the reference.  Same kernels and the same bits as pm_energies, but
 *  - the partial sums of the two reductions live in a file-scope array in device memory (16 KiB per GPU, held until
 *    the process ends) instead of a buffer the context allocated;
 *  - the results go to two doubles in DEVICE memory, device_ke_pe[0] = KE and device_ke_pe[1] = PE, written by the
 *    last kernels on the context's stream: the call returns without waiting and copies nothing to the host;
 *  - it calls no hipMalloc, hipFree or other allocating routine (the array's device address was looked up by
 *    pm_create).
 * At most PM_NOALLOC_MAX_PARTICLES particles (PM_EINVAL beyond; pm_energies has no such limit): the array holds 1024
 * partial sums per energy, one per block of 256 threads, so that each thread adds at most 1024 terms in sequence.
 * Every call on a GPU uses that GPU's one array, which keeps the last call's contents; each call overwrites every
 * entry it reads, so nothing needs clearing, but two calls that overlap on one GPU (two contexts on different
 * streams) race: order them. */
#define PM_NOALLOC_MAX_PARTICLES ((int64_t)1 << 28)
int pm_energies_noalloc(pm_ctx *ctx, double *device_ke_pe);

/* The nearest-grid-point branch (CIC_interpolation = False, main.py:9,40-41):
 * NGP(positions, grid_size, box_size, masses): functions.py:11-16 -- float64 mesh, read with
 * pm_get_mesh_f64(PM_MESH_DENSITY).  Returns PM_ERANGE where the reference raises IndexError (a
 * float32 position that wraps to exactly box_size).
 * force_NGP(potential_gradient, positions, grid_size, box_size): functions.py:92-97 -- the float32
 * gradient meshes the context holds, at the particle's cell; result via pm_get_forces. */
int pm_deposit_ngp(pm_ctx *ctx);
int pm_forces_ngp(pm_ctx *ctx);
int pm_get_mesh_f64(pm_ctx *ctx, int which, double *out /* G*G*G */);
/* The rest of that branch in float64, as NumPy runs it on NGP's float64 mesh:
 * pm_potential_f64: compute_potential(density, grid_size), functions.py:73-89, float64 density -> float64 potential;
 * pm_gradient_f64:  np.gradient(potential), main.py:54, on the float64 potential;
 * pm_forces_ngp_f64: force_NGP, functions.py:92-97, on the float64 gradient -> (n,3) float64 via pm_get_forces_f64.
 * pm_get_mesh_f64 / pm_set_mesh_f64 read and write these meshes (any PM_MESH_* selector).  A context created with
 * PM_FLAG_NGP runs exactly this chain in pm_step / pm_run, with main.py:58-60 behind it, and pm_energies reads its
 * float64 potential. */
int pm_potential_f64(pm_ctx *ctx);
int pm_gradient_f64(pm_ctx *ctx);
int pm_forces_ngp_f64(pm_ctx *ctx);
int pm_get_forces_f64(pm_ctx *ctx, double *forces /* (n,3) */);
int pm_set_mesh_f64(pm_ctx *ctx, int which, const double *in /* G*G*G */);
/* The cloud-in-cell calls of the float64 build mode (PM_FLAG_F64 contexts step through pm_deposit_f64 ->
 * pm_potential_f64 -> pm_gradient_f64 -> these forces with main.py:58-60 in float64 behind them):
 * pm_deposit_f64: CIC, functions.py:19-70, with a float64 mesh (read with pm_get_mesh_f64(PM_MESH_DENSITY));
 * pm_forces_f64:  force_CIC, functions.py:99-180, on the float64 gradient with float64 forces (pm_get_forces_f64).
 * Both need float64 positions and velocities (pm_set_particles with PM_F64, PM_F64), on any context. */
int pm_deposit_f64(pm_ctx *ctx);
int pm_forces_f64(pm_ctx *ctx);

/*
 * Slab decomposition of the same path over several GPUs.  The reference is a single-process NumPy
 * program and has no counterpart for these calls; each one is the part of a reference statement that
 * one slab can do alone, and the comments name that statement.  A slab context owns mesh planes
 * [plane0, plane0 + nplanes) (axis 0, i <-> x) and the particles whose cell (functions.py:27-39) has
 * its i in that range.  The caller moves what crosses slabs between the device buffers these calls
 * fill and read (see nbody-simulation_amd/pm/distributed.py for the sequence).  Particles are float32.
 * Device buffers are plain device pointers; a call that fills one returns when the data is there.
 */
pm_ctx *pm_create_slab(int grid_size, double box_size, int64_t n_capacity, int device, unsigned flags,
                       int plane0, int nplanes);
/* Particle ownership apart from the mesh slabs, for particle sets that do not spread evenly over the planes of the mesh
 * (two galaxies in mid-box: the fullest of eight equal slabs holds half of the particles).  The reference has one process
 * and no counterpart; what must stay as it is are the cells of functions.py:27-39, the wrap i1 = (i + 1) % G of
 * functions.py:56 and the positions after main.py:58-60.  A context made by pm_create_slab_windowed still transforms mesh
 * planes [plane0, plane0 + nplanes), but the particles it holds are those whose cell has its i in a WINDOW of planes that
 * pm_slab_set_window chooses, at most window_planes of them (nplanes <= window_planes <= grid_size - 3; needs
 * PM_FLAG_FUSED_GRADIENT on a grid size of the hand-written transforms when it is more than nplanes).  The windows of all
 * ranks partition [0, grid_size): cuts[r] is the first plane of rank r's window, cuts[ranks] = grid_size.  Then
 *   pm_slab_deposit writes the window's planes and the one above (pm_slab_planes PM_SLAB_ZSPECTRUM, planes 0 .. window);
 *     the caller sends each to the rank whose slab holds it, where it is copied or added into that slab's own
 *     z-transformed planes (PM_SLAB_MESH_ZSPECTRUM, nplanes planes) -- the window's plane first, then the plane above the
 *     window before it: a fixed order, so the sums have the same bits whatever the arrival order;
 *   the transforms run on the slab's planes as before and leave its potential planes (PM_SLAB_MESH_POTENTIAL);
 *   the caller brings the potential of the window's planes, the one below and the two above into PM_MESH_POTENTIAL
 *     (planes 0 .. window + 2), and pm_slab_gather_kick_drift sends a particle that leaves the window to the rank whose
 *     window holds its new cell.
 * Until pm_slab_set_window is called the window is the slab itself, and the two pairs of buffers are the same memory
 * when window_planes == nplanes (pm_create_slab).
 * pm_slab_rekey, after pm_slab_set_window on a context that holds particles: those outside the new window go to the
 * outbox like the leavers of a drift (pm_slab_outbox* and pm_slab_inbox follow).  pm_slab_plane_counts ADDS the number
 * of live particles in every mesh plane to grid_size uint64 counters in device memory (summed over the ranks, they are
 * what the caller cuts the windows by).  pm_slab_window reports the present window and the room for it. */
pm_ctx *pm_create_slab_windowed(int grid_size, double box_size, int64_t n_capacity, int device, unsigned flags,
                                int plane0, int nplanes, int window_planes);
int pm_slab_set_window(pm_ctx *ctx, const int *cuts, int ranks, int rank);
int pm_slab_window(const pm_ctx *ctx, int *plane0, int *nplanes, int *max_planes);
int pm_slab_rekey(pm_ctx *ctx);
/* The part of a window that lies in the context's own slab needs no exchange: to_mesh != 0 copies (add == 0) or adds
 * z-transformed planes [first, first + count) of the window's deposit into the slab's planes from mesh_first on;
 * to_mesh == 0 copies the slab's potential planes [mesh_first, mesh_first + count) to planes first ... of the window's
 * potential buffer. */
int pm_slab_copy_planes(pm_ctx *ctx, int to_mesh, int first, int mesh_first, int count, int add);
int pm_slab_plane_counts(pm_ctx *ctx, uint64_t *device_counts);
/* pm_generate for a slab context: every rank draws the whole set (a particle's draw depends on its number only) and
 * keeps the particles whose cell lies in its window, in the order of their numbers, which become their ids.
 * pm_generate_plane_counts needs no context: it draws the same set on `device` without keeping anything and writes the
 * number of particles in every mesh plane to host_counts (grid_size entries), so that every rank can compute the same
 * windows before it makes its context, without a collective (errors: pm_last_error(NULL)). */
int pm_generate_plane_counts(int device, int grid_size, double box_size, int kind, int64_t n, uint64_t seed, const double *params, int nparams,
                             int64_t *host_counts);
/* main.py:25-33 for one slab: host (n,3) float32 positions and velocities, masses (or NULL), and the
 * particles' numbers in the whole run (or NULL for 0..n-1), which travel with them between slabs. */
int pm_slab_set_particles(pm_ctx *ctx, const float *pos, const float *vel, const float *mass, const uint32_t *ids, int64_t n);
/* CIC (functions.py:19-70) of the owned particles into nplanes + 1 density planes: the last one is
 * plane 0 of the next slab (functions.py:56, i1 = (i + 1) % G) and has to be added there.  A context created with
 * PM_FLAG_FUSED_GRADIENT on a grid size of the hand-written transforms (512, 1024) writes every row of those planes
 * already transformed along z, as a row of the half spectrum (no density mesh: PM_SLAB_ZSPECTRUM below); the
 * transform is linear, so the extra plane is added to the next slab's plane 0 just the same. */
int pm_slab_deposit(pm_ctx *ctx);
/* Planes [first, first + count) of the context's density (nplanes + 1 planes) or potential (plane 0 =
 * mesh plane plane0 - 1, planes 1..nplanes the owned ones, then mesh planes plane0 + nplanes and + 1,
 * all modulo G) <-> a device buffer of count*G*G floats; or, with which = PM_SLAB_ZSPECTRUM, of the nplanes + 1
 * z-transformed planes such a deposit writes <-> a buffer of count*G*pitch complex64 (2*count*G*pitch floats).
 * op 0: mesh -> buffer, 1: buffer -> mesh, 2: mesh += buffer. */
int pm_slab_planes(pm_ctx *ctx, int which, int first, int count, void *device_buffer, int op);
/* compute_potential (functions.py:73-89), the transforms along z and y: dir < 0 owned density planes ->
 * half spectrum (nplanes, G, pitch); dir > 0 half spectrum -> owned potential planes. */
int pm_slab_fft2(pm_ctx *ctx, int dir);
/* The half spectrum as G/nplanes blocks for an all-to-all (block p: rows p*nplanes.. of every owned
 * plane; nplanes*G*pitch complex64 in all), and back from the blocks the inverse all-to-all delivers.
 * Call order: pm_slab_fft2(-1), pm_slab_pack, [all-to-all], pm_slab_solve_x, [all-to-all], pm_slab_unpack,
 * pm_slab_fft2(+1) -- where the hand-written transforms apply (grid 512 / 1024, nplanes a power of two) the
 * passes along y are done by pack and unpack themselves, straight into and out of the blocks. */
int pm_slab_pack(pm_ctx *ctx, void *device_blocks);
int pm_slab_unpack(pm_ctx *ctx, const void *device_blocks);
/* The same two steps for owned planes [first, first + count) only, so that a caller can exchange the blocks of some
 * planes while others are still being transformed: pm_slab_forward_planes = the transforms along z (unless the deposit
 * has done them) and y of those planes into their part of every block (block p, planes first .. first + count - 1:
 * count*nplanes*pitch consecutive complex64); pm_slab_inverse_planes = from those parts of the blocks the inverse
 * transforms along y and z into the potential planes.  Only where the hand-written transforms apply and nplanes is a
 * power of two (PM_ESTATE otherwise: use the whole-slab calls).  Part of compute_potential (functions.py:73-89), like
 * pm_slab_fft2. */
int pm_slab_forward_planes(pm_ctx *ctx, void *device_blocks, int first, int count);
int pm_slab_inverse_planes(pm_ctx *ctx, const void *device_blocks, int first, int count);
/* compute_potential, the x part (functions.py:75-89), in place on what the all-to-all delivered:
 * (G, nplanes, pitch) complex64 = rows plane0..plane0+nplanes-1 of every mesh plane. */
int pm_slab_solve_x(pm_ctx *ctx, void *device_spectrum_t);
/* np.gradient + force_CIC + kick / drift / wrap (main.py:54-60) for the owned particles; needs all
 * nplanes + 3 potential planes.  Particles whose new cell lies in another slab go to the outbox. */
int pm_slab_gather_kick_drift(pm_ctx *ctx, double dt);
/* The particles that left in the last drift as 32-byte records {x,y,z,m,vx,vy,vz,id}, sorted by
 * (destination slab, id), written to device_records; counts[d] (host, G/nplanes entries) = how many go
 * to slab d.  pm_slab_inbox adds records that arrived from other slabs.  Both belong between
 * pm_slab_gather_kick_drift and the next pm_slab_deposit, outbox first. */
int pm_slab_outbox(pm_ctx *ctx, int64_t *counts, void *device_records, int64_t capacity);
int pm_slab_inbox(pm_ctx *ctx, const void *device_records, int64_t n);
/* pm_slab_outbox in two halves, for a caller that exchanges the counts between the slabs on the device and reads
 * them back once: pm_slab_outbox_counts writes counts[d], d < nslabs, and then the number of leavers that did not fit
 * the outbox (must be 0) as nslabs + 1 int64 to DEVICE memory, on the context's stream; pm_slab_outbox_records takes
 * the counts back from the host and writes the sorted records, reading nothing from the device. */
int pm_slab_outbox_counts(pm_ctx *ctx, int64_t *device_counts);
int pm_slab_outbox_records(pm_ctx *ctx, const int64_t *counts, void *device_records, int64_t capacity);
/* Owned particles in storage order: ids (n), pos and vel (n,3); the buffers hold pm_num_particles() entries as
 * it is before the call, and pm_num_particles() after the call says how many were written (the particles
 * that left in the last drift are not among them). */
int pm_slab_get_particles(pm_ctx *ctx, uint32_t *ids, float *pos, float *vel);
/* Row pitch, in complex64 elements, of the half-spectrum buffers above (>= G/2 + 1). */
int pm_slab_pitch(const pm_ctx *ctx);

/* Accumulated milliseconds and call counts per stage since the last call
 * (needs PM_FLAG_TIMING); both arrays have PM_NSTAGES entries.  Synchronises. */
int pm_stage_times(pm_ctx *ctx, float *ms, int *calls);
/* The same for the single kernel launches listed under PM_KERNEL_*: accumulated milliseconds and launch counts since
 * the last call (needs PM_FLAG_TIMING); both arrays have PM_NKERNELS entries.  ms[k] / calls[k] is the average duration
 * of one launch, measured with HIP events on the stream the kernel is launched on.  The reference has no timers besides
 * the wall clock of main.py:20, 69. */
int pm_kernel_times(pm_ctx *ctx, float *ms, int *calls);

/* Replace the flags given to pm_create (e.g. switch stage timing on for a profiling pass). */
int pm_set_flags(pm_ctx *ctx, unsigned flags);

/* Run the context's work on the caller's HIP stream (a hipStream_t, NULL for the default stream) instead of the
 * stream the context made for itself: its kernels are then ordered with whatever else the caller puts on that
 * stream, and the pm_slab_* calls that fill or read caller-owned device buffers no longer wait for the device. */
int pm_set_stream(pm_ctx *ctx, void *hip_stream);
int pm_synchronize(pm_ctx *ctx);
/* particles the context holds (a slab: without those that left in the last drift and have been taken from the outbox) */
int64_t pm_num_particles(const pm_ctx *ctx);
const char *pm_last_error(const pm_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PM_H */
