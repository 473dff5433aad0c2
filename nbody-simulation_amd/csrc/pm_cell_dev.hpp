// Device arithmetic that places a particle on the mesh, shared by the particle kernels (pm_particles.hip) and the
// last pass of the sort, which moves the position records into cell order itself (pm_sort.hip).
//
// Arithmetic follows what numba compiles the reference loops to (functions.py:19-70, 99-180): wrap, scale and
// fractional parts in float64.  Files that include this are built with -ffp-contract=off.
#pragma once

#include "pm_internal.hpp"

namespace pmcell {

// Python-style x % L for L > 0 (functions.py:27-29): result in [0, L].
__device__ __forceinline__ double wrap_f64(double x, double L)
{
    if (x >= 0.0 && x < L) return (x == 0.0) ? 0.0 : x;    // already inside: fmod is the identity
    double r = fmod(x, L);
    if (r != 0.0) {
        if (r < 0.0) r += L;
    } else {
        r = 0.0;
    }
    return r;
}

struct cell_t {
    int i, j, k;
    double dx, dy, dz;
};

// functions.py:27-43.  A scaled coordinate that rounds up to G (the reference would index out
// of bounds there) is folded back to cell 0 of that axis.
__device__ __forceinline__ cell_t locate(double px, double py, double pz, const pm_geom &g)
{
    cell_t c;
    double sx = wrap_f64(px, g.L) * g.inv;
    double sy = wrap_f64(py, g.L) * g.inv;
    double sz = wrap_f64(pz, g.L) * g.inv;
    c.i = (int)sx; c.j = (int)sy; c.k = (int)sz;
    c.dx = sx - (double)c.i; c.dy = sy - (double)c.j; c.dz = sz - (double)c.k;
    if (c.i >= g.G) c.i -= g.G;
    if (c.j >= g.G) c.j -= g.G;
    if (c.k >= g.G) c.k -= g.G;
    return c;
}

template <typename TP>
__device__ __forceinline__ cell_t locate(const pm_prec<TP> &p, const pm_geom &g)
{
    return locate((double)p.x, (double)p.y, (double)p.z, g);
}

// Sort key of a particle: its cell, counted from the first plane the context owns; one key past all
// cells for a particle whose plane belongs to another slab.
__device__ __forceinline__ uint32_t cell_key(const cell_t &c, const pm_geom &g)
{
    int il = c.i - g.i0;
    if (g.Gs != g.G && (il < 0 || il >= g.Gs)) return (uint32_t)g.Gs * (uint32_t)g.G * (uint32_t)g.G;
    return (uint32_t)(((int64_t)il * g.G + c.j) * g.G + c.k);
}

// The fraction record of a particle: its offsets inside the cell (functions.py:41-43, computed in float64 and
// rounded to float32 once) and its mass -- all the deposit needs besides the cell key.
template <typename TP>
__device__ __forceinline__ pm_frac fraction_record(const pm_prec<TP> &rec, const pm_geom &g)
{
    const cell_t c = locate(rec, g);
    return pm_frac{(float)c.dx, (float)c.dy, (float)c.dz, rec.m};
}

}  // namespace pmcell
