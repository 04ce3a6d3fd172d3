// ca_kernels.h -- HIP kernels of the batched collision-avoidance step for gfx950 (MI355X).
//
// Mapping: one LANE per agent, one workgroup per group of arenas (a
// workgroup never splits an arena).  The arena's positions/velocities are staged once into LDS;
// every agent then scans its arena from LDS (broadcast reads), keeps its K nearest neighbours in
// registers, builds its ORCA half-planes into an LDS line table laid out [line][lane] (16 B per
// lane, conflict-free) and solves the 2-D LP over that table.  Arenas are independent, so there
// is no inter-workgroup traffic and no XCD affinity to exploit: the grid is simply arena-major.
//
// Numerics contract: fp32, no FMA contraction (-ffp-contract=off), IEEE sqrt and division, the
// operation order of SURVEY.md Appendix A.  The CPU oracle (oracle/) obeys the same contract, so
// trajectories agree bit for bit.
//
// Files: ca_common.h (types, launch arguments), ca_lp.h (LP1/LP2/LP3), ca_lines.h (ORCA half-planes),
// ca_nbr.h (neighbour search), ca_step.h (solve + reset kernels), ca_alan.h (ALAN bandit kernels), ca_obs.h (laser
// observation).
#pragma once
#include "ca_step.h"
#include "ca_alan.h"
#include "ca_obs.h"

namespace ca {

// ---- diagnostics for the numerics contract ----
__global__ void debug_math_kernel(int op, const void* in, void* out, int n, uint64_t seed) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (op == 0) {
        ((float*)out)[t] = sqrtf(((const float*)in)[t]);
    } else if (op == 1) {
        ((float*)out)[t] = ((const float*)in)[2 * t] / ((const float*)in)[2 * t + 1];
    } else if (op == 2) {
        double s, c;
        sincos64(((const double*)in)[t], &s, &c);
        ((double*)out)[2 * t] = s; ((double*)out)[2 * t + 1] = c;
    } else if (op == 3) {
        const float* f = (const float*)in + 4 * t;
        double x, y;
        pref_dir64(f[0], f[1], (double)f[2], (double)f[3], &x, &y);
        ((double*)out)[2 * t] = x; ((double*)out)[2 * t + 1] = y;
    } else if (op == 4) {
        const uint32_t* u = (const uint32_t*)in + 4 * t;
        double a, b;
        rng2(seed, (int64_t)u[0], (int)u[1], (int)u[2], u[3], &a, &b);
        ((double*)out)[2 * t] = a; ((double*)out)[2 * t + 1] = b;
    } else if (op == 5) {
        ((double*)out)[t] = exp64(((const double*)in)[t]);
    }
}

}  // namespace ca
