// ca_kernels.h -- HIP kernels of the batched collision-avoidance step for gfx950 (MI355X).
//
// Mapping: a workgroup never splits an arena; the arena's positions / velocities are staged once
// in LDS and every agent scans its arena from there (broadcast reads).
//   * step_kernel (ca_step.h): one LANE per agent.  The K nearest neighbours are kept in registers as sorted keys
//     (ca_nbr.h); the ORCA half-planes live in REGISTER slots (4 obstacle + KMAX neighbour slots, LP2 / LP1 fully
//     unrolled; every configuration with K <= 10 and <= 4 obstacle neighbours) or, for larger K / S, in an LDS line
//     table laid out [line][lane]; the lanes whose LP2 is infeasible solve LP3 four lanes per agent in a per-wave LDS
//     pool (ca_lp.h lp3_coop).
//   * quad kernel (ca_quad.h): FOUR lanes per agent for small arenas (a chip of 1024 SIMDs is otherwise left with a
//     few hundred waves): candidates, edges, lines and LP1 clips are dealt over the quad and merged with DPP moves;
//     one launch can advance T ORCA-only steps with the arena resident in registers / LDS (ca_rollout).
//   * pair kernel (ca_pair.h): TWO lanes per agent for large arenas (192 .. 512 agents: one arena per workgroup is otherwise
//     two waves per SIMD, each a long dependent chain): grid-scan candidates, half-planes and LP1 clips dealt over the pair,
//     lines in registers (slot m of the even / odd lane = neighbour 2 m / 2 m + 1), merges by DPP.
//   * obs_kernel (ca_obs.h): 16 lanes per agent, one lane per (source, ray) pair, ds_min_u64 merge per ray.
// Arenas are independent, so there is no inter-workgroup traffic; the grids are arena-major, and the observation workgroups of
// an arena are indexed so that they run on the XCD whose solve workgroup wrote the arena's state (ca_obs.h).
// Diagnostics: the CA_STAMPS build (tools/stamps.py, tools/diag/placement.py; never the product library) adds phase
// time stamps, a block order for the solve kernel and the two ca_debug_* entry points that read / install them.
// ca_nbr.h claims 128 VGPRs for the stand-alone neighbour kernel on purpose (four waves per SIMD, see there).
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
#include "ca_quad.h"
#include "ca_pair.h"
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
    } else if (op == 6) {
        ((float*)out)[t] = div_ir(((const float*)in)[2 * t], ((const float*)in)[2 * t + 1]);
    } else if (op == 7) {
        ((float*)out)[t] = sqrt_ir(((const float*)in)[t]);
    }
}

}  // namespace ca
