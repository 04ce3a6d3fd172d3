// ca_alan.h -- ALAN online action selection kernels
// Part of the HIP kernels of libcaenv.so (see ca_kernels.h for the overview and the numerics contract).
#pragma once
#include "ca_common.h"

namespace ca {

// ============================================================================================
// ALAN online action selection (ALAN_true.py:569-628), one lane per agent, around the ORCA step:
//   alan_select_kernel : softmax over the agent's action weights, one draw, preferred velocity =
//                        goal direction rotated by the chosen action (ALAN:578-598);
//   [nbr_kernel + step_kernel in ORCA mode: sim.doStep(), step counter, goal test (ALAN:601, 118-121)]
//   alan_update_kernel : reward of the executed action, sliding-window bandit update (ALAN:603-628).
// Weights, times and the reward that feeds them are fp64 like the reference's Python floats.
// ============================================================================================
enum { ALAN_MAX_ACTIONS = 32, ALAN_BS = 128 };
struct AlanArgs {
    const float *pos_x, *pos_y, *vel_x, *vel_y;
    const double *goal_x, *goal_y;
    float *pref_x, *pref_y, *reward;
    double *w, *t;        // [A][nA][N] action weights / time since the action's weight was set
    int* action;          // [A*N] the action of the current step (complemented while its arena sits out a step)
    double* dirs;         // [4][A*N] goal direction and rotated direction of the current step
    const double* u;      // [A*N] caller-supplied uniforms in [0,1), or null: Philox (RNG_ALAN, episode, step)
    const int *step_count, *arena_done, *episode;
    unsigned long long* arena_stats;
    double act_c[ALAN_MAX_ACTIONS], act_s[ALAN_MAX_ACTIONS];  // (cos, sin) of every action's angle
    double temp, window, dt, reward_scale;
    uint64_t seed;
    int64_t arena_offset;
    int A, N, nA;
    uint32_t flags;
};

// numpy's float64 add.reduce for n < 128: < 8 sequential, otherwise eight accumulators combined as a
// fixed tree plus a sequential tail -- the value np.sum(ps) has at ALAN_true.py:582
template <class Get>
__device__ __forceinline__ double np_sum(int n, Get get) {
    if (n < 8) {
        double res = 0.0;
        for (int k = 0; k < n; ++k) res += get(k);
        return res;
    }
    double r0 = get(0), r1 = get(1), r2 = get(2), r3 = get(3), r4 = get(4), r5 = get(5), r6 = get(6), r7 = get(7);
    int k = 8;
    for (; k < n - (n % 8); k += 8) {
        r0 += get(k); r1 += get(k + 1); r2 += get(k + 2); r3 += get(k + 3);
        r4 += get(k + 4); r5 += get(k + 5); r6 += get(k + 6); r7 += get(k + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; k < n; ++k) res += get(k);
    return res;
}

__global__ __launch_bounds__(ALAN_BS) void alan_select_kernel(const AlanArgs p) {
    extern __shared__ double s_ps[];  // [n_actions][lane]: sized by the launch (8 B x n_actions x ALAN_BS)
    const int q = blockIdx.x * ALAN_BS + threadIdx.x;
    if (q >= p.A * p.N) return;
    const int a = q / p.N, i = q - a * p.N, nA = p.nA;
    if ((p.flags & 16u) && p.arena_done[a] != 0) {  // CA_F_FREEZE: tell the update kernel, keep the last action
        p.action[q] = ~p.action[q];
        return;
    }
    double* ps = s_ps + threadIdx.x;
    // weights and times are stored [A][n_actions][N]: the lanes of a wave read consecutive doubles
    const double* w = p.w + (size_t)a * nA * p.N + i;
    for (int k = 0; k < nA; ++k) ps[k * ALAN_BS] = exp64(w[(size_t)k * p.N] / p.temp);  // ALAN:580-581
    const double sum = np_sum(nA, [&](int k) { return ps[k * ALAN_BS]; });
    double acc = 0.0;
    for (int k = 0; k < nA; ++k) {                                                 // ALAN:582
        const double v = ps[k * ALAN_BS] / sum;
        ps[k * ALAN_BS] = v;
        acc += v;
    }
    // np.random.choice(n, 1, p=ps) (ALAN:585): cdf = cumsum(p) / cdf[-1]; searchsorted(cdf, u, 'right')
    double ui;
    if (p.u) ui = p.u[q];
    else {  // stream (seed, global arena, agent, RNG_ALAN + 256 x episode, step): a new stream every episode of the arena
        double u1;
        rng2(p.seed, p.arena_offset + a, i, RNG_ALAN + (p.episode[a] << 8), (uint32_t)p.step_count[a], &ui, &u1);
    }
    int id = nA - 1;
    double run = 0.0;
    bool found = false;
    for (int k = 0; k < nA - 1; ++k) {
        run += ps[k * ALAN_BS];
        if (!found && run / acc > ui) { id = k; found = true; }
    }
    p.action[q] = id;
    double gx, gy;
    pref_dir64(p.pos_x[q], p.pos_y[q], p.goal_x[q], p.goal_y[q], &gx, &gy);        // ALAN:588
    const double cs = p.act_c[id], sn = p.act_s[id];                                // ALAN:592-595
    const double lx = gx * cs - gy * sn, ly = gx * sn + gy * cs;
    const size_t an = (size_t)p.A * p.N;  // dirs: [4][A*N]
    p.dirs[q] = gx; p.dirs[an + q] = gy; p.dirs[2 * an + q] = lx; p.dirs[3 * an + q] = ly;
    p.pref_x[q] = (float)lx; p.pref_y[q] = (float)ly;                               // ALAN:598
}

__global__ __launch_bounds__(ALAN_BS) void alan_update_kernel(const AlanArgs p) {
    const int q = blockIdx.x * ALAN_BS + threadIdx.x;
    if (q >= p.A * p.N) return;
    const int id = p.action[q];
    if (id < 0) {  // the arena was frozen when this step began
        p.action[q] = ~id;
        return;
    }
    const int a = q / p.N, nA = p.nA;
    const size_t an = (size_t)p.A * p.N;
    const double d[4] = {p.dirs[q], p.dirs[an + q], p.dirs[2 * an + q], p.dirs[3 * an + q]};
    const float vxf = p.vel_x[q], vyf = p.vel_y[q];
    {   // env.py:389-400 in fp32, as ca_step reports it
        const float scale = (float)p.reward_scale;
        const float r_goal = vxf * (float)d[0] + vyf * (float)d[1];
        const float r_polite = vxf * (float)d[2] + vyf * (float)d[3];
        const float rew = scale * r_goal + (1.0f - scale) * r_polite;
        p.reward[q] = rew;
        if (p.flags & 2u) {
            double* sum = reinterpret_cast<double*>(&p.arena_stats[(size_t)a * ST_STRIDE + ST_SUMREW]);
            if ((p.N & 63) == 0) {  // a wave lies inside one arena (and leaves the kernel as a whole): one atomic per wave
                double r = (double)rew;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off, 64);
                if ((threadIdx.x & 63) == 0) atomicAdd(sum, r);
            } else {
                atomicAdd(sum, (double)rew);
            }
        }
    }
    const double vx = (double)vxf, vy = (double)vyf;                                // ALAN:606-613
    const double R = p.reward_scale * (vx * d[0] + vy * d[1]) + (1.0 - p.reward_scale) * (vx * d[2] + vy * d[3]);
    const int i = q - a * p.N;
    double* w = p.w + (size_t)a * nA * p.N + i;
    double* t = p.t + (size_t)a * nA * p.N + i;
    for (int k = 0; k < nA; ++k) {                                                  // ALAN:616-628
        double tk = t[(size_t)k * p.N] + p.dt;
        double wk = w[(size_t)k * p.N];
        if (tk >= p.window) { tk = 0.0; wk = 0.0; }
        if (k == id) wk = R;
        t[(size_t)k * p.N] = tk; w[(size_t)k * p.N] = wk;
    }
    // the solve kernel left the goal direction in pref (its ORCA-mode epilogue); the reference's agent
    // still holds the velocity it was given at ALAN:598
    p.pref_x[q] = (float)d[2]; p.pref_y[q] = (float)d[3];
}

}  // namespace ca
