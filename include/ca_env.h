/*
 * ca_env.h -- C ABI of libcaenv.so: the MI355X (gfx950) batched collision-avoidance environment.
 *
 * This is the drop-in boundary for the hot path of navallo/collision_avoidance: one call
 * advances A independent arenas x N agents through what the reference does per agent in Python
 * plus per-scalar calls into the external `rvo2` extension.  Citations (file:line) are into
 * /root/reference/collision_avoidance/ : env.py = envs/collision_avoidence_env.py,
 * utils.py = envs/utils.py, ALAN = ALAN/ALAN_true.py.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success, a negative CA_E* code on failure;
 *     ca_last_error() returns the message of the last failure on that handle (NULL handle: of the
 *     last failed ca_create on this thread).
 *   - a handle is bound to one HIP device and one stream; calls are asynchronous on that stream
 *     unless stated otherwise.  A handle is not thread-safe; distinct handles are independent.
 *   - all per-agent arrays are struct-of-arrays fp32/int32 (targets: fp64) of shape [A, N] (agent index
 *     fastest).
 *   - there is NO CPU fallback: without a HIP device ca_create fails with CA_ENODEV.
 */
#ifndef CA_ENV_H
#define CA_ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CA_OBS_DIM 64 /* env.py:34,53: 16 rays x (hit x, hit y, vel x, vel y) */
#define CA_N_RAYS 16
#define CA_MAX_NEIGHBORS 16      /* largest supported max_neighbors       */
#define CA_MAX_OBST_NEIGHBORS 16 /* largest supported max_obst_neighbors: RVO2 keeps every edge in range (env.py:249,
                                    301-318); 16 covers the reference's own worlds.  A list that meets more edges in range
                                    than it holds drops the farthest, counts it (ca_stats.obst_overflow) AND makes every
                                    later step call fail with CA_ERANGE: see ca_allow_obstacle_overflow */
#define CA_MAX_AGENTS 1024       /* one workgroup owns one arena; above 256 agents max_neighbors <= 10
                                    is required (LDS capacity: the register-line solve kernels)         */

/* error codes */
#define CA_OK 0
#define CA_EINVAL (-1)
#define CA_ENODEV (-2)
#define CA_EHIP (-3)
#define CA_ESIZE (-4)
#define CA_ERANGE (-5)

/* done_mode: who decides that an agent has finished */
#define CA_DONE_XLESS 0  /* env.py:352-365: pos.x < done_x_thresh -> done, goal <- goal2      */
#define CA_DONE_GOAL 1   /* ALAN:547-566: |pos - goal| < 2 r -> done, goal <- goal2           */
#define CA_DONE_REGOAL 2 /* synthetic workload: |pos - goal| < 2 r -> draw a new goal         */

/* step flags */
#define CA_F_OBS 1u       /* write the laser observation (env.py:231-277, utils.py:42-113)     */
#define CA_F_STATS 2u     /* count collisions (build-defined)                   */
#define CA_F_AUTORESET 4u /* arenas that finished are reset (env.py:461-488) inside the call   */
#define CA_F_NODONE 8u    /* ca_orca_step only: no done test, no step counter (env.py:447-450) */
#define CA_F_FREEZE 16u   /* an arena whose arena_done flag is set is not advanced any more: the
                             `break` of the reference's episode loops (ALAN:121-123), per arena,
                             without a host round trip.  ca_reset clears the flag.                */
#define CA_ALAN_MAX_ACTIONS 32

/* scenarios for ca_init_scenario */
#define CA_SCN_CROWD 0   /* ALAN:270-294 random start / random goal                            */
#define CA_SCN_CIRCLE 1  /* ALAN:297-330 circle swap                                           */
#define CA_SCN_DOORWAY 2 /* env.py:77-123 the reference env's own world                        */
#define CA_SCN_CONGESTED 3 /* ALAN:175-210 */
#define CA_SCN_INCOMING 4  /* ALAN:213-267 */
#define CA_SCN_BLOCKS 5    /* ALAN:333-374 (block obstacles come through ca_set_obstacles)       */
#define CA_SCN_DEADLOCK 6  /* ALAN:377-457 */
#define CA_SCN_CROWD_SEPARATED 7 /* the crowd with rejection-sampled starts >= 2 r apart (SURVEY 8d bench variant) */

/* fields for ca_get / ca_set / ca_field_ptr */
enum ca_field {
    CA_FLD_POS_X = 0, CA_FLD_POS_Y, CA_FLD_VEL_X, CA_FLD_VEL_Y, CA_FLD_PREF_X, CA_FLD_PREF_Y,
    CA_FLD_GOAL_X, CA_FLD_GOAL_Y, CA_FLD_GOAL2_X, CA_FLD_GOAL2_Y, /* f64 [A,N]: the reference keeps its
                                   targets as Python floats (env.py:94, ALAN:186) and derives the
                                   preferred velocity from them in fp64 */
    CA_FLD_REWARD,       /* f32 [A,N]                                             */
    CA_FLD_AGENT_DONE,   /* i32 [A,N]                                             */
    CA_FLD_ARRIVE_STEP,  /* i32 [A,N]                                             */
    /* the neighbour lists left by the last doStep (sim.getAgentNumAgentNeighbors / getAgentAgentNeighbor /
       getAgentNumObstacleNeighbors / getAgentObstacleNeighbor, env.py:246-249, 283-285, 305-306).  i32 at this
       boundary; the device keeps them packed (u16 counts, u8 / u16 ids), so ca_get / ca_set convert and
       ca_field_ptr refuses them */
    CA_FLD_NB_COUNT,     /* i32 [A,N]   ORCA agent-neighbour count                */
    CA_FLD_NB_IDX,       /* i32 [A,K,N] ORCA agent neighbours, nearest first      */
    CA_FLD_OBST_COUNT,   /* i32 [A,N]                                             */
    CA_FLD_OBST_IDX,     /* i32 [A,S,N] ORCA obstacle-edge neighbours: edge ids of the arena's processed table */
    CA_FLD_OBS,          /* f32 [A,N,64]                                          */
    CA_FLD_STEP_COUNT,   /* i32 [A]                                               */
    CA_FLD_ARENA_DONE,   /* i32 [A]                                               */
    CA_FLD_EPISODE,      /* i32 [A]                                               */
    CA_FLD_REGOAL_COUNT, /* i32 [A,N]                                             */
    CA_FLD_ALAN_WEIGHTS, /* f64 [A,n_actions,N] ALAN:75 self.weights (after ca_alan_configure);
                            agent index fastest like every other per-agent array                  */
    CA_FLD_ALAN_TIMES,   /* f64 [A,n_actions,N] ALAN:76 self.times                                 */
    CA_FLD_ALAN_ACTION,  /* i32 [A,N] the action executed by the last ca_alan_step (read-only)     */
    CA_FLD_ARENA_STATS,  /* u64 [A,8] per-arena counters (read-only): episodes, collisions, obst_collisions,
                            goals_reached, obst_overflow, sum_reward (f64 bits), steps sat out under
                            CA_F_FREEZE, last finished episode (length << 32 | agents that arrived)   */
    CA_FLD__COUNT
};

/* Units and magnitudes.  The kernels compute in fp32 with a correctly rounded division and square root that omit the
 * range-scaling steps of the general sequences (exact where quotients, discriminants and lengths stay well inside the normal
 * range: csrc/ca_math.h).  That holds for worlds in units of O(1) -- the reference's are metres and seconds: radius 0.5,
 * max_speed 1, time_step 1/60, arenas of 10-50 (env.py:26-44, ALAN_true.py:14-20) -- and is ENFORCED at the boundary:
 * ca_create refuses (CA_ERANGE) a configuration outside the ranges below, ca_set_obstacles a vertex beyond CA_MAX_COORD
 * or an edge shorter than CA_MIN_EDGE.  Positions and targets handed in through ca_set / ca_reset are device data and are
 * not checked: keep them within CA_MAX_COORD of the origin. */
#define CA_MIN_LENGTH 1e-3f      /* radius, max_speed, neighbor_dist, time_horizon, time_horizon_obst: [1e-3, 1e3] */
#define CA_MAX_LENGTH 1e3f
#define CA_MIN_TIME_STEP 1e-4f   /* time_step: [1e-4, 10] */
#define CA_MAX_TIME_STEP 10.0f
#define CA_MAX_COORD 1e5f        /* obstacle vertices, spawn / goal boxes: |x|, |y| <= 1e5 */
#define CA_MIN_EDGE 1e-4f        /* an obstacle edge is at least this long */

/* Replaces the constants the reference hard-codes in Collision_Avoidance_Env.__init__
 * (env.py:27-44), the literal at env.py:130, and the per-call arguments of
 * rvo2.PyRVOSimulator / addAgent (env.py:62-68, 126-133; ALAN:22-28). */
typedef struct ca_config {
    int32_t n_arenas;           /* arenas owned by this handle (this GPU's shard)             */
    int32_t n_agents;           /* agents per arena, 1..CA_MAX_AGENTS                         */
    int64_t arena_offset;       /* global id of arena 0: keys the scenario RNG, so results do
                                   not depend on how arenas are sharded over GPUs             */
    uint64_t seed;
    double reward_scale;        /* env.py:396 (0.3); ALAN:47 gamma (0.6)                      */
    float time_step;            /* env.py:27                                                  */
    float neighbor_dist;        /* env.py:28 / ALAN:16                                        */
    int32_t max_neighbors;      /* env.py:29 / ALAN:17, 0..CA_MAX_NEIGHBORS                   */
    float time_horizon;         /* env.py:30                                                  */
    float time_horizon_obst;    /* env.py:130                                                 */
    float radius;               /* env.py:31                                                  */
    float max_speed;            /* env.py:32                                                  */
    int32_t max_obst_neighbors; /* capacity of the obstacle-neighbour list, 1..CA_MAX_OBST_NEIGHBORS;
                                   overflow is counted in ca_stats.obst_overflow and is an error
                                   (CA_ERANGE) unless ca_allow_obstacle_overflow accepted it    */
    int32_t max_step;           /* env.py:44; <= 0: no cap                                    */
    int32_t done_mode;
    float done_x_thresh;        /* env.py:359                                                 */
    float spawn_x0, spawn_x1, spawn_y0, spawn_y1; /* reset() spawn box, env.py:478            */
    float goal_x0, goal_x1, goal_y0, goal_y1;     /* CA_DONE_REGOAL draw box                  */
} ca_config;

typedef struct ca_stats {
    uint64_t agent_steps;     /* counted IN the solve kernels: per arena, the steps it was really advanced, x n_agents */
    uint64_t episodes;
    uint64_t collisions;      /* overlapping agent pairs after the update, summed over steps  */
    uint64_t obst_collisions; /* agents overlapping an obstacle edge, summed over steps       */
    uint64_t goals_reached;
    uint64_t obst_overflow;
    double sum_reward;
} ca_stats;

typedef struct ca_env ca_env;

/* Replaces Collision_Avoidance_Env.__init__ + rvo2.PyRVOSimulator(...) (env.py:26-74, 62-68).
 * device: HIP device ordinal.  stream: a hipStream_t to run on (e.g. PyTorch's current stream),
 * or NULL to let the handle create its own. */
int ca_create(const ca_config* cfg, int device, void* stream, ca_env** out);
int ca_destroy(ca_env* env);
const char* ca_last_error(const ca_env* env);
/* Run on the caller's stream from now on (hipStream_t; NULL = the device's default stream).
 * The previous stream is drained first. */
int ca_set_stream(ca_env* env, void* stream);

/* Replaces sim.addObstacle + sim.processObstacles (env.py:118-123, 143-149): the same polygons
 * for every arena (see ca_set_obstacles_per_arena for a world per arena).  verts_xy: host array [sum(poly_sizes), 2].  Like the RVO2 library's
 * processObstacles, edges that cross the supporting line of a splitting edge of its obstacle tree are cut
 * there; the cut points are appended to the vertex table behind the caller's vertices. */
int ca_set_obstacles(ca_env* env, const float* verts_xy, const int32_t* poly_sizes, int32_t n_poly);
/* Replaces sim.getObstacleVertex / getNextObstacleVertexNo (env.py:148, 208, 307-311): the processed
 * vertex table.  Edge i runs from vertex i to vertex next[i]; these are the ids in CA_FLD_OBST_IDX.
 * Host arrays of capacity `cap` (each may be NULL); *n_out = number of vertices. */
int ca_get_obstacles(ca_env* env, float* verts_xy, int32_t* next, int32_t* convex, int32_t cap, int32_t* n_out);
/* A world of its own for every arena: the reference builds a simulator per environment and draws the four blocks of
 * the "blocks" world anew for each (ALAN:359-372: addObstacle x 5 + processObstacles per simulator; reset() redraws
 * them, ALAN:92-100; the trainer averages over such worlds, Train_ALAN_action_space.py:53-66).
 * n_poly: host array [A], polygons of arena a; poly_sizes / verts_xy: all arenas' polygons back to back.
 * Each arena gets its own processed table; obstacle-neighbour ids (CA_FLD_OBST_IDX) are local to it. */
int ca_set_obstacles_per_arena(ca_env* env, const float* verts_xy, const int32_t* poly_sizes, const int32_t* n_poly);
/* ca_get_obstacles for the table of one arena (the common table if ca_set_obstacles installed one). */
int ca_get_obstacles_arena(ca_env* env, int32_t arena, float* verts_xy, int32_t* next, int32_t* convex, int32_t cap,
                           int32_t* n_out);

/* Replaces _init_world's agent loop (env.py:86-97) / ALAN's scenario generators (ALAN:175-457).  Runs on the device: every
 * agent's heading, start and targets come from the handle's counter-based streams (keyed by the global arena id) or, for
 * the layouts that do not depend on the arena, from a per-agent table made once on the host (its cos / sin / sqrt are
 * libm's); only the rejection-sampled starts of CA_SCN_CROWD_SEPARATED are drawn on the host (sequential per arena). */
int ca_init_scenario(ca_env* env, int32_t scenario);

/* Replaces the per-scalar getters/setters (sim.getAgentPosition, setAgentPosition, ...
 * env.py:157, 237, 479): whole-array copies.  *_is_device: the caller's pointer is device memory. */
int ca_set(ca_env* env, int32_t field, const void* src, size_t bytes, int32_t src_is_device);
int ca_get(ca_env* env, int32_t field, void* dst, size_t bytes, int32_t dst_is_device);
/* Page-locked, device-visible host memory owned by the handle (freed by ca_host_free or ca_destroy): the destination of the
 * host-array calls at the link's rate instead of a pageable copy's, and an action buffer the kernels read where it lies. */
int ca_host_alloc(ca_env* env, size_t bytes, void** out);
int ca_host_free(ca_env* env, void* p);
/* Zero-copy view of a field's device buffer (valid until ca_destroy / ca_bind_obs). */
int ca_field_ptr(ca_env* env, int32_t field, void** dev_ptr, size_t* bytes);
/* Let the caller own the observation buffer (e.g. a torch tensor [A,N,64] f32 on this device). */
int ca_bind_obs(ca_env* env, void* dev_ptr, size_t bytes);

/* Replaces reset() (env.py:461-488).  pos_x/pos_y: device or host arrays [A,N] with the new
 * positions, or NULL to draw them from the spawn box with the counter-based RNG. */
int ca_reset(ca_env* env, const float* pos_x, const float* pos_y, int32_t pos_is_device, uint32_t flags);

/* The same for the arenas with mask[a] != 0 only, positions drawn from the spawn box: what a vector-env
 * front end needs for reset_at(index) / try_reset(env_id).  mask: i32 [A], device or host. */
int ca_reset_masked(ca_env* env, const int32_t* mask, int32_t mask_is_device, uint32_t flags);

/* Replaces step(action) (env.py:367-416).  actions: DEVICE array [A,N] f32 of heading offsets. */
int ca_step(ca_env* env, const float* actions, uint32_t flags);
/* Same, with the actions in HOST memory (copied to the device on the handle's stream). */
int ca_step_host(ca_env* env, const float* actions_host, uint32_t flags);
/* A whole host-side step in ONE round trip -- what one environment per worker needs (run_rllib.py:77, 108; env.py:367-416: the
 * four dictionaries every step): actions in (NULL: the ORCA-only step, env.py:447-458), then
 *   out_host = [ observation A*N*64 f32 | reward A*N f32 | arena_done A i32 | step_count A i32 ]
 * in one device-to-host copy and one synchronisation (ca_step_host + three ca_get make four).  out_bytes must be the size of
 * that layout.  Buffers from ca_host_alloc make the copy run at the link's rate; an action buffer from ca_host_alloc is read by
 * the kernel where it lies.  Without CA_F_OBS the observation part holds the last one computed. */
int ca_step_packed(ca_env* env, const float* actions_host, uint32_t flags, void* out_host, size_t out_bytes);
/* Replaces orca_step (env.py:447-458; ALAN:631-636 + the done test of ALAN:118-121). */
int ca_orca_step(ca_env* env, uint32_t flags);
/* Replaces _get_obs() alone (env.py:231-277): recompute the observation of the current state. */
int ca_observe(ca_env* env);
/* `steps` consecutive ca_orca_step calls without returning to the host (with CA_F_FREEZE: every
 * arena runs to the end of its own episode, at most `steps` steps): the reference's ORCA-only loops, env.py:570-573
 * (`while True: orca_step()`) and ALAN:106-123 (run_sim).  One kernel launch per 256 steps where the handle uses
 * the four-lanes-per-agent kernel (ca_solver_info) and no observation is asked for (a launch is bounded so that a long
 * rollout stays a sequence of kernels of a few milliseconds; ca_profile_read reports such a launch per step). */
int ca_rollout(ca_env* env, int32_t steps, uint32_t flags);

/* ALAN online learning (ALAN_true.py:569-628 online_step + the counter / goal test of run_sim,
 * ALAN:118-121), for every agent of every arena on the device.
 * ca_alan_configure replaces the constructor's bandit state (ALAN:31-38, 47-49, 73-76):
 *   actions_xy : HOST array [n_actions, 2] of action vectors (ALAN:31-38); an action rotates the goal
 *                direction by atan2(y, x) (ALAN:592-595)
 *   temp       : softmax temperature (ALAN:49 online_temp = 0.2)
 *   timewindow : seconds after which an action's weight is forgotten (ALAN:48 = 2)
 *   time_step  : the fp64 time step the reference adds to `times` (ALAN:15 = 1/60.)
 * Weights and times start at zero.
 * ca_alan_step: softmax draw -> preferred velocity -> ORCA step -> reward -> bandit update -> step
 * counter -> goal test.  u: device or host array [A,N] f64 of uniforms in [0,1) that drive the draws (one per
 * agent, consumed like numpy's choice: first action whose normalised cdf exceeds u), or NULL to use
 * the handle's counter-based RNG keyed by (seed, global arena, agent, episode of the arena, step).  Flags: CA_F_OBS,
 * CA_F_STATS, CA_F_FREEZE. */
int ca_alan_configure(ca_env* env, const double* actions_xy, int32_t n_actions, double temp, double timewindow,
                      double time_step);
int ca_alan_step(ca_env* env, const double* u, int32_t u_is_device, uint32_t flags);
/* `steps` consecutive ca_alan_step(env, NULL, 0, flags) calls without returning to the host: with
 * CA_F_FREEZE this is run_sim(mode=1) (ALAN:106-123) for every arena at once.  Where the handle uses the four-lanes kernel
 * (ca_solver_info: rollout_one_launch) the bandit runs INSIDE that kernel -- one launch per 256 steps, weights and times
 * resident in LDS --, and a single ca_alan_step is one launch instead of three (select, solve, update). */
int ca_alan_rollout(ca_env* env, int32_t steps, uint32_t flags);

/* The reference's simulator keeps EVERY obstacle edge within range of an agent (sim.getAgentNumObstacleNeighbors /
 * getAgentObstacleNeighbor, env.py:249, 301-318, iterate them all); this library's lists hold max_obst_neighbors and drop the
 * farthest edges beyond that.  So that such a deviation cannot pass unnoticed, an overflow is a sticky error of the handle:
 * once a step has met an agent with more edges in range than the list holds, ca_step / ca_step_host / ca_step_packed /
 * ca_orca_step / ca_rollout / ca_alan_step / ca_alan_rollout and ca_sync return CA_ERANGE (ca_last_error names the global arena,
 * the agent and the number of edges) -- asynchronously, like a device fault: the first call made after the kernel that
 * overflowed has finished reports it, calls that synchronise report it at once -- until ca_reset_stats clears it.
 * ca_allow_obstacle_overflow(env, 1) accepts the truncation (nearest max_obst_neighbors edges kept, counted in
 * ca_stats.obst_overflow and per arena); 0 restores the default.  ca_get / ca_get_stats always work. */
int ca_allow_obstacle_overflow(ca_env* env, int32_t allow);

/* Blocks until the stream is idle, then returns the counters accumulated so far. */
int ca_get_stats(ca_env* env, ca_stats* out);
int ca_reset_stats(ca_env* env);
int ca_sync(ca_env* env);

/* Diagnostics for the numerics contract tests: evaluates device primitives on n inputs.
 * op 0: sqrtf(x)            in f32[n]        out f32[n]
 * op 1: a / b               in f32[2n]       out f32[n]
 * op 2: sincos64(x)         in f64[n]        out f64[2n]
 * op 3: pref_dir64          in f32[4n]       out f64[2n]
 * op 4: philox4x32 uniform  in u32[4n]       out f64[2n]   (key = seed of the handle)
 * op 5: exp64(x)            in f64[n]        out f64[n]
 * op 6: a / b by the in-range sequence (csrc/ca_math.h div_ir)   in f32[2n]   out f32[n]
 * op 7: sqrt(x) by the in-range sequence (csrc/ca_math.h sqrt_ir) in f32[n]    out f32[n]
 * Host pointers. */
int ca_debug_math(ca_env* env, int32_t op, const void* in, void* out, int32_t n);

/* Per-kernel timing for roofline reports.  ca_profile(env, k), k >= 1: every kernel launch of every k-th step of this
 * handle carries a start and a stop HIP event on its own dispatch (hipExtLaunchKernel on the handle's stream: the execution
 * time of the kernel, what rocprofv3 --kernel-trace reports; k = 1: every launch).  ca_profile_read synchronises and
 * returns, per kernel kind, the number of sampled launches and their mean duration in milliseconds since the last read
 * (kinds: 0 nbr_kernel -- only when the neighbour search runs as a launch of its own, CA_FUSE_NBR=0; normally it is the
 * head of step_kernel --, 1 step_kernel -- per STEP: a ca_rollout launch that advances T steps counts as one launch of
 * duration / T --, 2 obs_kernel, 3 the small kernels: reset_kernel, reset_arena_kernel, the ALAN select / update kernels, each
 * launch on its own), then clears them.  ca_profile(env, 0) switches
 * it off (default).  A sampled step costs ~10 us of dispatch serialisation; results never depend on it. */
int ca_profile(ca_env* env, int32_t period);
int ca_profile_read(ca_env* env, int32_t counts[4], float mean_ms[4]);

/* Launch geometry chosen for this handle (for reports): threads per block, blocks, LDS bytes. */
int ca_launch_info(ca_env* env, int32_t* block, int32_t* grid, int32_t* lds_bytes, int32_t* obs_grid);
/* Which solve kernel the handle uses: *lanes_per_agent = 1 (one lane per agent), 2 (two lanes per agent: arenas of 129 .. 512
 * agents with max_neighbors <= 10 and max_obst_neighbors <= 4 -- one arena per workgroup is otherwise two waves per SIMD, each
 * a long dependent chain) or 4 (four lanes per agent: chosen at ca_create for batches that would otherwise leave SIMDs without
 * a wave -- fewer than 1024 waves -- when n_agents <= 128 and max_neighbors <= 10 (n_agents <= 64 when, in addition,
 * max_neighbors > 5 and max_obst_neighbors > 4)); results are identical bit for bit.  Among the one-lane kernels the ORCA
 * lines live in registers when max_neighbors <= 10 and either max_obst_neighbors <= 4 or the installed world has at most 16
 * edges per arena (the reference env's own doorway world: an agent with more than four edges in range is solved apart,
 * exactly), else in an LDS table; the choice is re-made when obstacle tables are installed.
 * *rollout_one_launch = 1: ca_rollout(env, T, flags without CA_F_OBS) is ONE kernel launch that keeps every arena in
 * registers / LDS for its T steps (the four-lanes kernel; chosen up to 1024 waves inclusive); 0: it is T launches. */
int ca_solver_info(ca_env* env, int32_t* lanes_per_agent, int32_t* rollout_one_launch);
/* Hash of the kernel sources and compiler flags this library was built from (collision_avoidance_amd/build.py compiles
 * it in): reports and counter profiles quote it, so that they name the code that ran.  No reference counterpart. */
const char* ca_source_sha(void);

#ifdef __cplusplus
}
#endif
#endif
