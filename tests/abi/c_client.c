/* A plain C caller of the drop-in boundary (include/ca_env.h): no Python, no HIP headers.  Builds the doorway
 * world of the reference env (envs/collision_avoidence_env.py:77-123) for a few arenas, steps it with host
 * actions and prints a checksum of the state and of the observation; tests/test_gpu_parity.py compares the
 * checksum with the same run through the ctypes binding.
 *   gcc -std=c99 -O1 -Iinclude tests/abi/c_client.c -o c_client -Lcollision_avoidance_amd -lcaenv -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ca_env.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != CA_OK) {                                                           \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ca_last_error(env)); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv) {
    const int A = argc > 1 ? atoi(argv[1]) : 8, N = argc > 2 ? atoi(argv[2]) : 10, steps = argc > 3 ? atoi(argv[3]) : 40;
    ca_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_arenas = A; cfg.n_agents = N; cfg.arena_offset = 0; cfg.seed = 7; cfg.reward_scale = 0.3;
    cfg.time_step = (float)(1.0 / 60.0); cfg.neighbor_dist = 1.5f; cfg.max_neighbors = 5; cfg.time_horizon = 1.5f;
    cfg.time_horizon_obst = 1.5f; cfg.radius = 0.5f; cfg.max_speed = 1.0f; cfg.max_obst_neighbors = 8;
    cfg.max_step = 1000; cfg.done_mode = CA_DONE_XLESS; cfg.done_x_thresh = 2.0f;
    cfg.spawn_x0 = 5.0f; cfg.spawn_x1 = 10.0f; cfg.spawn_y0 = 0.0f; cfg.spawn_y1 = 10.0f;
    cfg.goal_x0 = 0.0f; cfg.goal_x1 = 10.0f; cfg.goal_y0 = 0.0f; cfg.goal_y1 = 10.0f;
    ca_env* env = NULL;
    int rc = ca_create(&cfg, 0, NULL, &env);
    if (rc != CA_OK) { fprintf(stderr, "ca_create failed (%d): %s\n", rc, ca_last_error(NULL)); return 1; }
    /* env.py:118-122: outer wall and the two door posts */
    const float verts[] = {-15, 0, -15, 10, 10, 10, 10, 0, 2, 0, 2.5f, 0, 2.5f, 4.4f, 2, 4.4f, 2, 5.6f, 2.5f, 5.6f, 2.5f, 10, 2, 10};
    const int32_t sizes[] = {4, 4, 4};
    CHECK(ca_set_obstacles(env, verts, sizes, 3));
    CHECK(ca_init_scenario(env, CA_SCN_DOORWAY));
    CHECK(ca_reset(env, NULL, NULL, 0, CA_F_OBS));
    const size_t an = (size_t)A * N;
    float* act = (float*)malloc(an * 4);
    float* buf = (float*)malloc(an * CA_OBS_DIM * 4);
    uint64_t h = 1469598103934665603ull;
    uint32_t lcg = 12345u;
    for (int s = 0; s < steps; ++s) {
        for (size_t q = 0; q < an; ++q) {
            lcg = lcg * 1664525u + 1013904223u;
            act[q] = ((float)(lcg >> 8) / 16777216.0f - 0.5f) * 1.5f;
        }
        CHECK(ca_step_host(env, act, CA_F_OBS | CA_F_STATS));
    }
    CHECK(ca_get(env, CA_FLD_POS_X, buf, an * 4, 0)); h = fnv(h, buf, an * 4);
    CHECK(ca_get(env, CA_FLD_POS_Y, buf, an * 4, 0)); h = fnv(h, buf, an * 4);
    CHECK(ca_get(env, CA_FLD_REWARD, buf, an * 4, 0)); h = fnv(h, buf, an * 4);
    CHECK(ca_get(env, CA_FLD_OBS, buf, an * CA_OBS_DIM * 4, 0)); h = fnv(h, buf, an * CA_OBS_DIM * 4);
    ca_stats st;
    CHECK(ca_get_stats(env, &st));
    printf("%016llx %llu %llu\n", (unsigned long long)h, (unsigned long long)st.agent_steps, (unsigned long long)st.collisions);
    free(act); free(buf);
    CHECK(ca_destroy(env));
    return 0;
}
