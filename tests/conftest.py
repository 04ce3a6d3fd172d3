import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the GPU suite under `-x`: the hot-path parity tests come first, the
# auxiliary layers last, so that a failure in an adapter can never hide the rows of the solver again.
# Earlier entry = earlier run; a test matches the first entry its node id contains.
_GPU_ORDER = (
    "test_numerics_contract",
    "test_orca_rollout_bit_exact[c2like", "test_orca_rollout_bit_exact[c3like", "test_orca_rollout_bit_exact[c5like",
    "test_orca_rollout_bit_exact",
    "test_step_with_actions_bit_exact",
    "test_full_size",
    "test_obs_adversarial_geometry",
    "test_gpu_replays_reference_env_loop_golden",
    "test_gpu_replays_reference_orca_episode_loop",
    "test_autoreset_and_explicit_reset", "test_regoal_and_rollout_call", "test_sharding_invariance",
    "test_processed_obstacle_table_equals_oracle", "test_world_without_obstacles",
    "test_per_arena_obstacle",
    "test_fresh_handle",
    "test_gpu_parity.py",
    "test_gpu_alan.py",
    "test_gpu_adapters.py",
)


def _gpu_rank(item):
    for k, key in enumerate(_GPU_ORDER):
        if key in item.nodeid:
            return k
    return len(_GPU_ORDER)


def pytest_collection_modifyitems(config, items):
    gpu = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu:
        return
    rest = [it for it in items if not it.get_closest_marker("gpu")]
    gpu.sort(key=_gpu_rank)   # stable: file order within one rank
    items[:] = rest + gpu


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
