"""A part of the GPU parity suite (tests/test_gpu_parity.py, `-m gpu`) re-run on the CPU: the same test functions, unmodified, against the library built
for the host SIMT emulator of tests/emu (MINIGRID_AMD_LIB, in a subprocess).  What that adds to test_emu_cpu.py (kernels vs the ORACLE): the kernels
against the REFERENCE's own golden vectors (tests/golden/, recorded from the unmodified reference) -- generators over three consecutive episodes,
400-step random and solver rollouts in both observation modes, the observation wrappers, NoDeath, stepping past termination, RGB frames,
BABYAI_DONE_ACTIONS -- for the four BASELINE levels, and the facade-level behaviour tests (reset masks and stream continuation, checkpoint round trip,
seed + index sharding invariance, max_steps truncation, unknown actions, the reference's LavaCrossing doctest, two handles with different LDS sizes).
The selection is by run time on the emulator (a 64-lane wavefront is 64 fibers here): everything below finishes in about a minute; batch sizes in the
thousands stay on the GPU.  The spare-episode rings are cut to four slots (MG_SPARE_RING=4: a ring of 128-256 episodes per env is generated at every
explicit reset)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

MAIN = "(Empty-8x8-v0 or DoorKey-8x8-v0 or LavaCrossingS9N1-v0 or GoToRedBall-v0 or rgb)"
FUNCTIONAL = ["test_autoreset_disabled_steps_past_max_steps_like_the_reference", "test_max_steps_truncation_kat", "test_unknown_action_raises_value_error",
              "test_reference_doctest_lava_seed2", "test_seed_int_means_seed_plus_index_and_is_shard_invariant", "test_reset_mask_and_stream_continuation",
              "test_state_and_rng_checkpoint_roundtrip", "test_img_and_fully_obs_wrappers", "test_philox_mode_generates_valid_deterministic_maps",
              "test_nodeath_and_onehot_view5_compose_vs_oracle", "test_dict_observation_space_wrapper", "test_two_handles_with_different_lds_sizes_coexist"]


def _rerun(module, expr, at_least):
    import build_emu
    lib = build_emu.build([])
    env = dict(os.environ, MINIGRID_AMD_LIB=lib, MINIGRID_AMD_NO_TORCH="1", MINIGRID_AMD_EMU_RERUN="1", MG_SPARE_RING="4")
    for k in [k for k in env if (k.startswith("MG_") and k != "MG_SPARE_RING") or k.startswith("EMU_")]:
        del env[k]
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", module), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", expr],
                         env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
    m = re.search(r"(\d+) passed", tail)
    assert out.returncode == 0 and m and "failed" not in tail and "error" not in tail, (out.returncode, out.stdout[-3000:], out.stderr[-2000:])
    assert int(m.group(1)) >= at_least, tail
    return int(m.group(1))


def test_reference_goldens_of_the_baseline_levels_on_the_emulated_kernels():
    _rerun("test_gpu_parity.py", "goldens and " + MAIN, 55)


def test_facade_behaviour_tests_on_the_emulated_kernels():
    _rerun("test_gpu_parity.py", " or ".join(FUNCTIONAL), len(FUNCTIONAL))


def test_philox_generator_kernels_on_the_emulator():
    """MG_RNG_PHILOX (tests/test_gpu_philox.py): the Philox-keyed generator kernels' episodes injected into the oracle and stepped side by side, and the
    fused path against stepping -- with 200 envs instead of thousands (the tests scale themselves down under MINIGRID_AMD_EMU_RERUN)."""
    _rerun("test_gpu_philox.py", "state_injection or deterministic", 5)


def test_composed_same_step_on_the_emulated_kernels():
    """SAME_STEP of DynamicObstacles / the sentence levels outside the default view (composed by the facade, round 5) against the oracle's SAME_STEP mode."""
    _rerun("test_gpu_roll.py", "composed_for_the_other_observation_modes", 4)
