"""ctypes binding of libminigrid_hip.so (include/minigrid_hip.h) — the thin layer the reference's Gymnasium
surface sits on.  There is NO CPU fallback: if the library cannot be built/loaded this module raises."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

MG_ABI_VERSION = 3
MG_OK, MG_ERR_INVALID, MG_ERR_HIP, MG_ERR_BAD_ACTION, MG_ERR_GENERATOR, MG_ERR_NO_DEVICE, MG_ERR_OOB, MG_ERR_TRACKED = 0, -1, -2, -3, -4, -5, -6, -7
OBS_PARTIAL, OBS_FULL, OBS_ONEHOT, OBS_SYMBOLIC, OBS_RGB_PARTIAL, OBS_RGB = 0, 1, 2, 3, 4, 5
AUTORESET_NEXT_STEP, AUTORESET_DISABLED, AUTORESET_SAME_STEP = 0, 1, 2
RNG_PCG64, RNG_PHILOX = 0, 1
ACT_U8, ACT_I32, ACT_I64 = 0, 1, 2


class MgConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "env_kind", "width", "height", "max_steps", "see_through_walls", "agent_view_size",
        "obs_mode", "autoreset_mode", "rng_mode", "num_envs", "agent_start_x", "agent_start_y", "agent_start_dir",
        "num_crossings", "obstacle_type", "num_dists", "null_stream_sync", "strip2_row", "no_death_mask")] + [
        ("death_cost", C.c_double), ("room_size", C.c_int32), ("random_length", C.c_int32), ("env_index_base", C.c_int64),
        ("tile_size", C.c_int32), ("rgb_highlight", C.c_int32), ("spare_ring", C.c_int32), ("traj_slots", C.c_int32),
        ("babyai_done_actions", C.c_int32)]


class MgOutputs(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("reward", C.c_void_p), ("terminated", C.c_void_p), ("truncated", C.c_void_p),
                ("direction", C.c_void_p), ("mission_id", C.c_void_p), ("obs_bytes_per_env", C.c_int64),
                ("num_envs", C.c_int64), ("action", C.c_void_p), ("traj_slots", C.c_int64), ("slot_bytes", C.c_int64),
                ("record_bytes", C.c_int64), ("max_fused_steps", C.c_int64), ("sentence", C.c_void_p), ("scalar_stride", C.c_int64)]


class MiniGridHipError(RuntimeError):
    pass


_lib = None

# every symbol include/minigrid_hip.h declares (tests/test_abi_cpu.py checks the built library exports all of them)
SYMBOLS = ["mg_create", "mg_destroy", "mg_set_obs_config", "mg_reset", "mg_step", "mg_rollout", "mg_rollout_block", "mg_step_many", "mg_get_outputs", "mg_copy_outputs",
           "mg_copy_slot", "mg_copy_sentence", "mg_selftest_stream",
           "mg_sync", "mg_get_state", "mg_set_state", "mg_state_size", "mg_save_state", "mg_load_state", "mg_get_rng", "mg_set_rng", "mg_timer_start", "mg_timer_stop",
           "mg_get_counters", "mg_ring_depth", "mg_last_error", "mg_abi_version", "mg_build_info", "mg_device_count", "mg_selftest_vis_row",
           "mg_selftest_reward_lut", "mg_selftest_pack_cell", "mg_selftest_vis_row_n", "mg_render_tiles",
           "mg_selftest_obs7", "mg_selftest_vis_row_carry", "mg_selftest_prims", "mg_selftest_dynobs", "mg_selftest_verify", "mg_selftest_transition", "mg_selftest_generate", "mg_selftest_obs_full"]


def lib_path() -> str:
    return _build.LIB


def load():
    """Load (building first if stale/missing) libminigrid_hip.so.  Imports torch first when it is available so
    that both use the same HIP runtime (see build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("MINIGRID_AMD_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401  (side effect: its libamdhip64.so becomes the process's HIP runtime)
        except Exception:
            pass
    path = os.environ.get("MINIGRID_AMD_LIB") or _build.LIB      # override: an alternative build of the same ABI
    if path == _build.LIB:
        # no-op when the library is newer than every source it is built from; a box without hipcc uses the shipped .so
        path = _build.build(force=os.environ.get("MINIGRID_AMD_REBUILD", "0") == "1", missing_hipcc_ok=True)
    L = C.CDLL(path)
    vp, i, u64 = C.c_void_p, C.c_int, C.c_uint64
    L.mg_create.argtypes = [C.POINTER(MgConfig), i, vp, C.POINTER(vp)]
    L.mg_destroy.argtypes = [vp]
    L.mg_set_obs_config.argtypes = [vp, C.POINTER(MgConfig)]
    L.mg_reset.argtypes = [vp, vp, vp]
    L.mg_step.argtypes = [vp, vp, i, i]
    L.mg_rollout.argtypes = [vp, i, u64, i]
    L.mg_step_many.argtypes = [vp, vp, i, i]
    L.mg_rollout_block.argtypes = [vp, i, u64, i]
    L.mg_copy_slot.argtypes = [vp, i, vp, vp, vp, vp, vp, vp, vp]
    L.mg_copy_sentence.argtypes = [vp, i, vp]
    L.mg_selftest_stream.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp]
    L.mg_get_outputs.argtypes = [vp, C.POINTER(MgOutputs)]
    L.mg_copy_outputs.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.mg_sync.argtypes = [vp]
    L.mg_get_state.argtypes = [vp, vp, vp]
    L.mg_set_state.argtypes = [vp, vp, vp]
    L.mg_get_rng.argtypes = [vp, vp]
    L.mg_state_size.argtypes = [vp, C.POINTER(C.c_int64)]
    L.mg_save_state.argtypes = [vp, vp, C.c_int64]
    L.mg_load_state.argtypes = [vp, vp, C.c_int64]
    L.mg_set_rng.argtypes = [vp, vp]
    L.mg_timer_start.argtypes = [vp]
    L.mg_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.mg_get_counters.argtypes = [vp, vp]
    L.mg_ring_depth.argtypes = [vp]
    L.mg_last_error.argtypes = [vp]
    L.mg_last_error.restype = C.c_char_p
    L.mg_build_info.argtypes = []
    L.mg_build_info.restype = C.c_char_p
    L.mg_selftest_vis_row.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mg_selftest_vis_row_n.argtypes = [C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mg_selftest_reward_lut.argtypes = [C.c_int32, vp]
    L.mg_render_tiles.argtypes = [C.c_int32, C.c_void_p]
    L.mg_selftest_obs7.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_int32, vp]
    L.mg_selftest_vis_row_carry.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mg_selftest_prims.argtypes = [C.c_int32, vp, vp, vp, vp, C.c_int32]
    L.mg_selftest_dynobs.argtypes = [C.c_int32] * 8 + [vp] * 6
    L.mg_selftest_verify.argtypes = [C.c_int32] * 4 + [vp] * 7
    L.mg_selftest_obs_full.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]
    L.mg_selftest_generate.argtypes = [C.POINTER(MgConfig), C.c_int32, C.c_int32] + [vp] * 7
    L.mg_selftest_transition.argtypes = [C.c_int32] * 8 + [C.c_double, C.c_int32] + [vp] * 8
    L.mg_selftest_pack_cell.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    if L.mg_abi_version() != MG_ABI_VERSION:
        raise MiniGridHipError(f"libminigrid_hip ABI {L.mg_abi_version()} != binding {MG_ABI_VERSION}")
    _lib = L
    return L


def check(rc: int, handle=None):
    """Map mg_status to the exception the reference would raise at the same point."""
    if rc == MG_OK:
        return
    msg = (load().mg_last_error(handle) or b"").decode(errors="replace")
    if rc == MG_ERR_BAD_ACTION:
        raise ValueError(msg or "Unknown action")                 # minigrid_env.py:584-585
    if rc == MG_ERR_GENERATOR:
        raise RecursionError(msg or "rejection sampling failed")  # minigrid_env.py:342-343
    if rc == MG_ERR_OOB:
        raise AssertionError(msg)                                 # core/grid.py:74-78
    if rc == MG_ERR_INVALID:
        raise ValueError(msg or "invalid argument")
    raise MiniGridHipError(f"libminigrid_hip error {rc}: {msg}")
