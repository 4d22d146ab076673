import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

MAIN_IDS = ["MiniGrid-Empty-8x8-v0", "MiniGrid-DoorKey-8x8-v0", "MiniGrid-LavaCrossingS9N1-v0", "BabyAI-GoToRedBall-v0"]
EXTRA_IDS = ["MiniGrid-Empty-5x5-v0", "MiniGrid-Empty-Random-6x6-v0", "MiniGrid-Empty-16x16-v0",
             "MiniGrid-DoorKey-5x5-v0", "MiniGrid-DoorKey-6x6-v0", "MiniGrid-DoorKey-16x16-v0",
             "MiniGrid-LavaCrossingS9N2-v0", "MiniGrid-LavaCrossingS9N3-v0", "MiniGrid-LavaCrossingS11N5-v0",
             "MiniGrid-SimpleCrossingS9N1-v0", "MiniGrid-SimpleCrossingS11N5-v0", "BabyAI-GoToRedBallNoDists-v0",
             "MiniGrid-Empty-6x6-v0", "MiniGrid-Empty-Random-5x5-v0", "MiniGrid-SimpleCrossingS9N2-v0", "MiniGrid-SimpleCrossingS9N3-v0"]
WIDE_IDS = ["MiniGrid-LavaGapS5-v0", "MiniGrid-LavaGapS6-v0", "MiniGrid-LavaGapS7-v0", "MiniGrid-DistShift1-v0",
            "MiniGrid-DistShift2-v0", "MiniGrid-FourRooms-v0", "MiniGrid-Fetch-5x5-N2-v0", "MiniGrid-Fetch-6x6-N2-v0",
            "MiniGrid-Fetch-8x8-N3-v0", "MiniGrid-GoToDoor-5x5-v0", "MiniGrid-GoToDoor-6x6-v0", "MiniGrid-GoToDoor-8x8-v0",
            "MiniGrid-Unlock-v0", "MiniGrid-UnlockPickup-v0", "MiniGrid-BlockedUnlockPickup-v0",
            "MiniGrid-RedBlueDoors-6x6-v0", "MiniGrid-RedBlueDoors-8x8-v0", "MiniGrid-MemoryS17Random-v0",
            "MiniGrid-MemoryS13Random-v0", "MiniGrid-MemoryS13-v0", "MiniGrid-MemoryS11-v0", "MiniGrid-MemoryS9-v0",
            "MiniGrid-MemoryS7-v0", "MiniGrid-KeyCorridorS3R1-v0", "MiniGrid-KeyCorridorS3R2-v0", "MiniGrid-KeyCorridorS3R3-v0",
            "MiniGrid-KeyCorridorS4R3-v0", "MiniGrid-KeyCorridorS5R3-v0", "MiniGrid-KeyCorridorS6R3-v0",
            "MiniGrid-Dynamic-Obstacles-5x5-v0", "MiniGrid-Dynamic-Obstacles-Random-5x5-v0", "MiniGrid-Dynamic-Obstacles-6x6-v0",
            "MiniGrid-Dynamic-Obstacles-Random-6x6-v0", "MiniGrid-Dynamic-Obstacles-8x8-v0", "MiniGrid-Dynamic-Obstacles-16x16-v0",
            "MiniGrid-GoToObject-6x6-N2-v0", "MiniGrid-GoToObject-8x8-N2-v0",
            "MiniGrid-LockedRoom-v0", "MiniGrid-Playground-v0", "MiniGrid-MultiRoom-N2-S4-v0", "MiniGrid-MultiRoom-N4-S5-v0",
            "MiniGrid-MultiRoom-N4-S5-v1", "MiniGrid-MultiRoom-N6-v0",
            "BabyAI-PickupDist-v0", "BabyAI-PickupDistDebug-v0", "BabyAI-OneRoomS8-v0", "BabyAI-OneRoomS12-v0",
            "BabyAI-OneRoomS16-v0", "BabyAI-OneRoomS20-v0", "BabyAI-OpenRedDoor-v0",
            "BabyAI-FindObjS5-v0", "BabyAI-FindObjS6-v0", "BabyAI-FindObjS7-v0",
            "BabyAI-UnlockLocal-v0", "BabyAI-UnlockLocalDist-v0", "BabyAI-KeyCorridor-v0", "BabyAI-KeyCorridorS3R1-v0",
            "BabyAI-KeyCorridorS3R2-v0", "BabyAI-KeyCorridorS3R3-v0", "BabyAI-KeyCorridorS4R3-v0", "BabyAI-KeyCorridorS5R3-v0",
            "BabyAI-KeyCorridorS6R3-v0",
            "BabyAI-GoToRedBallGrey-v0", "BabyAI-GoToRedBlueBall-v0", "BabyAI-GoToObj-v0", "BabyAI-GoToObjS4-v0",
            "BabyAI-GoToObjS6-v1", "BabyAI-GoToLocal-v0", "BabyAI-GoToLocalS5N2-v0", "BabyAI-GoToLocalS6N2-v0",
            "BabyAI-GoToLocalS6N3-v0", "BabyAI-GoToLocalS6N4-v0", "BabyAI-GoToLocalS7N4-v0", "BabyAI-GoToLocalS7N5-v0",
            "BabyAI-GoToLocalS8N2-v0", "BabyAI-GoToLocalS8N3-v0", "BabyAI-GoToLocalS8N4-v0", "BabyAI-GoToLocalS8N5-v0",
            "BabyAI-GoToLocalS8N6-v0", "BabyAI-GoToLocalS8N7-v0"]
# ids moved from the oracle-only list onto the device in round 2 (their goldens: 400-step rollouts, written by main_oracle_only)
WIDE2_IDS = ["MiniGrid-ObstructedMaze-1Dl-v0", "MiniGrid-ObstructedMaze-1Dlh-v0", "MiniGrid-ObstructedMaze-1Dlhb-v0",
             "MiniGrid-ObstructedMaze-2Dl-v0", "MiniGrid-ObstructedMaze-2Dlh-v0", "MiniGrid-ObstructedMaze-2Dlhb-v0",
             "MiniGrid-ObstructedMaze-1Q-v0", "MiniGrid-ObstructedMaze-2Q-v0", "MiniGrid-ObstructedMaze-Full-v0",
             "MiniGrid-ObstructedMaze-2Dlhb-v1", "MiniGrid-ObstructedMaze-1Q-v1", "MiniGrid-ObstructedMaze-2Q-v1",
             "MiniGrid-ObstructedMaze-Full-v1", "MiniGrid-PutNear-6x6-N2-v0", "MiniGrid-PutNear-8x8-N3-v0",
             "BabyAI-GoTo-v0", "BabyAI-GoToOpen-v0", "BabyAI-GoToObjMaze-v0", "BabyAI-GoToObjMazeOpen-v0",
             "BabyAI-GoToObjMazeS4R2-v0", "BabyAI-GoToObjMazeS4-v0", "BabyAI-GoToObjMazeS5-v0", "BabyAI-GoToObjMazeS6-v0",
             "BabyAI-GoToObjMazeS7-v0", "BabyAI-Pickup-v0", "BabyAI-Open-v0",
             "BabyAI-UnlockPickup-v0", "BabyAI-UnlockPickupDist-v0", "BabyAI-BlockedUnlockPickup-v0", "BabyAI-UnlockToUnlock-v0", "BabyAI-Unlock-v0", "BabyAI-KeyInBox-v0",
             "BabyAI-GoToDoor-v0", "BabyAI-GoToObjDoor-v0", "BabyAI-GoToImpUnlock-v0", "BabyAI-UnblockPickup-v0", "BabyAI-PickupAbove-v0",
             "BabyAI-PutNextLocal-v0", "BabyAI-PutNextLocalS5N3-v0", "BabyAI-PutNextLocalS6N4-v0", "BabyAI-PutNextS4N1-v0",
             "BabyAI-PutNextS5N2-v0", "BabyAI-PutNextS5N1-v0", "BabyAI-PutNextS6N3-v0", "BabyAI-PutNextS7N4-v0",
             "BabyAI-PutNextS5N2Carrying-v0", "BabyAI-PutNextS6N3Carrying-v0", "BabyAI-PutNextS7N4Carrying-v0", "BabyAI-ActionObjDoor-v0",
             "BabyAI-OpenDoor-v0", "BabyAI-OpenDoorDebug-v0", "BabyAI-OpenDoorColor-v0", "BabyAI-OpenDoorLoc-v0"]
# the sentence levels: instruction trees (Before / After / And), object identity, per-episode max_steps; missions are sentences
SENTENCE_IDS = ["BabyAI-OpenTwoDoors-v0", "BabyAI-OpenRedBlueDoors-v0", "BabyAI-OpenRedBlueDoorsDebug-v0", "BabyAI-OpenDoorsOrderN2-v0",
                "BabyAI-OpenDoorsOrderN4-v0", "BabyAI-OpenDoorsOrderN2Debug-v0", "BabyAI-OpenDoorsOrderN4Debug-v0", "BabyAI-MoveTwoAcrossS5N2-v0",
                "BabyAI-MoveTwoAcrossS8N9-v0", "BabyAI-PickupLoc-v0", "BabyAI-GoToSeq-v0", "BabyAI-GoToSeqS5R2-v0",
                "BabyAI-Synth-v0", "BabyAI-SynthLoc-v0", "BabyAI-SynthSeq-v0", "BabyAI-MiniBossLevel-v0",
                "BabyAI-BossLevel-v0", "BabyAI-BossLevelNoUnlock-v0"]
ALL_IDS = MAIN_IDS + EXTRA_IDS + WIDE_IDS + WIDE2_IDS + SENTENCE_IDS


# an abort inside the HIP / HSA runtime leaves no message: have the library print the native backtrace first (mg_api.hip)
os.environ.setdefault("MG_ABORT_BACKTRACE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))


# BabyAI-SynthS5R2-v0: the reference never returns from about 0.4 % of its resets (RoomGrid.place_agent, roomgrid.py:327-332), so its
# goldens are recorded on the seeds where it does (make_golden.py main_synths5r2, with the (seed, episode) pairs where it does not); the
# tests that draw thousands of episodes run it with stuck_place_agent="redraw" (tests/test_gpu_synths5r2.py)
STUCK_IDS = ["BabyAI-SynthS5R2-v0"]

# restated and pinned in the oracle only (oracle groundwork for the next widening step): not in the GPU lists
ORACLE_ONLY_IDS = []


def full_obs_supported(env_id: str) -> bool:
    """Round 1 refused FullyObs / Symbolic observations of the 25 x 25 MultiRoom maps (165 KB of LDS staging per 64 envs).
    With 4 lanes per env a wavefront holds 16 envs (41 KB): every id is supported now."""
    return True
