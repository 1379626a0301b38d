"""c4a0_amd -- MI355X-native self-play generator for c4a0 (Connect Four AlphaZero).

Hot path only: the GPU-resident batched MCTS behind the reference's `c4a0_rust.play_games`
API (reference rust/src/pybridge.rs:20-53).  See DESIGN.md.
"""
N_COLS = 7           # reference rust/src/lib.rs:28
N_ROWS = 6           # reference rust/src/lib.rs:29
BUF_N_CHANNELS = 2   # reference rust/src/lib.rs:30

from .results import GameMetadata, GameResult, PlayGamesResult, Sample  # noqa: E402,F401


def __getattr__(name):  # torch / the HIP library are loaded on first use of the entry points
    if name in ("play_games", "run_tui", "DeviceCallback", "trim_cached_memory"):
        from . import api
        return getattr(api, name)
    if name == "play_games_native":
        from .native import play_games_native
        return play_games_native
    if name == "play_games_sharded":
        from .distributed import play_games_sharded
        return play_games_sharded
    raise AttributeError(name)


def install_as_c4a0_rust():
    """Make `import c4a0_rust` resolve to this package and make pickles interchangeable with the
    reference's: its PyO3 classes are registered under module "c4a0_rust" (pybridge.rs:60,
    types.rs) and `games.pkl` stores `c4a0_rust.PlayGamesResult` + CBOR state
    (training.py:48-67).  After this call the reference's training.py / tournament.py run on the
    GPU generator unmodified, and either side can load the other's pickles."""
    import sys

    from . import results

    for cls in (results.GameMetadata, results.GameResult, results.PlayGamesResult, results.Sample):
        cls.__module__ = "c4a0_rust"
    sys.modules["c4a0_rust"] = sys.modules[__name__]
    return sys.modules[__name__]
