"""c4a0_amd -- MI355X-native self-play generator for c4a0 (Connect Four AlphaZero).

Hot path only: the GPU-resident batched MCTS behind the reference's `c4a0_rust.play_games`
API (reference rust/src/pybridge.rs:20-53).  See DESIGN.md.
"""
N_COLS = 7           # reference rust/src/lib.rs:28
N_ROWS = 6           # reference rust/src/lib.rs:29
BUF_N_CHANNELS = 2   # reference rust/src/lib.rs:30

from .results import GameMetadata, GameResult, PlayGamesResult, Sample  # noqa: E402,F401


def __getattr__(name):  # torch / the HIP library are loaded on first use of the entry points
    if name in ("play_games", "run_tui"):
        from . import api
        return getattr(api, name)
    raise AttributeError(name)
