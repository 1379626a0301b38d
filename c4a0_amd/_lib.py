"""ctypes binding of libc4a0_hip.so (the C ABI declared in include/c4a0_hip.h).

The library is the product: there is no CPU fallback.  If it is missing, `lib()` raises with
the build command instead of degrading to something else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("C4A0_HIP_LIB", "libc4a0_hip.so"))  # C4A0_HIP_LIB: diagnostic builds

OK, ERR_BAD_ARG, ERR_HIP, ERR_NAN_IN_TREE, ERR_DEGENERATE_POLICY, ERR_ARENA_OVERFLOW, ERR_NOT_BOUND, ERR_NO_DEVICE, ERR_ILLEGAL_MOVE, ERR_CANCELLED = range(10)
STATUS_NAMES = {
    0: "C4_OK", 1: "C4_ERR_BAD_ARG", 2: "C4_ERR_HIP", 3: "C4_ERR_NAN_IN_TREE", 4: "C4_ERR_DEGENERATE_POLICY",
    5: "C4_ERR_ARENA_OVERFLOW", 6: "C4_ERR_NOT_BOUND", 7: "C4_ERR_NO_DEVICE", 8: "C4_ERR_ILLEGAL_MOVE", 9: "C4_ERR_CANCELLED",
}
FLAG_NO_MOVES = 1
FLAG_ONE_SIM_PER_STEP = 2
FLAG_RECLAIM = 4          # include/c4a0_hip.h C4_FLAG_RECLAIM: the tree arena is reclaimed while a game is played
FLAG_NO_RECLAIM = 8       # ... never, also where the default sizing would
MAX_SAMPLES_PER_GAME = 43
ABI_VERSION = 10   # include/c4a0_hip.h C4_ABI_VERSION: the signatures below are that version's
STRUCT_LAYOUT_SINCE = 7   # the ABI version that last changed a structure's layout (c4_config.reclaim_period, c4_counters.reclaim_*)


class C4Error(RuntimeError):
    def __init__(self, status: int, detail: str = ""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {detail}" if detail else STATUS_NAMES.get(status, str(status)))


class GameMetadataC(C.Structure):
    _fields_ = [("game_id", C.c_uint64), ("player0_id", C.c_uint64), ("player1_id", C.c_uint64)]


class SampleRec(C.Structure):
    _fields_ = [("game_id", C.c_uint64), ("mask", C.c_uint64), ("value", C.c_uint64), ("policy", C.c_float * 7),
                ("q_penalty", C.c_float), ("q_no_penalty", C.c_float), ("meta", C.c_uint32)]


class Config(C.Structure):
    _fields_ = [("n_slots", C.c_uint32), ("blocks_per_slot", C.c_uint32), ("n_mcts_iterations", C.c_uint32),
                ("c_exploration", C.c_float), ("c_ply_penalty", C.c_float), ("planes_dtype", C.c_uint32),
                ("flags", C.c_uint32), ("device", C.c_int32), ("reclaim_period", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("sims", "select_levels", "backup_nodes", "expansions", "moves", "games_done",
                                          "ref_skipped_sims", "samples", "games_started", "step_kernel_ns", "step_launches",
                                          "eval_cache_probes", "eval_cache_hits", "reclaim_passes", "reclaim_blocks")] + \
               [("error", C.c_uint32), ("error_slot", C.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class NetworkBf16(C.Structure):
    """include/c4a0_hip.h c4_network_bf16."""
    _fields_ = [("channels", C.c_uint32), ("n_blocks", C.c_uint32), ("tower_w0", C.c_void_p), ("tower_w", C.c_void_p), ("tower_bias", C.c_void_p),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("n_policy_hidden", C.c_uint32), ("n_value_hidden", C.c_uint32),
                ("policy_w", C.c_void_p * 8), ("policy_b", C.c_void_p * 8), ("value_w", C.c_void_p * 8), ("value_b", C.c_void_p * 8),
                ("policy_out_w", C.c_void_p), ("value_out_w", C.c_void_p), ("policy_out_b", C.c_void_p), ("value_out_b", C.c_void_p)]


class PlayOptions(C.Structure):
    """include/c4a0_hip.h c4_play_options."""
    _fields_ = [("device", C.c_int32), ("resident_games", C.c_uint32), ("concurrent_sessions", C.c_uint32), ("steps_per_graph", C.c_uint32),
                ("tail_steps_per_graph", C.c_uint32), ("blocks_per_slot", C.c_uint32), ("flags", C.c_uint32), ("reclaim_period", C.c_uint32),
                ("dirichlet_alpha", C.c_float), ("dirichlet_epsilon", C.c_float), ("eval_cache_entries", C.c_uint64)]


class PlayPhases(C.Structure):
    """include/c4a0_hip.h c4_play_phases."""
    _fields_ = [("setup_s", C.c_double), ("capture_s", C.c_double), ("steady_s", C.c_double), ("tail_s", C.c_double), ("drain_s", C.c_double),
                ("rounds", C.c_uint64), ("rounds_until_all_started", C.c_uint64), ("graph_captures", C.c_uint32), ("resident_games", C.c_uint32),
                ("sessions", C.c_uint32), ("rows_at_end", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/c4a0_hip.h declares: name -> (restype, argtypes)
_P = C.POINTER
_vp = C.c_void_p
SIGNATURES = {
    "c4_last_error_string": (C.c_char_p, []),
    "c4_abi_version": (C.c_int, []),
    "c4_source_hash": (C.c_char_p, []),
    "c4_device_count": (C.c_int, [_P(C.c_int)]),
    "c4_session_create": (C.c_int, [_P(Config), _P(_vp)]),
    "c4_session_destroy": (C.c_int, [_vp]),
    "c4_session_set_games": (C.c_int, [_vp, _P(GameMetadataC), C.c_uint64, _P(C.c_uint64), _P(C.c_uint64)]),
    "c4_session_bind_io": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "c4_session_set_dirichlet": (C.c_int, [_vp, C.c_float, C.c_float]),
    "c4_session_bind_leaf_models": (C.c_int, [_vp, _vp]),
    "c4_session_set_eval_cache": (C.c_int, [_vp, C.c_uint64, C.c_uint32]),
    "c4_session_progress": (C.c_int, [_vp, _P(C.c_uint64), _P(C.c_uint64), _P(C.c_uint32)]),
    "c4_session_compact": (C.c_int, [_vp, C.c_uint32, _P(C.c_uint32), _P(C.c_uint32)]),
    "c4_session_start": (C.c_int, [_vp]),
    "c4_session_step": (C.c_int, [_vp]),
    "c4_session_set_step_shape": (C.c_int, [_vp, C.c_uint32]),
    "c4_session_step_head_out": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "c4_session_set_timing": (C.c_int, [_vp, C.c_int]),
    "c4_session_counters": (C.c_int, [_vp, _P(Counters)]),
    "c4_session_poll": (C.c_int, [_vp, _P(C.c_uint64), _P(C.c_uint32)]),
    "c4_session_arena": (C.c_int, [_vp, _P(C.c_uint64), _P(C.c_uint32), _P(C.c_uint32)]),
    "c4_session_sample_counts": (C.c_int, [_vp, _P(C.c_uint32), C.c_uint64]),
    "c4_session_drain_samples": (C.c_int, [_vp, _P(SampleRec), C.c_uint64, _P(C.c_uint64)]),
    "c4_session_pack_samples": (C.c_int, [_vp, _vp, C.c_uint64, _P(C.c_uint64)]),
    "c4_session_debug_phase_stamps": (C.c_int, [_vp, _P(C.c_uint64), C.c_uint64, _P(C.c_uint64)]),
    "c4_session_sample_store": (C.c_int, [_vp, _P(_vp), _P(_vp), _P(C.c_uint64)]),
    "c4_session_root_stats": (C.c_int, [_vp, C.c_uint32, _P(C.c_float), _P(C.c_float), _P(C.c_float),
                                        _P(C.c_uint64), _P(C.c_uint64), _P(C.c_uint64)]),
    "c4_trim_cached_memory": (C.c_int, []),
    "c4_records_to_cbor": (C.c_int, [_vp, _vp, C.c_uint64, _vp, C.c_uint64, _vp, C.c_uint64, _P(C.c_uint64)]),
    "c4_shuffle_games": (C.c_int, [C.c_uint64, C.c_uint64, _vp]),
    "c4_play_games_cancel": (None, []),
    "c4_play_games_bf16": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_float, C.c_float, _P(NetworkBf16), _P(PlayOptions), _vp, _vp, C.c_uint64,
                                     _P(C.c_uint64), _P(Counters), _P(PlayPhases)]),
    "c4_cbor_to_records": (C.c_int, [_vp, C.c_uint64, _vp, _vp, C.c_uint64, _vp, C.c_uint64, _P(C.c_uint64), _P(C.c_uint64)]),
    "c4_session_leaf_keys": (C.c_int, [_vp, _vp]),
    "c4_session_unique_leaves": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "c4_session_scatter_outputs": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "c4_session_step_gather": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "c4_session_leaves": (C.c_int, [_vp, _P(C.c_uint64), _P(C.c_uint64), _P(C.c_uint32), _P(C.c_uint32)]),
    "c4_pos_ops": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    "c4_encode_planes": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, _vp, _vp]),
    "c4_expf_logf": (C.c_int, [_vp, C.c_uint64, C.c_int, _vp, _vp]),
    "c4_softmax7": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp]),
    "c4_apply_temperature": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp]),
    "c4_conv_tower_bf16": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint32, _vp]),
    "c4_linear_bf16": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "c4_linear_bf16_tile_map": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P(C.c_uint32), C.c_uint32, _P(C.c_uint32)]),
    "c4_planes_from_f32": (C.c_int, [_vp, _vp, C.c_uint32, _vp, C.c_uint32, _vp]),
    "c4_head_out_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "c4_dirichlet": (C.c_int, [_vp, _vp, _vp, C.c_float, C.c_uint64, _vp, _vp]),
    "c4_sample_move": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension is the product and there is no fallback. "
                "Build it with `python c4a0_amd/csrc/build.py` (hipcc --offload-arch=gfx950).")
        # PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7, but its libraries ask
        # for "libamdhip64.so").  If our library -- linked against "libamdhip64.so.7" -- pulled in
        # /opt/rocm's copy first, torch would load a SECOND HIP runtime into the process and one
        # of the two would see no device.  Loading torch first makes the loader resolve our
        # dependency to the runtime torch already mapped, so kernels, streams and tensors share it.
        import torch  # noqa: F401

        L = C.CDLL(LIB_PATH)
        # C4A0_HIP_LIB (tools/build_variant.py: a library built from ANOTHER revision's kernels for a same-box A/B) may predate
        # entry points and the ABI version of this binding: only there is that tolerated -- the product library must match exactly
        diagnostic = "C4A0_HIP_LIB" in os.environ
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name, None)
            if fn is None:
                if diagnostic:
                    continue
                raise ImportError(f"{LIB_PATH} does not export {name}")
            fn.restype = res
            fn.argtypes = args
        if L.c4_abi_version() != ABI_VERSION and not diagnostic:
            raise ImportError(f"{LIB_PATH} implements C ABI version {L.c4_abi_version()}, this binding is written for {ABI_VERSION}")
        if diagnostic and L.c4_abi_version() < STRUCT_LAYOUT_SINCE:
            # an A/B library may lack newer entry points, but the structures this binding hands it (Config) and reads back (Counters:
            # error / error_slot sit behind the counters) must be the ones it was compiled with -- an older layout would report a failed
            # run as a clean one
            raise ImportError(f"{LIB_PATH} implements C ABI version {L.c4_abi_version()}: c4_config / c4_counters changed layout in version "
                              f"{STRUCT_LAYOUT_SINCE}; build the A/B library from a revision at or after it")
        # a library compiled from other sources than the ones beside it is refused, not used
        # (C4A0_HIP_LIB names a diagnostic build with its own flags: not compared)
        if "C4A0_HIP_LIB" not in os.environ:
            from .csrc import build as _build

            have, want = L.c4_source_hash().decode(), _build.source_hash()
            if have != want:
                raise ImportError(f"{LIB_PATH} was built from different sources (hash {have}, tree {want}): "
                                  "rebuild it with `python c4a0_amd/csrc/build.py`")
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != OK:
        msg = lib().c4_last_error_string()
        raise C4Error(status, msg.decode() if msg else "")
