"""ctypes binding of libzenv_hip.so (include/zenv.h).  No PyTorch, no CPU fallback.

The library is built in-tree by ``build.py`` (hipcc, gfx950).  Importing this module fails
loudly when the shared object is missing; every compute call fails with ``ZenvError`` when
no MI355X is present.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZENV_LIB_PATH") or os.path.join(_HERE, "lib", "libzenv_hip.so")   # env: diagnostic builds

MAX_ZONES = 32
OBS_DIM = 8

TASK_TSP, TASK_TIMED_TSP, TASK_COLOUR_MATCH = 0, 1, 2
POLICY_UNIFORM, POLICY_GREEDY, POLICY_MLP_MEAN, POLICY_MLP_SAMPLE = 0, 1, 2, 3
KERNEL_LANE_PER_ENV, KERNEL_WAVE_PER_ENV = 0, 1
HIP_STREAM_LEGACY = 1       # hipStreamLegacy: the null stream as an explicit handle (hip_runtime_api.h)
ROLLOUT_UNFUSED = 1
ROLLOUT_PER_STEP = 2
ROLLOUT_ASYNC = 4
ROLLOUT_CHUNK = 256
COMM_ID_BYTES = 128
ORDER_FRESH_FIRST_OBS = 1
CHUNK_NO_RESET, CHUNK_RESET_EVERY, CHUNK_RESET_LAST = 0, 1, 2

E_ARG, E_HIP, E_STATE, E_LAYOUT, E_DONE, E_RANGE = -1, -2, -3, -4, -5, -6

(F_OBS, F_ZONE_OBS, F_REWARD, F_DONE, F_GOAL_MET, F_EP_RETURN, F_EP_LEN, F_LAST_RETURN,
 F_LAST_LEN, F_EPISODES, F_VISIT_COUNT, F_SEED, F_ACTIONS, F_POLICY_MU, F_POLICY_STD, F_POLICY_VALUE,
 F_SHAPED_REWARD, F_NEED_GOAL, F_AVAILABLE_GOALS, F_GOAL,
 F_EXP_OBS, F_EXP_ZONE_OBS, F_EXP_ACTION, F_EXP_LOG_PROB, F_EXP_VALUE, F_EXP_REWARD, F_EXP_MASK,
 F_EXP_ADVANTAGE, F_EXP_RETURN, F_ORDER_VAL, F_EXCEPTION, F_POLICY_VALUE_SIGMA, F_ORDER_POS,
 F_CHUNK_REWARD, F_CHUNK_DONE, F_CHUNK_ACTIONS) = range(36)
(RESULT_OBS, RESULT_REWARD, RESULT_DONE, RESULT_GOAL_MET, RESULT_EXCEPTION, RESULT_ZONE_OBS) = range(6)
N_RESULTS = 6


MLP_TENSORS = ("zone_w1", "zone_b1", "zone_w2", "zone_b2", "zone_w3", "zone_b3", "comb_w", "comb_b",
               "enc_w", "enc_b", "mu_w", "mu_b", "std_w", "std_b")
MLP_CRITIC_TENSORS = ("critic_w1", "critic_b1", "critic_w2", "critic_b2")   # optional, all or none
MLP_SIGMA_TENSORS = ("critic_sigma_w", "critic_sigma_b")   # optional: the distributional critic (critic_w2 = critic_mu)
MLP_BF16, MLP_F32, MLP_BF16X3, MLP_F16X3, MLP_F16 = 0, 1, 2, 3, 4


class MlpWeights(C.Structure):
    """struct zenv_mlp_weights (include/zenv.h): host float32 tensors in state_dict layout."""
    _fields_ = [("h_dim", C.c_int32), ("precision", C.c_int32)] + [
        (n, C.c_void_p) for n in MLP_TENSORS + MLP_CRITIC_TENSORS + MLP_SIGMA_TENSORS]


class ZenvError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"zenv error {code}: {message}")
        self.code = code


class Config(C.Structure):
    """Mirror of ``zenv_config`` (include/zenv.h)."""
    _fields_ = [
        ("task", C.c_int32), ("num_zones", C.c_int32), ("num_steps", C.c_int32),
        ("max_cd", C.c_int32), ("frameskip", C.c_int32), ("kernel", C.c_int32),
        ("zones_size", C.c_double), ("zones_keepout", C.c_double), ("robot_keepout", C.c_double),
        ("extent", C.c_double), ("placements_margin", C.c_double),
        ("time_saved_reward", C.c_double), ("beta_a", C.c_double), ("beta_b", C.c_double),
        ("timestep", C.c_double), ("mass", C.c_double), ("com_x", C.c_double),
        ("inertia_zz", C.c_double), ("damping", C.c_double * 3), ("gear", C.c_double),
        ("forcerange", C.c_double), ("vel_kv", C.c_double), ("reward_exception", C.c_double),
        ("n_zones_locations", C.c_int32), ("n_robot_locations", C.c_int32), ("robot_rot_fixed", C.c_int32),
        ("visited0", C.c_uint32), ("robot_rot", C.c_double), ("robot_location", C.c_double * 2),
        ("zones_locations", (C.c_double * 2) * MAX_ZONES),
    ]

    def copy(self):
        out = Config()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(Config))
        return out


# every symbol include/zenv.h declares: name -> (restype, argtypes)
_H = C.c_void_p
_PROTOTYPES = {
    "zenv_last_error": (C.c_char_p, []),
    "zenv_version": (C.c_char_p, []),
    "zenv_device_count": (C.c_int, []),
    "zenv_build_flags": (C.c_char_p, []),
    "zenv_rollout_chunk": (C.c_int, []),
    "zenv_config_for_id": (C.c_int, [C.c_char_p, C.POINTER(Config)]),
    "zenv_default_config": (C.c_int, [C.c_int, C.c_int, C.POINTER(Config)]),
    "zenv_zone_feat": (C.c_int, [C.POINTER(Config)]),
    "zenv_config_size": (C.c_int, []),
    "zenv_sample_layout": (C.c_int, [C.POINTER(Config), C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_int32)]),
    "zenv_fixed_seed_sequence": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "zenv_create": (C.c_int, [C.POINTER(Config), C.c_int, C.c_int, C.POINTER(_H)]),
    "zenv_destroy": (C.c_int, [_H]),
    "zenv_num_envs": (C.c_int, [_H]),
    "zenv_get_config": (C.c_int, [_H, C.POINTER(Config)]),
    "zenv_bank_build": (C.c_int, [_H, C.c_int64, C.c_int, C.c_int]),
    "zenv_bank_build_seeds": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int]),
    "zenv_bank_set": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "zenv_bank_size": (C.c_int, [_H]),
    "zenv_bank_update": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "zenv_schedule_ring": (C.c_int, [_H, C.c_void_p, C.c_int32]),
    "zenv_schedule_sequential": (C.c_int, [_H, C.c_void_p, C.c_int32]),
    "zenv_schedule_fixed_seeds": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_int64]),
    "zenv_reset": (C.c_int, [_H, C.c_void_p]),
    "zenv_step": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int]),
    "zenv_step_many": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "zenv_policy": (C.c_int, [_H, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]),
    "zenv_rollout": (C.c_int, [_H, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int,
                               C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "zenv_set_rollout_slice": (C.c_int, [_H, C.c_int]),
    "zenv_collect": (C.c_int, [_H, C.c_int, C.c_uint64, C.c_uint64, C.c_float, C.c_float]),
    "zenv_order_enable": (C.c_int, [_H]),
    "zenv_order_configure": (C.c_int, [_H, C.c_int]),
    "zenv_route_ranks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "zenv_goal_enable": (C.c_int, [_H]),
    "zenv_set_goals": (C.c_int, [_H, C.c_void_p]),
    "zenv_solver_goals": (C.c_int, [_H, C.c_void_p]),
    "zenv_mlp_load": (C.c_int, [_H, C.c_void_p]),
    "zenv_mlp_forward": (C.c_int, [_H]),
    "zenv_get": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int]),
    "zenv_get_rows": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "zenv_device_ptr": (C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p)]),
    "zenv_field_bytes": (C.c_int64, [_H, C.c_int]),
    "zenv_sync": (C.c_int, [_H]),
    "zenv_query": (C.c_int, [_H]),
    "zenv_host_alloc": (C.c_void_p, [C.c_int64]),
    "zenv_host_free": (C.c_int, [C.c_void_p]),
    "zenv_get_many": (C.c_int, [_H, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "zenv_results_layout": (C.c_int64, [_H, C.POINTER(C.c_int64)]),
    "zenv_step_results": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p]),
    "zenv_host_io": (C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "zenv_step_host": (C.c_int, [_H, C.c_int]),
    "zenv_set_stream": (C.c_int, [_H, C.c_void_p]),
    "zenv_comm_unique_id": (C.c_int, [C.c_void_p]),
    "zenv_comm_init": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "zenv_comm_destroy": (C.c_int, [_H]),
    "zenv_comm_info": (C.c_int, [_H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "zenv_allgather": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int]),
    "zenv_comm_barrier": (C.c_int, [_H]),
    "zenv_comm_allreduce_max": (C.c_int, [_H, C.POINTER(C.c_double)]),
    "zenv_probe_store_stream": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_float)]),
    "zenv_step_count": (C.c_int64, [_H]),
    "zenv_state_bytes": (C.c_int64, [_H]),
    "zenv_get_state": (C.c_int, [_H, C.c_void_p, C.c_int64]),
    "zenv_set_state": (C.c_int, [_H, C.c_void_p, C.c_int64]),
    "zenv_debug_state": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def lib():
    """Load libzenv_hip.so once.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # If torch is (or will be) in this process its bundled libamdhip64 must be the one HIP
    # runtime: make sure it is loaded first so our DT_NEEDED resolves to the same copy.
    if "torch" in sys.modules:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
    for name, (restype, argtypes) in _PROTOTYPES.items():
        fn = getattr(L, name)   # AttributeError if the ABI is incomplete
        fn.restype = restype
        fn.argtypes = argtypes
    if L.zenv_config_size() != C.sizeof(Config):
        raise ImportError("zenv_config layout mismatch between include/zenv.h and _native.Config")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().zenv_last_error()
        raise ZenvError(rc, msg.decode() if msg else "")
    return rc


def exported_symbols():
    return sorted(_PROTOTYPES)
