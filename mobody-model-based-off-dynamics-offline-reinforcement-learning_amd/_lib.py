"""ctypes binding of libmobody_hip.so -- the only way product code reaches the GPU kernels.

Mirrors include/mobody_hip.h one to one.  Loading fails loudly (ImportError) when the
library has not been built: there is deliberately no CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmobody_hip.so")

i32, i64, u32, f32 = C.c_int32, C.c_int64, C.c_uint32, C.c_float
vp = C.c_void_p


class MobodyLayer(C.Structure):
    _fields_ = [("in_dim", i32), ("out_dim", i32), ("Kp", i32), ("Np", i32), ("w_off", i64), ("b_off", i64)]


DL_NAMES = ["zs1", "zs2", "zs3", "za_src1", "za_src2", "za_trg1", "za_trg2", "transition1", "transition2",
            "transition3", "reward_model1", "reward_model2", "reward_model3"]


class MobodyDynLayout(C.Structure):
    _fields_ = [("S", i32), ("A", i32), ("E", i32), ("_pad", i32), ("layer", MobodyLayer * len(DL_NAMES)),
                ("total_floats", i64)]


class MobodyMlpLayout(C.Structure):
    _fields_ = [("in_dim", i32), ("out_dim", i32), ("members", i32), ("Kp1", i32), ("Np3", i32), ("Np1t", i32),
                ("w1", i64), ("b1", i64), ("w2", i64), ("b2", i64), ("w3", i64), ("b3", i64),
                ("member_floats", i64), ("total_floats", i64), ("w3t", i64), ("w2t", i64), ("w1t", i64),
                ("t_member_floats", i64), ("t_total_floats", i64), ("w2p", i64), ("w2tp", i64)]


class MobodyPretrainLayout(C.Structure):
    _fields_ = [("S", i32), ("A", i32), ("za_in", i32), ("_pad", i32), ("enc", MobodyMlpLayout), ("tr", MobodyMlpLayout),
                ("rw", MobodyMlpLayout), ("off_enc", i64), ("off_tr", i64), ("off_rw", i64), ("off_za_src", i64),
                ("off_za_trg", i64), ("za_w1", i64), ("za_b1", i64), ("za_w2", i64), ("za_b2", i64),
                ("za_member_floats", i64), ("total_floats", i64), ("t_off_enc", i64), ("t_off_tr", i64),
                ("t_off_rw", i64), ("t_total_floats", i64)]


class MobodyBufferView(C.Structure):
    _fields_ = [("state", vp), ("action", vp), ("next_state", vp), ("reward", vp), ("not_done", vp), ("pitch", i64)]


class MobodyTrainDims(C.Structure):
    _fields_ = [("S", i32), ("A", i32), ("N", i64), ("Nt", i64), ("N_global", i64), ("Nt_global", i64)]


class MobodyHyper(C.Structure):
    _fields_ = [("gamma", f32), ("tau", f32), ("max_action", f32), ("weight", f32), ("bc_coef", f32),
                ("q_weighted", i32), ("scale_q", i32), ("precision", i32)]


PRECISIONS = {"f32": 0, "bf16": 1, "bf16x2": 2, "bf16x3": 3, "f16x2": 4}


TERM_IDS = {"never": 0, "halfcheetah": 1, "hopper": 2, "ant": 3, "walker2d": 4, "humanoid": 5, "pen": 6}

# name -> (restype, argtypes); must list every symbol include/mobody_hip.h declares
PROTOTYPES = {
    "mobody_last_error": (C.c_char_p, []),
    "mobody_abi_version": (C.c_int, []),
    "mobody_dyn_layout": (C.c_int, [C.c_int, C.c_int, C.POINTER(MobodyDynLayout)]),
    "mobody_mlp_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(MobodyMlpLayout)]),
    "mobody_prof_begin": (C.c_int, [C.c_int]),
    "mobody_prof_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(i64), C.c_int]),
    "mobody_rng_normal": (C.c_int, [u32, u32, u32, i64, vp, vp]),
    "mobody_rng_index": (C.c_int, [u32, u32, u32, i64, u32, vp, vp]),
    "mobody_dyn_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, i64, C.c_int, vp, vp]),
    "mobody_dyn_planes_floats": (i64, []),
    "mobody_dyn_planes": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, vp]),
    "mobody_dyn_step_workspace": (i64, [C.c_int, C.c_int, i64]),
    "mobody_dyn_step": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, i64, vp, vp, vp, C.POINTER(i32), C.c_int,
                                  u32, u32, vp, f32, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mobody_mopo_step": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, i64, vp, vp, vp, C.POINTER(i32), C.c_int,
                                   u32, u32, f32, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mobody_rollout_workspace": (i64, [C.c_int, C.c_int, i64]),
    "mobody_rollout": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp, i64, C.c_int, C.POINTER(i32), C.c_int, u32, u32, f32,
                                 C.c_int, C.c_int, f32, C.c_int, C.POINTER(MobodyBufferView), i64, vp, vp, vp]),
    "mobody_termination": (C.c_int, [C.c_int, vp, i64, C.c_int, vp, vp]),
    "mobody_rollout_mask": (C.c_int, [vp, vp, vp, f32, C.c_int, i64, vp, vp, vp]),
    "mobody_sample_indices": (C.c_int, [u32, u32, vp, i64, i64, vp, vp, vp]),
    "mobody_counter_add": (C.c_int, [vp, C.c_int, i64, vp]),
    "mobody_mlp3_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, i64, C.c_int, f32, vp,
                                      vp, vp, vp, vp]),
    "mobody_gather_batch": (C.c_int, [C.POINTER(MobodyBufferView), C.POINTER(vp), C.POINTER(i64), C.c_int, C.c_int,
                                      C.c_int, vp, vp, vp, vp, vp, vp]),
    "mobody_gather_batch_rng": (C.c_int, [C.POINTER(MobodyBufferView), C.POINTER(i64), C.c_int, C.c_int, C.c_int,
                                          C.POINTER(u32), C.POINTER(i64), vp, C.POINTER(vp), vp, vp, vp, vp, vp, C.POINTER(vp), C.c_int,
                                          vp]),
    "mobody_ring_append": (C.c_int, [C.POINTER(MobodyBufferView), i64, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, i64, vp,
                                     vp]),
    "mobody_ring_pitch": (i64, [C.c_int, C.c_int]),
    "mobody_train_workspace": (i64, [C.POINTER(MobodyTrainDims)]),
    "mobody_critic_step": (C.c_int, [C.POINTER(MobodyTrainDims), C.POINTER(MobodyHyper), vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                     vp, vp, vp, vp, vp, vp, C.c_int, vp]),
    "mobody_critic_update": (C.c_int, [C.POINTER(MobodyTrainDims), C.POINTER(MobodyHyper), vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                       vp, vp, vp, vp, vp, i64, vp, f32, vp, vp, C.c_int, vp, vp]),
    "mobody_critic_update_phase": (C.c_int, [C.POINTER(MobodyTrainDims), C.POINTER(MobodyHyper), vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                             vp, vp, vp, vp, vp, i64, vp, f32, vp, vp, C.c_int, vp, C.c_int, vp]),
    "mobody_actor_update": (C.c_int, [C.POINTER(MobodyTrainDims), C.POINTER(MobodyHyper), vp, vp, vp, vp, vp, vp, vp,
                                      vp, vp, vp, i64, vp, f32, vp, vp, vp]),
    "mobody_value_loss_grad": (C.c_int, [vp, vp, i64, i64, vp, vp, vp, vp]),
    "mobody_actor_forward": (C.c_int, [C.POINTER(MobodyTrainDims), C.POINTER(MobodyHyper), vp, vp, vp, vp, vp, vp, vp, vp,
                                       C.c_int, vp]),
    "mobody_actor_backward": (C.c_int, [C.POINTER(MobodyTrainDims), C.POINTER(MobodyHyper), vp, vp, vp, vp, vp, vp,
                                        vp, vp, vp, vp, vp, vp]),
    "mobody_adam_polyak": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, C.c_int, vp]),
    "mobody_adam_polyak_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, f32, C.c_int, vp]),
    "mobody_par_penalty": (C.c_int, [vp, vp, vp, f32, i64, C.c_int, vp]),
    "mobody_mlp3_backward_workspace": (i64, [C.c_int, C.c_int, C.c_int, i64]),
    "mobody_mlp3_backward": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, i64, vp, vp, vp]),
    "mobody_dara_inputs": (C.c_int, [vp, vp, vp, i64, C.c_int, C.c_int, f32, vp, vp, u32, u32, vp, vp, vp]),
    "mobody_dara_loss_grad": (C.c_int, [vp, vp, vp, i64, i64, vp, vp, vp, vp, vp]),
    "mobody_dara_penalty": (C.c_int, [vp, vp, i64, f32, vp, vp, vp]),
    "mobody_mlp_transpose": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "mobody_pretrain_layout": (C.c_int, [C.c_int, C.c_int, C.POINTER(MobodyPretrainLayout)]),
    "mobody_pretrain_transpose": (C.c_int, [C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "mobody_pretrain_workspace": (i64, [C.c_int, C.c_int, i64]),
    "mobody_pretrain_gather": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, vp, i64, C.c_int, C.c_int, vp, vp, vp, vp]),
    "mobody_pretrain_update": (C.c_int, [C.c_int, C.c_int, i64, C.c_int, f32, vp, vp, vp, vp, vp, vp, vp, u32, u32, vp, vp,
                                         vp, i64, i64, vp, f32, vp, vp, vp, C.c_int, vp]),
    "mobody_pretrain_grads": (C.c_int, [C.c_int, C.c_int, i64, i64, C.c_int, f32, vp, vp, vp, vp, vp, vp, vp, u32, u32,
                                        vp, vp, vp, C.c_int, f32, f32, vp]),
    "mobody_pretrain_adam": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, i64, i64, f32, f32, C.c_int, C.c_int, i64, vp]),
    "mobody_pretrain_za_adam": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, i64, f32, f32, vp]),
    "mobody_dyn_validate_workspace": (i64, [C.c_int, C.c_int, i64]),
    "mobody_dyn_validate": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, i64, C.c_int, vp, vp, vp]),
}

_lib = None


def load():
    """Load the shared library once; raise ImportError (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own HIP runtime; it has to be in the process BEFORE this library is mapped so that both bind
    # to the same runtime instance (the reverse order left the kernels with "no ROCm-capable device is detected").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python {os.path.join(_HERE, 'csrc', 'build.py')}` "
            "(or __graft_entry__.build()). There is no CPU fallback for the MOBODY hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.mobody_abi_version() != 6:
        raise ImportError("libmobody_hip.so ABI version mismatch")
    _lib = lib
    return lib


class MobodyError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = load().mobody_last_error().decode(errors="replace")
        raise MobodyError(f"{what} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C ABI needs contiguous tensors"
    return t.data_ptr()


_dev_index = None


def cur_stream():
    """Raw hipStream_t of torch's current stream (also the capture stream inside torch.cuda.graph).  Uses the
    raw-handle accessor: torch.cuda.current_stream() builds a Stream object and costs ~8 us per call, which at ~8
    calls per train() step was a fifth of the eager host overhead."""
    global _dev_index
    import torch
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_dev_index)


def dyn_layout(S, A):
    L = MobodyDynLayout()
    check(load().mobody_dyn_layout(S, A, C.byref(L)), "mobody_dyn_layout")
    return L


def mlp_layout(in_dim, out_dim, members):
    L = MobodyMlpLayout()
    check(load().mobody_mlp_layout(in_dim, out_dim, members, C.byref(L)), "mobody_mlp_layout")
    return L


def pretrain_layout(S, A):
    L = MobodyPretrainLayout()
    check(load().mobody_pretrain_layout(S, A, C.byref(L)), "mobody_pretrain_layout")
    return L
