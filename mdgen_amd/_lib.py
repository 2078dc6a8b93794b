"""ctypes binding of libmdgen_amd.so (include/mdgen_amd.h).  There is NO fallback: if the HIP
library is missing or fails to load, importing this module raises."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- must be imported first: provides the process-wide libamdhip64.so.7

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDGEN_AMD_LIB", os.path.join(_HERE, "libmdgen_amd.so"))   # override: debugging builds only

ABI_VERSION = 7   # include/mdgen_amd.h MDGEN_ABI_VERSION

EXPORTS = [
    "mdgen_last_error", "mdgen_abi_version", "mdgen_dev_build", "mdgen_ctx_create", "mdgen_ctx_destroy", "mdgen_ctx_set_weight",
    "mdgen_ctx_finalize", "mdgen_ctx_set_option", "mdgen_debug_view_plan", "mdgen_ctx_num_weights", "mdgen_ctx_weight_name", "mdgen_workspace_layout",
    "mdgen_denoiser_forward", "mdgen_sample_euler", "mdgen_rollout_euler", "mdgen_profile_enable", "mdgen_profile_report", "mdgen_profile_phase_trace", "mdgen_debug_dispatch_plan", "mdgen_debug_layout_maps", "mdgen_debug_mlp_stream_table", "mdgen_debug_train_linear", "mdgen_debug_train_dw", "mdgen_debug_train_attention",
    "mdgen_rigid_compose", "mdgen_rigid_invert",
    "mdgen_rigid_apply", "mdgen_quat_to_rot", "mdgen_rot_to_quat", "mdgen_prep_latents",
    "mdgen_samples_to_atom14", "mdgen_atom14_to_cond", "mdgen_path_plan", "mdgen_masked_mse", "mdgen_from_3_points",
    "mdgen_grad_sumsq", "mdgen_adam_step", "mdgen_ema_update", "mdgen_train_workspace_bytes",
    "mdgen_train_forward_backward",
    "mdgen_train_num_milestones", "mdgen_train_bind_params",
    "mdgen_train_set_milestone_events",
]


class ModelDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "embed_dim", "mha_heads", "num_layers", "latent_dim", "ipa_heads", "ipa_head_dim", "ipa_qk", "ipa_v",
        "abs_pos_emb", "crop", "tps_condition")] + [("time_multiplier", C.c_float)]


class Shape(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("L", C.c_int32)]


class WsLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "total_bytes", "h", "qf", "kf", "vf", "obuf", "mod", "silu_t", "ipa_out", "h_ipa", "ipa_proj",
        "ipa_feat", "mask_bl", "rel7", "tgrid", "f32_scratch", "split", "fold", "embase")]


class ResidueTables(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "default_frames", "lit_positions", "atom14_group", "atom14_mask", "atom37_to_atom14", "atom37_mask",
        "chi_atom_indices", "chi_angles_mask")]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m mdgen_amd.build` (hipcc, gfx950). "
            "mdgen_amd has no CPU/PyTorch fallback path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    lib.mdgen_last_error.restype = C.c_char_p
    lib.mdgen_abi_version.restype = i32
    lib.mdgen_ctx_create.argtypes = [C.POINTER(vp), C.POINTER(ModelDesc)]
    lib.mdgen_ctx_destroy.argtypes = [vp]
    lib.mdgen_ctx_set_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), i32, vp]
    lib.mdgen_ctx_finalize.argtypes = [vp, vp]
    lib.mdgen_ctx_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.mdgen_debug_view_plan.argtypes = [C.POINTER(Shape), i32, C.POINTER(i32), C.POINTER(i32)]
    lib.mdgen_ctx_num_weights.argtypes = [vp]
    lib.mdgen_ctx_weight_name.argtypes = [vp, i32]
    lib.mdgen_ctx_weight_name.restype = C.c_char_p
    lib.mdgen_workspace_layout.argtypes = [vp, C.POINTER(Shape), i32, i32, C.POINTER(WsLayout)]
    lib.mdgen_denoiser_forward.argtypes = [vp, C.POINTER(Shape)] + [vp] * 15 + [sz, vp]
    lib.mdgen_sample_euler.argtypes = [vp, C.POINTER(Shape), i32] + [vp] * 11 + [sz, i32, vp]
    lib.mdgen_rollout_euler.argtypes = [vp, C.POINTER(Shape), i32, i32] + [vp] * 8 + [C.POINTER(ResidueTables), vp, vp, sz, i32, vp]
    lib.mdgen_profile_enable.argtypes = [vp, i32]
    lib.mdgen_profile_phase_trace.argtypes = [vp, vp, i64]
    lib.mdgen_profile_report.argtypes = [vp, vp, C.c_char_p, sz]
    lib.mdgen_debug_dispatch_plan.argtypes = [C.POINTER(Shape), i32, i32, i32, i32, i32, i32, C.c_char_p, C.c_char_p, sz]
    lib.mdgen_debug_layout_maps.argtypes = [vp] * 5
    lib.mdgen_debug_mlp_stream_table.argtypes = [vp, i32]
    lib.mdgen_debug_train_linear.argtypes = [i32, vp, i32, vp, i32, vp, i64, i32, i32, vp, i32, vp, vp]
    lib.mdgen_debug_train_dw.argtypes = [i32, vp, i32, vp, i32, i64, i32, i32, vp, vp, vp, i64, vp]
    lib.mdgen_debug_train_attention.argtypes = [i32, vp, i64, i32, i32, i32, i32, i32, i32] + [vp] * 11
    lib.mdgen_rigid_compose.argtypes = [i64] + [vp] * 7
    lib.mdgen_rigid_invert.argtypes = [i64] + [vp] * 5
    lib.mdgen_rigid_apply.argtypes = [i64, i64, vp, vp, vp, vp, i32, vp]
    lib.mdgen_quat_to_rot.argtypes = [i64, vp, i32, vp, vp]
    lib.mdgen_rot_to_quat.argtypes = [i64, vp, vp, vp]
    lib.mdgen_prep_latents.argtypes = [C.POINTER(Shape), i32, i32] + [vp] * 7
    lib.mdgen_samples_to_atom14.argtypes = [C.POINTER(Shape), i32, i32] + [vp] * 10
    lib.mdgen_atom14_to_cond.argtypes = [i32, i32] + [vp] * 11
    lib.mdgen_from_3_points.argtypes = [i64] + [vp] * 6
    lib.mdgen_path_plan.argtypes = [i64, i64, i32] + [vp] * 6
    lib.mdgen_masked_mse.argtypes = [i64, i64] + [vp] * 5
    f32 = C.c_float
    lib.mdgen_grad_sumsq.argtypes = [i64, vp, f32, vp, i32, vp, vp]
    lib.mdgen_adam_step.argtypes = [i64, vp, vp, vp, vp, i32, f32, f32, f32, f32, f32, i32, f32, vp, f32, vp]
    lib.mdgen_ema_update.argtypes = [i64, vp, vp, f32, vp]
    lib.mdgen_train_workspace_bytes.argtypes = [vp, C.POINTER(Shape), C.POINTER(sz)]
    lib.mdgen_train_num_milestones.argtypes = [vp]
    lib.mdgen_train_bind_params.argtypes = [vp, vp, C.POINTER(i64)]
    lib.mdgen_train_set_milestone_events.argtypes = [vp, C.POINTER(vp), i32]
    lib.mdgen_train_forward_backward.argtypes = [vp, C.POINTER(Shape)] + [vp] * 17 + [vp, sz, vp, sz, vp]
    for n in EXPORTS:
        getattr(lib, n)
        if n not in ("mdgen_last_error", "mdgen_ctx_weight_name"):
            getattr(lib, n).restype = i32
    if lib.mdgen_dev_build() and "MDGEN_AMD_LIB" not in os.environ:
        raise ImportError(f"{LIB_PATH} is an experiment build (csrc/dev.h switches): rebuild with `python -m mdgen_amd.build --force`")
    got = lib.mdgen_abi_version()
    if got != ABI_VERSION:   # a stale .so would take struct writes / argument lists of another layout
        raise ImportError(f"{LIB_PATH} has ABI version {got}, this package expects {ABI_VERSION}: "
                          "rebuild it with `python -m mdgen_amd.build --force`")
    return lib


lib = _load()


class MdgenError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        raise MdgenError(f"libmdgen_amd error {rc}: {lib.mdgen_last_error().decode()}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MdgenError("mdgen_amd runs on the GPU only (no CPU fallback): got a CPU tensor")


def ptr(t, dtype=None):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        raise MdgenError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise MdgenError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """hipStream_t of torch's current stream on `device` (default: the current device)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device_of(t):
    """Context manager making `t`'s device current: kernels are enqueued on THAT device's current stream
    (a launch on device 0's stream with device-1 pointers faults or loses its stream dependency)."""
    return torch.cuda.device(t.device)


def launch(fn, like, *args):
    """Call a stream-taking entry point `fn(*args, stream)` on `like`'s device: that device is made current for
    the call and the launch goes on ITS current stream (centralised device guard)."""
    with torch.cuda.device(like.device):
        check(fn(*args, stream_ptr()))


def dispatch_plan(B, T, L_, n_steps=1, mode=0, tps=False, num_layers=5, ncu=256, xcd_round_robin=True, options=None):
    """`mdgen_debug_dispatch_plan` (host only): {"streams", "prepare": {kernel class: launches}, "views": [{"B", "classes"}]} of a call of this shape.
    mode 0: sample_euler (product path), 1: forward, 2: sample_euler under the profiler (one stream)."""
    import json
    buf = C.create_string_buffer(1 << 14)
    sh = Shape(B, T, L_)
    opts = ",".join(f"{k}={int(v)}" for k, v in (options or {}).items()).encode()
    check(lib.mdgen_debug_dispatch_plan(C.byref(sh), n_steps, mode, int(tps), num_layers, ncu, int(xcd_round_robin), opts, buf, len(buf)))
    return json.loads(buf.value.decode())
