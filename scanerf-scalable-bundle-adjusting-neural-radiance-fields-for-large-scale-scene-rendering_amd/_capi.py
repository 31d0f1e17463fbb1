"""ctypes view of libscanerf_hip.so (the C ABI declared in include/scanerf_hip.h).

There is no CPU fallback: if the HIP library is missing, or a tensor is not a
contiguous tensor of the expected dtype on a HIP device, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCANERF_LIB") or os.path.join(_HERE, "lib", "libscanerf_hip.so")

F32, F16, BF16 = 0, 1, 2
RAY_OUT = 16
PARAMSIZE = 13994

_lib = None

# every symbol include/scanerf_hip.h declares
SYMBOLS = [
    "scanerf_last_error", "scanerf_abi_version", "scanerf_compute_ray_forward", "scanerf_compute_ray_backward",
    "scanerf_ray_aabb_intersection", "scanerf_sample_points_grid", "scanerf_sample_insideout_block",
    "scanerf_background_sampling", "scanerf_adam_step", "scanerf_adam_step_fp16", "scanerf_embedding_bg_forward",
    "scanerf_embedding_bg_backward", "scanerf_embedding_forward", "scanerf_embedding_backward",
    "scanerf_render_workspace_floats", "scanerf_pack_decoder", "scanerf_render_forward_packed",
    "scanerf_embedding_bg_forward_ex", "scanerf_embedding_bwd_workspace_bytes",
    "scanerf_embedding_bg_backward_binned", "scanerf_embedding_bg_backward_binned_adam", "scanerf_render_backward_grid", "scanerf_render_backward",
    "scanerf_sort_tracing_blocks", "scanerf_render_forward_packed_plan", "scanerf_render_forward_plan_supported", "scanerf_ray_grad_epilogue", "scanerf_photometric_loss_scratch_floats", "scanerf_photometric_loss_grad", "scanerf_render_scatter_workspace_bytes", "scanerf_render_scatter_plan", "scanerf_render_scatter_accumulate", "scanerf_render_scatter_accumulate_adam", "scanerf_render_scatter_accumulate_adam2", "scanerf_photometric_loss_grad_fgbg",
    "scanerf_ray_block_intersection", "scanerf_render_sample_points", "scanerf_prepare_points", "scanerf_pts_inference",
    "scanerf_accumulate_color", "scanerf_render_inverse_z_sampling", "scanerf_bg_pts_inference_v2",
    "scanerf_update_outgoing_bidx", "scanerf_update_outgoing_bidx_v2", "scanerf_get_last_block",
    "scanerf_ray_firsthit_block", "scanerf_process_occupied_grid", "scanerf_embedding_bg_point_grad", "scanerf_voxelize_mesh", "scanerf_ray_valid", "scanerf_compact_rays",
    "scanerf_decoder_forward", "scanerf_decoder_backward_grid", "scanerf_decoder_backward", "scanerf_pts_inference_tracing",
    "scanerf_table_grad_scatter_adam_rays", "scanerf_experiments_enabled", "scanerf_composite_forward", "scanerf_composite_backward",
]

# test / measurement entry points (include/scanerf_hip.h declares them, the default build exports them): NOT required --
# a lean product build may drop csrc/h3_selftest.hip and the probes; their users (tests, bench.py's live ceiling) check first
OPTIONAL_SYMBOLS = ["scanerf_h3_selftest", "scanerf_rec8_selftest", "scanerf_icache_sweep", "scanerf_gather_rate_probe"]


def has_symbol(name):
    return hasattr(lib(), name)


class RenderCfg(ctypes.Structure):
    _fields_ = [("contract_mode", ctypes.c_int), ("infinity", ctypes.c_int),
                ("min_bbox", ctypes.c_float * 3), ("bbox_size", ctypes.c_float * 3), ("arith", ctypes.c_int),
                ("skip_levels", ctypes.c_uint)]


ARITH_F32, ARITH_H3, ARITH_T16, ARITH_T16S = 0, 1, 2, 3
T16_FAMILY = (ARITH_T16, ARITH_T16S)   # backward on 16-sample tiles: x-stash, plan counted by the forward, in-kernel pose path


def lib():
    """Load the HIP library or fail loudly (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"scanerf: HIP library not built: {LIB_PATH} is missing. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, gfx950).")
        l = ctypes.CDLL(LIB_PATH)
        l.scanerf_last_error.restype = ctypes.c_char_p
        l.scanerf_embedding_bwd_workspace_bytes.restype = ctypes.c_size_t
        l.scanerf_render_scatter_workspace_bytes.restype = ctypes.c_size_t
        for name in SYMBOLS:
            if not hasattr(l, name):
                raise RuntimeError(f"scanerf: {LIB_PATH} does not export {name}")
        _lib = l
        st = audit_state()
        if st["status"] != "passed":
            import warnings
            warnings.warn(f"scanerf: {LIB_PATH} is NOT the audited build (ISA audit: {st['status']}; {st.get('why', '')}).  Its kernels' "
                          "listings were not checked against the validated ones (csrc/isa_manifest.json, DESIGN.md 4.10: no "
                          "packed-f32 instructions, pinned listings): launch-to-launch reproducibility is not established for it.",
                          RuntimeWarning, stacklevel=2)
    return _lib


def audit_state():
    """What tools/isa_audit.py (run by `make`) said about the library file this process loads: {"status": "passed" | "unvalidated"
    (another compiler build than the validated one; no packed-f32 instruction found) | "skipped" (built with
    SCANERF_SKIP_ISA_AUDIT=1) | "failed" | "stale" (the file changed after the audit) | "missing"}."""
    import hashlib
    import json
    side = os.path.join(os.path.dirname(LIB_PATH), "isa_audit.json")
    if os.environ.get("SCANERF_LIB"):
        return {"status": "variant", "why": "SCANERF_LIB selects an investigation build (tools/build_variant.py)"}
    try:
        st = json.load(open(side))
    except (OSError, ValueError):
        return {"status": "missing", "why": f"{side} not found: the library was not built by csrc/Makefile's `all`"}
    sha = hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()
    if st.get("library_sha256") != sha:
        return {"status": "stale", "why": "the library file is not the one the audit saw"}
    return {"status": st.get("status", "missing"), "why": "; ".join(st.get("detail") or [])[:400], "compiler": st.get("compiler")}


def audit_required():
    """True where an "unvalidated" library is NOT acceptable: SCANERF_REQUIRE_AUDITED=1 (the project's own CI / bench box), or the
    installed hipcc IS the compiler build csrc/isa_manifest.json was validated with -- then nothing excuses a library that was
    not built and audited by it.  Elsewhere (a downstream user's other ROCm) "unvalidated" stays a loud warning."""
    import json
    import subprocess
    if os.environ.get("SCANERF_REQUIRE_AUDITED") == "1":
        return True
    try:
        want = json.load(open(os.path.join(_HERE, "csrc", "isa_manifest.json"))).get("compiler", "")
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
        have = " | ".join(ln.strip() for ln in out.splitlines() if ln.startswith(("HIP version", "AMD clang version")))
        return bool(want) and want == have
    except (OSError, ValueError, subprocess.SubprocessError):
        return False


SWEEP_ICACHE = False   # tests only: evict the instruction caches after every library call (scanerf_icache_sweep)


def check(status, what):
    if status != 0:
        msg = lib().scanerf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"scanerf {what} failed ({status}): {msg}")
    if SWEEP_ICACHE:
        if not has_symbol("scanerf_icache_sweep"):
            raise RuntimeError("scanerf: SWEEP_ICACHE needs scanerf_icache_sweep (a test entry point this build does not export)")
        if lib().scanerf_icache_sweep(stream()) != 0:
            raise RuntimeError("scanerf icache_sweep failed: " + lib().scanerf_last_error().decode("utf-8", "replace"))


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_DT = {torch.float32: "f32", torch.int32: "i32", torch.bool: "bool", torch.uint8: "u8", torch.float16: "f16",
       torch.bfloat16: "bf16", torch.int64: "i64", torch.int16: "i16"}


def dev_ptr(t, dtypes, name, allow_none=False):
    """Device pointer of a contiguous GPU tensor; validates what the reference leaves unchecked
    (the reference casts data_ptr blindly and calls .contiguous() even on outputs, so a
    non-contiguous output silently receives nothing)."""
    if t is None:
        if allow_none:
            return ctypes.c_void_p(0)
        raise RuntimeError(f"scanerf: {name} is None")
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"scanerf: {name} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"scanerf: {name} must live on the GPU (no CPU path exists); got device {t.device}")
    if not isinstance(dtypes, (tuple, list)):
        dtypes = (dtypes,)
    if t.dtype not in dtypes:
        raise RuntimeError(f"scanerf: {name} has dtype {t.dtype}, expected {[_DT.get(d, d) for d in dtypes]}")
    if not t.is_contiguous():
        raise RuntimeError(f"scanerf: {name} must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def feat_dtype_code(t):
    return {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}[t.dtype]


_workspaces = {}


def workspace(device, nbytes, tag="plan"):
    """Grow-only scratch buffer per (device, stream, tag) (torch's caching allocator owns the memory; the C ABI never
    allocates).  Keyed by the CURRENT stream: work queued on one stream never shares scratch with another stream's, and a
    buffer that is replaced by a larger one is released to the allocator on the stream all of its users were queued on,
    so it cannot be handed out again while they are still in flight.  tag: a fused plan lives in its buffer from
    scatter_plan / render_forward(plan=True) until the accumulate; the stand-alone scatter (one call, "scatter") has its own
    buffer so that it cannot land between the two."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if str(device).startswith("cuda") else 0, tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _workspaces.pop(key, None)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
