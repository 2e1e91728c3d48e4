"""ctypes binding of libmi355q.so (include/mi355q.h).

The shared object is built in-tree by `__graft_entry__.build()` into
ai-edge-quantizer_amd/lib/. There is deliberately NO CPU fallback: if the
library or a GPU is missing, calls raise.
"""
from __future__ import annotations

import ctypes
import os

# torch first: its bundled libamdhip64.so (SONAME libamdhip64.so.7) must be the
# HIP runtime in the process so that torch device pointers / streams and our
# kernels share one runtime.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libmi355q.so")

c_i32, c_i64, c_f32, c_f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double
c_ptr, c_size = ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/mi355q.h one to one.
PROTOTYPES = {
    "mi355q_version": (c_i32, []),
    "mi355q_last_error": (ctypes.c_char_p, []),
    "mi355q_device_info": (c_i32, [c_ptr, c_ptr, c_ptr, c_i32]),
    "mi355q_minmax_workspace_bytes": (c_size, [c_i64, c_i64, c_i64]),
    "mi355q_minmax_f32": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_requant_sym_f32": (c_i32, [c_ptr, c_i64, c_i64, c_i32, c_i32, c_ptr, c_ptr, c_ptr,
                                       c_ptr, c_ptr, c_ptr]),
    "mi355q_requant_sym_f32_batched": (c_i32, [c_ptr, c_i32, c_i64, c_i64, c_i32, c_i32, c_ptr,
                                               c_ptr, c_ptr, c_ptr, c_ptr]),
    "mi355q_requant_sym_f32_batched_hostptrs": (c_i32, [c_ptr, c_i32, c_i64, c_i64, c_i32, c_i32, c_ptr,
                                                        c_ptr, c_ptr, c_ptr, c_ptr]),
    "mi355q_quantize_f32": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_ptr, c_i32, c_ptr, c_i32, c_i32,
                                    c_i32, c_i32, c_ptr, c_ptr]),
    "mi355q_dequantize_f32": (c_i32, [c_ptr, c_i32, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_i32,
                                      c_i32, c_ptr, c_ptr]),
    "mi355q_pack_bits": (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr]),
    "mi355q_unpack_bits": (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr]),
    "mi355q_act_minmax_workspace_bytes": (c_size, [c_i32]),
    "mi355q_act_minmax_f32": (c_i32, [c_ptr, c_ptr, c_i32, c_f32, c_f32, c_i32, c_ptr, c_ptr,
                                      c_size, c_ptr]),
    "mi355q_octav_workspace_bytes": (c_size, [c_i64, c_i32]),
    "mi355q_octav_rows_workspace_bytes": (c_size, [c_i64, c_i64, c_i32]),
    "mi355q_octav_clip_f32": (c_i32, [c_ptr, c_i64, c_i64, c_i32, c_i32, c_f32, c_i32, c_i32,
                                      c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_octav_clip_fast_f32": (c_i32, [c_ptr, c_i64, c_i64, c_i32, c_i32, c_f32, c_i32, c_ptr, c_ptr, c_ptr,
                                           c_size, c_ptr]),
    "mi355q_octav_clip_nd_f32": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_i32, c_i32, c_f32, c_i32,
                                         c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_mse_scale_f32": (c_i32, [c_ptr, c_i64, c_i64, c_f32, c_ptr, c_ptr]),
    "mi355q_mse_scale_nd_f32": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_f32, c_ptr, c_ptr]),
    "mi355q_mse_requant_f32": (c_i32, [c_ptr, c_i64, c_i64, c_f32, c_i32, c_i32, c_ptr, c_ptr, c_ptr]),
    "mi355q_hadamard_rotate_f32": (c_i32, [c_ptr, c_i64, c_i32, c_ptr, c_ptr]),
    "mi355q_gemm_f32": (c_i32, [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64,
                                c_i64, c_i64, c_i64, c_f32, c_f32, c_i32, c_ptr]),
    "mi355q_gemm_f64": (c_i32, [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64,
                                c_i64, c_i64, c_i64, c_f64, c_f64, c_i32, c_ptr]),
    "mi355q_gptq_xtx_workspace_bytes": (c_size, [c_i64, c_i64]),
    "mi355q_gptq_xtx_f32": (c_i32, [c_ptr, c_i64, c_i64, c_f64, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_gptq_xtx_accum_workspace_bytes": (c_size, [c_i64, c_i64]),
    "mi355q_gptq_xtx_accum_f32": (c_i32, [c_ptr, c_i64, c_i64, c_ptr, c_i32, c_ptr, c_size, c_ptr]),
    "mi355q_gptq_xtx_finish_f64": (c_i32, [c_ptr, c_i64, c_f64, c_ptr, c_ptr]),
    "mi355q_gptq_hessian_merge_f64": (c_i32, [c_ptr, c_f64, c_ptr, c_f64, c_i64, c_ptr, c_ptr]),
    "mi355q_gptq_hinv_workspace_bytes": (c_size, [c_i64]),
    "mi355q_shutdown": (c_i32, []),
    "mi355q_gptq_hinv_f64": (c_i32, [c_ptr, c_i64, c_f64, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_gptq_hinv_from_product_f32": (c_i32, [c_ptr, c_i64, c_f64, c_f64, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_gptq_hinv_batched_workspace_bytes": (c_size, [c_i32, c_i64]),
    "mi355q_gptq_hinv_f64_batched": (c_i32, [c_ptr, c_i32, c_i64, c_f64, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_gptq_hinv_from_product_batched_workspace_bytes": (c_size, [c_i32, c_i64]),
    "mi355q_gptq_hinv_from_product_f32_batched": (c_i32, [c_ptr, c_ptr, c_i32, c_i64, c_f64, c_ptr, c_ptr, c_ptr, c_size,
                                                          c_ptr]),
    "mi355q_gptq_apply_workspace_bytes": (c_size, [c_i64, c_i64]),
    "mi355q_gptq_apply_f32": (c_i32, [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i32,
                                      c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_gptq_apply_wide_f32": (c_i32, [c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i32,
                                           c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_cast_f32_to_f16": (c_i32, [c_ptr, c_i64, c_ptr, c_ptr]),
    "mi355q_oscar_col_sumsq_f32": (c_i32, [c_ptr, c_i64, c_i64, c_i32, c_ptr, c_ptr]),
    "mi355q_oscar_group_terms_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_i64, c_i32, c_ptr, c_ptr, c_ptr,
                                             c_ptr, c_ptr]),
    "mi355q_oscar_winner_energy_f64": (c_i32, [c_ptr, c_ptr, c_i64, c_i64, c_i32, c_ptr, c_ptr]),
    "mi355q_oscar_clip_workspace_bytes": (c_i32, [c_i64, c_i64, c_i64, ctypes.POINTER(ctypes.c_size_t)]),
    "mi355q_oscar_clip_bounds_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr,
                                             c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_oscar_quantize_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i32, c_i32,
                                          c_ptr, c_ptr]),
    "mi355q_dwr_scales_f32": (c_i32, [c_ptr, c_i64, c_i64, c_i64, c_i32, c_ptr, c_ptr, c_size, c_ptr]),
    "mi355q_dwr_max_error_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr]),
    "mi355q_comm_unique_id": (c_i32, [c_ptr]),
    "mi355q_comm_init_rank": (c_i32, [ctypes.POINTER(ctypes.c_void_p), c_i32, c_ptr, c_i32]),
    "mi355q_comm_info": (c_i32, [c_ptr, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "mi355q_comm_destroy": (c_i32, [c_ptr]),
    "mi355q_allgather_minmax": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr]),
    "mi355q_allreduce_minmax_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr]),
    "mi355q_allreduce_sum_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr]),
    "mi355q_allreduce_sum_f64": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr]),
    "mi355q_allreduce_hessian_f64": (c_i32, [c_ptr, c_ptr, c_i64, c_f64, c_ptr]),
    "mi355q_hessian_exchange_workspace_bytes": (c_size, [c_i64]),
    "mi355q_reduce_hessian_f64": (c_i32, [c_ptr, c_ptr, c_i64, c_f64, c_i32, c_ptr, c_size, c_ptr]),
    "mi355q_product_exchange_workspace_bytes": (c_size, [c_i64]),
    "mi355q_reduce_product_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_size, c_ptr]),
    "mi355q_file_to_device": (c_i32, [c_i32, c_i64, c_i64, c_ptr, c_ptr]),
    "mi355q_device_to_file": (c_i32, [c_ptr, c_i64, c_i32, c_i64, c_ptr]),
    "mi355q_file_io_finish": (c_i32, []),
    "mi355q_file_io_submit_upload": (c_i32, [c_i32, c_i64, c_i64, c_ptr, c_ptr, ctypes.POINTER(c_i64)]),
    "mi355q_file_io_wait": (c_i32, [c_i64]),
    "mi355q_file_io_submit_download": (c_i32, [c_ptr, c_i64, c_i32, c_i64, c_ptr, c_ptr]),
    "mi355q_file_io_submit_download_mapped": (c_i32, [c_ptr, c_i64, c_ptr, c_ptr, c_ptr]),
    "mi355q_prepare_device": (c_i32, []),
    "mi355q_clock_probe": (c_i32, [c_f64, c_ptr, c_ptr]),
    "mi355q_device_alloc": (c_i32, [c_size, ctypes.POINTER(ctypes.c_void_p)]),
    "mi355q_device_free": (c_i32, [c_ptr]),
}

STATUS_NAMES = {0: "OK", -1: "BAD_ARG", -2: "BAD_SHAPE", -3: "UNSUPPORTED", -4: "HIP_ERROR",
                -5: "RCCL_ERROR", -6: "IO_ERROR"}


class Mi355qError(RuntimeError):
  """A libmi355q call returned a negative status."""

  def __init__(self, status: int, message: str):
    super().__init__(f"libmi355q {STATUS_NAMES.get(status, status)}: {message}")
    self.status = status
    self.message = message


_lib = None


def lib() -> ctypes.CDLL:
  """Loads libmi355q.so once; raises if it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise RuntimeError(
          f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
          " (hipcc --offload-arch=gfx950). mi355q has no CPU fallback.")
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
      fn = getattr(handle, name)  # AttributeError = header/library drift
      fn.restype, fn.argtypes = res, args
    _lib = handle
  return _lib


def check(status: int) -> None:
  if status != 0:
    raise Mi355qError(status, lib().mi355q_last_error().decode("utf-8", "replace"))
