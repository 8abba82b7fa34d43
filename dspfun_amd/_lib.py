"""ctypes binding of the C ABI declared in include/dspfft.h.

The product library is dspfun_amd/csrc/libdspfft_hip.so (hand-written HIP kernels for gfx950).
There is no CPU fallback: if the library is missing, load() raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSPFFT_LIB_PATH: another build of the same library (same-box A/B runs of a kernel change: tools/ab_oldlib.sh); nothing else is ever loaded
LIB_PATH = os.environ.get("DSPFFT_LIB_PATH") or os.path.join(_HERE, "csrc", "libdspfft_hip.so")

REDFT01, REDFT10 = 4, 5

#: every symbol include/dspfft.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "dspfft_plan_many_r2r", "dspfft_plan_r2r_2d", "dspfft_plan_set_scale", "dspfft_plan_set_axis_scale0",
    "dspfft_plan_many_r2r_f64", "dspfft_plan_set_scale_f64", "dspfft_plan_set_axis_scale0_f64", "dspfft_execute_f64", "dspfft_execute_masked_accumulate_f64", "dspfft_plan_scan_prepare", "dspfft_plan_set_input_window", "dspfft_plan_set_output_alternate", "dspfft_set_plan_effort", "dspfft_get_plan_effort",
    "dspfft_plan_many_r2r_ordered", "dspfft_plan_guru_r2r", "dspfft_execute_roundtrip", "dspfft_execute_roundtrip_u8",
    "dspfft_execute", "dspfft_plan_num_passes", "dspfft_execute_pass", "dspfft_destroy_plan", "dspfft_plan_describe", "dspfft_plan_algorithmic_bytes",
    "dspfft_execute_many", "dspfft_execute_many_repeat", "dspfft_execute_sum2", "dspfft_cosrows_create", "dspfft_cosrows_execute", "dspfft_cosrows_destroy", "dspfft_cztrows_create", "dspfft_cztrows_execute", "dspfft_cztrows_length", "dspfft_cztrows_destroy", "dspfft_transpose_f32", "dspfft_plan_set_input_modulation", "dspfft_stream_create", "dspfft_stream_destroy", "dspfft_stream_synchronize", "dspfft_event_create", "dspfft_event_destroy", "dspfft_event_synchronize", "dspfft_event_elapsed_ms",
    "dspfft_last_error", "dspfft_version", "dspfft_set_thread_plan_effort", "dspfft_get_thread_plan_effort", "dspfft_fftw_sparse_uploads",
    "dspfft_scan_zigzag", "dspfft_scan_zigzag_frame_ids", "dspfft_execute_masked_accumulate", "dspfft_scan_scatter", "dspfft_accumulate", "dspfft_broadcast_dc",
    "dspfft_scan_limit", "dspfft_scan_max_interval", "dspfft_scan_coord_slots", "dspfft_scan_owner_index", "dspfft_scan_frame_ids", "dspfft_scan_coords", "dspfft_scan_stamp",
    "dspfft_scan_index_to_frame_ids", "dspfft_scan_magnitude_work_bytes", "dspfft_scan_magnitude_index",
    "dspfft_u8_to_f32", "dspfft_f32_to_u8",
    "dspfft_zoom_ncomponents", "dspfft_zoom_basis", "dspfft_zoom_work_floats", "dspfft_zoom_product", "dspfft_gemm_nt_f32",
    "dspfft_zoom_last_error", "dspfft_zoomfft_create", "dspfft_zoomfft_work_floats", "dspfft_zoomfft_execute", "dspfft_zoomfft_destroy", "dspfft_zoomfft_last_error", "dspfft_zoomczt_create", "dspfft_zoomczt_work_floats", "dspfft_zoomczt_execute", "dspfft_zoomczt_destroy",
    "dspfft_applybasis_work_floats", "dspfft_applybasis_partsums",
    "dspfft_applybasis_work_floats_ex", "dspfft_applybasis_partsums_ex", "dspfft_applybasis_render",
    "dspfft_motion_load_u8", "dspfft_motion_store_u8", "dspfft_motion_load_f32", "dspfft_motion_store_f32", "dspfft_motion_topn_work_bytes", "dspfft_motion_topn", "dspfft_motion_last_error",
    "dspfft_spec_encode", "dspfft_ispec_decode", "dspfft_ispec_signmap", "dspfft_motion_filter", "dspfft_scan_pruned_accumulate", "dspfft_scan_pruned_work_floats", "dspfft_scan_pruned_accumulate_ws", "dspfft_pointwise_last_error",
]

class IoDim(C.Structure):
    """dspfft_iodim (include/dspfft.h): extent, input stride, output stride in elements"""
    _fields_ = [("n", C.c_int), ("is_", C.c_int), ("os", C.c_int)]


class MotionFilterParams(C.Structure):
    """dspfft_motion_filter_params (include/dspfft.h)"""
    _fields_ = [("active", C.c_int * 3), ("minbuf_hw", C.c_int * 2), ("block_depth", C.c_int), ("band_begin", C.c_int * 3), ("band_end", C.c_int * 3),
                ("damp", C.c_float), ("boost", C.c_float), ("threshold_lo", C.c_float), ("threshold_hi", C.c_float),
                ("preserve_dc", C.c_int), ("grey_add", C.c_float), ("quantizer", C.c_float)]


_lib = None


def bind(lib):
    """Attach argtypes/restypes to an already opened CDLL exporting the dspfft_* ABI."""
    ip = C.POINTER(C.c_int)
    vp = C.c_void_p
    lib.dspfft_plan_many_r2r.argtypes = [C.POINTER(vp), C.c_int, ip, C.c_int, ip, C.c_int, C.c_int, ip, C.c_int, C.c_int, ip]
    lib.dspfft_plan_r2r_2d.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.dspfft_plan_set_scale.argtypes = [vp, C.c_float]
    lib.dspfft_plan_set_axis_scale0.argtypes = [vp, C.c_int, C.c_float, C.c_float]
    lib.dspfft_execute.argtypes = [vp, vp, vp, vp]
    lib.dspfft_execute_sum2.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.dspfft_cosrows_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.dspfft_cosrows_execute.argtypes = [vp, vp, C.c_longlong, vp, C.c_longlong, C.c_double, C.c_double, vp]
    lib.dspfft_cosrows_destroy.argtypes = [vp]
    lib.dspfft_cosrows_destroy.restype = None
    lib.dspfft_cztrows_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.dspfft_cztrows_execute.argtypes = [vp, vp, C.c_longlong, C.c_longlong, C.c_int, vp, C.c_longlong, C.c_longlong, C.c_int, C.c_double, C.c_double, C.c_double, vp]
    lib.dspfft_cztrows_length.argtypes = [vp]
    lib.dspfft_cztrows_destroy.argtypes = [vp]
    lib.dspfft_cztrows_destroy.restype = None
    lib.dspfft_transpose_f32.argtypes = [vp, C.c_longlong, vp, C.c_longlong, C.c_int, C.c_int, vp]
    lib.dspfft_plan_set_input_modulation.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.dspfft_plan_set_input_window.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.dspfft_plan_set_output_alternate.argtypes = [vp, C.c_int, C.c_int]
    lib.dspfft_execute_many.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.POINTER(vp)]
    lib.dspfft_execute_many_repeat.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.dspfft_stream_create.restype = vp
    lib.dspfft_stream_create.argtypes = []
    lib.dspfft_stream_destroy.argtypes = [vp]
    lib.dspfft_stream_synchronize.argtypes = [vp]
    lib.dspfft_event_create.restype = vp
    lib.dspfft_event_create.argtypes = []
    lib.dspfft_event_destroy.argtypes = [vp]
    lib.dspfft_event_destroy.restype = None
    lib.dspfft_event_synchronize.argtypes = [vp]
    lib.dspfft_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    lib.dspfft_plan_many_r2r_f64.argtypes = lib.dspfft_plan_many_r2r.argtypes
    lib.dspfft_plan_many_r2r_ordered.argtypes = lib.dspfft_plan_many_r2r.argtypes + [C.c_int]
    lib.dspfft_plan_guru_r2r.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(IoDim), C.c_int, C.POINTER(IoDim), ip, C.c_int]
    lib.dspfft_execute_roundtrip.argtypes = [vp, vp, vp, vp, C.POINTER(MotionFilterParams), vp, vp]
    lib.dspfft_execute_roundtrip_u8.argtypes = [vp, vp, vp, vp, vp, C.c_double, C.POINTER(MotionFilterParams), vp, vp]
    lib.dspfft_plan_set_scale_f64.argtypes = [vp, C.c_double]
    lib.dspfft_plan_set_axis_scale0_f64.argtypes = [vp, C.c_int, C.c_double, C.c_double]
    lib.dspfft_execute_f64.argtypes = [vp, vp, vp, vp]
    lib.dspfft_execute_masked_accumulate_f64.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.c_int, vp]
    lib.dspfft_plan_scan_prepare.argtypes = [vp, vp, C.c_int, vp]
    lib.dspfft_set_plan_effort.argtypes = [C.c_int]
    lib.dspfft_set_plan_effort.restype = None
    lib.dspfft_get_plan_effort.argtypes = []
    lib.dspfft_plan_num_passes.argtypes = [vp]
    lib.dspfft_execute_pass.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.dspfft_destroy_plan.argtypes = [vp]
    lib.dspfft_destroy_plan.restype = None
    lib.dspfft_plan_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.dspfft_plan_algorithmic_bytes.argtypes = [vp]
    lib.dspfft_plan_algorithmic_bytes.restype = C.c_size_t
    lib.dspfft_last_error.restype = C.c_char_p
    lib.dspfft_version.restype = C.c_char_p
    lib.dspfft_scan_zigzag.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp]
    lib.dspfft_scan_zigzag_frame_ids.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, vp]
    lib.dspfft_execute_masked_accumulate.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, C.c_int, vp]
    lib.dspfft_scan_limit.restype = C.c_uint64
    lib.dspfft_scan_limit.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
    lib.dspfft_scan_max_interval.restype = C.c_uint64
    lib.dspfft_scan_max_interval.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
    lib.dspfft_scan_coord_slots.restype = C.c_uint64
    lib.dspfft_scan_coord_slots.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
    lib.dspfft_scan_owner_index.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, vp]
    lib.dspfft_scan_frame_ids.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, vp]
    lib.dspfft_scan_coords.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, vp]
    lib.dspfft_scan_stamp.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp]
    lib.dspfft_scan_index_to_frame_ids.argtypes = [vp, C.c_uint64, C.c_uint64, vp]
    lib.dspfft_scan_magnitude_work_bytes.restype = C.c_size_t
    lib.dspfft_scan_magnitude_work_bytes.argtypes = [C.c_uint32, C.c_uint32]
    lib.dspfft_scan_magnitude_index.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_double, vp, C.c_size_t, C.POINTER(C.c_uint32), vp]
    lib.dspfft_scan_scatter.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_int, vp]
    lib.dspfft_accumulate.argtypes = [vp, vp, C.c_uint64, vp]
    lib.dspfft_broadcast_dc.argtypes = [vp, vp, C.c_uint64, C.c_int, vp]
    lib.dspfft_u8_to_f32.argtypes = [vp, vp, C.c_uint64, vp]
    lib.dspfft_f32_to_u8.argtypes = [vp, vp, C.c_double, C.c_uint64, vp]
    if hasattr(lib, "dspfft_zoom_product"):      # HIP-only entry points (absent from the CPU emulation used in tests)
        lib.dspfft_zoom_ncomponents.restype = C.c_size_t
        lib.dspfft_zoom_ncomponents.argtypes = [C.c_double, C.c_double, C.c_size_t]
        lib.dspfft_zoom_basis.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_size_t, C.c_size_t, vp]
        lib.dspfft_zoom_work_floats.restype = C.c_size_t
        lib.dspfft_zoom_work_floats.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_int]
        lib.dspfft_zoom_product.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_int, C.c_int, vp, vp]
        lib.dspfft_gemm_nt_f32.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int,
                                           C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.c_float, vp]
        lib.dspfft_zoom_last_error.restype = C.c_char_p
        lib.dspfft_zoomfft_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        lib.dspfft_zoomfft_work_floats.restype = C.c_size_t
        lib.dspfft_zoomfft_work_floats.argtypes = [vp]
        lib.dspfft_zoomfft_execute.argtypes = [vp, vp, C.c_double, C.c_double, vp, vp, vp]
        lib.dspfft_zoomfft_destroy.argtypes = [vp]
        lib.dspfft_zoomfft_last_error.restype = C.c_char_p
        lib.dspfft_zoomczt_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        lib.dspfft_zoomczt_work_floats.restype = C.c_size_t
        lib.dspfft_zoomczt_work_floats.argtypes = [vp]
        lib.dspfft_zoomczt_execute.argtypes = [vp, vp, C.c_double, C.c_double, vp, vp, vp]
        lib.dspfft_zoomczt_destroy.argtypes = [vp]
        lib.dspfft_spec_encode.argtypes = [vp, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, vp]
        lib.dspfft_ispec_decode.argtypes = [vp, C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, vp]
        lib.dspfft_motion_filter.argtypes = [vp, ip, ip, ip, ip, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, vp, vp]
        lib.dspfft_pointwise_last_error.restype = C.c_char_p
        lib.dspfft_ispec_signmap.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
        lib.dspfft_motion_load_u8.argtypes = [vp, vp, ip, ip, C.c_int, C.c_double, C.c_double, vp]
        lib.dspfft_motion_store_u8.argtypes = [vp, vp, ip, ip, C.c_int, C.c_double, C.c_double, C.c_double, vp]
        lib.dspfft_motion_load_f32.argtypes = [vp, vp, ip, ip, C.c_int, C.c_double, C.c_double, vp]
        lib.dspfft_motion_store_f32.argtypes = [vp, vp, ip, ip, C.c_int, C.c_double, C.c_double, C.c_double, vp]
        lib.dspfft_motion_topn_work_bytes.restype = C.c_size_t
        lib.dspfft_motion_topn_work_bytes.argtypes = [C.c_size_t]
        lib.dspfft_motion_topn.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, vp]
        lib.dspfft_motion_last_error.restype = C.c_char_p
        lib.dspfft_scan_pruned_accumulate.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        lib.dspfft_scan_pruned_work_floats.restype = C.c_size_t
        lib.dspfft_scan_pruned_work_floats.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.dspfft_scan_pruned_accumulate_ws.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        lib.dspfft_applybasis_work_floats.restype = C.c_size_t
        lib.dspfft_applybasis_work_floats.argtypes = [C.c_int] * 7
        lib.dspfft_applybasis_work_floats_ex.restype = C.c_size_t
        lib.dspfft_applybasis_work_floats_ex.argtypes = [C.c_int] * 7
        lib.dspfft_applybasis_partsums_ex.argtypes = [vp, vp, vp] + [C.c_int] * 10 + [C.c_longlong, C.c_longlong, C.c_int, vp, vp]
        lib.dspfft_applybasis_render.argtypes = [vp, vp] + [C.c_int] * 11 + [C.c_double, C.c_double, C.POINTER(C.c_float), vp]
        lib.dspfft_applybasis_partsums.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, vp, vp]
    return lib


def load():
    """The HIP product library.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  dspfun_amd has no CPU fallback.")
        _lib = bind(C.CDLL(LIB_PATH))
    return _lib
