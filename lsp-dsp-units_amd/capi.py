"""ctypes declarations for include/mi_dspu.h (one line per exported symbol)."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_uint32, c_void_p

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmi_dspu.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libmi_dspu.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C lsp-dsp-units_amd`. There is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)


class MiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mi_dspu error %d: %s" % (code, message))
        self.code = code


class FilterParams(ctypes.Structure):
    """mi_filter_params_t == filter_params_t layout."""
    _fields_ = [("nType", c_uint32), ("nSlope", c_uint32), ("fFreq", c_float), ("fFreq2", c_float),
                ("fGain", c_float), ("fQuality", c_float)]


class FilterCascade(ctypes.Structure):
    _fields_ = [("t", c_float * 4), ("b", c_float * 4)]


class BiquadX1(ctypes.Structure):
    """mi_biquad_x1_t == dsp::biquad_x1_t layout."""
    _fields_ = [(n, c_float) for n in ("b0", "b1", "b2", "a1", "a2", "p0", "p1", "p2")]


# name -> (restype, argtypes); every symbol declared in include/mi_dspu.h must be listed here
# (tests/test_abi.py parses the header and checks both directions).
PROTOTYPES = {
    "mi_dspu_abi_version": (c_int, []),
    "mi_dspu_last_error": (c_char_p, []),
    "mi_dspu_device_count": (c_int, []),
    "mi_dspu_set_device": (c_int, [c_int]),
    "mi_dspu_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "mi_dspu_free": (c_int, [c_void_p]),
    "mi_dspu_memset": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "mi_dspu_copy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mi_dspu_copy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mi_dspu_copy_d2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mi_dspu_stream_create": (c_int, [POINTER(c_void_p)]),
    "mi_dspu_stream_destroy": (c_int, [c_void_p]),
    "mi_dspu_stream_synchronize": (c_int, [c_void_p]),
    "mi_dspu_event_create": (c_int, [POINTER(c_void_p)]),
    "mi_dspu_event_destroy": (c_int, [c_void_p]),
    "mi_dspu_event_record": (c_int, [c_void_p, c_void_p]),
    "mi_dspu_event_synchronize": (c_int, [c_void_p]),
    "mi_dspu_event_elapsed_ms": (c_int, [POINTER(c_float), c_void_p, c_void_p]),
    "mi_dspu_profile_next_launch": (c_int, [c_void_p, c_void_p]),
    "mi_dspu_last_launch": (c_char_p, []),
    "mi_dspu_source_sha": (c_char_p, [c_char_p]),
    "mi_dspu_last_stream_clock": (c_int, [POINTER(c_double), POINTER(c_double)]),
    "mi_dspu_graph_begin_capture": (c_int, [c_void_p]),
    "mi_dspu_graph_end_capture": (c_int, [c_void_p, POINTER(c_void_p)]),
    "mi_dspu_graph_launch": (c_int, [c_void_p, c_void_p]),
    "mi_dspu_graph_destroy": (c_int, [c_void_p]),
    "mi_dspu_comm_unique_id": (c_int, [c_void_p]),
    "mi_dspu_comm_create": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int]),
    "mi_dspu_comm_adopt": (c_int, [POINTER(c_void_p), c_void_p]),
    "mi_dspu_comm_destroy": (c_int, [c_void_p]),
    "mi_dspu_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "mi_analyzer_bank_allreduce_bins": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mi_analyzer_bank_allreduce_bins_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_void_p]),
    "mi_dspu_comm_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "mi_dynfilter_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_uint32]),
    "mi_dynfilter_bank_destroy": (c_int, [c_void_p]),
    "mi_dynfilter_bank_set_sample_rate": (c_int, [c_void_p, c_uint32]),
    "mi_dynfilter_bank_set_params": (c_int, [c_void_p, c_uint32, POINTER(FilterParams)]),
    "mi_dynfilter_bank_get_params": (c_int, [c_void_p, c_uint32, POINTER(FilterParams), POINTER(c_int)]),
    "mi_dynfilter_bank_set_filter_active": (c_int, [c_void_p, c_uint32, c_int]),
    "mi_dynfilter_bank_process": (c_int, [c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_dynfilter_sections": (c_int, [POINTER(FilterParams), c_uint32, c_float, POINTER(BiquadX1), c_uint32, POINTER(c_uint32)]),
    "mi_dynfilter_freq_chart": (c_int, [POINTER(FilterParams), c_uint32, c_void_p, c_void_p, c_float, c_size_t]),
    "mi_biquad_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_uint32]),
    "mi_biquad_bank_destroy": (c_int, [c_void_p]),
    "mi_biquad_bank_set_chains": (c_int, [c_void_p, c_uint32, POINTER(BiquadX1), c_uint32, c_int]),
    "mi_biquad_bank_set_all_chains": (c_int, [c_void_p, POINTER(BiquadX1), c_uint32, c_int]),
    "mi_biquad_bank_size": (c_int, [c_void_p, c_uint32, POINTER(c_uint32)]),
    "mi_biquad_bank_set_row_enabled": (c_int, [c_void_p, c_uint32, c_int]),
    "mi_biquad_bank_set_exact": (c_int, [c_void_p, c_int]),
    "mi_dspu_set_exact_iir_default": (c_int, [c_int]),
    "mi_biquad_bank_commit": (c_int, [c_void_p, c_void_p]),
    "mi_biquad_bank_reset": (c_int, [c_void_p, c_uint32, c_void_p]),
    "mi_biquad_bank_process": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_biquad_bank_process_blocks": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_biquad_bank_impulse_response": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "mi_biquad_bank_get_state": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mi_biquad_bank_set_state": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mi_filter_design": (c_int, [POINTER(FilterParams), c_uint32, c_void_p, c_uint32, POINTER(c_uint32),
                                 c_void_p, c_uint32, POINTER(c_uint32), POINTER(c_int)]),
    "mi_filter_limit": (c_int, [POINTER(FilterParams), c_uint32]),
    "mi_filter_freq_chart": (c_int, [POINTER(FilterParams), c_uint32, c_void_p, c_void_p, c_size_t]),
    "mi_convolver_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_void_p, c_size_t, c_void_p, c_uint32, c_uint32,
                                         c_float, c_void_p]),
    "mi_convolver_bank_destroy": (c_int, [c_void_p]),
    "mi_convolver_bank_faults": (c_int, [c_void_p, POINTER(c_uint32), c_void_p]),
    "mi_convolver_bank_reset": (c_int, [c_void_p, c_void_p]),
    "mi_convolver_bank_info": (c_int, [c_void_p, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32),
                                       POINTER(c_uint32)]),
    "mi_convolver_bank_process": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_convolver_bank_process_blocks": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_window": (c_int, [c_void_p, c_size_t, c_int]),
    "mi_window_general": (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_uint32]),
    "mi_envelope_reverse_noise_lin": (c_int, [c_void_p, c_float, c_float, c_float, c_size_t, c_int]),
    "mi_envelope_noise_lin": (c_int, [c_void_p, c_float, c_float, c_float, c_size_t, c_int]),
    "mi_crossover_bank_needs_reconfiguration": (c_int, [c_void_p, POINTER(c_int)]),
    "mi_loudness_bank_needs_update": (c_int, [c_void_p, POINTER(c_int)]),
    "mi_loudness_bank_update_settings": (c_int, [c_void_p, c_void_p]),
    "mi_ilufs_bank_needs_update": (c_int, [c_void_p, POINTER(c_int)]),
    "mi_ilufs_bank_update_settings": (c_int, [c_void_p, c_void_p]),
    "mi_analyzer_bank_reset": (c_int, [c_void_p]),
    "mi_envelope_noise_log": (c_int, [c_void_p, c_float, c_float, c_float, c_size_t, c_int, c_int]),
    "mi_envelope_noise_list": (c_int, [c_void_p, c_void_p, c_float, c_size_t, c_int, c_int]),
    "mi_spectral_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_uint32]),
    "mi_spectral_bank_destroy": (c_int, [c_void_p]),
    "mi_spectral_bank_set_rank": (c_int, [c_void_p, c_uint32]),
    "mi_spectral_bank_set_phase": (c_int, [c_void_p, c_float]),
    "mi_spectral_bank_get": (c_int, [c_void_p, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
    "mi_spectral_bank_bind": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "mi_spectral_bank_unbind": (c_int, [c_void_p]),
    "mi_spectral_bank_bind_mask": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mi_spectral_bank_bind_channels": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "mi_spectral_bank_reset": (c_int, [c_void_p, c_void_p]),
    "mi_spectral_bank_process": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_spectral_bank_process_blocks": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_spectral_bank_set_timing": (c_int, [c_void_p, c_int]),
    "mi_analyzer_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_uint32, c_uint32, c_float, c_uint32]),
    "mi_analyzer_bank_destroy": (c_int, [c_void_p]),
    "mi_analyzer_bank_configure": (c_int, [c_void_p, c_int, ctypes.c_double]),
    "mi_analyzer_bank_channel": (c_int, [c_void_p, c_uint32, c_int, c_uint32]),
    "mi_analyzer_bank_process": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "mi_analyzer_bank_get_spectrum": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_uint32, c_void_p]),
    "mi_analyzer_bank_reduce_bins": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "mi_analyzer_bank_process_reduce": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_int, c_void_p]),
    "mi_analyzer_bank_process_reduce_frames": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t, c_int, c_void_p]),
    "mi_analyzer_bank_info": (c_int, [c_void_p, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
    "mi_convolver_bank_set_irs_device": (c_int, [c_void_p, c_void_p, c_size_t, c_uint32, c_void_p, c_void_p]),
    "mi_convolver_bank_crossfade_irs_device": (c_int, [c_void_p, c_void_p, c_size_t, c_uint32, c_void_p, c_void_p]),
    "mi_spectral_bank_set_windows": (c_int, [c_void_p, c_int, c_int]),
    "mi_equalizer_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_uint32, c_uint32]),
    "mi_equalizer_bank_destroy": (c_int, [c_void_p]),
    "mi_equalizer_bank_set_params": (c_int, [c_void_p, c_uint32, c_uint32, POINTER(FilterParams)]),
    "mi_equalizer_bank_get_params": (c_int, [c_void_p, c_uint32, c_uint32, POINTER(FilterParams)]),
    "mi_equalizer_bank_set_mode": (c_int, [c_void_p, c_int]),
    "mi_equalizer_bank_set_sample_rate": (c_int, [c_void_p, c_uint32]),
    "mi_equalizer_bank_set_actual_sample_rate": (c_int, [c_void_p, c_uint32]),
    "mi_equalizer_bank_get_latency": (c_int, [c_void_p, POINTER(c_uint32), c_void_p]),
    "mi_equalizer_bank_set_smooth": (c_int, [c_void_p, c_int]),
    "mi_loudness_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_uint32, c_float]),
    "mi_loudness_bank_destroy": (c_int, [c_void_p]),
    "mi_loudness_bank_set_sample_rate": (c_int, [c_void_p, c_uint32, c_void_p]),
    "mi_loudness_bank_set_period": (c_int, [c_void_p, c_float]),
    "mi_loudness_bank_set_weighting": (c_int, [c_void_p, c_int]),
    "mi_loudness_bank_set_designation": (c_int, [c_void_p, c_uint32, c_int]),
    "mi_loudness_bank_set_link": (c_int, [c_void_p, c_uint32, c_float]),
    "mi_loudness_bank_set_active": (c_int, [c_void_p, c_uint32, c_int, c_void_p]),
    "mi_loudness_bank_clear": (c_int, [c_void_p, c_void_p]),
    "mi_loudness_bank_set_bound": (c_int, [c_void_p, c_uint32, c_int]),
    "mi_loudness_bank_latency": (c_int, [c_void_p, POINTER(c_uint32)]),
    "mi_loudness_bank_process": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_loudness_bank_process_gain": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_float, c_void_p]),
    "mi_loudness_bank_loudness": (c_int, [c_void_p, POINTER(c_float), c_void_p]),
    "mi_ilufs_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_uint32, c_float, c_float]),
    "mi_ilufs_bank_destroy": (c_int, [c_void_p]),
    "mi_ilufs_bank_set_sample_rate": (c_int, [c_void_p, c_uint32, c_void_p]),
    "mi_ilufs_bank_set_integration_period": (c_int, [c_void_p, c_float, c_void_p]),
    "mi_ilufs_bank_set_weighting": (c_int, [c_void_p, c_int]),
    "mi_ilufs_bank_set_designation": (c_int, [c_void_p, c_uint32, c_int]),
    "mi_ilufs_bank_set_active": (c_int, [c_void_p, c_uint32, c_int]),
    "mi_ilufs_bank_clear": (c_int, [c_void_p, c_void_p]),
    "mi_ilufs_bank_process": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_float, c_void_p]),
    "mi_ilufs_bank_loudness": (c_int, [c_void_p, POINTER(c_float), c_void_p]),
    "mi_ilufs_bank_history": (c_int, [c_void_p, c_void_p, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), c_void_p]),
    "mi_splitter_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_uint32, c_uint32]),
    "mi_splitter_bank_destroy": (c_int, [c_void_p]),
    "mi_splitter_bank_set_rank": (c_int, [c_void_p, c_uint32]),
    "mi_splitter_bank_set_chunk_rank": (c_int, [c_void_p, ctypes.c_int32]),
    "mi_splitter_bank_set_phase": (c_int, [c_void_p, c_float]),
    "mi_splitter_bank_get": (c_int, [c_void_p, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
    "mi_splitter_bank_bind_copy": (c_int, [c_void_p, c_uint32, c_void_p]),
    "mi_splitter_bank_bind_mask": (c_int, [c_void_p, c_uint32, POINTER(c_float), c_size_t, c_void_p]),
    "mi_splitter_bank_bind_callback": (c_int, [c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mi_splitter_bank_unbind": (c_int, [c_void_p, c_uint32]),
    "mi_splitter_bank_clear": (c_int, [c_void_p, c_void_p]),
    "mi_splitter_bank_process": (c_int, [c_void_p, POINTER(c_void_p), c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_splitter_bank_process_blocks": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_crossover_hipass": (c_float, [c_float, c_float, c_float]),
    "mi_crossover_lopass": (c_float, [c_float, c_float, c_float]),
    "mi_crossover_hipass_set": (None, [POINTER(c_float), POINTER(c_float), c_float, c_float, c_size_t]),
    "mi_crossover_hipass_apply": (None, [POINTER(c_float), POINTER(c_float), c_float, c_float, c_size_t]),
    "mi_crossover_lopass_set": (None, [POINTER(c_float), POINTER(c_float), c_float, c_float, c_size_t]),
    "mi_crossover_lopass_apply": (None, [POINTER(c_float), POINTER(c_float), c_float, c_float, c_size_t]),
    "mi_crossover_hipass_fft_set": (None, [POINTER(c_float), c_float, c_float, c_float, c_size_t]),
    "mi_crossover_hipass_fft_apply": (None, [POINTER(c_float), c_float, c_float, c_float, c_size_t]),
    "mi_crossover_lopass_fft_set": (None, [POINTER(c_float), c_float, c_float, c_float, c_size_t]),
    "mi_crossover_lopass_fft_apply": (None, [POINTER(c_float), c_float, c_float, c_float, c_size_t]),
    "mi_crossover_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_uint32]),
    "mi_crossover_bank_destroy": (c_int, [c_void_p]),
    "mi_crossover_bank_set_sample_rate": (c_int, [c_void_p, c_uint32]),
    "mi_crossover_bank_set_slope": (c_int, [c_void_p, c_uint32, c_uint32]),
    "mi_crossover_bank_set_frequency": (c_int, [c_void_p, c_uint32, c_float]),
    "mi_crossover_bank_set_mode": (c_int, [c_void_p, c_uint32, c_int]),
    "mi_crossover_bank_set_gain": (c_int, [c_void_p, c_uint32, c_float]),
    "mi_crossover_bank_get_split": (c_int, [c_void_p, c_uint32, POINTER(c_uint32), POINTER(c_float), POINTER(c_int)]),
    "mi_crossover_bank_get_band": (c_int, [c_void_p, c_uint32, POINTER(c_float), POINTER(c_float), POINTER(c_float), POINTER(c_int), c_void_p]),
    "mi_crossover_bank_process": (c_int, [c_void_p, POINTER(c_void_p), c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_crossover_bank_process_blocks": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_crossover_bank_freq_chart": (c_int, [c_void_p, c_uint32, POINTER(c_float), POINTER(c_float), c_size_t, c_void_p]),
    "mi_equalizer_bank_reset": (c_int, [c_void_p, c_void_p]),
    "mi_equalizer_bank_process": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_equalizer_bank_process_blocks": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]),
    "mi_equalizer_bank_info": (c_int, [c_void_p, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_int), POINTER(c_uint32)]),
    "mi_delay_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_size_t]),
    "mi_delay_bank_destroy": (c_int, [c_void_p]),
    "mi_delay_bank_set_delay": (c_int, [c_void_p, c_uint32, c_size_t]),
    "mi_delay_bank_get": (c_int, [c_void_p, c_uint32, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
    "mi_delay_bank_clear": (c_int, [c_void_p, c_void_p]),
    "mi_delay_bank_append": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "mi_delay_bank_process": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_int, c_int, c_float,
                                      c_void_p, c_size_t, c_void_p]),
    "mi_delay_bank_append_rows": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p, c_size_t, c_size_t, c_void_p]),
    "mi_delay_bank_process_rows": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_int, c_int,
                                   c_float, c_void_p, c_size_t, c_void_p]),
    "mi_delay_bank_process_ramping": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_int,
                                              c_float, c_void_p, c_size_t, c_void_p]),
    "mi_delay_bank_process_ramping_rows": (c_int, [c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_int,
                                              c_float, c_void_p, c_size_t, c_void_p]),
    "mi_ring_bank_create": (c_int, [POINTER(c_void_p), c_uint32, c_size_t, c_float]),
    "mi_ring_bank_create_shared": (c_int, [POINTER(c_void_p), c_uint32, c_size_t, c_float, POINTER(c_void_p)]),
    "mi_ring_bank_destroy": (c_int, [c_void_p]),
    "mi_ring_bank_fill": (c_int, [c_void_p, c_float, c_void_p]),
    "mi_ring_bank_append": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, POINTER(c_size_t), c_void_p]),
    "mi_ring_bank_get": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, POINTER(c_size_t), c_void_p]),
    "mi_ring_bank_info": (c_int, [c_void_p, c_size_t, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
    "mi_biquad_section_tables": (c_int, [POINTER(BiquadX1), c_int, POINTER(c_float), POINTER(c_uint32)]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    _fn = getattr(lib, _name)          # AttributeError here == symbol missing from the .so
    _fn.restype = _res
    _fn.argtypes = _args


def check(code):
    if code != 0:
        raise MiError(code, (lib.mi_dspu_last_error() or b"").decode("utf-8", "replace"))
    return code


SPECTRAL_FUNC = ctypes.CFUNCTYPE(None, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p)
SPLITTER_FUNC = ctypes.CFUNCTYPE(None, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p)
