"""ctypes bindings for the two in-tree libraries (built by `make -C gliclass/c_amd`).

libgliclass_hip.so   — include/gliclass_hip.h  (HIP engine C-ABI)
libgliclass_model.so — include/model.h, parallel_processor.h, postprocessor.h, glc_weights.h (pure-C host)

There is no Python/CPU fallback for compute: if the HIP library is missing, importing the engine
raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.environ.get("GLC_HIP_SO") or os.path.join(_HERE, "libgliclass_hip.so")   # override: developer A/B of two builds
MODEL_SO = os.environ.get("GLC_MODEL_SO") or os.path.join(_HERE, "libgliclass_model.so")   # override: sanitizer build of the host layer


class ModelConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab", "hidden", "layers", "heads", "head_dim", "inter", "pos_buckets",
                                         "max_rel_pos", "pad_id", "cls_id", "sep_id", "class_token_index",
                                         "text_token_index", "pooling", "scorer", "embed_class_token",
                                         "normalize_features", "backbone", "kv_heads", "causal")] + \
               [("ln_eps", C.c_float), ("logit_scale", C.c_float), ("rope_theta", C.c_float)]


class Weights(C.Structure):
    _fields_ = [("cfg", ModelConfig), ("n_tensors", C.c_int), ("tensors", C.POINTER(C.POINTER(C.c_float))),
                ("_map", C.c_void_p), ("_map_len", C.c_size_t), ("_owned", C.POINTER(C.c_float))]


class OrtValue(C.Structure):
    _fields_ = [("type", C.c_int), ("ndim", C.c_size_t), ("dims", C.c_int64 * 4), ("data", C.c_void_p), ("owns_data", C.c_int)]


class TokenizerEncodeResult(C.Structure):
    _fields_ = [("token_ids", C.POINTER(C.c_int)), ("len", C.c_size_t)]


class TokenizedInputs(C.Structure):
    _fields_ = [("input_ids", C.POINTER(C.POINTER(C.c_int))), ("token_type_ids", C.POINTER(C.POINTER(C.c_int))),
                ("attention_mask", C.POINTER(C.POINTER(C.c_int))), ("batch_size", C.c_size_t), ("seq_length", C.c_size_t)]


HIP_SYMBOLS = ["glc_device_count", "glc_last_error", "glc_engine_create", "glc_engine_destroy", "glc_engine_forward",
               "glc_engine_forward_device", "glc_engine_sync", "glc_device_malloc", "glc_device_free", "glc_memcpy_h2d",
               "glc_memcpy_d2h", "glc_timer_start", "glc_timer_stop_ms", "glc_profile_enable", "glc_profile_read",
               "glc_debug_keep_hidden", "glc_debug_get_hidden", "glc_debug_set_attention_impl", "glc_delta_table",
               "glc_engine_config", "glc_engine_dtype", "glc_debug_gemm_bench", "glc_debug_attn_bench", "glc_engine_set_prune_last_layer", "glc_engine_set_length_buckets", "glc_plan_length_buckets", "glc_debug_last_forward_groups", "glc_debug_set_group_split", "glc_debug_last_forward_group_split", "glc_debug_set_ln_fused", "glc_debug_last_forward_ln_folded", "glc_debug_set_precision_mask", "glc_debug_set_gemm_full_lines", "glc_debug_gemm_mx_check", "glc_debug_range_retries", "glc_debug_fp8_range_retries", "glc_debug_fp8_range_sticky", "glc_debug_activation_exponent", "glc_debug_set_mx2", "glc_debug_is_developer_build", "glc_engine_device_forward_valid", "glc_debug_mx_weight_bytes", "glc_debug_set_mx", "glc_debug_last_forward_mx", "glc_debug_last_forward_mx_attention", "glc_debug_set_stop", "glc_debug_read_workspace", "glc_debug_set_mx_attention"]
MODEL_SYMBOLS = ["flatten_int_array", "create_tensor", "prepare_input_tensors", "initialize_ort_api",
                 "initialize_ort_environment", "create_ort_session", "run_inference", "parallel_inference",
                 "glc_session_num_devices", "parallel_preprocess", "parallel_postprocess", "sigmoid",
                 "process_output_tensor", "OrtGetApiBase", "g_ort", "glc_weights_load", "glc_weights_free",
                 "glc_prng_fill", "glc_fnv1a64", "glc_named_config", "glc_tensor_spec", "prepare_input", "prepare_inputs",
                 "free_prepared_inputs", "tokenizers_new_from_str", "tokenizers_encode", "tokenizers_encode_batch",
                 "tokenizers_free_encode_results", "tokenizers_decode", "tokenizers_get_decode_str", "tokenizers_get_vocab_size",
                 "tokenizers_id_to_token", "tokenizers_token_to_id", "tokenizers_free", "glc_tokenizer_normalize",
                 "tokenize_inputs", "print_tokenized_inputs", "free_tokenized_inputs", "create_tokenizer",
                 "read_file", "parse_json", "string_to_bool", "free_parsed_data", "glc_load_hf_checkpoint", "parallel_classify", "glc_config_prompt_first"]

_hip = None
_model = None


def hip():
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise RuntimeError(f"{HIP_SO} is missing — build it with `make -C gliclass/c_amd` (no fallback path exists)")
        L = C.CDLL(HIP_SO, mode=C.RTLD_GLOBAL)
        L.glc_last_error.restype = C.c_char_p
        L.glc_engine_create.restype = C.c_void_p
        L.glc_engine_create.argtypes = [C.POINTER(ModelConfig), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
        L.glc_engine_destroy.argtypes = [C.c_void_p]
        L.glc_engine_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.glc_engine_forward_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.glc_engine_sync.argtypes = [C.c_void_p]
        L.glc_engine_device_forward_valid.argtypes = [C.c_void_p]
        L.glc_device_malloc.restype = C.c_void_p
        L.glc_device_malloc.argtypes = [C.c_void_p, C.c_size_t]
        L.glc_device_free.argtypes = [C.c_void_p, C.c_void_p]
        L.glc_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.glc_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.glc_timer_start.argtypes = [C.c_void_p]
        L.glc_timer_stop_ms.argtypes = [C.c_void_p]
        L.glc_timer_stop_ms.restype = C.c_float
        L.glc_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.glc_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]
        L.glc_debug_keep_hidden.argtypes = [C.c_void_p, C.c_int]
        L.glc_debug_get_hidden.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.glc_debug_set_attention_impl.argtypes = [C.c_void_p, C.c_int]
        L.glc_delta_table.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.glc_engine_config.restype = C.POINTER(ModelConfig)
        L.glc_engine_config.argtypes = [C.c_void_p]
        L.glc_engine_dtype.argtypes = [C.c_void_p]
        L.glc_engine_set_prune_last_layer.argtypes = [C.c_void_p, C.c_int]
        L.glc_debug_gemm_bench.restype = C.c_float
        L.glc_debug_gemm_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.glc_engine_set_length_buckets.argtypes = [C.c_void_p, C.c_int]
        L.glc_debug_last_forward_groups.argtypes = [C.c_void_p]
        L.glc_debug_set_group_split.argtypes = [C.c_void_p, C.c_int]
        L.glc_debug_last_forward_group_split.argtypes = [C.c_void_p]
        L.glc_debug_set_ln_fused.argtypes = [C.c_void_p, C.c_int]
        L.glc_debug_last_forward_ln_folded.argtypes = [C.c_void_p]
        L.glc_debug_set_precision_mask.argtypes = [C.c_void_p, C.c_int]
        L.glc_debug_set_gemm_full_lines.argtypes = [C.c_int]
        L.glc_debug_range_retries.argtypes = [C.c_void_p]
        L.glc_debug_fp8_range_retries.argtypes = [C.c_void_p]
        L.glc_debug_fp8_range_sticky.argtypes = [C.c_void_p]
        L.glc_debug_activation_exponent.argtypes = [C.c_void_p]
        L.glc_debug_set_mx2.argtypes = [C.c_void_p, C.c_int]
        L.glc_debug_mx_weight_bytes.argtypes = [C.c_void_p]
        L.glc_debug_mx_weight_bytes.restype = C.c_longlong
        L.glc_debug_set_mx.argtypes = [C.c_void_p, C.c_int]
        L.glc_debug_last_forward_mx.argtypes = [C.c_void_p]
        L.glc_debug_last_forward_mx_attention.argtypes = [C.c_void_p]
        L.glc_debug_set_stop.argtypes = [C.c_void_p, C.c_int]
        L.glc_debug_set_mx_attention.argtypes = [C.c_void_p, C.c_int]
        L.glc_debug_read_workspace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.glc_debug_gemm_mx_check.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_double)]
        L.glc_plan_length_buckets.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.glc_debug_attn_bench.restype = C.c_float
        L.glc_debug_attn_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        _hip = L
    return _hip


def model():
    global _model
    if _model is None:
        hip()
        if not os.path.exists(MODEL_SO):
            raise RuntimeError(f"{MODEL_SO} is missing — build it with `make -C gliclass/c_amd`")
        L = C.CDLL(MODEL_SO, mode=C.RTLD_GLOBAL)
        L.flatten_int_array.restype = C.POINTER(C.c_int64)
        L.flatten_int_array.argtypes = [C.POINTER(C.POINTER(C.c_int)), C.c_size_t, C.c_size_t]
        L.create_tensor.restype = C.POINTER(OrtValue)
        L.create_tensor.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.prepare_input_tensors.argtypes = [C.POINTER(TokenizedInputs), C.POINTER(C.POINTER(OrtValue)), C.POINTER(C.POINTER(OrtValue))]
        L.initialize_ort_environment.restype = C.c_void_p
        L.create_ort_session.restype = C.c_void_p
        L.create_ort_session.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.run_inference.restype = C.POINTER(OrtValue)
        L.run_inference.argtypes = [C.c_void_p, C.POINTER(OrtValue), C.POINTER(OrtValue)]
        L.parallel_inference.argtypes = [C.c_void_p, C.POINTER(C.POINTER(OrtValue)), C.POINTER(C.POINTER(OrtValue)), C.c_size_t,
                                         C.POINTER(C.POINTER(OrtValue))]
        L.glc_session_num_devices.argtypes = [C.c_void_p]
        L.sigmoid.restype = C.c_float
        L.sigmoid.argtypes = [C.c_float]
        L.prepare_input.restype = C.c_void_p
        L.prepare_input.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_size_t, C.c_bool]
        L.prepare_inputs.restype = C.POINTER(C.c_void_p)
        L.prepare_inputs.argtypes = [C.POINTER(C.c_char_p), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_bool, C.c_bool]
        L.free_prepared_inputs.argtypes = [C.c_void_p, C.c_size_t]
        L.glc_weights_load.argtypes = [C.c_char_p, C.POINTER(Weights)]
        L.glc_weights_free.argtypes = [C.POINTER(Weights)]
        L.glc_prng_fill.argtypes = [C.c_uint64, C.c_char_p, C.c_size_t, C.c_double, C.c_double, C.c_void_p]
        L.glc_fnv1a64.restype = C.c_uint64
        L.glc_fnv1a64.argtypes = [C.c_char_p]
        L.glc_named_config.argtypes = [C.c_char_p, C.POINTER(ModelConfig)]
        L.glc_tensor_spec.argtypes = [C.POINTER(ModelConfig), C.c_int, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.tokenizers_new_from_str.restype = C.c_void_p
        L.tokenizers_new_from_str.argtypes = [C.c_char_p, C.c_size_t]
        L.tokenizers_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(TokenizerEncodeResult)]
        L.tokenizers_encode_batch.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int,
                                              C.POINTER(TokenizerEncodeResult)]
        L.tokenizers_free_encode_results.argtypes = [C.POINTER(TokenizerEncodeResult), C.c_size_t]
        L.tokenizers_decode.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t, C.c_int]
        L.tokenizers_get_decode_str.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
        L.tokenizers_get_vocab_size.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.tokenizers_id_to_token.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.tokenizers_token_to_id.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]
        L.tokenizers_free.argtypes = [C.c_void_p]
        L.glc_tokenizer_normalize.restype = C.c_void_p
        L.glc_tokenizer_normalize.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.create_tokenizer.restype = C.c_void_p
        L.create_tokenizer.argtypes = [C.c_char_p]
        L.tokenize_inputs.restype = TokenizedInputs
        L.tokenize_inputs.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, C.c_size_t]
        L.free_tokenized_inputs.argtypes = [C.POINTER(TokenizedInputs)]
        L.print_tokenized_inputs.argtypes = [C.POINTER(TokenizedInputs)]
        L.read_file.restype = C.c_void_p
        L.read_file.argtypes = [C.c_char_p]
        L.string_to_bool.restype = C.c_bool
        L.string_to_bool.argtypes = [C.c_char_p]
        _model = L
    return _model
