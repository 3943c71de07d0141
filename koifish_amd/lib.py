"""ctypes loader for libkf_hip.so / libkf_host.so (built in-tree by koifish_amd/build.py)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# KF_LIB_DIR (development only: scratch/build_variant.py): a directory holding an A/B build of the two libraries
LIB_HIP = os.path.join(os.environ.get("KF_LIB_DIR", HERE), "libkf_hip.so")
LIB_HOST = os.path.join(os.environ.get("KF_LIB_DIR", HERE), "libkf_host.so")

# typNUMBER (src/g_float.hpp:84-117)
F32, F64, F16, BF16, F8E5M2, F8E4M3, U8, I8, U16, I16, U32, I32, U64, I64, Q4, Q3, Q2, T_SIGN, T_SEQ, BOOL1, T_BINARY, T_BINARY_3, T_BINARY_TILE = range(23)
BITS = {BF16: 16, F8E5M2: 8, Q4: 4, T_SIGN: 2, BOOL1: 1, T_BINARY: 1}
KF_EPI_RESIDUAL = 1
QUANT_GROUP, QUANT_ROW_LUT, QUANT_ROW_RTN = 0, 1, 2  # kf_weight.quant
NF4 = 1000  # not a typNUMBER: "Q4 with the normal-float quant card" (QUANT_MODE::RTNf) for the helpers that take a storage type

# every symbol include/kf_abi.h declares (tests/test_abi_symbols.py checks the header against this list and the .so)
ABI_SYMBOLS = [
    "kf_init", "kf_destroy", "kf_sync", "kf_last_error", "kf_version", "kf_malloc", "kf_free", "kf_memset", "kf_memset32", "kf_h2d", "kf_d2h", "kf_d2d",
    "kf_graph_begin", "kf_graph_end", "kf_graph_launch", "kf_graph_destroy", "kf_event_create", "kf_event_record", "kf_event_elapsed_ms",
    "kf_event_destroy", "kf_dequant", "kf_quantize", "kf_linear", "kf_rmsnorm", "kf_qknorm_rope", "kf_rope_table_host", "kf_attn_decode",
    "kf_attn_scratch_bytes", "kf_linear_f32", "kf_tp_reduce", "kf_swiglu", "kf_add", "kf_embed", "kf_lm_head", "kf_head_scratch_bytes", "kf_norm_linear",
    "kf_norm_gateup_swiglu", "kf_attn_block", "kf_norm_lm_head", "kf_set_state", "kf_embed_state", "kf_embed_batch", "kf_qknorm_rope_batch", "kf_attn_prefill", "kf_sample", "kf_linear_multi", "kf_linear_multi_scratch_bytes", "kf_gateup_swiglu_batch", "kf_adamw", "kf_layernorm", "kf_gelu", "kf_sample_topk", "kf_fused_classifier", "kf_gelu_backward", "kf_swiglu_backward", "kf_rope_backward", "kf_norm_backward", "kf_norm_backward_scratch_bytes", "kf_linear_backward", "kf_linear_backward_scratch_bytes", "kf_embed_backward", "kf_attn_backward", "kf_attn_backward_scratch_bytes", "kf_attn_prefill_batch", "kf_attn_prefill_batch_strided", "kf_embed_pos", "kf_memset2d", "kf_copy_blocks", "kf_argmax_rows_state", "kf_qknorm_rope_train",
    "kf_hot_rows", "kf_linear_masked", "kf_norm_gateup_swiglu_masked", "kf_linear_scratch_bytes", "kf_set_scratch", "kf_engine_workspace_bytes", "kf_engine_create", "kf_engine_step", "kf_engine_check", "kf_engine_destroy", "kf_engine_set_embedding", "kf_engine_set_head", "kf_engine_step_head", "kf_engine_steps_head", "kf_engine_reset", "kf_set_canonical", "kf_get_canonical", "kf_engine_served", "kf_engine_tune", "kf_engine_stats", "kf_qkv_rope_batch", "kf_qkv_rope_seqs", "kf_set_dequant_arena", "kf_dequant_arena_used", "kf_resident_scratch_bytes",
    "kf_xengine_workspace_bytes", "kf_xengine_create", "kf_xengine_served", "kf_xengine_set_embedding", "kf_xengine_set_head", "kf_xengine_steps", "kf_xengine_check", "kf_xengine_reset", "kf_xengine_destroy", "kf_xengine_workspace_bytes_tp", "kf_xengine_create_tp", "kf_xengine_set_head_tp",
    "kf_tp_recv_bytes", "kf_tp_push_bytes", "kf_tp_commit", "kf_tp_alloc", "kf_tp_ipc_export", "kf_tp_ipc_open", "kf_tp_ipc_close", "kf_linear_f32_push", "kf_tp_reduce_recv", "kf_tp_lm_head", "kf_tp_pick",
]


class KFError(RuntimeError):
    pass


class Weight(C.Structure):
    """struct kf_weight (include/kf_abi.h)"""
    _fields_ = [("data", C.c_void_p), ("gama", C.c_void_p), ("type", C.c_int32), ("ne0", C.c_int32), ("ne1", C.c_int32), ("nGroup", C.c_int32),
                ("lGroup", C.c_int32), ("qMin", C.c_int32), ("qMax", C.c_int32), ("qBias", C.c_int32), ("qzeros", C.c_void_p), ("qscales", C.c_void_p), ("quant", C.c_int32), ("reserved_", C.c_int32)]


_libs = None


def load():
    """Returns (hip, host) CDLLs.  No fallback: a missing library is an error the caller must see."""
    global _libs
    if _libs is None:
        for p in (LIB_HIP, LIB_HOST):
            if not os.path.exists(p):
                raise KFError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                              "koifish_amd has no CPU fallback." % p)
        hip = C.CDLL(LIB_HIP, mode=C.RTLD_GLOBAL)
        host = C.CDLL(os.environ.get("KF_HOST_LIB", LIB_HOST))   # KF_HOST_LIB: the sanitizer build of the host library (tests/test_host_asan_cpu.py)
        hip.kf_last_error.restype = C.c_char_p
        hip.kf_version.restype = C.c_char_p
        hip.kf_attn_scratch_bytes.restype = C.c_size_t
        hip.kf_head_scratch_bytes.restype = C.c_size_t
        hip.kf_linear.argtypes = [C.c_void_p, C.POINTER(Weight), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_void_p]
        hip.kf_linear_f32.argtypes = [C.c_void_p, C.POINTER(Weight), C.c_void_p, C.c_void_p]
        hip.kf_tp_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        hip.kf_rmsnorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        hip.kf_qknorm_rope.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
        hip.kf_rope_table_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
        hip.kf_attn_decode.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        hip.kf_attn_block.argtypes = [C.c_void_p] * 9 + [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        hip.kf_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        hip.kf_linear_multi.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        hip.kf_linear_multi_scratch_bytes.argtypes, hip.kf_linear_multi_scratch_bytes.restype = [C.c_int, C.c_void_p, C.c_int], C.c_size_t
        hip.kf_gateup_swiglu_batch.argtypes = [C.c_void_p, C.POINTER(Weight), C.POINTER(Weight), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        hip.kf_adamw.argtypes = [C.c_void_p] * 5 + [C.c_size_t, C.c_int] + [C.c_float] * 8 + [C.c_uint32, C.c_void_p]
        hip.kf_sample_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        hip.kf_layernorm.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        hip.kf_qknorm_rope_train.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        hip.kf_attn_prefill_batch.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        hip.kf_attn_backward.argtypes = [C.c_void_p] * 4 + [C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        hip.kf_attn_backward_scratch_bytes.argtypes, hip.kf_attn_backward_scratch_bytes.restype = [C.c_int, C.c_int, C.c_int], C.c_size_t
        hip.kf_attn_prefill_batch_strided.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        hip.kf_embed_pos.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        hip.kf_argmax_rows_state.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        hip.kf_memset32.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_size_t]
        hip.kf_copy_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        hip.kf_memset2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t]
        hip.kf_embed_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        hip.kf_linear_backward.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p]
        hip.kf_linear_backward_scratch_bytes.argtypes, hip.kf_linear_backward_scratch_bytes.restype = [C.c_int, C.c_int, C.c_int], C.c_size_t
        hip.kf_norm_backward.argtypes = [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p]
        hip.kf_norm_backward_scratch_bytes.argtypes, hip.kf_norm_backward_scratch_bytes.restype = [C.c_int, C.c_int, C.c_int], C.c_size_t
        hip.kf_rope_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int]
        hip.kf_gelu_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        hip.kf_swiglu_backward.argtypes = [C.c_void_p] * 5 + [C.c_size_t]
        hip.kf_fused_classifier.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        hip.kf_gelu.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        hip.kf_embed_batch.argtypes = [C.c_void_p, C.POINTER(Weight), C.c_void_p, C.c_int, C.c_void_p]
        hip.kf_qknorm_rope_batch.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float]
        hip.kf_qkv_rope_batch.argtypes = [C.c_void_p] * 8 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        hip.kf_qkv_rope_seqs.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
        hip.kf_attn_prefill.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
        hip.kf_norm_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        hip.kf_norm_gateup_swiglu.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.POINTER(Weight), C.POINTER(Weight), C.c_void_p]
        hip.kf_norm_lm_head.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.POINTER(Weight), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        hip.kf_lm_head.argtypes = [C.c_void_p, C.POINTER(Weight), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        hip.kf_embed.argtypes = [C.c_void_p, C.POINTER(Weight), C.c_int, C.c_void_p, C.c_void_p]
        hip.kf_swiglu.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        hip.kf_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        hip.kf_dequant.argtypes = [C.c_void_p, C.POINTER(Weight), C.c_void_p]
        hip.kf_quantize.argtypes = [C.c_void_p, C.POINTER(Weight), C.c_void_p, C.c_int]
        hip.kf_init.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        hip.kf_destroy.argtypes = [C.c_void_p]
        hip.kf_sync.argtypes = [C.c_void_p]
        hip.kf_event_create.argtypes = [C.POINTER(C.c_void_p)]
        hip.kf_event_record.argtypes = [C.c_void_p, C.c_void_p]
        hip.kf_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        hip.kf_event_destroy.argtypes = [C.c_void_p]
        host.kfh_create.restype = C.c_void_p
        host.kfh_create.argtypes = [C.c_int, C.c_void_p] + [C.c_int] * 8 + [C.c_float] * 3 + [C.POINTER(C.c_int)]
        host.kfh_destroy.argtypes = [C.c_void_p]
        host.kfh_ctx.restype = C.c_void_p
        host.kfh_ctx.argtypes = [C.c_void_p]
        host.kfh_set_fuse_level.argtypes = [C.c_void_p, C.c_int]
        host.kfh_set_weight.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int]
        host.kfh_set_weight_lut.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        host.kfh_tie_head.argtypes = [C.c_void_p]
        host.kfh_set_norm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        host.kfh_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        host.kfh_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        host.kfh_set_forced.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        host.kfh_set_state.argtypes = [C.c_void_p, C.c_int, C.c_int]
        host.kfh_last_error.restype = C.c_char_p
        host.kfh_host_error.restype = C.c_char_p
        host.kfh_st_open.restype = C.c_void_p
        host.kfh_st_open.argtypes = [C.c_char_p, C.c_int]
        host.kfh_st_close.argtypes = [C.c_void_p]
        host.kfh_st_count.argtypes = [C.c_void_p]
        host.kfh_st_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        host.kfh_st_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        host.kfh_load_hf.restype = C.c_void_p
        host.kfh_load_hf.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        host.kfh_get_config.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        host.kfh_tp_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        host.kfh_tp_area.restype = C.c_void_p
        host.kfh_tp_area.argtypes = [C.c_void_p]
        host.kfh_tp_set_peer.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        host.kfh_tp_export.argtypes = [C.c_void_p, C.c_void_p]
        host.kfh_tp_open_peer.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        host.kfh_tp_check.argtypes = [C.c_void_p]
        host.kfh_tp_group_run.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]
        host.kfh_save_kun.argtypes = [C.c_void_p, C.c_char_p]
        host.kfh_load_kun.restype = C.c_void_p
        host.kfh_load_kun.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        host.kfh_kun_write.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                       C.c_char_p]
        host.kfh_st_config_json.restype = C.c_int64
        host.kfh_st_config_json.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        host.kfh_st_blob_sizes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        host.kfh_json_to_msgpack.restype = C.c_int64
        host.kfh_json_to_msgpack.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
        host.kfh_msgpack_to_json.restype = C.c_int64
        host.kfh_msgpack_to_json.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_int64]
        host.kfh_set_sampler.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_uint64]
        host.kfh_prefill.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        host.kfh_set_prefill_mode.argtypes = [C.c_void_p, C.c_int, C.c_int]
        host.kfh_run_steps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        host.kfh_sync.argtypes = [C.c_void_p]
        host.kfh_get_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for f in ("kfh_kcache", "kfh_vcache", "kfh_logits", "kfh_hidden"):
            getattr(host, f).restype = C.c_void_p
            getattr(host, f).argtypes = [C.c_void_p]
        host.kfh_num_graphs.argtypes = [C.c_void_p]
        host.kfh_set_engine.argtypes = [C.c_void_p, C.c_int]
        host.kfh_engine_steps.argtypes = [C.c_void_p]
        host.kfh_engine_check.argtypes = [C.c_void_p]
        host.kfh_set_hot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        host.kfh_n_hot.argtypes = [C.c_void_p, C.c_int]
        host.kfh_engine_only.argtypes = [C.c_void_p, C.c_int]
        hip.kf_hot_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        hip.kf_linear_masked.argtypes = [C.c_void_p, C.POINTER(Weight), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        hip.kf_norm_gateup_swiglu_masked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.POINTER(Weight), C.POINTER(Weight), C.c_void_p, C.c_void_p, C.c_int]
        hip.kf_linear_scratch_bytes.argtypes, hip.kf_linear_scratch_bytes.restype = [C.POINTER(Weight), C.c_int], C.c_size_t
        hip.kf_set_scratch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        hip.kf_set_dequant_arena.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        hip.kf_dequant_arena_used.argtypes, hip.kf_dequant_arena_used.restype = [C.c_void_p], C.c_size_t
        hip.kf_resident_scratch_bytes.argtypes, hip.kf_resident_scratch_bytes.restype = [], C.c_size_t
        host.kfh_set_prefill_resident.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        host.kfh_weights_changed.argtypes = [C.c_void_p]
        host.kfh_resident_bytes.argtypes, host.kfh_resident_bytes.restype = [C.c_void_p], C.c_size_t
        hip.kf_engine_workspace_bytes.argtypes, hip.kf_engine_workspace_bytes.restype = [C.c_void_p], C.c_size_t
        hip.kf_engine_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        hip.kf_engine_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        hip.kf_engine_check.argtypes = [C.c_void_p, C.c_void_p]
        hip.kf_engine_set_head.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Weight), C.c_void_p, C.c_void_p, C.c_void_p]
        hip.kf_engine_step_head.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        hip.kf_engine_reset.argtypes = [C.c_void_p, C.c_void_p]
        hip.kf_set_canonical.argtypes = [C.c_void_p, C.c_int]
        hip.kf_get_canonical.argtypes = [C.c_void_p]
        hip.kf_engine_destroy.argtypes = [C.c_void_p]
        # the config-3 training step's sequencer (koifish::GPT2Trainer, host/kf_train.cpp)
        host.kfh_gpt2_create.restype = C.c_void_p
        host.kfh_gpt2_create.argtypes = [C.c_void_p] + [C.c_int] * 7
        host.kfh_gpt2_destroy.restype = None
        host.kfh_gpt2_destroy.argtypes = [C.c_void_p]
        host.kfh_gpt2_n_params.argtypes = [C.c_void_p]
        host.kfh_gpt2_set_param.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_int]
        host.kfh_gpt2_set_block_acts.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        host.kfh_gpt2_set_buffers.argtypes = [C.c_void_p, C.c_void_p]
        host.kfh_gpt2_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        host.kfh_gpt2_backward.argtypes = [C.c_void_p]
        host.kfh_gpt2_update.argtypes = [C.c_void_p, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float, C.c_uint32]
        host.kfh_gpt2_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float, C.c_uint32]
        host.kfh_gpt2_steps_taken.restype = C.c_longlong
        host.kfh_gpt2_steps_taken.argtypes = [C.c_void_p]
        # eight decoders, one per XCD (kf_xengine_*): handles are koifish::XcdReplicas* of the host library
        host.kfh_xr_create.restype = C.c_void_p
        host.kfh_xr_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        host.kfh_xr_destroy.argtypes = [C.c_void_p]
        host.kfh_xr_set_forced.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        host.kfh_xr_set_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        host.kfh_xr_prefill.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        host.kfh_xr_run_steps.argtypes = [C.c_void_p, C.c_int]
        host.kfh_xr_check.argtypes = [C.c_void_p]
        host.kfh_xr_park.argtypes = [C.c_void_p, C.c_int, C.c_int]
        host.kfh_xr_status.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        host.kfh_xr_prefill_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        host.kfh_xr_set_prefill_batch.argtypes = [C.c_void_p, C.c_int]
        host.kfh_xr_set_sampler.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_uint64]
        host.kfh_xr_chat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        host.kfh_xr_chat_each.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        host.kfh_xr_set_steps_per_launch.argtypes = [C.c_void_p, C.c_int]
        host.kfh_xr_get_tokens.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        host.kfh_xr_get_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for f in ("kfh_xr_logits", "kfh_xr_hidden", "kfh_xr_kcache", "kfh_xr_vcache"):
            getattr(host, f).restype = C.c_void_p
            getattr(host, f).argtypes = [C.c_void_p, C.c_int]
        host.kfh_xr_variant.argtypes = [C.c_void_p, C.c_int, C.c_int]
        host.kfh_xr_stamps_enable.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        host.kfh_xr_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        # one sequence over eight tensor-parallel ranks = the eight XCDs of one launch (kf_xengine_create_tp): handles are koifish::XcdTP*
        host.kfh_xtp_create.restype = C.c_void_p
        host.kfh_xtp_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        host.kfh_xtp_destroy.argtypes = [C.c_void_p]
        host.kfh_xtp_set_forced.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        host.kfh_xtp_set_state.argtypes = [C.c_void_p, C.c_int, C.c_int]
        host.kfh_xtp_run_steps.argtypes = [C.c_void_p, C.c_int]
        host.kfh_xtp_check.argtypes = [C.c_void_p]
        host.kfh_xtp_set_steps_per_launch.argtypes = [C.c_void_p, C.c_int]
        host.kfh_xtp_get_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        host.kfh_xtp_vocab.argtypes = [C.c_void_p]
        host.kfh_xtp_variant.argtypes = [C.c_void_p, C.c_int, C.c_int]
        host.kfh_xtp_stamps_enable.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        host.kfh_xtp_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for f in ("kfh_xtp_logits", "kfh_xtp_hidden", "kfh_xtp_kcache", "kfh_xtp_vcache"):
            getattr(host, f).restype = C.c_void_p
            getattr(host, f).argtypes = [C.c_void_p]
        _libs = (hip, host)
    return _libs


def check(rc, what=""):
    if rc != 0:
        hip, _ = load()
        raise KFError("%s failed: code %d: %s" % (what, rc, hip.kf_last_error().decode()))
    return rc
