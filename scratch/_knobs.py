"""Development knobs for the scripts in this directory.  The product library reads no environment variables (round 3): the forms a script wants to compare are
set through the non-ABI hook kfdbg_set_knob of libkf_hip.so.  apply(hip) translates the environment names the round-1/2 scripts used:
    KF_Q4_PERM, KF_Q2_TAB, KF_Q1_TAB, KF_GEMV_WAVES, KF_GEMV_STREAM, KF_GEMM_MIN, KF_G3_TILES, KF_G3_FIRST
Call it once after koifish_amd.load(); a script that times `bench.py` in a child process has to do its A/B inside one process instead."""
import ctypes as C
import os

_MAP = {"KF_Q4_PERM": "q4_perm", "KF_Q2_TAB": "q2_tab", "KF_Q1_TAB": "q1_tab", "KF_GEMV_WAVES": "gemv_waves", "KF_GEMV_STREAM": "gemv_stream", "KF_GEMV_XF2": "gemv_xf2", "KF_GEMM_MIN": "gemm_min", "KF_G3_TILES": "g3_tiles", "KF_G3_FIRST": "g3_first", "KF_RESIDENT_MIN": "resident_min", "KF_ATTN_PAIR_MIN": "attn_pair_min", "KF_G3_WIDE": "g3_wide", "KF_G3_MID_MIN": "g3_mid_min", "KF_ATTN_GQ_SPLIT": "attn_gq_split"}


def apply(hip, env=None):
    env = os.environ if env is None else env
    hip.kfdbg_set_knob.argtypes = [C.c_char_p, C.c_long]
    for k, name in _MAP.items():
        if k in env:
            assert hip.kfdbg_set_knob(name.encode(), int(env[k])) == 0, k
