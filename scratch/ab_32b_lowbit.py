"""Qwen3-32B shapes on one GPU: decode rate with 4-bit, ternary and 1-bit layers; the LDS selector-table forms against the per-weight forms (A/B in one box)."""
import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = [("q4", {}), ("ternary", {"KF_Q2_TAB": "0"}), ("ternary", {"KF_Q2_TAB": "1"}), ("1bit", {"KF_Q1_TAB": "0"}), ("1bit", {"KF_Q1_TAB": "1"})]
for layers, env in runs:
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "qwen3-32b", "--steps", "128", "--warmup", "8", "--cpu-seconds", "0", "--streams", "0",
                        "--layers", layers, "--head", "q4"], env=dict(os.environ, **env), capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-8s %-16s %.1f tok/s, %.3f ms/step, step HBM %.0f GB/s" % (layers, env, j["value"], j["ms_per_step"], j["step_roofline"]["achieved"]))
    except Exception as e:
        print(layers, env, "failed", r.stderr[-400:])
