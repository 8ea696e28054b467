"""Full-path replay: first-party server (own process, engine core in a second process) + the preserved
harness (``traffic_generator/main.py`` run as a script from its own directory, contract I1) + analysis.

This is the measurement BASELINE.md §5 asks for: BurstGPT replay → ``log.json`` (reference schema) →
goodput / TTFT / TPOT.  ``bench.py`` measures the same engine in-process; this script adds the HTTP,
NDJSON and harness costs on top.

    python benchmarks/replay_http.py --model llama3-8b --qps-scale 20 --max-trace 1500 --max-tokens 200
    python benchmarks/replay_http.py --model opt-125m --device cpu --trace ../data/trace1.csv   # config 1
"""
import argparse
import json
import os
import subprocess
import sys
import time
import urllib.request

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "traffic_generator"))
import analyze  # noqa: E402


def wait_health(url, timeout, server=None):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if server is not None and server.poll() is not None:
            raise RuntimeError(f"the server exited with code {server.returncode} before it became healthy")
        try:
            with urllib.request.urlopen(url + "/health", timeout=2) as r:
                if r.status == 200:
                    return
        except Exception:
            time.sleep(0.5)
    raise TimeoutError("server did not come up")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--device", default="auto")
    ap.add_argument("--port", type=int, default=11434)
    ap.add_argument("--trace", default="../data/BurstGPT_1.csv", help="relative to traffic_generator/")
    ap.add_argument("--data", default="../data/conversations.json")
    ap.add_argument("--qps-scale", type=float, default=1.0, help="divide trace timestamps by this factor")
    ap.add_argument("--max-trace", type=int, default=1000)
    ap.add_argument("--max-tokens", type=int, default=200)
    ap.add_argument("--slo", default="llama3-8b-relaxed")
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "gpurun_out", "replay"))
    ap.add_argument("--router-replicas", type=int, default=0, help="put the DP router in front of N replicas")
    ap.add_argument("--tp", type=int, default=1, help="tensor-parallel degree of the served replica (api_server --tp)")
    ap.add_argument("--ep", type=int, default=1, help="expert-parallel degree (MoE models; 1 or tp)")
    ap.add_argument("--kv-dtype", default="auto", choices=["auto", "fp8"])
    ap.add_argument("--server-args", default="", help="extra args for api_server, space separated")
    a = ap.parse_args()
    a.out_dir = os.path.abspath(a.out_dir)
    os.makedirs(a.out_dir, exist_ok=True)
    tg_dir = os.path.join(ROOT, "traffic_generator")

    trace_path = a.trace
    if a.qps_scale != 1.0:
        df = pd.read_csv(os.path.join(tg_dir, a.trace), nrows=a.max_trace)
        df["Timestamp"] = df["Timestamp"] / a.qps_scale
        trace_path = os.path.join(a.out_dir, "trace_scaled.csv")
        df.to_csv(trace_path, index=False)

    base = f"http://127.0.0.1:{a.port}"
    extra = ["--tp", str(a.tp), "--ep", str(a.ep), "--kv-dtype", a.kv_dtype] + (a.server_args.split() if a.server_args else [])
    if a.router_replicas:
        cmd = [sys.executable, "-m", "dli_amd.router.router", "--spawn", str(a.router_replicas), "--port", str(a.port),
               "--host", "127.0.0.1", "--model", a.model, "--device", a.device] + (["--cpu"] if a.device == "cpu" else []) + extra
    else:
        cmd = [sys.executable, "-m", "dli_amd.server.api_server", "--model", a.model, "--device", a.device, "--host",
               "127.0.0.1", "--port", str(a.port), "--max-model-len", "2112"] + extra
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    server = subprocess.Popen(cmd, env=env, cwd=ROOT)
    try:
        wait_health(base, 900, server)
        log_path = os.path.join(a.out_dir, "log.json")
        side_path = os.path.join(a.out_dir, "sidecar.json")
        t0 = time.time()
        harness = subprocess.run([sys.executable, "main.py", "--url", base + "/api/generate", "--trace-path", trace_path,
                                  "--data-path", a.data, "--max-trace", str(a.max_trace), "--max-tokens", str(a.max_tokens),
                                  "--log-path", log_path, "--sidecar-path", side_path, "--connector-limit", "0",
                                  "--timeout", "3600"], cwd=tg_dir, stdout=subprocess.DEVNULL)
        wall = time.time() - t0
        assert harness.returncode == 0, "harness failed"
        metrics = json.load(open(log_path))
        n_out = analyze.n_out_from_sidecar(json.load(open(side_path)))
        ttft, tpot = analyze.SLOS[a.slo]
        summary = analyze.summarize(metrics, n_out, slo_ttft=ttft, slo_tpot=tpot)
        summary.update(model=a.model, qps_scale=a.qps_scale, max_tokens=a.max_tokens, harness_wall_s=wall,
                       router_replicas=a.router_replicas, tp=a.tp, ep=a.ep, kv_dtype=a.kv_dtype)
        try:
            with urllib.request.urlopen(base + ("/router/stats" if a.router_replicas else "/stats"), timeout=5) as r:
                summary["server_stats"] = json.loads(r.read())
        except Exception:
            pass
        with open(os.path.join(a.out_dir, "summary.json"), "w") as fh:
            json.dump(summary, fh, indent=1)
        st = summary.get("server_stats") or {}
        brief = {k: v for k, v in summary.items() if k != "server_stats"}
        if "comm" in st:                                     # TP replica: what it ran on, like the bench line says
            brief["comm"] = {k: st["comm"].get(k) for k in ("tp", "ep", "device_backend", "custom_ar_enabled", "pipelined_steps",
                                                            "rccl_ranks_joined", "ranks_share_gpu")}
            brief["selftest"] = st.get("selftest")
            brief["graph_steps"], brief["eager_steps"] = st.get("graph_steps"), st.get("eager_steps")
        print(json.dumps(brief))
    finally:
        server.terminate()
        try:
            server.wait(20)
        except subprocess.TimeoutExpired:
            server.kill()


if __name__ == "__main__":
    main()
