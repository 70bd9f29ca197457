"""Prometheus-format metrics endpoint next to the gRPC service: per-stage prover time, request
counters and the HBM rate of the commitment stages, so an operator can compare backends in situ.  The reference
has no counterpart; the closest thing is the GetStatus reply (proto/prover/v1/prover.proto:159-195), which the
service also fills.  Plain http.server, text exposition format 0.0.4."""
from __future__ import annotations

import http.server
import threading


def _algorithmic_bytes(tm_key, logn, logb, W):
    """LDE 8N(1+b) per column + Merkle 8MW + 32(2M-1)"""
    N, M = 1 << logn, 1 << (logn + logb)
    if tm_key == "lde+merkle(trace)":
        return 8 * N * (1 + (1 << logb)) * W + 8 * M * W + 32 * (2 * M - 1)
    return None


class Metrics:
    def __init__(self):
        self.lock = threading.Lock()
        self.requests = {}        # (kind, outcome) -> count
        self.stage_seconds = {}   # stage -> [count, sum]
        self.proofs = 0
        self.last_rate = {}       # stage -> GB/s (algorithmic bytes / wall time) of the latest chunk

    def count_request(self, kind, ok):
        with self.lock:
            k = (kind, "ok" if ok else "error")
            self.requests[k] = self.requests.get(k, 0) + 1

    def record_proof(self, timings, logn=None, logb=None, W=None):
        with self.lock:
            self.proofs += 1
            for st, sec in timings.items():
                c = self.stage_seconds.setdefault(st, [0, 0.0])
                c[0] += 1
                c[1] += float(sec)
                if logn is not None and sec > 0:
                    b = _algorithmic_bytes(st, logn, logb, W)
                    if b:
                        self.last_rate[st] = b / sec / 1e9

    def record_stage(self, stage, sec):
        with self.lock:
            c = self.stage_seconds.setdefault(stage, [0, 0.0])
            c[0] += 1
            c[1] += float(sec)

    def render(self):
        esc = lambda s: s.replace("\\", "\\\\").replace('"', '\\"')
        out = ["# HELP zeth_prover_requests_total ProverStream requests by type and outcome",
               "# TYPE zeth_prover_requests_total counter"]
        with self.lock:
            for (kind, oc), n in sorted(self.requests.items()):
                out.append('zeth_prover_requests_total{type="%s",outcome="%s"} %d' % (esc(kind), oc, n))
            out += ["# HELP zeth_prover_chunk_proofs_total chunk STARKs produced", "# TYPE zeth_prover_chunk_proofs_total counter",
                    "zeth_prover_chunk_proofs_total %d" % self.proofs,
                    "# HELP zeth_prover_stage_seconds wall-clock per prover stage", "# TYPE zeth_prover_stage_seconds summary"]
            for st, (n, s) in sorted(self.stage_seconds.items()):
                out.append('zeth_prover_stage_seconds_count{stage="%s"} %d' % (esc(st), n))
                out.append('zeth_prover_stage_seconds_sum{stage="%s"} %.6f' % (esc(st), s))
            out += ["# HELP zeth_prover_stage_hbm_gbps algorithmic bytes / wall time of the latest chunk (HBM-bound stages)",
                    "# TYPE zeth_prover_stage_hbm_gbps gauge"]
            for st, r in sorted(self.last_rate.items()):
                out.append('zeth_prover_stage_hbm_gbps{stage="%s"} %.3f' % (esc(st), r))
        return "\n".join(out) + "\n"

    def serve(self, port=0, host="127.0.0.1"):
        m = self

        class H(http.server.BaseHTTPRequestHandler):
            def do_GET(self):
                if self.path.split("?")[0] != "/metrics":
                    self.send_response(404)
                    self.end_headers()
                    return
                body = m.render().encode()
                self.send_response(200)
                self.send_header("Content-Type", "text/plain; version=0.0.4")
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)

            def log_message(self, *a):
                pass

        httpd = http.server.ThreadingHTTPServer((host, port), H)
        threading.Thread(target=httpd.serve_forever, daemon=True).start()
        return httpd
