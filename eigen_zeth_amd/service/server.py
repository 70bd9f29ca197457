"""prover.v1.ProverService server (proto/prover/v1/prover.proto:9-11).

One bidi stream per client connection; the client keeps exactly one request in flight and matches
responses by oneof type, ignoring `id` (src/prover/provider.rs:649-667), so a stream is served by one
sequential loop.  Application errors are result_code = COMPLETED_ERROR + error_message (the client
logs and retries, provider.rs:332-335); an OK response always carries 32-byte state roots, a
non-empty chunk_proofs list and a final_proof (provider.rs:323-324,384-387,492-509)."""
from __future__ import annotations

import base64
import os
import threading
import time
from concurrent import futures

import grpc

from . import proto
from .engine import Engine, EngineConfig
from .store import BatchStore

VERSION = "zeth-prover-mi355x/0.1"


class ProverService:
    def __init__(self, engine, store, metrics=None):
        self.engine, self.store = engine, store
        self.metrics = metrics
        if metrics is not None:
            engine.metrics = metrics
        self.last_id, self.last_end, self.cur_start = "", 0, 0
        # gRPC serves every stream on its own worker thread.  load -> compute -> save of a request is atomic PER BATCH
        # (`_batch_locks`): a replay that arrives on a new stream while the old handler is still proving the same batch
        # waits and then finds the stored result (in-flight dedup per batch: nothing is computed twice); requests of other
        # batches are not held up here -- the engine serialises the use of its GPU contexts itself (Engine._serial).
        # `_stat` guards what GetStatus reads: the set of request ids being computed (busy = non-empty; one handler
        # finishing does not clear another one's entry) and the last finished request.
        self._stat = threading.Lock()
        self._in_flight = set()
        self._batch_locks = {}

    # ---- the stream
    def prover_stream(self, request_iterator, context):
        for req in request_iterator:
            kind = req.WhichOneof("request_type")
            resp = proto.ProverResponse(id=req.id)
            if kind == "get_status":      # never waits for a running proof
                self._status(resp)
                yield resp
                continue
            token = object()                    # request ids are the client's: two streams may replay the same one
            bkey = _batch_of(req, kind)
            with self._stat:
                self._in_flight.add(token)
                self.cur_start = int(time.time())
                # [lock, users]: a handler counts itself in BEFORE it leaves _stat, so a lock somebody is about to take is never
                # pruned (a fresh lock for the same batch would let two handlers run load -> compute -> save side by side)
                ent = self._batch_locks.setdefault(bkey, [threading.Lock(), 0])
                ent[1] += 1
                lock = ent[0]
            lock.acquire()
            try:
                if kind == "gen_batch_proof":
                    step = req.gen_batch_proof.WhichOneof("step")
                    if step == "gen_batch_chunks":
                        self._batch_chunks(req.gen_batch_proof.gen_batch_chunks, resp.gen_batch_proof.gen_batch_chunks)
                    elif step == "gen_chunk_proof":
                        self._chunk_proof(req.gen_batch_proof.gen_chunk_proof, resp.gen_batch_proof.gen_chunk_proof)
                    else:
                        r = resp.gen_batch_proof.gen_batch_chunks
                        r.result_code, r.error_message = proto.COMPLETED_ERROR, "GenBatchProofRequest without a step"
                elif kind == "gen_aggregated_proof":
                    self._aggregate(req.gen_aggregated_proof, resp.gen_aggregated_proof)
                elif kind == "gen_final_proof":
                    self._final(req.gen_final_proof, resp.gen_final_proof)
                else:
                    self._status(resp, error="request without request_type")
            finally:
                lock.release()
                with self._stat:
                    self._in_flight.discard(token)
                    self.last_id, self.last_end = req.id, int(time.time())
                    ent[1] -= 1
                    if len(self._batch_locks) > 256:     # locks of finished batches: nobody holds or awaits them (users == 0)
                        for k in [k for k, e in self._batch_locks.items() if e[1] == 0][:128]:
                            del self._batch_locks[k]
            if self.metrics is not None:
                self.metrics.count_request(*_outcome(resp))
            yield resp

    # ---- handlers (idempotent per batch_id)
    def _batch_chunks(self, q, r):
        r.batch_id = q.batch_id
        try:
            rec = self.store.load(q.batch_id)
            key = [list(q.batch.block_number), int(q.chain_id), q.program_name]
            if rec.get("chunks_key") != key:
                res = self.engine.gen_batch_chunks(q.batch_id, list(q.batch.block_number), q.chain_id, q.program_name)
                rec = {"chunks_key": key,
                       "chunks": {"task_id": res["task_id"], "chunk_count": res["chunk_count"], "batch_data": res["batch_data"],
                                  "pre": base64.b64encode(res["pre_state_root"]).decode(),
                                  "post": base64.b64encode(res["post_state_root"]).decode()}}
                self.store.save(q.batch_id, rec)
            c = rec["chunks"]
            pre, post = base64.b64decode(c["pre"]), base64.b64decode(c["post"])
            assert len(pre) == 32 and len(post) == 32
            r.task_id, r.chunk_count, r.batch_data = c["task_id"], c["chunk_count"], c["batch_data"]
            r.pre_state_root, r.post_state_root = pre, post
            r.result_code = proto.COMPLETED_OK
        except Exception as e:  # application error -> the client retries the same request
            r.result_code, r.error_message = proto.COMPLETED_ERROR, "%s: %s" % (type(e).__name__, e)

    def _chunk_proof(self, q, r):
        r.batch_id, r.task_id = q.batch_id, q.task_id
        try:
            rec = self.store.load(q.batch_id)
            key = [q.task_id, int(q.chunk_count), q.batch_data]
            if rec.get("proofs_key") != key:
                proofs = self.engine.gen_chunk_proofs(q.batch_id, q.task_id, int(q.chunk_count), q.batch_data)
                if not proofs:
                    raise ValueError("no chunks to prove")
                rec["proofs_key"], rec["proofs"] = key, proofs
                rec["timings"] = {k: v for k, v in self.engine.stage_timings.items() if k.startswith(q.task_id + "/")}
                self.store.save(q.batch_id, rec)
            r.batch_proof_result.task_id = q.task_id
            for p in rec["proofs"]:
                cp = r.batch_proof_result.chunk_proofs.add()
                cp.chunk_id, cp.proof_key, cp.proof = p["chunk_id"], p["proof_key"], p["proof"]
            r.result_code = proto.COMPLETED_OK
        except Exception as e:
            r.ClearField("batch_proof_result")
            r.result_code, r.error_message = proto.COMPLETED_ERROR, "%s: %s" % (type(e).__name__, e)

    def _aggregate(self, q, r):
        r.batch_id = q.batch_id
        try:
            rec = self.store.load(q.batch_id)
            key = [self.engine._digest(q.recursive_proof_1), self.engine._digest(q.recursive_proof_2)]
            if rec.get("agg_key") != key:
                rec["agg_key"], rec["agg"] = key, self.engine.aggregate(q.batch_id, q.recursive_proof_1, q.recursive_proof_2)
                self.store.save(q.batch_id, rec)
            r.result_string, r.result_code = rec["agg"], proto.COMPLETED_OK
        except Exception as e:
            r.result_code, r.error_message = proto.COMPLETED_ERROR, "%s: %s" % (type(e).__name__, e)

    def _final(self, q, r):
        r.batch_id = q.batch_id
        try:
            rec = self.store.load(q.batch_id)
            key = [self.engine._digest(q.recursive_proof), q.curve_name, q.aggregator_addr]
            if rec.get("final_key") != key:
                proof, pub = self.engine.final(q.batch_id, q.recursive_proof, q.curve_name, q.aggregator_addr)
                rec["final_key"], rec["final"] = key, {"proof": proof, "public_input": pub}
                self.store.save(q.batch_id, rec)
            r.final_proof.proof, r.final_proof.public_input = rec["final"]["proof"], rec["final"]["public_input"]
            r.result_string, r.result_code = "ok", proto.COMPLETED_OK
        except Exception as e:
            r.ClearField("final_proof")
            r.result_code, r.error_message = proto.COMPLETED_ERROR, "%s: %s" % (type(e).__name__, e)

    def _status(self, resp, error=None):
        s = resp.get_status
        s.id = "zeth-prover-mi355x"
        s.result_code = 1 if error else 0
        with self._stat:
            busy, last_id, last_end = bool(self._in_flight), self.last_id, self.last_end
        s.status = getattr(proto, "STATUS_COMPUTING", proto.STATUS_IDLE) if busy else proto.STATUS_IDLE
        if error:
            s.error_message = error
        ps = s.prover_status
        ps.last_computed_request_id, ps.last_computed_end_time = last_id, last_end
        ps.version_proto, ps.version_server = "v0_0_1", VERSION
        ps.prover_name, ps.prover_id = "zeth-prover-mi355x", str(os.getpid())
        ps.number_of_cores = os.cpu_count() or 0
        try:
            import psutil
            vm = psutil.virtual_memory()
            ps.total_memory, ps.free_memory = vm.total, vm.available
        except Exception:
            pass
        ps.fork_id = 0


def _batch_of(req, kind):
    """the batch a request belongs to (the unit of in-flight dedup)"""
    if kind == "gen_batch_proof":
        step = req.gen_batch_proof.WhichOneof("step")
        return getattr(req.gen_batch_proof, step).batch_id if step else ""
    if kind in ("gen_aggregated_proof", "gen_final_proof"):
        return getattr(req, kind).batch_id
    return ""


def _outcome(resp):
    """(request kind, ok?) of a response, for the request counters"""
    kind = resp.WhichOneof("response_type")
    if kind == "gen_batch_proof":
        step = resp.gen_batch_proof.WhichOneof("step")
        return step or kind, getattr(resp.gen_batch_proof, step).result_code == proto.COMPLETED_OK if step else False
    if kind in ("gen_aggregated_proof", "gen_final_proof"):
        return kind, getattr(resp, kind).result_code == proto.COMPLETED_OK
    return kind or "unknown", True


def make_server(service, port=50061, host="127.0.0.1", max_workers=16):
    """max_workers: one gRPC worker per open stream; replays of a running request wait on their batch's lock, so the pool is
    sized well above the one-stream-per-client norm -- a GetStatus on a new stream must still find a free worker"""
    handler = grpc.method_handlers_generic_handler(proto.SERVICE, {
        "ProverStream": grpc.stream_stream_rpc_method_handler(
            service.prover_stream, request_deserializer=proto.ProverRequest.FromString,
            response_serializer=proto.ProverResponse.SerializeToString)})
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=max_workers),
                         options=[("grpc.max_send_message_length", 1 << 30), ("grpc.max_receive_message_length", 1 << 30)])
    server.add_generic_rpc_handlers((handler,))
    bound = server.add_insecure_port("%s:%d" % (host, port))
    return server, bound


def default_backend_factory(device=0):
    def make(hash_mode="gl"):
        """hash_mode "bn128": the backend of the final STARK (16-ary Poseidon-BN254 trees, transcript over F_r)"""
        from ..stark.backend_hip import HipBackend
        return HipBackend(device, hash_mode=hash_mode)   # raises without the HIP library / an MI355X: no CPU fallback
    return make


def serve(port=50061, host="127.0.0.1", state_dir="prover_state", config=None, device=0, metrics_port=None, devices=None, prewarm=False):
    """devices: GPUs the chunk proofs of a batch are spread over (default: just `device`).  prewarm: build keys, plans, tables and kernels
    before the port opens (Engine.prewarm: one synthetic batch proven end to end), so that the first block costs what every block costs"""
    devs = list(devices) if devices else [device]
    engine = Engine([default_backend_factory(d) for d in devs], config or EngineConfig())
    engine.be  # fail at start-up, not at the first request, when no GPU is present
    if prewarm:
        engine.prewarm_report = engine.prewarm()
    metrics = None
    if metrics_port is not None:
        from .metrics import Metrics
        metrics = Metrics()
        metrics.serve(metrics_port, host)
    service = ProverService(engine, BatchStore(state_dir), metrics)
    server, bound = make_server(service, port, host)
    server.start()
    return server, bound
