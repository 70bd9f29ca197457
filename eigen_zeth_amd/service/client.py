"""Python mirror of eigen-zeth's ProverChannel state machine (src/prover/provider.rs:243-544):
Start -> GenBatchChunks -> GenChunkProof -> GenAggregatedProof(first,last) -> GenFinalProof -> End,
one request in flight, responses matched by oneof type, same field population
(provider.rs:292-310,358-377,422-433,472-483) and the same acceptance predicates
(provider.rs:315-330,381-390,436-451,486-503).  Used as the integration test client and as a tool."""
from __future__ import annotations

import queue
import uuid

import grpc

from . import proto

DEFAULT_AGGREGATOR_ADDR = "479881985774944702531460751064278034642760119942"  # src/commands/run.rs:36-79


class ProverClientError(Exception):
    pass


class ProverChannel:
    def __init__(self, addr="127.0.0.1:50061", chain_id=12345, program_name="evm", curve="BN128",
                 aggregator_addr=DEFAULT_AGGREGATOR_ADDR):
        self.addr, self.chain_id, self.program_name = addr, chain_id, program_name
        self.curve, self.aggregator_addr = curve, aggregator_addr
        self._q = queue.Queue()
        self._channel = grpc.insecure_channel(addr, options=[("grpc.max_receive_message_length", 1 << 30),
                                                             ("grpc.max_send_message_length", 1 << 30)])
        stub = self._channel.stream_stream(proto.METHOD, request_serializer=proto.ProverRequest.SerializeToString,
                                           response_deserializer=proto.ProverResponse.FromString)
        self._resp = stub(iter(self._q.get, None))
        self.trace = []   # (request kind, response kind) pairs, for tests

    def close(self):
        self._q.put(None)
        self._channel.close()

    def _call(self, req):
        self._q.put(req)
        resp = next(self._resp)
        self.trace.append((req.WhichOneof("request_type"), resp.WhichOneof("response_type")))
        return resp

    def get_status(self):
        req = proto.ProverRequest(id=str(uuid.uuid4()))
        req.get_status.SetInParent()
        return self._call(req).get_status

    def execute(self, block_number, batch_id=None, max_retries=3):
        """returns the ProofResult dict eigen-zeth stores (src/db/mod.rs:63-71)"""
        batch_id = batch_id or str(uuid.uuid4())
        # GenChunk
        for _ in range(max_retries):
            req = proto.ProverRequest(id=str(uuid.uuid4()))
            g = req.gen_batch_proof.gen_batch_chunks
            g.batch_id, g.chain_id, g.program_name = batch_id, self.chain_id, self.program_name
            g.batch.block_number.append(block_number)
            r = self._call(req).gen_batch_proof.gen_batch_chunks
            if r.result_code == proto.COMPLETED_OK:
                break
        else:
            raise ProverClientError("gen batch chunk failed: " + r.error_message)
        if len(r.pre_state_root) != 32 or len(r.post_state_root) != 32:
            raise ProverClientError("parse the state root failed")   # provider.rs:323-324
        task_id, chunk_count, batch_data = r.task_id, r.chunk_count, r.batch_data
        pre, post = bytes(r.pre_state_root), bytes(r.post_state_root)
        # GenProof
        for _ in range(max_retries):
            req = proto.ProverRequest(id=str(uuid.uuid4()))
            g = req.gen_batch_proof.gen_chunk_proof
            g.batch_id, g.task_id, g.chunk_count = batch_id, task_id, chunk_count
            g.chain_id, g.program_name, g.batch_data = self.chain_id, self.program_name, batch_data
            r = self._call(req).gen_batch_proof.gen_chunk_proof
            if r.result_code == proto.COMPLETED_OK:
                break
        else:
            raise ProverClientError("gen chunk proof failed: " + r.error_message)
        proofs = r.batch_proof_result.chunk_proofs
        if not proofs:
            raise ProverClientError("empty chunk_proofs")             # provider.rs:384-387 unwraps first()/last()
        first, last = proofs[0].proof, proofs[len(proofs) - 1].proof
        # Aggregate
        for _ in range(max_retries):
            req = proto.ProverRequest(id=str(uuid.uuid4()))
            g = req.gen_aggregated_proof
            g.batch_id, g.recursive_proof_1, g.recursive_proof_2 = batch_id, first, last
            r = self._call(req).gen_aggregated_proof
            if r.result_code == proto.COMPLETED_OK:
                break
        else:
            raise ProverClientError("gen aggregated proof failed: " + r.error_message)
        recursive = r.result_string
        # Final
        for _ in range(max_retries):
            req = proto.ProverRequest(id=str(uuid.uuid4()))
            g = req.gen_final_proof
            g.batch_id, g.recursive_proof, g.curve_name, g.aggregator_addr = batch_id, recursive, self.curve, self.aggregator_addr
            resp = self._call(req).gen_final_proof
            if resp.result_code == proto.COMPLETED_OK and resp.HasField("final_proof"):
                break
        else:
            raise ProverClientError("gen final proof failed: " + resp.error_message)
        return {"block_number": block_number, "proof": resp.final_proof.proof, "public_input": resp.final_proof.public_input,
                "pre_state_root": list(pre), "post_state_root": list(post), "batch_id": batch_id,
                "chunk_proofs": [p.proof for p in proofs], "aggregated": recursive}
