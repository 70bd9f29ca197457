"""prover.v1 message classes built from a hand-written FileDescriptorProto.

The image has no protoc / grpc_tools, so the schema of the reference's
proto/prover/v1/prover.proto (195 lines) is restated here field by field -- names, numbers, types and
oneofs exactly as in that file (note ChunkProof.proof_key = 3 / proof = 2, prover.proto:107-111, and
ProverStatus starting at field 2, :176-190).  tests/test_service.py checks the numbering against the
.proto text when /root/reference is present and against a frozen table otherwise."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_T = descriptor_pb2.FieldDescriptorProto
STRING, UINT64, BYTES, ENUM, MESSAGE = _T.TYPE_STRING, _T.TYPE_UINT64, _T.TYPE_BYTES, _T.TYPE_ENUM, _T.TYPE_MESSAGE

# message -> [(name, number, type, type_name|None, repeated, oneof|None)]
SCHEMA = {
    "Version": [("v0_0_1", 1, STRING, None, False, None)],
    "ProverRequest": [("id", 1, STRING, None, False, None),
                      ("get_status", 2, MESSAGE, "GetStatusRequest", False, "request_type"),
                      ("gen_batch_proof", 3, MESSAGE, "GenBatchProofRequest", False, "request_type"),
                      ("gen_aggregated_proof", 4, MESSAGE, "GenAggregatedProofRequest", False, "request_type"),
                      ("gen_final_proof", 5, MESSAGE, "GenFinalProofRequest", False, "request_type")],
    "ProverResponse": [("id", 1, STRING, None, False, None),
                       ("get_status", 2, MESSAGE, "GetStatusResponse", False, "response_type"),
                       ("gen_batch_proof", 3, MESSAGE, "GenBatchProofResponse", False, "response_type"),
                       ("gen_aggregated_proof", 4, MESSAGE, "GenAggregatedProofResponse", False, "response_type"),
                       ("gen_final_proof", 5, MESSAGE, "GenFinalProofResponse", False, "response_type")],
    "GenBatchProofRequest": [("gen_batch_chunks", 1, MESSAGE, "GenBatchChunks", False, "step"),
                             ("gen_chunk_proof", 2, MESSAGE, "GenChunkProof", False, "step")],
    "GenBatchChunks": [("batch_id", 1, STRING, None, False, None), ("batch", 2, MESSAGE, "Batch", False, None),
                       ("chain_id", 3, UINT64, None, False, None), ("program_name", 4, STRING, None, False, None)],
    "GenChunkProof": [("batch_id", 1, STRING, None, False, None), ("task_id", 2, STRING, None, False, None),
                      ("chunk_count", 3, UINT64, None, False, None), ("chain_id", 4, UINT64, None, False, None),
                      ("program_name", 5, STRING, None, False, None), ("batch_data", 6, STRING, None, False, None)],
    "Batch": [("block_number", 1, UINT64, None, True, None)],
    "GenBatchProofResponse": [("gen_batch_chunks", 1, MESSAGE, "GenBatchChunksResult", False, "step"),
                              ("gen_chunk_proof", 2, MESSAGE, "GenChunkProofResult", False, "step")],
    "GenBatchChunksResult": [("batch_id", 1, STRING, None, False, None), ("task_id", 2, STRING, None, False, None),
                             ("result_code", 3, ENUM, "ProofResultCode", False, None),
                             ("chunk_count", 4, UINT64, None, False, None), ("batch_data", 5, STRING, None, False, None),
                             ("pre_state_root", 6, BYTES, None, False, None), ("post_state_root", 7, BYTES, None, False, None),
                             ("error_message", 8, STRING, None, False, None)],
    "GenChunkProofResult": [("batch_id", 1, STRING, None, False, None), ("task_id", 2, STRING, None, False, None),
                            ("result_code", 3, ENUM, "ProofResultCode", False, None),
                            ("batch_proof_result", 4, MESSAGE, "BatchProofResult", False, None),
                            ("error_message", 5, STRING, None, False, None)],
    "BatchProofResult": [("task_id", 1, STRING, None, False, None), ("chunk_proofs", 2, MESSAGE, "ChunkProof", True, None)],
    "ChunkProof": [("chunk_id", 1, UINT64, None, False, None), ("proof_key", 3, STRING, None, False, None),
                   ("proof", 2, STRING, None, False, None)],
    "GenAggregatedProofRequest": [("batch_id", 1, STRING, None, False, None), ("recursive_proof_1", 2, STRING, None, False, None),
                                  ("recursive_proof_2", 3, STRING, None, False, None)],
    "GenAggregatedProofResponse": [("batch_id", 1, STRING, None, False, None),
                                   ("result_code", 2, ENUM, "ProofResultCode", False, None),
                                   ("result_string", 3, STRING, None, False, None), ("error_message", 4, STRING, None, False, None)],
    "GenFinalProofRequest": [("batch_id", 1, STRING, None, False, None), ("recursive_proof", 2, STRING, None, False, None),
                             ("curve_name", 3, STRING, None, False, None), ("aggregator_addr", 4, STRING, None, False, None)],
    "GenFinalProofResponse": [("batch_id", 1, STRING, None, False, None), ("result_code", 2, ENUM, "ProofResultCode", False, None),
                              ("result_string", 3, STRING, None, False, None), ("final_proof", 4, MESSAGE, "FinalProof", False, None),
                              ("error_message", 5, STRING, None, False, None)],
    "FinalProof": [("proof", 1, STRING, None, False, None), ("public_input", 2, STRING, None, False, None)],
    "GetStatusRequest": [],
    "GetStatusResponse": [("id", 1, STRING, None, False, None), ("result_code", 2, ENUM, "GetStatusResultCode", False, None),
                          ("status", 3, ENUM, "GetStatusResponse.Status", False, None),
                          ("prover_status", 4, MESSAGE, "ProverStatus", False, None), ("error_message", 5, STRING, None, False, None)],
    "ProverStatus": [("last_computed_request_id", 2, STRING, None, False, None), ("last_computed_end_time", 3, UINT64, None, False, None),
                     ("current_computing_request_id", 4, STRING, None, False, None),
                     ("current_computing_start_time", 5, UINT64, None, False, None), ("version_proto", 6, STRING, None, False, None),
                     ("version_server", 7, STRING, None, False, None), ("pending_request_queue_ids", 8, STRING, None, True, None),
                     ("prover_name", 9, STRING, None, False, None), ("prover_id", 10, STRING, None, False, None),
                     ("number_of_cores", 11, UINT64, None, False, None), ("total_memory", 12, UINT64, None, False, None),
                     ("free_memory", 13, UINT64, None, False, None), ("fork_id", 14, UINT64, None, False, None)],
}
ENUMS = {"ProofResultCode": [("COMPLETED_OK", 0), ("COMPLETED_ERROR", 1)],
         "GetStatusResultCode": [("OK", 0), ("FAIL", 1)]}
NESTED_ENUMS = {"GetStatusResponse": {"Status": [("STATUS_UNSPECIFIED", 0), ("STATUS_BOOTING", 1), ("STATUS_COMPUTING", 2),
                                                 ("STATUS_IDLE", 3), ("STATUS_HALT", 4)]}}

PACKAGE = "prover.v1"
SERVICE = "prover.v1.ProverService"
METHOD = "/prover.v1.ProverService/ProverStream"


def _build():
    fd = descriptor_pb2.FileDescriptorProto(name="prover/v1/prover.proto", package=PACKAGE, syntax="proto3")
    for ename, vals in ENUMS.items():
        e = fd.enum_type.add(name=ename)
        for n, v in vals:
            e.value.add(name=n, number=v)
    for mname, fields in SCHEMA.items():
        m = fd.message_type.add(name=mname)
        for ename, vals in NESTED_ENUMS.get(mname, {}).items():
            e = m.enum_type.add(name=ename)
            for n, v in vals:
                e.value.add(name=n, number=v)
        oneofs = []
        for (fname, num, typ, tname, rep, oneof) in fields:
            f = m.field.add(name=fname, number=num, type=typ,
                            label=_T.LABEL_REPEATED if rep else _T.LABEL_OPTIONAL)
            if tname:
                f.type_name = "." + PACKAGE + "." + tname
            if oneof:
                if oneof not in oneofs:
                    oneofs.append(oneof)
                    m.oneof_decl.add(name=oneof)
                f.oneof_index = oneofs.index(oneof)
    svc = fd.service.add(name="ProverService")
    svc.method.add(name="ProverStream", input_type="." + PACKAGE + ".ProverRequest",
                   output_type="." + PACKAGE + ".ProverResponse", client_streaming=True, server_streaming=True)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return pool


_POOL = _build()


def cls(name):
    return message_factory.GetMessageClass(_POOL.FindMessageTypeByName(PACKAGE + "." + name))


ProverRequest, ProverResponse = cls("ProverRequest"), cls("ProverResponse")
GetStatusResponse, ProverStatus = cls("GetStatusResponse"), cls("ProverStatus")
COMPLETED_OK, COMPLETED_ERROR = 0, 1
STATUS_COMPUTING, STATUS_IDLE = 2, 3
