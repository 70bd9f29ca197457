"""What the four ProverStream requests compute (prover.proto:49-66,115-148).

GenBatchChunks : synthetic executor -- the real zkVM executor/EVM program is not obtainable offline
                 (SURVEY.md par.7); every block becomes `chunks_per_block` chunks of a synthetic AIR.
GenChunkProof  : one real STARK per chunk on the GPU backend (eigen_zeth_amd/stark).
GenAggregated  : a STARK over the Merkle-verifier AIR (stark/verifier_air.py): its witness is the verification trace of the
                 two chunk proofs' query openings, its public inputs their roots and query indices.
GenFinalProof  : a final STARK in BN128-hash mode over the same verifier AIR applied to the aggregated proof's STARK, then
                 a Groth16 proof over BN254 (service/groth16.py: QAP step and MSMs on the GPU) of a circuit that verifies the HASHING of
                 that STARK's verifier (service/wrap_circuit.py: leaf sponges and 16-ary paths at every query, one public input), in
                 the exact JSON grammar eigen-zeth parses (src/settlement/ethereum/mod.rs:445-481) -- under a locally generated,
                 seeded key: a ceremony does not exist offline.
"""
from __future__ import annotations

import contextlib
import gc
import hashlib
from concurrent.futures import ThreadPoolExecutor
import json
import queue
import threading
import os
import struct
import time

from ..stark import air as AIR
from ..stark import prover as PR
from ..stark import verifier_air as VA
from .. import native
from . import bn254
from . import groth16
from . import statement
from . import wrap_circuit as WC
from . import wrap_arith as WA
from ..stark import verifier as SV


@contextlib.contextmanager
def _no_cyclic_gc():
    """parsing a 1-2 MB proof text makes a few hundred thousand small objects, none of them in a cycle: the generational collector
    would walk them (and everything older) several times per request -- tens of milliseconds of a 100 ms request"""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


AGGREGATED_STATEMENT = ("for every query slot, inner proof and committed tree there are values that hash, as a leaf and up a path "
                        "along the bits of the public index, to the public root; they give the DEEP quotient at the query point, "
                        "every FRI layer's opened coset interpolates the value the layer before claims and folds to the next, the "
                        "last fold is the public final-layer value; the Fiat-Shamir sponge of every inner proof absorbs the public "
                        "blocks and yields the public rates, its grinding hash the public digest (what is left to the checker per "
                        "inner proof: the out-of-domain identity, the final layer's degree, reading the transcript -- on 'inner', "
                        "which holds no openings)")


def aggregated_head(batch_id, shape, level, vair_digest):
    """the text an aggregated proof starts with, up to (not including) its "inner" member -- one definition for the engine and for the files a
    compiled host works from (tools/export_recursion_shape.py -> host/aggregate.cpp)"""
    return json.dumps({"kind": "aggregated", "version": 1, "batch_id": batch_id, "statement": AGGREGATED_STATEMENT, "shape": shape.to_dict(),
                       "level": level, "slots": shape.n_slots(), "verifier_air_digest": vair_digest}, separators=(",", ":"))[:-1]


class EngineConfig:
    """Default chunk-STARK security: 80 queries x blow-up 2 (1 bit each) + 20 bits of proof-of-work grinding = 100 bits
    CONJECTURED (ethSTARK-style, stark/prover.py StarkParams.security_bits; the provable FRI bound at these parameters is about
    half of that); every value is bound into the proof's transcript."""

    def __init__(self, air="chunk64", logn=12, logb=1, chunks_per_block=1, n_queries=80, fri_logf=3, fri_final_log=5,
                 crs_dir=None, l2_addr=None, txs_per_chunk=64,
                 witness_threads=8, prover_streams=8, pow_bits=20,
                 final_air="chunk16", final_logn=10, final_logb=2, final_queries=50, native_prover=True,
                 agg_queries=50, agg_pow_bits=0, aggregate_all_chunks=False, groth16_seed=None, witness="device",
                 speculate_recursion=False, verify_before_wrap=True, final_ranks=1, final_devices=None):
        self.air, self.logn, self.logb = air, logn, logb
        # GenFinalProof: check natively, before the final STARK and the wrap are made, what no query of the aggregation STARK covers (its
        # constraint identity at the out-of-domain point, its final layer, its grinding: stark/verifier.py).  Off only for timing experiments.
        self.verify_before_wrap = verify_before_wrap
        # GenAggregatedProof names two proofs -- the client sends the first and the last chunk proof of a batch
        # (src/prover/provider.rs:385-388).  False: exactly those two are verified (the wire contract taken literally).  True: when
        # they ARE the first and last chunk proof of a batch this engine has just proven, every chunk proof of the batch is verified
        # by the one aggregation STARK (n_proofs = chunk count: the final proof then covers the whole batch, at ~ chunk count / 2
        # times the aggregation cost).
        self.aggregate_all_chunks = aggregate_all_chunks
        # Blinding of the final Groth16 proof.  None (the service default): fresh (r, s) from the OS per proof -- zero knowledge.  A string:
        # (r, s) = SHA-256(seed | final-STARK digest | recursive proof | aggregator address | "r" / "s") -- every run, on any backend, writes
        # the SAME proof.json for the same request ("identical proof/proof.json" of BASELINE.json made testable).  A test /
        # reproduction mode: blinding that anybody can recompute hides nothing.
        self.groth16_seed = groth16_seed
        self.chunks_per_block, self.n_queries, self.pow_bits = chunks_per_block, n_queries, pow_bits
        self.fri_logf, self.fri_final_log = fri_logf, fri_final_log
        self.crs_dir = crs_dir          # (unused since round 4: wrap keys are made in memory per engine; kept for callers that pass it)
        self.l2_addr, self.txs_per_chunk = l2_addr, txs_per_chunk   # optional block-input fetcher
        self.witness_threads = witness_threads
        # where the synthetic witnesses (the stand-in for the zkVM executor) are made.  "host": zp_synth_trace_bound on witness_threads
        # threads, uploaded from page-locked memory.  "device": the same traces generated in HBM (csrc/synth.hip) on backends that can --
        # the recurrences of the whole batch are walked once, one wave per chunk, while the first witness_threads chunks still come from
        # the host generator and are being proven; every later chunk is filled from its checkpoints in a few milliseconds.
        assert witness in ("host", "device")
        self.witness = witness
        # The reference's client always follows GenChunkProof with GenAggregatedProof(first chunk proof, last chunk proof) and GenFinalProof of
        # the result (src/prover/provider.rs:385-388, 422-503).  True: a batch of >= 4 chunks proves its first and last chunk FIRST and, on a
        # backend of its own, makes that aggregated proof and its final STARK while the other chunks are still being proven; the two requests
        # that follow are then answered from what is already there (same texts, byte for byte: both proofs are deterministic) and only the
        # Groth16 wrap -- which needs the request's aggregator address and fresh blinding -- is left.  Any other request is computed as before.
        # OFF by default: measured on one MI355X (profiles/r4_speculation_ab.txt) the two recursion STARKs are 0.15 s of GPU-saturating work
        # (Merkle commitments), not idle latency -- made during the chunk proofs they slow those by what they take (16 chunks of 2^20 rows:
        # 0.336 + 0.229 s in sequence, 0.500 + 0.043 s overlapped; BASELINE configs[4]: 7.51 against 7.45 s).
        self.speculate_recursion = speculate_recursion
        # final_ranks > 1 (a power of two): the final STARK is ONE proof over that many ranks of this process -- zp_stark_prove_sharded_bn128
        # on an in-process communicator, rank r on final_devices[r] (default: all on the engine's GPU, which only rehearses the path).  Same
        # text and wrap as with one rank (tests/test_gpu_verifier_air.py).  For a node whose GPUs would otherwise idle during GenFinalProof.
        if not isinstance(final_ranks, int) or isinstance(final_ranks, bool) or not 1 <= final_ranks <= 64 or final_ranks & (final_ranks - 1):
            raise ValueError("final_ranks must be a power of two between 1 and 64, not %r" % (final_ranks,))
        if final_devices is not None and (len(final_devices) != final_ranks or any(not isinstance(d, int) or d < 0 for d in final_devices)):
            raise ValueError("final_devices must name one GPU id per rank of final_ranks (%d), not %r" % (final_ranks, final_devices))
        self.final_ranks, self.final_devices = final_ranks, final_devices
        self.prover_streams = prover_streams   # chunk proofs in flight on one GPU (each on its own ctx / stream); 8 measured best (profiles/r2_streams_sweep.txt)
        # the final STARK (BN128-hash mode, no grinding: 50 queries x blow-up 4 = 100 bits conjectured)
        self.final_air, self.final_logn, self.final_logb, self.final_queries = final_air, final_logn, final_logb, final_queries
        self.native_prover = native_prover     # chunk proofs through zp_stark_prove (False: the Python orchestration, per-stage timings)
        # the aggregation STARK (Merkle-verifier AIR, degree-4 constraints: blow-up 4): 50 queries x 2 bits = 100 bits conjectured
        self.agg_queries, self.agg_pow_bits = agg_queries, agg_pow_bits


def recursion_airs_for(cfg, root32, shift, rc, mds, n_proofs=2, logn=None):
    """(the verifier AIR an aggregation STARK over `n_proofs` chunk proofs of 2^logn rows proves, the one the final STARK over THAT proof proves) under
    the configuration `cfg` -- statements made per shape (stark/verifier_air.py).  No backend needed: __graft_entry__.build() calls this to
    compile the generated constraint kernels of the default service's recursion programs ahead of time (they travel with the tree)."""
    air = AIR.get_air(cfg.air)
    lg = cfg.logn if logn is None else logn
    chunk_shape = VA.Shape(lg, cfg.logb, air.width, air.width2, 3 * AIR.quotient_chunks(air), cfg.n_queries, cfg.fri_logf, cfg.fri_final_log, n_proofs,
                           air.n_pub, cfg.pow_bits, int(root32), int(shift))
    ap = VA.aggregation_params(chunk_shape, cfg.agg_queries, cfg.fri_logf, cfg.fri_final_log, cfg.agg_pow_bits)
    agg_shape = VA.Shape(ap.logn, ap.logb, VA.WIDTH, 0, 3 * VA.Q_PIECES, ap.n_queries, ap.fri_logf, ap.fri_final_log, 1, chunk_shape.n_pub(), ap.pow_bits,
                         int(root32), int(shift))
    return VA.verifier_air(chunk_shape, rc, mds), VA.verifier_air(agg_shape, rc, mds)


class Engine:
    FAST_PARSE_MIN = 1 << 16      # proof texts from this size on have their openings parsed by the library (csrc/proofparse.hip)

    def __init__(self, backend_factory, config=None):
        self._factories = list(backend_factory) if isinstance(backend_factory, (list, tuple)) else [backend_factory]
        self._factory = self._factories[0]
        self._be = None
        self._be_bn = None        # backend of the final STARK (BN128-hash mode), created at the first GenFinalProof
        self._last_wrap = None    # (circuit, key) of the most recent GenFinalProof
        self.final_starks = {}    # batch_id -> final STARK JSON of the most recent batches (inspection / tests)
        self.final_programs = {}  # batch_id -> constraint program of that final STARK's statement
        self._batch_chunk_proofs = {}   # batch_id -> chunk proof texts of the most recent batches (cfg.aggregate_all_chunks)
        self.cfg = config or EngineConfig()
        self.stage_timings = {}
        self.metrics = None   # service/metrics.py Metrics, attached by serve()
        self._g16 = {}            # final-STARK layout -> (wrap circuit, Groth16 key)
        self._extra_be, self._be_lock = [], threading.Lock()
        # One request at a time per engine: every backend (zp_ctx: scratch buffers, pinned staging, buffer pools,
        # last-error slot) is single-threaded by contract, and backend 0 also serves state roots, witness uploads
        # and the Groth16 MSMs.  A reconnecting client may replay a request while the old handler still runs
        # (src/prover/provider.rs:671-700): the replay waits here, it never drives the same ctx concurrently.
        self._serial = threading.RLock()
        self._free_be = None    # engine-owned pool of idle proving backends, shared by all calls
        self._tables_cache = None
        self._be_spec = None    # backend of the speculative aggregation (cfg.speculate_recursion): never in the proving pool
        self._spec = {}         # "agg": ((sha(p1), sha(p2)), text, timings), "final": (sha(aggregated text), final-STARK tuple)
        self.pregenerate_witnesses = False   # measurement hook (bench.py): witnesses made by prepare_witnesses() are reused
        self._witness_cache = {}

    @property
    def be(self):
        if self._be is None:
            self._be = self._factory()   # HipBackend(): raises without libzethprover.so / a GPU
        return self._be

    @property
    def be_bn128(self):
        if self._be_bn is None:
            self._be_bn = self._factory(hash_mode="bn128")
        return self._be_bn

    def final_stark_params(self, agg_stark=None):
        """parameters of the final STARK (BN128-hash mode, Merkle-verifier AIR: blow-up 4, no grinding); its trace length follows
        the shape of the aggregated proof's STARK (agg_stark: that proof object) -- without one, the configured stand-alone size"""
        if agg_stark is not None:
            sh = VA.Shape.of_proof(agg_stark, 1)
            return VA.aggregation_params(sh, self.cfg.final_queries, self.cfg.fri_logf, self.cfg.fri_final_log, 0, hash="bn128")
        return PR.StarkParams(self.cfg.final_logn, self.cfg.final_logb, self.cfg.fri_logf, min(self.cfg.fri_final_log, self.cfg.final_logn - 1),
                              self.cfg.final_queries, 0, hash="bn128")

    def stark_params(self, logn=None):
        return PR.StarkParams(self.cfg.logn if logn is None else logn, self.cfg.logb, self.cfg.fri_logf, self.cfg.fri_final_log,
                              self.cfg.n_queries, self.cfg.pow_bits)

    def _backends(self, n):
        """the first n proving backends (one ctx / stream each); backend 0 is self.be.  With several factories
        (one per GPU: Engine(backend_factory=[f0, f1, ...])) backends are dealt round-robin over the GPUs, so
        independent chunk proofs of a batch spread over the devices with no exchange between them."""
        with self._be_lock:
            while len(self._extra_be) < n - 1:
                k = len(self._extra_be) + 1
                self._extra_be.append(self._factories[k % len(self._factories)]())
            return [self.be] + self._extra_be[:n - 1]

    # ---- helpers
    def _state_root(self, chain_id, block):
        st = self.be.poseidon_perm([chain_id & 0xFFFFFFFFFFFFFFFF, block & 0xFFFFFFFFFFFFFFFF] + [0] * 10)
        return b"".join(struct.pack("<Q", v) for v in st[:4])   # exactly 32 bytes (provider.rs:323-324)

    # ---- GenBatchChunks
    def gen_batch_chunks(self, batch_id, blocks, chain_id, program_name):
        with self._serial:
            return self._gen_batch_chunks(batch_id, blocks, chain_id, program_name)

    def _gen_batch_chunks(self, batch_id, blocks, chain_id, program_name):
        if not blocks:
            raise ValueError("empty batch")
        if program_name and program_name.lower() != "evm":
            raise ValueError("unknown program %r (only 'evm' is served)" % program_name)
        chunks, fetched = [], []
        l2 = None
        if self.cfg.l2_addr:
            from .l2client import L2Client
            l2 = L2Client(self.cfg.l2_addr)
        n_bind = min(4, native.BIND_SLOTS.get(AIR.get_air(self.cfg.air).trace_kind, 0))
        for b in blocks:
            nchunks = self.cfg.chunks_per_block
            bhash, txd = b"", b""
            if l2 is not None:   # real block: chunk count follows the transaction count, roots come from the node
                info = l2.block(int(b))
                fetched.append(info)
                nchunks = max(1, -(-info["n_tx"] // self.cfg.txs_per_chunk))
                pre = info["parent_state_root"] or self._state_root(chain_id, int(b) - 1)
                post, bhash, txd = info["state_root"], bytes.fromhex(info["hash"][2:]), statement.tx_digest(info["tx_hashes"])
            else:                # no node configured: the synthetic roots the response reports
                pre, post = self._state_root(chain_id, int(b) - 1), self._state_root(chain_id, int(b))
            for c in range(nchunks):
                # the statement this chunk proof is bound to (service/statement.py): its limbs become public inputs of the proof
                st = {"chain_id": int(chain_id), "block": int(b), "chunk": c, "n_chunks": nchunks, "pre_state_root": pre.hex(),
                      "post_state_root": post.hex(), "block_hash": bhash.hex(), "tx_digest": txd.hex()}
                chunks.append({"block": int(b), "chunk": c, "air": self.cfg.air, "logn": self.cfg.logn,
                               "seed": (int(chain_id) * 1000003 + int(b) * 1009 + c) & 0xFFFFFFFFFFFFFFFF,
                               "statement": st,
                               "bind": statement.statement_limbs(chain_id, b, c, nchunks, pre, post, bhash, txd, n_bind)})
        batch_data = json.dumps({"version": 2, "chain_id": int(chain_id), "blocks": [int(b) for b in blocks],
                                 "block_hashes": [f["hash"] for f in fetched], "chunks": chunks}, separators=(",", ":"))
        return {"task_id": str(int(blocks[0])).rjust(10, "0"),  # prover.proto:82-83
                "chunk_count": len(chunks), "batch_data": batch_data,
                "pre_state_root": (fetched[0]["parent_state_root"] if fetched and fetched[0]["parent_state_root"] else
                                   self._state_root(chain_id, int(blocks[0]) - 1)),
                "post_state_root": fetched[-1]["state_root"] if fetched else self._state_root(chain_id, int(blocks[-1]))}

    @staticmethod
    def _chunk_witness(air, ch, out=None):
        """the witness of one chunk: the synthetic generator (stand-in for the zkVM executor) started from the statement limbs"""
        return native.synth_trace(air.trace_kind, ch["logn"], air.width, ch["seed"], out=out, bind=ch.get("bind"))

    def prepare_witnesses(self, batch_data):
        """generate (host) and keep the synthetic witnesses of a batch, so that a following gen_chunk_proofs with
        pregenerate_witnesses = True times the prover alone (bench.py reports both forms)"""
        self._witness_cache = {}
        for ch in json.loads(batch_data)["chunks"]:
            air = AIR.get_air(ch["air"])
            self._witness_cache[(ch["air"], ch["logn"], ch["seed"])] = self._chunk_witness(air, ch)

    @staticmethod
    def _chunk_label(ch):
        """what a chunk proof says about itself next to the STARK: the block, the chunk and the statement whose limbs are its
        first public inputs (a verifier recomputes the limbs from the block: service/statement.py)"""
        lab = {"block": ch["block"], "chunk": ch["chunk"]}
        if "statement" in ch:
            lab["statement"] = ch["statement"]
        return lab

    # ---- GenChunkProof
    def gen_chunk_proofs(self, batch_id, task_id, chunk_count, batch_data):
        with self._serial:
            return self._gen_chunk_proofs(batch_id, task_id, chunk_count, batch_data)

    def _gen_chunk_proofs(self, batch_id, task_id, chunk_count, batch_data):
        plan = json.loads(batch_data)
        chunks = plan["chunks"]
        if len(chunks) != chunk_count:
            raise ValueError("chunk_count %d does not match batch_data (%d chunks)" % (chunk_count, len(chunks)))
        # witness generation is host code (stand-in for the zkVM executor); run it ahead of the GPU on a small
        # thread pool (ctypes releases the GIL) so that chunk i+1.. are generated while chunk i is being proven
        from concurrent.futures import ThreadPoolExecutor

        # proving runs on `prover_streams` backends per GPU (ctxs with their own streams) in as many threads: the latency-bound
        # tail of one proof (FRI layers, queries, transcript) overlaps the Poseidon-bound head of the next.  Backends are
        # grouped by device (backend k sits on GPU k % ndev, _backends()): chunk i belongs to GPU i % ndev from the start, so
        # its witness is uploaded THROUGH a ctx of that GPU and proven by a ctx of that GPU -- a device pointer never
        # crosses GPUs (round 2 uploaded every witness through backend 0 and handed the pointer to whichever ctx was free).
        ndev = len(self._factories)
        n_streams = max(ndev, min(self.cfg.prover_streams * ndev, -(-len(chunks) // ndev) * ndev))
        if self._free_be is None or self._free_be[0] < n_streams:   # (size, queues, uploaders): grown under the engine lock, all idle here
            bes = self._backends(n_streams)
            free_qs = [queue.SimpleQueue() for _ in range(ndev)]
            for k, b in enumerate(bes):
                free_qs[k % ndev].put(b)
            self._free_be = (n_streams, free_qs, [bes[d] for d in range(ndev)])
        free_qs, uploaders = self._free_be[1], self._free_be[2]

        # processing order: with speculation the two proofs the client aggregates -- the first and the last -- are proven first
        order = list(range(len(chunks)))
        speculate = (self.cfg.speculate_recursion and len(chunks) >= 4 and not self.cfg.aggregate_all_chunks and self.cfg.native_prover
                     and hasattr(self.be, "prove_native"))
        if speculate:
            order = [0, len(chunks) - 1] + order[1:-1]
        self._spec = {}
        if speculate:
            self._tables(self.be)       # read on this thread, before a speculation thread exists (one thread per ctx)
        spec_pool = ThreadPoolExecutor(max_workers=1) if speculate else None
        spec_state = {"texts": {}, "fut": None, "lock": threading.Lock()}
        # device-made witnesses: one recurrence walk per GPU over the chunks it will prove, started now on a thread of its own
        n_host = min(len(chunks), self.cfg.witness_threads)
        host_set = set(order[:n_host])
        # (the walk is one wave per chunk and takes what ONE chunk takes -- 0.4 s at 2^20 rows, 1.6 s at 2^22 -- whatever the chunk count: it pays
        # where the host threads would need more than two rounds of chunks)
        dev_witness = (self.cfg.witness == "device" and not self.pregenerate_witnesses and len(chunks) > 2 * n_host
                       and all(hasattr(u, "synth_checkpoints") for u in uploaders)
                       and len({(ch["air"], ch["logn"], len(ch.get("bind") or [])) for ch in chunks}) == 1
                       and AIR.get_air(chunks[0]["air"]).trace_kind in (1, 3))
        ck_futs, ck_pool = {}, None
        if dev_witness:
            ck_pool = ThreadPoolExecutor(max_workers=ndev)
            air0 = AIR.get_air(chunks[0]["air"])
            for d in range(ndev):
                mine = [i for i in range(len(chunks)) if i not in host_set and i % ndev == d]
                if mine:
                    ck_futs[d] = (ck_pool.submit(uploaders[d].synth_checkpoints, air0, chunks[0]["logn"], [chunks[i]["seed"] for i in mine],
                                                 [chunks[i].get("bind") or [] for i in mine]), {i: k for k, i in enumerate(mine)})

        def witness(i, ch):
            air = AIR.get_air(ch["air"])
            up = uploaders[i % ndev]             # a backend on the GPU that will prove this chunk
            t0 = time.perf_counter()
            cached = self._witness_cache.get((ch["air"], ch["logn"], ch["seed"])) if self.pregenerate_witnesses else None
            if cached is not None:
                trace, pubs = cached
                if hasattr(up, "prefetch_trace"):
                    trace = up.prefetch_trace(trace)
                return air, trace, pubs, 0.0, "host"
            if dev_witness and i not in host_set:
                fut, index = ck_futs[i % ndev]
                try:
                    ck = fut.result()
                    t0 = time.perf_counter()
                    trace, pubs = up.synth_trace_device(air, ch["logn"], ch["seed"], ch.get("bind"), ck, index[i])
                    return air, trace, pubs, time.perf_counter() - t0, "device"
                except native.ZpError:          # (device memory, say): the host generator makes the same trace
                    t0 = time.perf_counter()
            out = None
            if hasattr(up, "witness_buffer"):   # generate straight into page-locked memory: the copy is then plain DMA
                out = up.witness_buffer(air.width, 1 << ch["logn"])
            trace, pubs = self._chunk_witness(air, ch, out)
            tw = time.perf_counter() - t0
            if hasattr(up, "prefetch_trace"):   # copy to the GPU from this worker thread, on its own stream
                trace = up.prefetch_trace(trace)
            return air, trace, pubs, tw, "host"
        ahead = threading.Semaphore(self.cfg.witness_threads + 2)   # witnesses generated but not yet proven (memory bound)

        def witness_bounded(i, ch):
            ahead.acquire()
            try:
                return witness(i, ch)
            except BaseException:
                ahead.release()
                raise

        def prove_chunk(i, ch, wfut):
            air, trace, pubs, tw, where = wfut.result()
            free_be = free_qs[i % ndev]
            be = free_be.get()
            try:
                tm = {"witness(%s)" % where: tw}
                params = self.stark_params(ch["logn"])
                if self.cfg.native_prover and hasattr(be, "prove_native"):
                    # one C-ABI call per chunk (zp_stark_prove): the orchestration runs in the library, Python only frames the result
                    t0 = time.perf_counter()
                    text = be.prove_native(air, trace, pubs, params)
                    tm["total"] = time.perf_counter() - t0
                    text = text[:-1] + ',"chunk":%s}' % json.dumps(self._chunk_label(ch), separators=(",", ":"))
                else:
                    proof = PR.prove(air, trace, pubs, params, be, timings=tm)
                    proof["chunk"] = self._chunk_label(ch)
                    text = PR.proof_to_json(proof)
            finally:
                free_be.put(be)
                ahead.release()
            del trace
            self.stage_timings["%s/%d" % (task_id, i)] = tm
            if self.metrics is not None:
                self.metrics.record_proof(tm, ch["logn"], self.cfg.logb, air.width)
            if speculate and i in (0, len(chunks) - 1):
                with spec_state["lock"]:
                    spec_state["texts"][i] = text
                    if len(spec_state["texts"]) == 2 and spec_state["fut"] is None:
                        spec_state["fut"] = spec_pool.submit(self._speculate, batch_id, spec_state["texts"][0], spec_state["texts"][len(chunks) - 1])
            return {"chunk_id": i, "proof_key": "chunk-%s-%d" % (task_id, i), "proof": text}

        try:
            with ThreadPoolExecutor(max_workers=max(1, min(self.cfg.witness_threads, len(chunks)))) as wpool, \
                    ThreadPoolExecutor(max_workers=n_streams) as ppool:
                wfuts = {i: wpool.submit(witness_bounded, i, chunks[i]) for i in order}
                pfuts = {i: ppool.submit(prove_chunk, i, chunks[i], wfuts[i]) for i in order}
                out = [pfuts[i].result() for i in range(len(chunks))]
        finally:
            # whatever a chunk proof raised: the speculation thread is joined, the device checkpoints are freed, both pools end
            if spec_pool is not None:
                if spec_state["fut"] is not None:
                    spec_state["fut"].result()     # never raises (_speculate keeps what it has); the next request finds it finished
                spec_pool.shutdown()
            if ck_pool is not None:
                for fut, _ in ck_futs.values():
                    try:
                        fut.result().free()
                    except Exception:              # the walk itself failed (nothing to free), or the free did: the ctx reports it on its next call
                        pass
                ck_pool.shutdown()
        self._batch_chunk_proofs[batch_id] = [o["proof"] for o in out]
        while len(self._batch_chunk_proofs) > 4:
            self._batch_chunk_proofs.pop(next(iter(self._batch_chunk_proofs)))
        return out

    def _speculate(self, batch_id, p_first, p_last):
        """the aggregated proof of (first, last) and its final STARK, made while the rest of the batch is proven (cfg.speculate_recursion);
        kept under the digests of their inputs.  Whatever fails here is simply not there when the request comes."""
        try:
            if self._be_spec is None:
                self._be_spec = self._factory()
            t0 = time.perf_counter()
            text = self._aggregate(batch_id, p_first, p_last, be=self._be_spec)
            tm = dict(self.stage_timings.get("aggregate/" + batch_id, {}))
            tm["made-during-chunk-proofs"] = time.perf_counter() - t0
            self._spec["agg"] = ((batch_id, self._digest(p_first), self._digest(p_last)), text, tm)
            t0 = time.perf_counter()
            fin = self._final_stark(text, hdr_be=self._be_spec)     # never self.be off the engine's thread: chunk proofs are running on it
            self._spec["final"] = (self._digest(text), fin + (time.perf_counter() - t0,))
        except Exception:       # a proof that does not aggregate is reported when the client asks for it, by the ordinary path
            pass

    # ---- GenAggregatedProof
    @staticmethod
    def _digest(s):
        return hashlib.sha256(s.encode()).hexdigest()

    @staticmethod
    def _check_params(obj, n_queries_seen):
        pr, lim = obj["params"], {"logn": 28, "logb": 8, "fri_logf": 8, "fri_final_log": 16, "n_queries": 4096, "pow_bits": 64}
        if any(not isinstance(pr.get(k), int) or isinstance(pr.get(k), bool) or not 0 <= pr[k] <= m for k, m in lim.items()) \
                or pr["logn"] < 1 or pr["logb"] < 1 or pr["n_queries"] < 1 or pr["fri_logf"] < 1 or n_queries_seen != pr["n_queries"]:
            raise ValueError("chunk proof parameters out of range")       # nothing below is sized by numbers the text could choose freely

    @staticmethod
    def _parse_and_prepare(text):
        """(text, whole object, prepared opening arrays) of a recursive proof text.  The openings of a proof in the provers' own grammar are
        parsed by the library (csrc/proofparse.hip: no Python object per number); any other text goes through the JSON module, which also
        words the errors."""
        fast = None
        data = text.encode() if isinstance(text, str) else text
        if len(data) >= Engine.FAST_PARSE_MIN:
            span = None
            if data.startswith(b'{"kind":"aggregated"'):
                span = native.json_key_span(data, "stark")       # an aggregated proof is folded through ITS STARK (one recursion level up)
            inner = data[span[0]:span[1]] if span else data
            got = VA.prepare_proof_text(inner)
            if got is not None:
                obj, arr = got
                if "roots" in obj:
                    Engine._check_params(obj, len(arr["index"]))
                    prep = VA.prepared_from_arrays(obj, arr)
                    if span:
                        whole = json.loads(data[:span[0]] + b"null" + data[span[1]:])
                        if not isinstance(whole, dict) or whole.get("kind") != "aggregated":
                            raise ValueError("not a recursive proof of this prover")
                        whole["stark"] = obj
                    else:
                        whole = obj
                    fast = (text, whole, prep)
        if fast is not None:
            return fast
        obj = json.loads(text)
        whole = obj
        if isinstance(obj, dict) and obj.get("kind") == "aggregated" and isinstance(obj.get("stark"), dict):
            obj = obj["stark"]          # an aggregated proof is folded through ITS STARK (one recursion level up)
        if not isinstance(obj, dict) or "queries" not in obj or "roots" not in obj or not isinstance(obj.get("params"), dict):
            raise ValueError("not a recursive proof of this prover")
        if not isinstance(obj["queries"], list):
            raise ValueError("chunk proof parameters out of range")
        Engine._check_params(obj, len(obj["queries"]))
        return text, whole, VA.prepare_proof(obj)

    def _parsed_chunk_proof(self, text):
        """(object, prepared arrays) of a recursive proof text.  (Preparing the engine's own chunk proofs on a background thread
        while the rest of the batch is proven was tried: the parser holds the interpreter lock for milliseconds at a time and the
        eight proving threads starve on it -- chunk proofs 0.34 -> 0.47 s per 16 for 7 ms saved here.  Inline it is.)"""
        _, obj, prep = self._parse_and_prepare(text)
        return obj, prep

    def aggregate(self, batch_id, p1, p2):
        with self._serial, _no_cyclic_gc():
            return self._aggregate(batch_id, p1, p2)

    def _tables(self, be):
        """the Poseidon tables of this prover (configuration: the same for every backend of the engine).  Read ONCE, from the caller's own
        backend, and kept: the speculation thread (which has a backend of its own) never calls into a ctx another thread is proving on."""
        if hasattr(be, "rc"):
            return be.rc, be.mds
        if self._tables_cache is None:
            self._tables_cache = (be.p.get_constants(native.ZP_CONST_POSEIDON_RC, 360), be.p.get_constants(native.ZP_CONST_POSEIDON_MDS, 144))
        return self._tables_cache

    @staticmethod
    def _header(proof):
        """an inner proof as an aggregated proof carries it: everything a verifier reads WITHOUT an opening -- parameters, publics, roots,
        out-of-domain evaluations, FRI roots, final layer, grinding nonce.  The query openings (values and paths) stay with the prover: the
        verifier-AIR STARK vouches for everything that happens at the queries (stark/verifier_air.py)."""
        return {k: v for k, v in proof.items() if k != "queries"}

    def _prove_merkle_verifier(self, proofs, params_of, be, timings, inner_air, prepared=None):
        """STARK over the verifier AIR (stark/verifier_air.py) for inner proof objects `proofs` of one shape, proofs of `inner_air`"""
        rc, mds = self._tables(be)
        shape = VA.Shape.of_proof(proofs[0], len(proofs))
        for pr in proofs:
            if pr.get("air_digest") != inner_air.digest():
                raise ValueError("recursive proof is for another statement than this prover's")
        vair = VA.verifier_air(shape, rc, mds)
        t0 = time.perf_counter()
        # raises ValueError: an opening does not hash to its root / the transcript does not give the indices -> no witness
        trace, pubs = VA.build_witness(shape, proofs, be, inner_air.digest_words(), prepared)
        timings["verifier-witness"] = time.perf_counter() - t0
        self._last_verifier_pubs = pubs                 # the public inputs of the STARK made next (GenFinalProof's wrap needs the final STARK's)
        params = params_of(shape)
        t0 = time.perf_counter()
        self._sharded_openings = None
        if self.cfg.native_prover and self.cfg.final_ranks > 1 and params.hash == "bn128" and hasattr(be, "prove_native_sharded"):
            text, self._sharded_openings = be.prove_native_sharded(vair, trace, pubs, params, self.cfg.final_ranks, self.cfg.final_devices)
        elif self.cfg.native_prover and hasattr(be, "prove_native"):
            text = be.prove_native(vair, trace, pubs, params)
        else:
            text = PR.proof_to_json(PR.prove(vair, trace, pubs, params, be))
        timings["verifier-stark"] = time.perf_counter() - t0
        return shape, vair, params, text

    def _aggregate(self, batch_id, p1, p2, be=None):
        """GenAggregatedProof: a STARK whose witness is the verification trace of the two recursive proofs (every Poseidon
        permutation of their Merkle paths and transcripts), public inputs = their roots, indices, opened values and transcripts
        (stark/verifier_air.py).  With one chunk the client sends the same proof twice (provider.rs:386-387): it is then verified
        once.  The two proofs are chunk proofs (level 1) or aggregated proofs of one shape (level n + 1: their aggregation STARKs are
        the inner proofs, what they aggregated travels along as "children") -- "recursive proofs" in the contract's words."""
        if not p1 or not p2:
            raise ValueError("empty recursive proof")
        if be is None:
            hit = self._spec.get("agg")
            if hit is not None and hit[0] == (batch_id, self._digest(p1), self._digest(p2)):      # made while this batch was being proven
                self.stage_timings["aggregate/" + batch_id] = dict(hit[2], **{"answered-from-speculation": 1.0})
                return hit[1]
            be = self.be
        try:
            texts = [p1] if p1 == p2 else [p1, p2]
            known = self._batch_chunk_proofs.get(batch_id)
            if self.cfg.aggregate_all_chunks and known and len(known) > 2 and known[0] == p1 and known[-1] == p2:
                texts = known                      # the whole batch, not only its two ends
            parsed = [self._parsed_chunk_proof(t) for t in texts]
            wholes, prepared = [a for a, _ in parsed], [b for _, b in parsed]
            agg_params = self._agg_params
            kinds = {w.get("kind", "chunk") for w in wholes}
            if kinds == {"aggregated"}:
                # recursion one level up: the inner proofs are the aggregation STARKs of the two aggregated proofs (same shape: they
                # came out of this prover under its parameters); what they aggregated travels along as "children"
                if len({json.dumps(w.get("shape"), sort_keys=True) for w in wholes}) != 1:
                    raise ValueError("aggregated proofs of different shapes cannot be folded together")
                below = self._own_shape(wholes[0])          # before anything is sized by it
                inner_air = VA.verifier_air(below, *self._tables(self.be))
                proofs = [w["stark"] for w in wholes]
                for w, pr in zip(wholes, proofs):
                    if w.get("verifier_air_digest") != inner_air.digest() or pr["params"] != agg_params(below).to_dict():
                        raise ValueError("aggregated proof was not made under this prover's parameters")
                children = [{k: w[k] for k in ("shape", "slots", "level", "inner", "children") if k in w} for w in wholes]
                level = 1 + max(int(w.get("level", 1)) for w in wholes)
                if level > 8:
                    raise ValueError("recursion deeper than 8 levels")
            elif kinds == {"chunk"}:
                proofs, inner_air, children, level = wholes, AIR.get_air(self.cfg.air), None, 1
                for pr in proofs:
                    # this prover aggregates chunk proofs made under ITS security parameters (the text comes from the client:
                    # nothing below is sized by numbers it could choose freely)
                    lg = pr["params"].get("logn")
                    if not isinstance(lg, int) or not 1 <= lg <= 28 or pr["params"] != self.stark_params(lg).to_dict():
                        raise ValueError("chunk proof was not made under this prover's parameters")
            else:
                raise ValueError("a chunk proof and an aggregated proof cannot be folded together (aggregate the chunk proof first)")
        except (json.JSONDecodeError, TypeError, KeyError, IndexError, AssertionError) as e:
            raise ValueError("recursive proof is not a proof of this prover: %s" % e)
        tm = {}
        shape, vair, params, text = self._prove_merkle_verifier(proofs, agg_params, be, tm, inner_air, prepared)
        self.stage_timings["aggregate/" + batch_id] = tm
        if self.metrics is not None:
            for k, v in tm.items():
                self.metrics.record_stage(k, v)
        head = aggregated_head(batch_id, shape, level, vair.digest())
        head += ',"inner":' + json.dumps([self._header(pr) for pr in proofs], separators=(",", ":"))
        if children:
            head += ',"children":' + json.dumps(children, separators=(",", ":"))
        return head + ',"stark":' + text + "}"

    def _agg_params(self, sh):
        return VA.aggregation_params(sh, self.cfg.agg_queries, self.cfg.fri_logf, self.cfg.fri_final_log, self.cfg.agg_pow_bits)

    def _own_shape(self, node, depth=0):
        """The shape an aggregated proof names is CLIENT TEXT, and a verifier AIR is sized by it (schedule, transcript script and
        fixed-column entry lists grow with queries x proofs x widths): it is accepted only when it is a shape THIS prover makes.
        Level 1: inner proofs are chunk proofs of the configured AIR under the engine's STARK parameters (any trace length, 1..64
        proofs).  Level n + 1: inner proofs are aggregation STARKs over the shape of the level below (named by "children"), under the
        engine's aggregation parameters.  Returns the Shape; ValueError otherwise -- nothing is built before this has passed."""
        if not isinstance(node, dict) or not isinstance(node.get("shape"), dict) or depth > 8:
            raise ValueError("aggregated proof names no shape")
        d = node["shape"]
        if sorted(d) != sorted(VA.Shape.KEY_NAMES) or any(not isinstance(v, int) or isinstance(v, bool) for v in d.values()):
            raise ValueError("malformed shape")
        if not 1 <= d["n_proofs"] <= 64 or not 1 <= d["logn"] <= 28:
            raise ValueError("shape out of range")
        kids = node.get("children")
        if not kids:
            air = AIR.get_air(self.cfg.air)
            sp = self.stark_params(d["logn"]).to_dict()
            want = dict(d, logb=sp["logb"], W=air.width, W2=air.width2, Wq=3 * AIR.quotient_chunks(air), n_queries=sp["n_queries"],
                        fri_logf=sp["fri_logf"], fri_final_log=sp["fri_final_log"], n_pub_inner=air.n_pub, pow_bits=sp["pow_bits"],
                        root32=int(self.be.root32), shift=int(self.be.shift))
        else:
            if not isinstance(kids, list) or len(kids) != d["n_proofs"] or d["n_proofs"] > 2:
                raise ValueError("children do not match the shape")
            below = self._own_shape(kids[0], depth + 1)
            if any(not isinstance(k, dict) or k.get("shape") != kids[0]["shape"] for k in kids):
                raise ValueError("children of different shapes")
            ap = self._agg_params(below).to_dict()
            want = dict(d, logn=ap["logn"], logb=ap["logb"], W=VA.WIDTH, W2=0, Wq=3 * VA.Q_PIECES, n_queries=ap["n_queries"], fri_logf=ap["fri_logf"],
                        fri_final_log=ap["fri_final_log"], n_pub_inner=below.n_pub(), pow_bits=ap["pow_bits"], root32=int(self.be.root32),
                        shift=int(self.be.shift))
        if d != want:
            raise ValueError("aggregated proof was not made under this prover's parameters (shape)")
        shape = VA.Shape.from_dict(d)
        if shape.logn_trace() > 26:
            raise ValueError("aggregation of this size is not served")
        return shape

    # ---- GenFinalProof
    def wrap_layout(self, n_proofs=2, logn=None):
        """the layout of the final STARK this engine makes over an aggregation of `n_proofs` chunk proofs of 2^logn rows -- what sizes the wrap
        circuit (service/wrap_circuit.py); derived from the configuration alone, so that the key can be made before the first request"""
        air = AIR.get_air(self.cfg.air)
        sp = self.stark_params(logn)
        chunk_shape = VA.Shape(sp.logn, sp.logb, air.width, air.width2, 3 * AIR.quotient_chunks(air), sp.n_queries, sp.fri_logf, sp.fri_final_log, n_proofs,
                               air.n_pub, sp.pow_bits, int(self.be.root32), int(self.be.shift))
        ap = self._agg_params(chunk_shape)
        agg_shape = VA.Shape(ap.logn, ap.logb, VA.WIDTH, 0, 3 * VA.Q_PIECES, ap.n_queries, ap.fri_logf, ap.fri_final_log, 1, chunk_shape.n_pub(), ap.pow_bits,
                             int(self.be.root32), int(self.be.shift))
        fp = VA.aggregation_params(agg_shape, self.cfg.final_queries, self.cfg.fri_logf, self.cfg.fri_final_log, 0, hash="bn128")
        return WC.Layout(fp, VA.WIDTH, 3 * VA.Q_PIECES, agg_shape.n_pub())     # = Layout.of_air(the final STARK's verifier AIR, fp)

    def recursion_airs(self, n_proofs=2, logn=None):
        """(the verifier AIR the aggregation STARK proves, the one the final STARK proves) for the usual request -- two chunk proofs of the
        configured size"""
        return recursion_airs_for(self.cfg, self.be.root32, self.be.shift, *self._tables(self.be), n_proofs=n_proofs, logn=logn)

    def _wrap_key(self, fair, fp):
        """(wrap circuit, Groth16 key) for final STARKs of the statement `fair` (the verifier AIR over an aggregated proof's STARK) under the
        parameters `fp`: built once per statement (seconds at the service's size: the circuit in Python, the key's scalars on the host, its group
        elements on the GPU) and kept -- key generation is setup, not proving.  Stage B-2: the circuit runs the verifier's field arithmetic, so it is
        built FOR this statement (its constraint program, domain and parameter block), not only for its shape."""
        layout = WC.Layout.of_air(fair, fp)
        st = WA.Statement(fair.program(), int(self.be.root32), int(self.be.shift), WC.head_values(fair, fp, int(self.be.root32), int(self.be.shift)))
        k = layout.key() + st.key()
        if k not in self._g16:
            wc = WC.wrap_circuit(layout, st)
            key = groth16.Key(wc.blob)
            if hasattr(self.be, "p"):          # the GPU backend: the key's points are made now and stay in HBM
                key.load_points(self.be)
            self._g16[k] = (wc, key)
        return self._g16[k]

    def groth16_keys(self, n_proofs=2, logn=None):
        """make (or fetch) the wrap circuit and key for the usual request -- two chunk proofs of the configured size"""
        fair = self.recursion_airs(n_proofs, logn)[1]
        return self._wrap_key(fair, self.wrap_layout(n_proofs, logn).params)

    def prewarm(self, n_chunks=None, compile_kernels=True):
        """everything a first request would otherwise pay for, done at service start: the wrap circuit and its key (seconds: the circuit is built on
        the host, the key's points on the GPU), and -- by proving one synthetic batch of the configured shape end to end and throwing it away --
        the proving backends and their streams, transform plans and twiddle tables, coset tables, fixed-column extensions, the device buffer
        pools, the generated constraint kernels, the verifier AIRs of both recursion layers.  Round 4 measured 2.8-5.8 s for the first block
        against 0.65 s in the steady state (profiles/r4_service_e2e.txt).  Returns the seconds it took, by part."""
        t0 = time.perf_counter()
        self.groth16_keys()
        t_key = time.perf_counter() - t0
        # generated constraint kernels for the two recursion programs (hipcc at the first start on a host, a file afterwards); a host without a
        # compiler keeps the interpreter
        t_k = time.perf_counter()
        kernels = None
        if compile_kernels and hasattr(self.be, "compile_air_kernel"):
            agg_air, fin_air = self.recursion_airs()
            kernels = [self.be.compile_air_kernel(agg_air), self.be_bn128.compile_air_kernel(fin_air)]
        t_k = time.perf_counter() - t_k
        cfg_l2, self.cfg.l2_addr = self.cfg.l2_addr, None            # no node is asked for blocks that do not exist
        try:
            n = max(2, n_chunks or min(8, self.cfg.prover_streams))
            blocks = list(range(1, 1 + -(-n // max(1, self.cfg.chunks_per_block))))
            ch = self.gen_batch_chunks("__prewarm__", blocks, 0, "evm")
            proofs = self.gen_chunk_proofs("__prewarm__", ch["task_id"], ch["chunk_count"], ch["batch_data"])
            t_chunks = time.perf_counter() - t0 - t_key
            agg = self.aggregate("__prewarm__", proofs[0]["proof"], proofs[-1]["proof"])
            self.final("__prewarm__", agg, "BN128", "0")
        finally:
            self.cfg.l2_addr = cfg_l2
            for d in (self._batch_chunk_proofs, self.final_starks, self.final_programs):
                d.pop("__prewarm__", None)
            for k in [k for k in self.stage_timings if "__prewarm__" in k]:
                self.stage_timings.pop(k, None)
        total = time.perf_counter() - t0
        return {"wrap_key_s": t_key, "recursion_kernels_s": t_k, "recursion_kernels": kernels, "chunk_proofs_s": t_chunks - t_k,
                "recursion_s": total - t_key - t_chunks, "total_s": total}

    def verifying_key_json(self, n_proofs=None, logn=None):
        """the verifying key final proofs of this engine verify under.  A stage B-2 key belongs to ONE statement (the final STARK's AIR depends on how
        many chunk proofs the aggregation folded): without arguments the key of the most recent GenFinalProof, else of the usual request (two chunk
        proofs of the configured size); with arguments, of that request."""
        if n_proofs is None and logn is None and self._last_wrap is not None:
            return groth16.vk_to_json(self._last_wrap[1].vk)
        return groth16.vk_to_json(self.groth16_keys(2 if n_proofs is None else n_proofs, logn)[1].vk)

    def final(self, batch_id, recursive_proof, curve_name, aggregator_addr):
        with self._serial, _no_cyclic_gc():
            return self._final(batch_id, recursive_proof, curve_name, aggregator_addr)

    def _final(self, batch_id, recursive_proof, curve_name, aggregator_addr):
        if (curve_name or "").upper() not in ("BN128", "BN254"):
            raise ValueError("unsupported curve %r" % curve_name)
        if not recursive_proof:
            raise ValueError("empty recursive proof")
        # 1. the final STARK: BN128-hash mode (16-ary Poseidon-BN254 trees, transcript over the BN254 scalar field), the form a
        #    Groth16 circuit over that field can verify.  Its statement: the Merkle-verifier AIR over the aggregated proof's own
        #    STARK -- every query opening of the aggregated proof hashes to its roots (public inputs: those roots and indices).
        #    A recursive proof that is not an aggregated proof of this service (a client of another prover) cannot be verified
        #    here and is an application error (the client retries: provider.rs:504-523).
        t0 = time.perf_counter()
        hit = self._spec.get("final")
        if hit is not None and hit[0] == self._digest(recursive_proof):       # made while the batch was being proven
            fshape, fair, fp, final_stark, tmf, openings, fpubs, t_made = hit[1]
            tmf = dict(tmf, **{"made-during-chunk-proofs": t_made, "answered-from-speculation": 1.0})
        else:
            fshape, fair, fp, final_stark, tmf, openings, fpubs = self._final_stark(recursive_proof)
        t_fs = time.perf_counter() - t0
        return self._final_wrap(batch_id, aggregator_addr, fair, fp, final_stark, tmf, openings, t_fs, fpubs)

    def _check_inner_headers(self, node, stark_publics, be, depth=0):
        """What NO query of a recursion STARK covers, for the proofs it vouches for, level by level (round-5 advisor item).  node: an aggregated
        proof (or a "children" entry of one) -- {"shape", "inner": headers, ["children"]}; stark_publics: the public inputs of the STARK over the
        verifier AIR of that shape, already known to verify (the level above checked its header; the final STARK proves its query phase).  Per
        inner header: the verifier's own parameters, the constraint identity at the out-of-domain point, the final layer's degree, the grinding
        (stark/verifier.py) -- against the chunk AIR under this prover's STARK parameters at the leaves, against the verifier AIR of the level
        below otherwise.  And the link: stark_publics are EXACTLY what these headers dictate (roots, indices and transcripts from each header's
        own Fiat-Shamir stream, arithmetic constants, final-layer values) -- without it the STARK could vouch for the queries of other proofs than
        the ones whose headers the text shows.  Raises ValueError."""
        if depth > 8 or not isinstance(node.get("inner"), list):
            raise ValueError("aggregated proof carries no inner headers")
        shape = self._own_shape(node)
        kids = node.get("children")
        if not kids:
            inner_air, params = AIR.get_air(self.cfg.air), self.stark_params(shape.logn)
        else:
            below = self._own_shape(kids[0])
            inner_air, params = VA.verifier_air(below, *self._tables(self.be_bn128)), self._agg_params(below)
        if len(node["inner"]) != shape.n_proofs:
            raise ValueError("inner headers do not match the shape")
        for h in node["inner"]:
            if not isinstance(h, dict) or "queries" in h:
                raise ValueError("an aggregated proof carries the HEADERS of its inner proofs")
            SV.verify_header(h, inner_air, params, be)
        want = VA.publics_from_headers(shape, node["inner"], inner_air.digest_words(), be)
        if [int(v) for v in stark_publics] != want:
            raise ValueError("the recursion STARK's public inputs are not what the inner proofs' headers dictate")
        if kids:
            for kid, hdr in zip(kids, node["inner"]):
                self._check_inner_headers(kid, hdr["publics"], be, depth + 1)

    def _final_stark(self, recursive_proof, hdr_be=None):
        """(shape, AIR, parameters, text, timings, binary openings) of the final STARK over an aggregated proof text.  hdr_be: the backend the
        header checks hash on -- the engine's own (self.be) on the engine's thread, the speculation thread's backend when that thread calls
        (a zp_ctx is single-threaded; self.be is proving chunks while a speculation runs)"""
        hb = self.be if hdr_be is None else hdr_be
        try:
            _, agg, prep = self._parse_and_prepare(recursive_proof)
            outer = agg["stark"]
            if agg.get("kind") != "aggregated" or "queries" not in outer:
                raise KeyError("kind")
            agg_shape = self._own_shape(agg)                # client text: only a shape this prover makes is ever built
            if outer.get("params") != self._agg_params(agg_shape).to_dict():
                raise ValueError("aggregated proof was not made under this prover's aggregation parameters")
            agg_air = VA.verifier_air(agg_shape, *self._tables(self.be_bn128))   # the statement of agg["stark"]
            if agg.get("verifier_air_digest") != agg_air.digest() or outer.get("air_digest") != agg_air.digest():
                raise ValueError("aggregated proof is over another verifier AIR")
            # The text is the client's.  The final STARK below proves the aggregation STARK's QUERY phase (its witness does not exist when a
            # path, the DEEP quotient or a fold fails); what no query covers -- the constraint identity at the out-of-domain point, the final
            # layer's degree, the grinding -- is checked here, natively, before anything is proven (stark/verifier.py; round-4 advisor item)
            # It runs BESIDE the final STARK (host arithmetic + a handful of sponge calls on self.be; the final STARK runs on be_bn128, another
            # ctx) and is joined before anything leaves this function: a header that does not verify raises, whatever the STARK did meanwhile.
            hdr = {"t": 0.0, "err": None}

            def check_header():
                t0 = time.perf_counter()
                try:
                    SV.verify_header(outer, agg_air, self._agg_params(agg_shape), hb)
                    self._check_inner_headers(agg, outer["publics"], hb)       # ... and the same for every proof below it, down to the chunk proofs
                except Exception as e:          # re-raised on the engine's thread below
                    hdr["err"] = e
                hdr["t"] = time.perf_counter() - t0
            th = None
            if self.cfg.verify_before_wrap:
                if self.be_bn128 is not hb and hasattr(hb, "p"):
                    th = threading.Thread(target=check_header, name="verify-aggregated-header")
                    th.start()
                else:                           # one ctx for both (the CPU checker's backend in tests): one after the other
                    check_header()
                    if hdr["err"] is not None:
                        raise hdr["err"]
        except (json.JSONDecodeError, TypeError, KeyError, AssertionError, ValueError) as e:
            raise ValueError("recursive proof is not an aggregated proof of this prover (%s)" % e)
        tmf = {}
        try:
            fshape, fair, fp, final_stark = self._prove_merkle_verifier([outer], lambda sh: self.final_stark_params(outer), self.be_bn128, tmf, agg_air, [prep])
            fpubs = [int(v) for v in self._last_verifier_pubs]
        finally:
            if th is not None:
                th.join()
        if hdr["err"] is not None:
            e = hdr["err"]
            if isinstance(e, ValueError):
                raise ValueError("recursive proof is not an aggregated proof of this prover (%s)" % e)
            raise e
        tmf["verify-aggregated-header(beside the final STARK)" if th is not None else "verify-aggregated-header"] = hdr["t"]
        openings = None
        if getattr(self, "_sharded_openings", None) is not None:
            openings = self._sharded_openings               # rank 0's record (cfg.final_ranks > 1)
        elif self.cfg.native_prover and hasattr(self.be_bn128, "stark_openings"):
            openings = self.be_bn128.stark_openings()       # the prover's own binary record of what the text carries: no text round trip
        return fshape, fair, fp, final_stark, tmf, openings, fpubs

    def _final_wrap(self, batch_id, aggregator_addr, fair, fp, final_stark, tmf, openings, t_fs, final_publics):
        self.final_starks[batch_id] = final_stark
        self.final_programs[batch_id] = fair.program()         # the statement of that STARK (a checker recomputes the wrap's public input from both)
        while len(self.final_starks) > 4:
            k0 = next(iter(self.final_starks))
            self.final_starks.pop(k0)
            self.final_programs.pop(k0, None)
        fs_digest = hashlib.sha256(final_stark.encode()).hexdigest()
        # 2. the Groth16 wrap: an R1CS that verifies the hashing of that STARK's verifier at its queries (service/wrap_circuit.py); its one
        #    public input commits to the roots, indices and leaf elements it vouches for, and to the aggregator address of the request
        t0 = time.perf_counter()
        wc, key = self._wrap_key(fair, fp)
        self._last_wrap = (wc, key)
        try:
            aux = int(aggregator_addr or "0")
        except ValueError:
            aux = int(hashlib.sha256((aggregator_addr or "").encode()).hexdigest(), 16)
        if openings is None:      # a final STARK made by the Python orchestration: its transcript is replayed for the record (the one-call prover logs it)
            fs = json.loads(final_stark)
            tlog = WC.TranscriptLog(fs, wc.layout, WC.head_values(fair, fp, fs["root32"], fs["shift"]), getattr(self.be_bn128, "publics_digest", None),
                                    getattr(self.be_bn128, "poseidon_bn254_perm17", WC.perm17))
            openings = WC.openings_record(fs, wc.layout, tlog)
        # stage B-2: the circuit takes the statement's sparse fixed columns at zeta from its caller -- a function of (statement, public inputs, zeta);
        # zeta is challenge element 1 of the prover's record.  The proof text carries zeta's words: a reader recomputes the columns and d from public data.
        aux_list, zeta_words = native.wrap_aux(openings, fair.program(), final_publics, fp.logn, int(self.be.root32), aux % bn254.R)
        set_idx, set_val = native.wrap_assign(wc.script, openings, aux_list)
        t_wit = time.perf_counter() - t0
        # fresh blinding per proof (zero knowledge); replays of a request are answered from the batch store (server.py),
        # which keeps the finished proof, so the client still sees one proof per batch
        if self.cfg.groth16_seed is None:
            rnd = (int.from_bytes(os.urandom(32), "big") % bn254.R or 1, int.from_bytes(os.urandom(32), "big") % bn254.R or 1)
        else:
            det = lambda tag: int(hashlib.sha256(("%s|%s|%s|%s" % (self.cfg.groth16_seed, fs_digest, aggregator_addr or "", tag)).encode()).hexdigest(), 16) % bn254.R or 1
            rnd = (det("r"), det("s"))
        t0 = time.perf_counter()
        proof, pub, g16_ms = groth16.prove(key, set_idx, set_val, self.be, rnd)      # ValueError: the openings do not hash to the roots -> no witness
        self.stage_timings["final/" + batch_id] = {"final_stark(bn128)": t_fs, "wrap-assign": t_wit, "groth16": time.perf_counter() - t0,
                                                   "groth16/witness": g16_ms[0] / 1e3, "groth16/qap": g16_ms[1] / 1e3, "groth16/msm": g16_ms[2] / 1e3,
                                                   **({"groth16/msm/" + k: v / 1e3 for k, v in zip(("A", "B1", "B2", "l", "h"), g16_ms[3:8])} if len(g16_ms) >= 8 else {}),
                                                   **{"final/" + k: v for k, v in tmf.items()}}
        n_v = key.dev["v_wires"][1] if key.dev else int((key.v != 0).any(axis=1).sum())      # wires with a non-zero column in B
        self.wrap_info = {"constraints": wc.c.n_constraints, "qap_domain_log2": wc.c.logm(), "wires": wc.c.n_wires,
                          "msm_points": {"A (G1)": wc.c.n_wires + 2, "B (G1)": n_v + 2, "B (G2)": n_v + 2, "C: l (G1)": wc.c.n_wires,
                                         "C: h (G1)": (1 << wc.c.logm()) - 1}}
        if self.metrics is not None:
            self.metrics.record_stage("groth16", self.stage_timings["final/" + batch_id]["groth16"])
        js = groth16.proof_to_json(proof, {"circuit": groth16.circuit_text(wc, key), "final_stark_sha256": fs_digest, "zeta": [str(v) for v in zeta_words]})
        return js, json.dumps([str(pub[0])])
