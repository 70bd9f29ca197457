"""GPU backend of the STARK prover: every method is one or two C-ABI calls on device-resident data.
No CPU fallback: construction fails without libzethprover.so and an MI355X."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import numpy as np

from .. import native
from . import air as air_mod
from . import build_airs

_u64p = C.POINTER(C.c_uint64)


class Commit:
    def __init__(self, root, tree, ext=None, coef=None):
        self.root, self.tree, self.ext, self.coef = root, tree, ext, coef


class HipBackend:
    def __init__(self, device=0, prover=None, quotient="kernel", hash_mode="gl"):
        """quotient: "kernel" = the generated per-AIR constraint kernel (AIR plug-in ABI, needs hipcc at build time);
        "program" = zp_eval_quotient interpreting the AIR's constraint program blob (what a host without a compiler uses).
        hash_mode: "gl" = Poseidon-Goldilocks binary trees; "bn128" = Poseidon-BN254 16-ary trees + transcript (final STARK)"""
        assert quotient in ("kernel", "program") and hash_mode in ("gl", "bn128")
        self.quotient_mode = quotient
        self._native_kernels = set()
        self.hash_mode = hash_mode
        self.p = prover or native.Prover(device)
        self.p.pooling = True   # chunk after chunk has the same shapes: reuse device buffers
        self.root32 = int(self.p.get_constants(native.ZP_CONST_ROOT32, 1)[0])
        self.shift = int(self.p.get_constants(native.ZP_CONST_COSET_SHIFT, 1)[0])
        self._fixed = {}
        self._airlibs = {}
        self._perm_buf = self.p.alloc(12)
        self.p_device = device
        self._up = None
        self._up_lock, self._copy_lock = threading.Lock(), threading.Lock()
        if hash_mode == "bn128":
            self.p.install_poseidon_bn254(17)
        if os.environ.get("ZP_MERKLE_COOP_LOG"):      # experiment knob: tree levels handled by the 12-lanes-per-node kernels
            self.p.set_tuning("merkle_coop_log", int(os.environ["ZP_MERKLE_COOP_LOG"]))

    # ---- Merkle trees of the configured hash mode
    def _tree_alloc(self, M):
        return self.p.alloc(self.p.merkle16_nodes(M) * 4 if self.hash_mode == "bn128" else (2 * M - 1) * 4)

    def _commit(self, d_cols, M, W, tree):
        if self.hash_mode == "bn128":
            self.p.merkle16_commit_bn254(d_cols, M, W, tree)
        else:
            self.p.merkle_commit(d_cols, M, W, tree)

    def poseidon_bn254_perm17(self, state):
        return self.p.poseidon_bn254_perm([state])[0]

    def sync(self):
        self.p.sync()

    # ---- transcript permutation (one tiny launch)
    def poseidon_perm(self, state):
        self.p.h2d(self._perm_buf, np.array(state, dtype=np.uint64))
        self.p.poseidon_perm(self._perm_buf, 1)
        return [int(v) for v in self.p.download(self._perm_buf, (12,))]

    def poseidon_perm_batch(self, states):
        """states uint64 [B][12] -> permuted copy (one launch)"""
        st = np.ascontiguousarray(np.asarray(states, dtype=np.uint64))
        if st.shape[0] == 0:
            return st.copy()
        d = self.p.upload(st)
        self.p.poseidon_perm(d, st.shape[0])
        out = self.p.download(d, st.shape)
        d.free()
        return out

    def poseidon_trace(self, inputs):
        """inputs uint64 [B][12] -> (states [12][32 B], cubes [12][32 B]): round-by-round witness of a Poseidon AIR (zp_poseidon_trace)"""
        a = np.ascontiguousarray(np.asarray(inputs, dtype=np.uint64))
        B = a.shape[0]
        d_in, d_out = self.p.upload(a), self.p.alloc(24 * 32 * B)
        self.p.poseidon_trace(d_in, B, d_out, d_out.offset(12 * 32 * B), 32 * B)
        out = self.p.download(d_out, (24, 32 * B))
        d_in.free()
        d_out.free()
        return out[:12], out[12:]

    def publics_digest(self, pubs):
        """commitment to a long public-input vector (stark/prover.py: publics_rows): the root of its tree in this hash mode"""
        from .prover import publics_rows
        mat = publics_rows(pubs, self.hash_mode == "bn128")
        if self.hash_mode == "bn128":
            M = mat.shape[1]
            d, tree = self.p.upload(mat), self._tree_alloc(M)
            self.p.merkle16_commit_bn254(d, M, 48, tree)
        else:
            M = mat.shape[0]
            d, tree = self.p.upload(mat), self._tree_alloc(M)
            self.p.merkle_commit_rows(d, M, 8, tree)
        root = self._root(tree, M)
        d.free()
        tree.free()
        return root

    def publics_digest_gl(self, pubs):
        """the Goldilocks-mode digest whatever this backend's hash mode (the transcript of an INNER proof, replayed for the
        verifier AIR's witness: stark/verifier_air.py)"""
        from .prover import publics_rows
        mat = publics_rows(pubs, False)
        M = mat.shape[0]
        d, tree = self.p.upload(mat), self.p.alloc((2 * M - 1) * 4)
        try:
            self.p.merkle_commit_rows(d, M, 8, tree)
            return [int(v) for v in self.p.download(tree, (4,), offset_elems=(2 * M - 2) * 4)]
        finally:
            d.free()
            tree.free()

    def poseidon_sponge_caps(self, state, blocks, extra):
        return self.p.poseidon_sponge_caps(state, blocks, extra)

    def verifier_trace_device(self, inputs, dbit, idxv, arith):
        """the trace of the verifier AIR (stark/verifier_air.py) assembled IN HBM: zp_poseidon_trace writes the 24 state / cube columns of
        every permutation block; the direction-bit and index columns (one value per block, repeated over its 32 rows) and the arithmetic
        columns (`arith`: host array u64[WIDTH - 26][N], or a device buffer of that shape from verifier_arith_columns) follow.
        Returns a device buffer shaped [WIDTH][32 * blocks] that commit_trace / prove_native take as they take an uploaded witness."""
        from . import verifier_air as VA
        a = np.ascontiguousarray(np.asarray(inputs, dtype=np.uint64))
        B = a.shape[0]
        N = 32 * B
        d_in = self.p.upload(a)
        d_tr = self.p.alloc(VA.WIDTH * N)
        self.p.poseidon_trace(d_in, B, d_tr, d_tr.offset(12 * N), N)
        tail = np.empty((2, N), dtype=np.uint64)
        tail[0] = np.repeat(np.asarray(dbit, dtype=np.uint64), 32)
        tail[1] = np.repeat(np.asarray(idxv, dtype=np.uint64), 32)
        self.p.h2d(d_tr.offset(24 * N), tail)
        if isinstance(arith, native.DeviceBuffer):
            self.p.d2d(d_tr.offset(VA.HR0 * N), arith, (VA.WIDTH - VA.HR0) * N * 8)
            arith.free()
        else:
            assert tuple(arith.shape) == (VA.WIDTH - VA.HR0, N)
            self.p.h2d(d_tr.offset(VA.HR0 * N), arith)
        d_in.free()
        d_tr.shape = (VA.WIDTH, N)
        return d_tr

    def recursion_witness(self, shape, proofs, prepared, digest_words):
        """(device trace u64[WIDTH][N], public inputs) of the verifier AIR through ONE library call (zp_recursion_witness: hashing walk,
        transcript replays, public inputs, trace assembly in HBM) -- what VA.build_witness does step by step in Python"""
        from . import verifier_air as VA
        k, periods, pb = shape.layout()
        N = 32 * pb * periods
        index, values, paths, streams = [], [], [], []
        for pr, pp in zip(proofs, prepared):
            index.append(pp["index"])
            values.append(np.concatenate([np.ascontiguousarray(v).reshape(-1) for v in pp["values"]]))
            paths.append(np.concatenate([np.ascontiguousarray(v).reshape(-1) for v in pp["paths"]]))
            streams.append(VA.transcript_stream(shape, pr, digest_words))
        d_tr = self.p.alloc(VA.WIDTH * N)
        try:
            pubs = self.p.recursion_witness(VA.arith_descriptor(shape), index, values, paths, streams, d_tr)
        except BaseException:
            d_tr.free()
            raise
        d_tr.shape = (VA.WIDTH, N)
        return d_tr, pubs

    def verifier_arith_columns(self, shape, arith_in):
        """the 21 arithmetic columns of the verifier trace, built by the library (host walk + expansion kernel: zp_verifier_arith_trace) into a
        device buffer u64[21][N] that verifier_trace_device splices behind the hashing columns"""
        from . import verifier_air as VA
        N = 32 * len(arith_in["dbit"])
        d = self.p.alloc((VA.WIDTH - VA.HR0) * N)
        try:
            self.p.verifier_arith_trace(VA.arith_descriptor(shape), arith_in, d)
        except BaseException:
            d.free()
            raise
        return d

    def stark_openings(self):
        """the binary openings of the last BN128-mode proof prove_native made on this backend (zp_stark_openings)"""
        return self.p.stark_openings()

    def prove_native(self, air, trace, pubs, params):
        """the whole chunk STARK through zp_stark_prove (one C-ABI call, orchestration in the library's host C++): proof TEXT,
        byte-identical to proof_to_json(prove(...)) over this backend.  trace: host array or a device buffer from prefetch_trace."""
        assert self.hash_mode == params.hash
        kkey = (air.name, air.digest())
        if self.quotient_mode == "kernel" and kkey not in self._native_kernels and (not air.fixed_cols or os.path.exists(build_airs.lib_path(air))):
            # the one-call prover evaluates this AIR's constraints through its generated kernel too (zp_stark_set_air_kernel); a host without the
            # AIR's library (and without a compiler to make it) simply stays with the interpreter: same proofs.  An AIR made per shape (a
            # verifier AIR: sparse periodic fixed columns) is never compiled on the way -- its library exists when somebody asked for it
            # (compile_air_kernel: the service's prewarm does, for the two recursion programs it will prove)
            try:
                self.p.set_air_kernel(air.program(), self._airlib(air))
            except (OSError, RuntimeError, AttributeError, subprocess.CalledProcessError):
                pass
            self._native_kernels.add(kkey)
        d_tr = trace if isinstance(trace, native.DeviceBuffer) else self.p.upload(trace)
        try:
            if self.hash_mode == "bn128":       # zp_stark_prove_bn128: 16-ary Poseidon-BN254 trees, transcript over F_r, no grinding
                return self.p.stark_prove_bn128(air.name, air.program(), d_tr, [int(v) for v in pubs], params.logn, params.logb, params.fri_logf,
                                                params.fri_final_log, params.n_queries)
            return self.p.stark_prove(air.name, air.program(), d_tr, [int(v) for v in pubs], params.logn, params.logb, params.fri_logf,
                                      params.fri_final_log, params.n_queries, params.pow_bits)
        finally:
            d_tr.free()

    def prove_native_sharded(self, air, trace, pubs, params, ranks, devices=None):
        """ONE proof over `ranks` thread-ranks of this process (zp_stark_prove_sharded / zp_stark_prove_sharded_bn128 on an in-process
        communicator, csrc/comm.hip: zp_comm_create_local): rank r drives devices[r] with a ctx of its own (default: every rank on this
        backend's GPU -- the ranks then share its CUs, which is a rehearsal, not a speed-up).  Rank r takes ITS columns of the trace
        (ceil(W / ranks) per rank, the tail ranks fewer: 47 columns of a verifier AIR over 8 ranks = 6 x 7 + 5).  Returns (proof text --
        byte-identical to prove_native's --, rank 0's binary openings record in BN128 mode, else None).  Round 6: every rank evaluates its row
        window of the quotient through the AIR's GENERATED kernel (`<symbol>_rows`, zp_stark_set_air_kernel_rows) when this backend runs in
        kernel mode and the AIR's library exists; the interpreter serves otherwise -- same proof bytes."""
        assert self.hash_mode == params.hash and ranks >= 1 and (ranks & (ranks - 1)) == 0
        bn = params.hash == "bn128"
        devs = list(devices) if devices else [self.p_device] * ranks
        assert len(devs) == ranks
        W, N = air.width, 1 << params.logn
        with self._up_lock:
            provers = getattr(self, "_rank_provers", None)
            if provers is None or [d for d, _ in provers] != devs:
                for _, q in provers or []:
                    q.close()
                provers = []
                for d in devs:
                    q = native.Prover(d)
                    for kind, n in ((native.ZP_CONST_POSEIDON_RC, 360), (native.ZP_CONST_POSEIDON_MDS, 144), (native.ZP_CONST_ROOT32, 1),
                                    (native.ZP_CONST_COSET_SHIFT, 1)):
                        q.set_constants(kind, self.p.get_constants(kind, n))         # this backend's field / hash configuration
                    if bn:
                        q.install_poseidon_bn254(17)
                    provers.append((d, q))
                self._rank_provers = provers
        dev_trace = isinstance(trace, native.DeviceBuffer)
        if dev_trace:
            self.p.sync()                                   # the witness builder's last kernels, before other ctxs read the buffer
        host = None if dev_trace and all(d == self.p_device for d in devs) else (self.p.download(trace, (W, N)) if dev_trace else np.asarray(trace, dtype=np.uint64))
        group = native.CommGroup(ranks)
        texts, errs, rec = [None] * ranks, [None] * ranks, [None]
        prog, pl = air.program(), [int(v) for v in pubs]
        rows_fn = None
        if self.quotient_mode == "kernel" and (not air.fixed_cols or os.path.exists(build_airs.lib_path(air))):
            try:
                rows_fn = self._airlib_rows(air)
            except (OSError, RuntimeError, AttributeError, subprocess.CalledProcessError):
                rows_fn = None
        for _, q in provers:
            q.set_air_kernel_rows(prog, rows_fn)

        def body(r):
            q = provers[r][1]
            c, d_l = None, None
            try:
                c = native.Comm(q, r, ranks, group=group)
                first, count = c.my_columns(W)
                if count:
                    if host is None:                        # same GPU: a device copy of my columns
                        d_l = q.alloc(count * N)
                        q.d2d(d_l, trace.ptr + first * N * 8, count * N * 8)
                    else:
                        d_l = q.upload(np.ascontiguousarray(host[first:first + count]))
                texts[r] = c.stark_prove_sharded(air.name, prog, d_l, pl, params.logn, params.logb, params.fri_logf, params.fri_final_log, params.n_queries,
                                                 0 if bn else params.pow_bits, bn128=bn)
                if bn and r == 0:
                    rec[0] = q.stark_openings()
            except BaseException as e:                      # noqa: every rank's error is looked at below
                errs[r] = e
                if c is not None:
                    try:
                        c.abort()                           # the peers are inside the same collective call: free them
                    except Exception:
                        pass
            finally:
                if d_l is not None:
                    d_l.free()
                if c is not None:
                    c.close()
        ts = [threading.Thread(target=body, args=(r,), name="final-rank-%d" % r) for r in range(ranks)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        group.close()
        if dev_trace:
            trace.free()
        bad = [e for e in errs if e is not None]
        if bad:
            real = [e for e in bad if not (isinstance(e, native.ZpError) and e.code == -6)]       # ZP_ERR_COMM: a peer of the rank that failed
            raise (real or bad)[0]
        assert all(t == texts[0] for t in texts)
        return texts[0], rec[0]

    def poseidon_sponge(self, state, blocks, extra):
        return self.p.poseidon_sponge(state, blocks, extra)

    def pow_grind(self, seed4, bits):
        return self.p.pow_grind(seed4, bits)

    # ---- commitments
    def _root(self, tree, M):
        if self.hash_mode == "bn128":   # one field element: the last node of the 16-ary tree
            return self.p._fr_ints(self.p.download(tree, (4,), offset_elems=(self.p.merkle16_nodes(M) - 1) * 4))
        return [int(v) for v in self.p.download(tree, (4,), offset_elems=(2 * M - 2) * 4)]

    def _uploader(self):
        with self._up_lock:
            if self._up is None:
                self._up = native.Prover(self.p_device)
                self._up.pooling = True
        return self._up

    # ---- synthetic witnesses generated in HBM (csrc/synth.hip): a third ctx, so that the recurrence walk of a batch (a second or so)
    # never holds the ctx the uploads of host-made witnesses go through
    def _generator(self):
        with self._up_lock:
            if getattr(self, "_gen", None) is None:
                self._gen = native.Prover(self.p_device)
                self._gen.pooling = True
                self._gen_lock = threading.Lock()
        return self._gen

    def synth_checkpoints(self, air, logn, seeds, binds):
        """checkpoints of the wide-mix recurrences of a batch of chunks (synchronous; call from a worker thread)"""
        g = self._generator()
        with self._gen_lock:
            return g.synth_checkpoints(air.trace_kind, logn, air.width, seeds, binds)

    def synth_trace_device(self, air, logn, seed, bind, ckpt, index):
        """the witness of one chunk from its checkpoints, as a device buffer prove_native / commit_trace accept"""
        g = self._generator()
        with self._gen_lock:
            d, pubs = g.synth_trace_device(air.trace_kind, logn, air.width, seed, bind=bind, ckpt=ckpt, ckpt_index=index)
            g.sync()            # the proof runs on another ctx's stream
        return d, pubs

    def witness_buffer(self, W, N):
        """page-locked host array for a witness generator to fill (pooled); pass it to prefetch_trace"""
        return self._uploader().host_array((W, N))

    def prefetch_trace(self, trace):
        """H2D of a witness on a second ctx/stream (safe to call from a worker thread): lets the engine copy
        chunk i+1 while chunk i is being proven.  One copy in flight at a time (PCIe is the limit anyway).
        Returns a handle commit_trace accepts in place of the array; a witness_buffer is recycled."""
        up = self._uploader()
        with self._copy_lock:
            buf = up.upload(trace)
        buf.shape = trace.shape
        if trace.ctypes.data in getattr(up, "_host_ptrs", {}):
            up.release_host_array(trace)
        return buf

    def commit_trace(self, trace, logn, logb, extra_cols=0, group=0):
        """ext is allocated with room for `extra_cols` stage-2 columns behind the trace columns; group: log2 of the rows
        per leaf of the tree (BN128 mode: stark/prover.py bn128_rows_per_leaf_log)"""
        W = trace.shape[0]
        M = 1 << (logn + logb)
        d_tr = trace if isinstance(trace, native.DeviceBuffer) else self.p.upload(trace)
        ext = self.p.alloc((W + extra_cols) * M)
        tree = self._tree_alloc(M >> group)
        self.p.lde(d_tr, ext, logn, logb, W, self.shift)         # no coefficient store: the out-of-domain evaluations read ext (ood_evals)
        self._commit(ext, M >> group, W << group, tree)
        c = Commit(self._root(tree, M >> group), tree, ext)
        if extra_cols:
            c.trace, c.W = d_tr, W   # stage 2 reads witness columns
        else:
            d_tr.free()
        return c

    def column_view(self, d_mat, col, rows):
        return d_mat.offset(col * rows)

    def commit_stage2(self, air, c1, chal, logn, logb, group=0):
        """stage-2 witness columns (grand products Z, LogUp h1/h2/S) -> LDE into columns W.. of c1.ext -> their tree"""
        N, M, W, W2 = 1 << logn, 1 << (logn + logb), c1.W, air.width2
        d_s2 = self.p.alloc(W2 * N)
        at = 0
        for st in air.stage2:
            col = lambda k: c1.trace.offset(st[k] * N)
            if st["kind"] == "perm":
                self.p.grand_product(col("a"), col("b"), N, chal, d_s2.offset(at * N))
            else:
                self.p.logup_columns(col("a"), col("t"), col("m"), N, chal, d_s2.offset(at * N))
            at += air_mod.STAGE2_WIDTH[st["kind"]]
        self.p.lde(d_s2, c1.ext.offset(W * M), logn, logb, W2, self.shift)
        tree = self._tree_alloc(M >> group)
        self._commit(c1.ext.offset(W * M), M >> group, W2 << group, tree)
        self.p.sync()
        d_s2.free()
        c1.trace.free()
        return Commit(self._root(tree, M >> group), tree)

    def commit_cols(self, d_cols, M, W):
        tree = self._tree_alloc(M)
        self._commit(d_cols, M, W, tree)
        return Commit(self._root(tree, M), tree)

    def fixed_ext(self, logn, logb, air=None, pubs=None):
        if air is not None and air.fixed_cols:      # sparse periodic fixed columns: selectors + one extended period each (program mode)
            key = (logn, logb, air.digest())
            if any(fc.public for fc in air.fixed_cols):
                return self.p.fixed_columns(air.program(), pubs, logn, logb, self.shift)
            if key not in self._fixed:
                self._fixed[key] = self.p.fixed_columns(air.program(), pubs, logn, logb, self.shift)
            return self._fixed[key]
        key = (logn, logb)
        if key not in self._fixed:
            N = 1 << logn
            ind = np.zeros((2, N), dtype=np.uint64)
            ind[0, 0] = 1
            ind[1, N - 1] = 1
            d_in = self.p.upload(ind)
            d_out = self.p.alloc(2 << (logn + logb))
            self.p.lde(d_in, d_out, logn, logb, 2, self.shift)
            d_in.free()
            self._fixed[key] = d_out
        return self._fixed[key]

    # ---- N4: generated constraint kernel
    def _airlib(self, air):
        k = (air.name, air.digest())
        if k not in self._airlibs:
            lib = C.CDLL(build_airs.build_air(air))
            fn = getattr(lib, air.symbol)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p] * 3 + [C.c_uint64, C.c_uint64] + [C.c_void_p] * 5 + [C.c_int, C.c_uint64,
                                                                                                 C.c_uint64, C.c_void_p]
            self._airlibs[k] = fn
        return self._airlibs[k]

    def _airlib_rows(self, air):
        """the row-window entry point of the AIR's generated library (one kernel, two launchers: stark/air.py emit_quotient_source)"""
        k = (air.name, air.digest(), "rows")
        if k not in self._airlibs:
            lib = C.CDLL(build_airs.build_air(air))
            fn = getattr(lib, air.symbol + "_rows")
            fn.restype = C.c_int
            self._airlibs[k] = fn
        return self._airlibs[k]

    def compile_air_kernel(self, air):
        """generate, compile (hipcc: seconds to a minute for a verifier AIR) and load the constraint kernel of `air` and hand it to the one-call
        prover of this ctx; returns False on a host without a compiler (the interpreter then serves: same proofs)"""
        try:
            self.p.set_air_kernel(air.program(), self._airlib(air))
            self._native_kernels.add((air.name, air.digest()))
            return True
        except (OSError, RuntimeError, AttributeError, subprocess.CalledProcessError):
            return False

    def quotient(self, air, c1, fixed, pubs, apow, zhinv, logn, logb, wlast):
        M = 1 << (logn + logb)
        if self.quotient_mode == "program" or air.fixed_cols:     # periodic fixed columns exist in the interpreter only
            out = self.p.alloc(3 * M)
            self.p.eval_quotient(air.program(), c1.ext, fixed, logn + logb, logb, [int(v) for v in pubs], apow, zhinv, self.shift, wlast, out)
            return out
        fn = self._airlib(air)
        d_pub = self.p.upload(np.array(list(pubs) + [0], dtype=np.uint64))
        d_ap = self.p.upload(np.array(apow, dtype=np.uint64).reshape(-1))
        d_zh = self.p.upload(np.array(zhinv, dtype=np.uint64))
        lo, hi, lb = self.p.domain_tables(logn + logb)
        out = self.p.alloc(3 * M)
        rc = fn(self.p.stream_handle(), c1.ext.ptr, fixed.ptr, M, 1 << logb, d_pub.ptr, d_ap.ptr, d_zh.ptr, lo, hi, lb, self.shift,
                wlast, out.ptr)
        if rc != 0:
            raise native.ZpError(-2, "constraint kernel launch failed (hip error %d)" % rc)
        self.p.sync()
        for b in (d_pub, d_ap, d_zh):
            b.free()
        return out

    def quotient_pieces(self, d_q, logn, logb, Q):
        """d_q u64[3][M] (the quotient on the coset) -> LDEs of its Q pieces u64[3Q][M], piece-major.  Piece j of plane c =
        coefficients [jN, (j+1)N) of q_c(shift X): one inverse transform, slices zero-padded to M, one forward transform."""
        N, M = 1 << logn, 1 << (logn + logb)
        d_coef = self.p.alloc(3 * M)
        self.p.intt(d_q, d_coef, logn + logb, 3)
        pad = self.p.alloc(3 * Q * M)
        self.p.memset(pad, 0, 3 * Q * M * 8)
        for j in range(Q):
            for c in range(3):
                self.p.d2d(pad.offset((3 * j + c) * M), d_coef.offset(c * M + j * N), N * 8)
        ext = self.p.alloc(3 * Q * M)
        self.p.ntt(pad, ext, logn + logb, 3 * Q)
        self.p.sync()
        pad.free()
        d_coef.free()
        return ext

    def ood_evals(self, c1, Wt, d_q, Wq, q_logn, logn, logb, zeta, zeta_w):
        """(p_k(zeta))[Wt], (p_k(zeta w))[Wt], (q_k(zeta))[Wq] from the resident extensions (zp_ood_eval, barycentric form): a
        polynomial of degree < 2^d is read on the 2^d-point sub-coset of its extension (row stride M / 2^d); zeta_w is implied"""
        logm = logn + logb
        M = 1 << logm
        ev_z, ev_zw = self.p.ood_eval(c1.ext, M, 1 << logb, Wt, logn, self.shift, zeta, want_next=True)
        ev_q = self.p.ood_eval(d_q, M, 1 << (logm - q_logn), Wq, q_logn, self.shift, zeta)
        return ev_z, ev_zw, ev_q

    def deep(self, d_a, Wa, d_b, Wb, logm, n_next, z, zw, gamma, ev_z, ev_zw):
        out = self.p.alloc(3 << logm)
        self.p.deep_quotient(d_a, Wa, d_b, Wb, logm, n_next, z, zw, gamma, ev_z, ev_zw, self.shift, out)
        return out

    def fri_fold(self, d_in, logn, logf, beta, shift):
        out = self.p.alloc(3 << (logn - logf))
        self.p.fri_fold(d_in, out, logn, logf, beta, shift)
        return out

    def download(self, d, shape):
        return self.p.download(d, shape)

    def gather_rows(self, d_cols, M, W, idx):
        return self.p.gather_rows(d_cols, M, W, idx)

    def open_paths(self, tree, M, idx):
        if self.hash_mode == "bn128":   # per query: per level the 16 digests of the group on the path
            return self.p.merkle16_open_batch_bn254(tree, M, idx)
        return self.p.merkle_open_batch(tree, M, idx)

    # ---- N6 (Groth16 wrap)
    def groth16(self, key, set_idx, set_val, rand):
        """the Groth16 wrap's prover (service/groth16.py: prove): witness completion, QAP quotient and the five MSMs behind zp_groth16_prove"""
        from ..service import groth16 as G16
        return G16.prove_on_gpu(key, set_idx, set_val, self, rand)

    def msm_g1(self, points, scalars):
        return self.p.msm_bn254([p if p is not None else (0, 0) for p in points], [int(s) for s in scalars])

    def msm_g2(self, points, scalars):
        return self.p.msm_bn254_g2(points, [int(s) for s in scalars])

    def qap_quotient(self, a_ev, b_ev, c_ev, logm, coset):
        """coefficients of H = (A B - C) / (x^m - 1): seven F_r transforms + one pointwise kernel (zp_qap_quotient_bn254)"""
        return self.p.qap_quotient_bn254(a_ev, b_ev, c_ev, logm, coset)
