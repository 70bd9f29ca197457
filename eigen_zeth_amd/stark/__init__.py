"""STARK stage of the batch prover (what answers GenChunkProof, proto/prover/v1/prover.proto:56-66).

The real zkEVM AIR/PIL is not in the reference and not obtainable offline; the
stage is exercised with synthetic AIRs (air.py).  Host orchestration is Python over the C-ABI; every
O(trace) computation runs in HIP kernels (csrc/*.hip and the generated constraint kernels)."""
