"""prover.v1.ProverService -- the gRPC boundary eigen-zeth's ProverChannel talks to
(proto/prover/v1/prover.proto; client src/prover/provider.rs)."""
