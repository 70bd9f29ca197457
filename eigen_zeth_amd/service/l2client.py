"""Block-input fetcher: GenBatchChunks carries only block numbers
(proto/prover/v1/prover.proto:49-54,68-70), so the prover pulls the block from the L2 node's JSON-RPC
(the same endpoint eigen-zeth's L2Watcher polls, src/batch_proposer/mod.rs:68-106; default
ZETH_L2_ADDR http://localhost:8546, src/config/env.rs:19-35).  Standard library only."""
import json
import urllib.request


class L2Client:
    def __init__(self, url, timeout=5.0):
        self.url, self.timeout, self._id = url, timeout, 0

    def _call(self, method, params):
        self._id += 1
        req = urllib.request.Request(self.url, data=json.dumps({"jsonrpc": "2.0", "id": self._id, "method": method,
                                                                "params": params}).encode(),
                                     headers={"Content-Type": "application/json"})
        with urllib.request.urlopen(req, timeout=self.timeout) as r:
            out = json.loads(r.read().decode())
        if "error" in out:
            raise RuntimeError("L2 node error: %s" % out["error"])
        return out["result"]

    def block(self, number):
        """{number, hash, state_root (32 B), parent_state_root (32 B or None), n_tx, tx_hashes}"""
        b = self._call("eth_getBlockByNumber", [hex(int(number)), False])
        if b is None:
            raise RuntimeError("block %d not found on the L2 node" % number)
        parent = self._call("eth_getBlockByNumber", [hex(int(number) - 1), False]) if int(number) > 0 else None
        root = bytes.fromhex(b["stateRoot"][2:])
        proot = bytes.fromhex(parent["stateRoot"][2:]) if parent else None
        if len(root) != 32 or (proot is not None and len(proot) != 32):
            raise RuntimeError("malformed stateRoot from the L2 node")
        txs = [t if isinstance(t, str) else t["hash"] for t in b.get("transactions", [])]
        return {"number": int(b["number"], 16), "hash": b["hash"], "state_root": root, "parent_state_root": proot,
                "n_tx": len(txs), "tx_hashes": txs}
