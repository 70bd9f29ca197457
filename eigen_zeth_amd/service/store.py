"""Per-batch artefact store.  The client persists its own step (PROVE_STEP_RECORD,
src/prover/provider.rs:232-274) and after a restart resumes with e.g. GenChunkProof or GenFinalProof
for a batch_id first seen on an earlier connection, and it replays requests verbatim after errors
(provider.rs:332-343) -- so results are kept per batch_id, on disk, and handlers are idempotent."""
import json
import os
import threading


class BatchStore:
    def __init__(self, root):
        self.root = root
        os.makedirs(root, exist_ok=True)
        self._lock = threading.Lock()

    def _path(self, batch_id):
        safe = "".join(ch if ch.isalnum() or ch in "-_" else "_" for ch in batch_id)[:128] or "_"
        return os.path.join(self.root, safe + ".json")

    def load(self, batch_id):
        with self._lock:
            try:
                with open(self._path(batch_id)) as f:
                    return json.load(f)
            except (OSError, ValueError):
                return {}

    def save(self, batch_id, rec):
        with self._lock:
            tmp = self._path(batch_id) + ".tmp"
            with open(tmp, "w") as f:
                json.dump(rec, f)
            os.replace(tmp, self._path(batch_id))
