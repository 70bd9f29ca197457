"""The tests' CPU backend with the Groth16 stage of the backend interface (eigen_zeth_amd/service/groth16.py: prove -> be.groth16): the
witness completed by the library's HOST evaluator (zp_r1cs_eval: no GPU), the three proof elements by the trapdoor of the seeded test key
(oracle/groth16_trapdoor.py) -- the same group elements the GPU's transforms and MSMs must give.  Test infrastructure."""
import numpy as np

from eigen_zeth_amd import native
from oracle.stark_cpu import CpuBackend


class CpuWrapBackend(CpuBackend):
    def groth16(self, key, set_idx, set_val, rand):
        w = np.zeros((key.n_wires, 4), dtype=np.uint64)
        mask = np.zeros(key.n_wires, dtype=np.uint8)
        w[set_idx.astype(np.int64)] = set_val
        mask[set_idx.astype(np.int64)] = 1
        wf, _, _, _ = native.r1cs_eval(key.blob, w, mask)
        return self.groth16_prove(key, native.fr_ints(wf), rand), native.fr_ints(wf[1:1 + key.n_pub]), [0.0, 0.0, 0.0]
