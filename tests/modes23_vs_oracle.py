"""Modes 2 and 3 against the CPU oracle at a size with chance repeats (GPU-generated inputs)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ebwt2indel_amd import sim, api
from tests import _oracle
n_each = int(sys.argv[1]); L = 150
glen = int(2 * n_each * L / 30)
g = sim.random_genome(glen, 5); h2 = sim.mutate_haplotype(g, 5)
text, offs = sim.haplotype_text([g, h2])
rng = np.random.Generator(np.random.PCG64(99))
s0 = (rng.integers(0, len(g) - L + 1, size=n_each, dtype=np.int64) + offs[0]).astype(np.uint64)
s1 = (rng.integers(0, len(h2) - L + 1, size=n_each, dtype=np.int64) + offs[1]).astype(np.uint64)
d = sim.build_ebwt_gpu(text, np.concatenate([s0, s1]), L, n_first=n_each, want_da=True); m, da = d.to_host(); d.free()
d = sim.build_ebwt_gpu(text, s0, L); a, _ = d.to_host(); d.free()
d = sim.build_ebwt_gpu(text, s1, L); b, _ = d.to_host(); d.free()
orc = _oracle.Oracle(); p = _oracle.Params.from_args([])
t = time.time(); w3, _ = orc.run(m, da=da, params=p); t3 = time.time() - t
t = time.time(); w2, _ = orc.run(a, bwt2=b, params=p); t2 = time.time() - t
t = time.time(); g3 = api.run_two_datasets_da(m, da); tg3 = time.time() - t
t = time.time(); g2 = api.run_two_datasets(a, b); tg2 = time.time() - t
print(f"n={len(m)}: mode 3 gpu==oracle {g3 == w3} ({w3.count(b'>')} records, cpu {t3:.1f}s gpu {tg3:.2f}s); mode 2 gpu==oracle {g2 == w2} ({w2.count(b'>')} records, cpu {t2:.1f}s gpu {tg2:.2f}s); mode2==mode3 {w2 == w3}")
