import os, sys
os.environ.setdefault("GPUART_LIBDIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpuart_amd", "lib_test"))  # uses test hooks (include/gpuart_hip_test.h)
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpuart_amd import binding as B
g = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/aabb_irregular.npz"))
pad4 = lambda a: np.concatenate([a, np.zeros((len(a), 1), np.float32)], 1)
be = B.Backend(0)
got = be.test_aabb(pad4(g["rs"]), pad4(g["rd"]), pad4(g["bmin"]), pad4(g["bmax"]))[:, :2]
exp = g["out"]
bad = ~((got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))).all(1)
print("mismatches:", int(bad.sum()), "kinds (i%16):", np.bincount(np.nonzero(bad)[0] % 16, minlength=16), "ray kinds ((i//16)%8):", np.bincount((np.nonzero(bad)[0] // 16) % 8, minlength=8))
for i in np.nonzero(bad)[0][:12]:
    print(i, "kind", i % 16, (i // 16) % 8, "o", g["rs"][i], "d", g["rd"][i], "lo", g["bmin"][i], "hi", g["bmax"][i], "gpu", got[i], "ref", exp[i])
be.close()
