import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import splat_renderer_amd as sr
from oracle import oracle as O
from tests.helpers import make_case, oracle_pipeline
from tests.test_gpu_stages import run_gpu_pipeline
n,w,h,seed,rs = 1000,256,256,3,1.0
dev = sr.Device(0)
props, normals, u = make_case(n,w,h,seed,rs)
ref = oracle_pipeline(props, normals, u, w, h)
want, want8, _ = O.composite(0, False, props[:, 4:], normals, ref["proj"], ref["indices"], ref["counts"], ref["offsets"], w, h)
g = run_gpu_pipeline(dev, props, normals, u, n, w, h)
r = sr.ComputeShaderRenderer(dev, None, "rgba8unorm", mode=0, earlyOut=False)
b = g["binner"]
r.render(u, g["pm"].getPropertyBuffer(), b.getTileIndicesBuffer(), g["nbuf"], g["proj"].getProjectedBuffer(), b.getTileCountsBuffer(), b.getTileOffsetsBuffer(), 16, 16, w, h, wantFloat=True)
got = r.readPixelsFloat()
err = np.abs(got-want).max(axis=2)
ys,xs = np.nonzero(err>1e-4)
print(len(ys), 'bad pixels')
for y,x in list(zip(ys,xs))[:20]:
    print(y,x,'tile',y//16,x//16,'in-tile',y%16,x%16, err[y,x], got[y,x], want[y,x], 'count', ref['counts'][(y//16)*16+x//16])
