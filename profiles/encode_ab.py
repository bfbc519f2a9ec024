import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lrp_imagecaptioning_amd.engine import LRPEngine, switches
from lrp_imagecaptioning_amd.synthetic import images
B, V = 32, 1000
w = bench.synth_weights(0, V)
rs = np.random.RandomState(1)
X = torch.as_tensor(images(rs, B)).cuda()
eng = LRPEngine(decoder="adaptive", V=V, max_images=B, max_tokens=B, max_caption_len=11)
eng.set_weights(w)
def t_enc(n=20):
    eng.encode_images(X); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): eng.encode_images(X)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for rep in range(3):
    a = t_enc()
    with switches(LRP_POOL_FUSED=0):
        b = t_enc()
    print("encode of 32 images: fused pool %.3f ms, pool pass %.3f ms" % (a, b), flush=True)
