"""A float32 product as a sum of bfloat16 x bfloat16 products of three-way split operands (what a bf16 MFMA computes exactly, accumulated in
float32 per K = 16 chunk), against the float32 multiply-add chain the classifier's kernels run today and against float64: a 1 x 1 convolution
384 -> 64 over 4,096 pixels of post-ReLU activations (CPU, numpy).  DESIGN section 10."""
import numpy as np
rng = np.random.default_rng(0)
def bf16(x):
    # round-to-nearest-even to bfloat16, returned as float32
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)
def split3(x):
    a = bf16(x); r = (x - a).astype(np.float32); b = bf16(r); r2 = (r - b).astype(np.float32); c = bf16(r2)
    return a, b, c
K, N, P = 384, 64, 4096
x = np.maximum(rng.standard_normal((P, K)).astype(np.float32), 0) * 3      # post-ReLU activations
w = (rng.standard_normal((K, N)) * (2.0 / K) ** 0.5).astype(np.float32)
truth = x.astype(np.float64) @ w.astype(np.float64)
# (a) float32 chain
acc = np.zeros((P, N), np.float32)
for k in range(K):
    acc = (acc + x[:, k:k+1] * w[k:k+1, :]).astype(np.float32)      # not fused, close enough to the fmaf chain
ea = np.abs(acc - truth).max() / np.abs(truth).max()
# (b) three-way split, six products, float32 accumulation in chunks of 16 (an MFMA's K), exact products
x1, x2, x3 = split3(x); w1, w2, w3 = split3(w)
def run(terms):
    acc = np.zeros((P, N), np.float32)
    for k0 in range(0, K, 16):
        s = np.zeros((P, N), np.float64)
        for a, b in terms:
            s += a[:, k0:k0+16].astype(np.float64) @ b[k0:k0+16, :].astype(np.float64)
        acc = (acc + s.astype(np.float32)).astype(np.float32)
    return np.abs(acc - truth).max() / np.abs(truth).max()
e6 = run([(x1, w1), (x1, w2), (x2, w1), (x1, w3), (x2, w2), (x3, w1)])
e3 = run([(x1, w1), (x1, w2), (x2, w1)])
e1 = run([(x1, w1)])
print("relative to the largest output: f32 chain %.2e, bf16 x 6 products %.2e, x 3 products %.2e, plain bf16 %.2e" % (ea, e6, e3, e1))
