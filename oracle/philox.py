"""NumPy restatement of the device noise generator (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference draws add_noise's normals from NumPy's MT19937 on the host
(model/net.py:13); a GPU cannot reproduce that stream, so in "perf mode" the build draws
sigma*N(0,1) in-kernel from Philox4x32-10 (Salmon et al. 2011, Random123) + Box-Muller.  This
file states the same generator so the in-kernel draw is checkable:
counter = (idx_lo, idx_hi, stream_lo, stream_hi), key = (seed_lo, seed_hi), one counter per
group of 4 consecutive elements.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def randn(n, sigma, seed, stream_id):
    """The first n elements of the stream (float64 math; the device uses float32 intrinsics)."""
    n4 = (n + 3) // 4
    idx = np.arange(n4, dtype=np.uint64)
    z = np.zeros(n4, np.uint32)
    r = philox4x32_10((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                      z + np.uint32(stream_id & 0xFFFFFFFF), z + np.uint32(stream_id >> 32),
                      seed & 0xFFFFFFFF, seed >> 32)
    S = 2.0 ** -32
    f = [np.float32(x).astype(np.float64) for x in r]            # the device converts uint32 -> float32 first
    u1 = np.minimum(np.float32((f[0] + 1.0)).astype(np.float64) * S, 1.0)
    u2 = f[1] * S
    u3 = np.minimum(np.float32((f[2] + 1.0)).astype(np.float64) * S, 1.0)
    u4 = f[3] * S
    ra, rb = np.sqrt(-2 * np.log(u1)), np.sqrt(-2 * np.log(u3))
    out = np.stack([ra * np.cos(2 * np.pi * u2), ra * np.sin(2 * np.pi * u2),
                    rb * np.cos(2 * np.pi * u4), rb * np.sin(2 * np.pi * u4)], axis=1).reshape(-1)
    return sigma * out[:n]


def randint(n, modulus, seed, stream_id):
    """The first n integers of the stream (mcg_randint): element i = word i & 3 of counter i >> 2, modulo `modulus` -- the
    device's stand-in for the generator's label draw xp.random.randint(dim_zl, size=batchsize) (model/net.py:91-92)."""
    n4 = (n + 3) // 4
    idx = np.arange(n4, dtype=np.uint64)
    z = np.zeros(n4, np.uint32)
    r = philox4x32_10((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                      z + np.uint32(stream_id & 0xFFFFFFFF), z + np.uint32(stream_id >> 32),
                      seed & 0xFFFFFFFF, seed >> 32)
    return (np.stack(r, axis=1).reshape(-1)[:n] % np.uint32(modulus)).astype(np.int64)


def randn_rowquad(M, C, sigma, seed, stream_id):
    """[M][C] noise in the element order of the fused first-layer epilogue (mcg_conv_epilogue.sigma,
    mcg_randn_rowquad): element (m, c) is normal m & 3 of counter (m >> 2) * C + c -- the same generator, one
    counter per (row quad, channel) instead of per 4 consecutive channels.  M % 4 == 0."""
    assert M % 4 == 0
    z = randn(M * C, sigma, seed, stream_id).reshape(M // 4, C, 4)          # counter (mq, c) -> its 4 normals
    return np.ascontiguousarray(z.transpose(0, 2, 1)).reshape(M, C)


# ---- the randomness of one perf-mode iteration (what bench.py times and Updater.update_core runs) ------------------------------
# Written from the SPECIFICATION of the ids, independently of step.py / nets.py:
#   base(it, rank) = (it * 64 + rank + 1) * 64; call k of the iteration owns ids base + 8 k ..: k = 0 D_I(real), 1 D_V(real),
#   2 the generator's latent draw, 3 D_I(fake), 4 D_V(fake); a discriminator call uses id + l - 1 for the add_noise in front of layer
#   l = 1..4 (model/net.py:148-154,189-195), the latent draw id + 0 / 1 / 2 / 3 for h0 / e / zc / labels (model/net.py:66,71,102,92).
# Used by tests/test_gpu_step.py and __graft_entry__.smoke() to feed the oracle the draws the device makes in-kernel.
def perf_mode_randomness(seed, it, rank, model, n, nf, dim_zl, c_img=3, T=16, dim_zc=50, dim_zm=10, sigma=0.2):
    base = (it * 64 + rank + 1) * 64
    cd = c_img + (dim_zl if model == 'cgan' else 0)
    cp = (cd + 3) // 4 * 4

    def to_ref(z, ndim):                                  # [n][T][H][W][C] -> (n,C,T,H,W) / (n,C,H,W)
        z = z.transpose(0, 4, 1, 2, 3)
        return np.ascontiguousarray(z[:, :, 0] if ndim == 2 else z)

    def dis_noise(ndim, k):
        sid = base + 8 * k
        t = T if ndim == 3 else 1
        out = [to_ref(randn(n * t * 64 * 64 * cp, sigma, seed, sid).reshape(n, t, 64, 64, cp)[..., :cd], ndim)]
        t = t - 3 if ndim == 3 else 1
        out.append(to_ref(randn_rowquad(n * t * 32 * 32, nf, sigma, seed, sid + 1).reshape(n, t, 32, 32, nf), ndim))
        for l, (h, c) in ((3, (16, 2 * nf)), (4, (8, 4 * nf))):
            t = t - 3 if ndim == 3 else 1
            out.append(to_ref(randn(n * t * h * h * c, sigma, seed, sid + l - 1).reshape(n, t, h, h, c), ndim))
        return out

    import torch                                          # (the frame index is a torch host generator draw)
    g = torch.Generator()
    g.manual_seed(seed * 7919 + it)                       # the frame index: one host draw per iteration, shared by all ranks (Q7)
    rnd = {'t': int(torch.randint(0, T, (1,), generator=g))}
    rnd['noise_i_real'], rnd['noise_v_real'] = dis_noise(2, 0), dis_noise(3, 1)
    sid = base + 16
    rnd['gen'] = {'h0': randn(n * dim_zm, 0.33, seed, sid).reshape(n, dim_zm),
                  'e': randn(T * n * dim_zm, 0.33, seed, sid + 1).reshape(T, n, dim_zm),
                  'zc': randn(n * dim_zc, 0.33, seed, sid + 2).reshape(n, dim_zc),
                  'labels': randint(n, dim_zl, seed, sid + 3) if dim_zl else None}
    rnd['noise_i_fake'], rnd['noise_v_fake'] = dis_noise(2, 3), dis_noise(3, 4)
    return rnd
