"""Python twin of the device random-number path (csrc/smc.hip): Philox4x32-10 (Salmon et al.,
SC'11), the 53-bit uniform and the Box-Muller pair.  Test infrastructure: the GPU tests compare the
kernels' draws with it; tests/test_samplers_cpu.py pins it to the published known-answer vectors."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """counters: uint32 arrays of equal shape; key: two python ints -> four uint32 arrays"""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xffffffff, int(k1) & 0xffffffff
    for _ in range(10):
        p0 = M0 * c0.astype(np.uint64)
        p1 = M1 * c2.astype(np.uint64)
        n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ np.uint32(k0)
        n1 = (p1 & np.uint64(0xffffffff)).astype(np.uint32)
        n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ np.uint32(k1)
        n3 = (p0 & np.uint64(0xffffffff)).astype(np.uint32)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xffffffff
        k1 = (k1 + W1) & 0xffffffff
    return c0, c1, c2, c3


def u53(hi, lo):
    return ((hi >> np.uint32(5)).astype(np.float64) * 67108864.0
            + (lo >> np.uint32(6)).astype(np.float64) + 0.5) / 9007199254740992.0


def _pair(j, chain, step, stream, seed):
    r = philox4x32_10(j, chain, np.full_like(j, step), np.full_like(j, stream), seed & 0xffffffff, seed >> 32)
    u1, u2 = u53(r[0], r[1]), u53(r[2], r[3])
    rad, th = np.sqrt(-2.0 * np.log(u1)), 6.283185307179586476925286766559 * u2
    return rad * np.cos(th), rad * np.sin(th), u1


def normals(C, K, seed, step, first_chain=0):
    """z [C, K] as beatamd_proposal_draw generates them (stream 0)"""
    npair = (K + 1) // 2
    cc, jj = np.meshgrid(np.arange(C, dtype=np.uint32) + np.uint32(first_chain),
                         np.arange(npair, dtype=np.uint32), indexing="ij")
    a, b, _ = _pair(jj.ravel(), cc.ravel(), step, 0, seed)
    z = np.empty((C, 2 * npair))
    z[:, 0::2] = a.reshape(C, npair)
    z[:, 1::2] = b.reshape(C, npair)
    return z[:, :K]


def log_uniforms(C, seed, step, first_chain=0):
    chain = np.arange(C, dtype=np.uint32) + np.uint32(first_chain)
    _, _, u1 = _pair(np.zeros(C, dtype=np.uint32), chain, step, 2, seed)
    return np.log(u1)


def t_row_scale(C, seed, step, df, first_chain=0):
    chain = np.arange(C, dtype=np.uint32) + np.uint32(first_chain)
    x = np.zeros(C)
    for m in range(0, df, 2):
        g0, g1, _ = _pair(np.full(C, m // 2, dtype=np.uint32), chain, step, 1, seed)
        x += g0 * g0
        if m + 1 < df:
            x += g1 * g1
    return 1.0 / np.sqrt(x / df)


def univariate(C, npar, kind, scale, seed, step, first_chain=0):
    """delta [C, npar] as beatamd_proposal_draw_univariate generates it (streams 3, 4):
    kind 0 normal, 1 Cauchy, 2 Laplace; each times scale[j]; kind 3 Poisson: poisson(lam=scale[j]) - scale[j] by
    inversion of ONE uniform (sequential search)"""
    npair = (npar + 1) // 2
    cc, jj = np.meshgrid(np.arange(C, dtype=np.uint32) + np.uint32(first_chain),
                         np.arange(npair, dtype=np.uint32), indexing="ij")
    j, c = jj.ravel(), cc.ravel()
    r = philox4x32_10(j, c, np.full_like(j, step), np.full_like(j, 3), seed & 0xffffffff, seed >> 32)
    u1, u2 = u53(r[0], r[1]), u53(r[2], r[3])
    if kind == 0:
        rad, th = np.sqrt(-2.0 * np.log(u1)), 6.283185307179586476925286766559 * u2
        a, b = rad * np.cos(th), rad * np.sin(th)
    elif kind == 1:
        a, b = np.tan(np.pi * (u1 - 0.5)), np.tan(np.pi * (u2 - 0.5))
    elif kind == 2:
        q = philox4x32_10(j, c, np.full_like(j, step), np.full_like(j, 4), seed & 0xffffffff, seed >> 32)
        a, b = np.log(u53(q[0], q[1])) - np.log(u1), np.log(u53(q[2], q[3])) - np.log(u2)
    else:
        sc = np.concatenate([np.asarray(scale, dtype=np.float64), [0.0]])

        def inv(u, lam):
            out = np.zeros(u.size)
            for i in range(u.size):
                if lam[i] <= 0:
                    continue
                p = F = np.exp(-lam[i])
                k = 0
                while u[i] > F and k < 4096:
                    k += 1
                    p *= lam[i] / k
                    Fn = F + p
                    if Fn == F and k > lam[i]:      # the cumulative sum has stopped growing: the far tail
                        break
                    F = Fn
                out[i] = k
            return out - lam
        la, lb = sc[np.minimum(2 * j, npar)], sc[np.minimum(2 * j + 1, npar)]
        z = np.empty((C, 2 * npair))
        z[:, 0::2] = inv(u1, la).reshape(C, npair)
        z[:, 1::2] = inv(u2, lb).reshape(C, npair)
        return z[:, :npar]
    z = np.empty((C, 2 * npair))
    z[:, 0::2] = a.reshape(C, npair)
    z[:, 1::2] = b.reshape(C, npair)
    return z[:, :npar] * np.asarray(scale)[None, :]
