"""The random stream, checked without the oracle: a numpy restatement of Philox4x32 (pinned to Random123's published
known answers here), the word-stream layout the renderer draws from -- counter (block, bounce, sample, pixel), key =
seed, words of blocks 0, 1, 2, ... in order -- and sample statistics of what is drawn from it over neighbouring
(pixel, sample, bounce) counters: the pixel jitter (main.cu:90-91) and the rejection samplers of cuda_functions.h:14-20
and :65-71.  Seven rounds is the fewest the Philox paper certifies; the oracle and the device share the stream, so a
test through either cannot notice a weak one.  Every bound below is four standard errors of the exact moment."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
ROUNDS = 7
SEED = 1984


def philox4x32(c, key, rounds):
    """c: (4, n) uint64 holding 32-bit values; key: two ints.  Returns (4, n) uint32."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in c)
    k0, k1 = key[0] & 0xFFFFFFFF, key[1] & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0, p1 = np.uint64(M0) * c0, np.uint64(M1) * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3]).astype(np.uint32)


def test_numpy_philox_known_answers():
    def one(ctr, key, rounds):
        return [int(x) for x in philox4x32([[v] for v in ctr], key, rounds)[:, 0]]

    assert one([0, 0, 0, 0], [0, 0], 10) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert one([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], 10) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    assert one([0, 0, 0, 0], [0, 0], 7) == [0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48]
    assert one([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, 7) == [0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662]
    assert one([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], 7) == [0x4DFCCABA, 0x190A87F0, 0xC47362BA, 0xB6B5242A]


def stream_words(pixel, sample, bounce, n_blocks):
    """(n, 4 * n_blocks) words of the streams of the given counters."""
    out = [philox4x32([np.full_like(pixel, blk), bounce, sample, pixel], (SEED, 0), ROUNDS).T for blk in range(n_blocks)]
    return np.concatenate(out, axis=1)


def uniform(x):  # (0, 1] in steps of 2^-24, curand_uniform's range
    return ((x >> 8).astype(np.float64) + 1.0) * 2.0 ** -24


def signed_unit(x):  # 2u - 1
    return 2.0 * uniform(x) - 1.0


def _counters():
    """10^6 neighbouring counters as a render uses them: a 125 x 100 block of pixels of a 1920-wide frame, 8 samples, 10 bounces."""
    px, s, b = np.meshgrid(np.arange(12500), np.arange(8), np.arange(10), indexing="ij")
    pixel = (px // 125 * 1920 + px % 125 + 700 * 1920).ravel().astype(np.uint64)
    return pixel, s.ravel().astype(np.uint64) + 40, b.ravel().astype(np.uint64)


def test_uniforms_of_neighbouring_counters():
    pixel, sample, bounce = _counters()
    n = len(pixel)
    w = stream_words(pixel, sample, bounce, 2)
    u = uniform(w)
    se = np.sqrt(1.0 / 12.0 / n)
    assert np.all(np.abs(u.mean(axis=0) - (0.5 + 2.0 ** -25)) < 4 * se)          # every word position: mean
    assert np.all(np.abs(u.var(axis=0) - 1.0 / 12.0) < 4 * np.sqrt(1.0 / 180.0 / n))  # variance (4th central moment 1/80)
    # no correlation between word positions of one stream ...
    c = np.corrcoef(u.T)
    assert np.abs(c - np.eye(8)).max() < 4.0 / np.sqrt(n)
    # ... nor between the same word of neighbouring counters: next bounce, next sample, next pixel, next row
    cube = u[:, 0].reshape(100, 125, 8, 10)
    for a, b in ((cube[..., :-1], cube[..., 1:]), (cube[:, :, :-1], cube[:, :, 1:]), (cube[:, :-1], cube[:, 1:]), (cube[:-1], cube[1:])):
        r = np.corrcoef(a.ravel(), b.ravel())[0, 1]
        assert abs(r) < 4.0 / np.sqrt(a.size), r
    # equidistribution of the top byte (chi-square, 255 degrees of freedom: mean 255, sd 22.6)
    counts = np.bincount((w[:, 0] >> 24).astype(np.int64), minlength=256)
    chi2 = ((counts - n / 256.0) ** 2 / (n / 256.0)).sum()
    assert abs(chi2 - 255.0) < 4 * np.sqrt(2 * 255.0), chi2


def test_unit_sphere_sampler_over_the_word_stream():
    """cuda_functions.h:14-20 over the stream: try k takes words 3k, 3k + 1, 3k + 2.  The accepted point is uniform in
    the unit ball: mean 0 (variance 1/5 per axis), E|p|^2 = 3/5 (variance 12/175), acceptance pi/6 per try."""
    pixel, sample, bounce = _counters()
    n = len(pixel)
    w = stream_words(pixel, sample, bounce + 1, 9)  # 12 tries; a stream still without a point then: (1 - pi/6)^12 = 1.4e-4
    p = signed_unit(w).reshape(n, 12, 3)
    inside = (p ** 2).sum(axis=2) < 1.0
    first = inside.argmax(axis=1)
    got = inside.any(axis=1)
    pts = p[np.arange(n), first][got]
    m = len(pts)
    assert m > 0.999 * n
    assert np.all(np.abs(pts.mean(axis=0)) < 4 * np.sqrt(0.2 / m))
    assert abs((pts ** 2).sum(axis=1).mean() - 0.6) < 4 * np.sqrt(12.0 / 175.0 / m)
    acc = inside[:, 0].mean()
    assert abs(acc - np.pi / 6) < 4 * np.sqrt(np.pi / 6 * (1 - np.pi / 6) / n)
    tries = first[got] + 1  # geometric: mean 6 / pi
    q = np.pi / 6
    assert abs(tries.mean() - 1 / q) < 4 * np.sqrt((1 - q) / q ** 2 / m) + 12 * (1 - q) ** 12
    # the direction of the point (what lambertian scatter uses, material.h:91) is uniform on the sphere: z uniform in [-1, 1]
    z = pts[:, 2] / np.linalg.norm(pts, axis=1)
    hist = np.bincount(np.minimum((0.5 * (z + 1) * 20).astype(np.int64), 19), minlength=20)
    chi2 = ((hist - m / 20.0) ** 2 / (m / 20.0)).sum()
    assert abs(chi2 - 19.0) < 4 * np.sqrt(2 * 19.0), chi2


def test_jitter_and_lens_disk_of_the_camera_stream():
    """Stream 0 of a sample: words 0, 1 are the pixel jitter (main.cu:90-91), then two words per unit-disk try
    (cuda_functions.h:65-71).  The accepted point is uniform in the unit disk: mean 0, variance 1/4 per axis,
    acceptance pi/4; it is uncorrelated with the jitter."""
    pixel, sample, _ = _counters()
    pixel, sample = pixel[::10], sample[::10]  # bounce 0 only: 10^5 streams ... of 8 x as many pixels
    pixel = np.concatenate([pixel + np.uint64(k * 200) for k in range(8)])
    sample = np.tile(sample, 8)
    n = len(pixel)
    w = stream_words(pixel, sample, np.zeros_like(pixel), 5)
    jit = uniform(w[:, :2])
    assert np.all(np.abs(jit.mean(axis=0) - 0.5) < 4 * np.sqrt(1.0 / 12.0 / n))
    p = signed_unit(w[:, 2:]).reshape(n, 9, 2)
    inside = (p ** 2).sum(axis=2) < 1.0
    first = inside.argmax(axis=1)
    got = inside.any(axis=1)
    pts = p[np.arange(n), first][got]
    m = len(pts)
    assert m > 0.99999 * n
    assert np.all(np.abs(pts.mean(axis=0)) < 4 * np.sqrt(0.25 / m))
    assert np.all(np.abs((pts ** 2).mean(axis=0) - 0.25) < 4 * np.sqrt((1.0 / 8.0 - 1.0 / 16.0) / m))
    assert abs(inside[:, 0].mean() - np.pi / 4) < 4 * np.sqrt(np.pi / 4 * (1 - np.pi / 4) / n)
    r = np.corrcoef(np.concatenate([jit[got], pts], axis=1).T)
    assert np.abs(r - np.eye(4)).max() < 4.0 / np.sqrt(m)
