"""Shared by the statistics tests: the per-pixel comparison of SURVEY 8(c)(iii).

Two independent estimates of the same per-pixel expectation, each with its own standard error:
z = |a - b| / sqrt(se_a^2 + se_b^2 + (1e-5 * |b|)^2).  The last term is the float-rounding floor of a 4096-term
single-precision sum: corner pixels whose paths graze an empty edge of the grid are the background colour to seven
digits with a standard error of 1e-8 on both sides, and differ by 1e-7 for the order of the additions alone (they
were the whole 0.2 % of "z up to 17" when the oracle's two RNG modes were first compared).  Pixels with zero
variance on both sides (pure background) must agree exactly up to that floor as well.

The bar: |z| < 4 on at least 99.9 % of the pixel-channels and the image mean within 0.5 %.

`compare` is the symmetric two-sample form (both sides 4096 samples, each normalised by its own spread): the
oracle's two random schemes against each other.  `compare_frames` is the form for the GPU, which delivers frames,
not samples, and so many of them that its mean is nearly exact: there the reference's OWN standard error must not
normalise the difference — for a skewed per-pixel distribution (a delta-tracking pixel is a Bernoulli variable with
p near 0 or 1, a ratio-tracking pixel has rare bright NEE spikes) a 4096-sample mean and its estimated error are
correlated, and the one-sided statistic is skewed: measured on the GPU, 16 x 4096 spp against the fixture gave
mean z = -0.19 +- 0.016 and 97.5 % below 3 for the delta tracker although the same streams in the oracle pass the
symmetric test with mean z = 0.000.  Under the hypothesis being tested both sides have the same variance, and the
GPU's K frames measure it far better: z = (mean of K frames - reference mean) / (s * sqrt(1 + 1/K)), s = standard
deviation of the K frame values of the pixel (each a 4096-sample mean, like the reference's).
"""
import numpy as np

PIXEL_Z = 4.0
PIXEL_FRACTION = 0.999
IMAGE_MEAN_REL = 0.005


def compare(mean_a, se_a, mean_b, se_b):
    """-> dict(frac_z4, frac_z3, z_max, image_mean_rel); a, b float64 arrays (H, W, 3)."""
    mean_a, se_a, mean_b, se_b = (np.asarray(x, dtype=np.float64) for x in (mean_a, se_a, mean_b, se_b))
    den = np.sqrt(se_a ** 2 + se_b ** 2 + (1e-5 * np.abs(mean_b)) ** 2)
    z = np.abs(mean_a - mean_b) / np.maximum(den, 1e-300)
    return {"frac_z4": float(np.mean(z < PIXEL_Z)), "frac_z3": float(np.mean(z < 3.0)), "z_max": float(z.max()),
            "image_mean_rel": float(abs(mean_a.mean() - mean_b.mean()) / mean_b.mean())}


def passes(r):
    return r["frac_z4"] >= PIXEL_FRACTION and r["image_mean_rel"] <= IMAGE_MEAN_REL


def blocks(a, b):
    """(..., H, W, 3) -> means over b x b pixel blocks (..., H/b, W/b, 3)"""
    *lead, h, w, c = a.shape
    return a.reshape(*lead, h // b, b, w // b, b, c).mean(axis=(-4, -2))


def compare_frames(frames, ref_mean, ref_se=None, block=1, ref_samples_per_frame=1.0, own_se=False):
    """frames: (K, H, W, 3) float64, K independent renders; the reference mean was taken over `ref_samples_per_frame` times
    the samples of one frame (1: the reference's sample count; 16: the 65 536-spp ratio fixtures against 4096-spp frames),
    so its variance is the frames' spread squared over that factor."""
    frames = np.asarray(frames, dtype=np.float64)
    ref_mean = np.asarray(ref_mean, dtype=np.float64)
    if block > 1:
        frames, ref_mean = blocks(frames, block), blocks(ref_mean, block)
        if ref_se is not None:
            ref_se = np.sqrt(blocks(np.asarray(ref_se, dtype=np.float64) ** 2, block)) / block
    k = frames.shape[0]
    mean = frames.mean(axis=0)
    s = frames.std(axis=0, ddof=1)
    den = np.sqrt(s ** 2 * (1.0 / ref_samples_per_frame + 1.0 / k) + (1e-5 * np.abs(ref_mean)) ** 2)
    if own_se:
        # SURVEY 8(c)(iii) read literally: every side brings its own standard error — the reference the one of its samples
        # (the fixture's), the GPU the spread of its K frames
        den = np.sqrt(s ** 2 / k + np.asarray(ref_se, dtype=np.float64) ** 2 + (1e-5 * np.abs(ref_mean)) ** 2)
    z = (mean - ref_mean) / np.maximum(den, 1e-300)
    r = {"frac_z4": float(np.mean(np.abs(z) < PIXEL_Z)), "frac_z3": float(np.mean(np.abs(z) < 3.0)), "z_max": float(np.abs(z).max()),
         "z_mean": float(z.mean()), "z_std": float(z.std()),
         "image_mean_rel": float(abs(mean.mean() - ref_mean.mean()) / ref_mean.mean())}
    if ref_se is not None:      # same estimator, same spread: the reference's standard error against the frames'
        ref_se = np.asarray(ref_se, dtype=np.float64)
        ok = (ref_se > 0) & (s > 0)
        r["median_se_ratio"] = float(np.median(s[ok] / (ref_se[ok] * np.sqrt(ref_samples_per_frame))))
    return r
