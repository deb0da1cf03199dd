"""Feature rows padded to a multiple of four floats for the autograd operators."""
import torch

# Below this many rows a call is bound by launch latency, not by how its lanes move the rows:
# the two extra copies would only add launches (Cora-sized graphs).
PAD_MIN_ROWS = 1 << 15


def pad4(x):
    """[n,h,f] -> the same rows zero-padded to the next multiple of four features.  A row of
    4 x 41 or 1 x 47 floats is not 16-byte aligned, which leaves the kernels scalar lanes (one
    head per walk over the edges, and the one-pass backward out of range for three 41-wide
    kernels); padded to 4 x 44 / 1 x 48 the same rows move as float4 in one walk.  Zero features
    gather, accumulate and dot to zero, so the first f features are unchanged."""
    f = x.size(2)
    fp = (f + 3) // 4 * 4
    if fp == f or x.size(0) < PAD_MIN_ROWS:
        return x
    return torch.nn.functional.pad(x, (0, fp - f))


def pad_to(x, width):
    """[m,h,f] zero-padded to exactly `width` features: the output gradient of a padded forward.
    The forward decided from ITS tensor's row count whether to pad; the gradient has the row
    count of the other side of a rectangular block, so it follows the saved tensor's width
    instead of deciding again (m < PAD_MIN_ROWS <= n would leave the two at different widths)."""
    f = x.size(2)
    return x if f == width else torch.nn.functional.pad(x, (0, width - f))


def unpad(x, f):
    """View of the first f features (the caller's width)."""
    return x if x.size(2) == f else x[..., :f]
